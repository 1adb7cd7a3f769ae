"""bf16 train step of bench.py under values of the brcnn_tuning switches (one process, interleaved rounds, median of the
per-step HIP-event times):   python tools/experiments/tuning_sweep.py [rounds]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 2
sys.argv = ['bench.py', '--mode', 'train', '--steps', '12', '--warmup', '4', '--no-cpu-baseline']
import torch
import bench
from brcnn import lib

results = {}
cells = [('default', {})]
for v in (50, 60, 90, 100):
    cells.append((f'wgrad_generation_percent={v}', dict(wgrad_generation_percent=v)))
for v in (50, 60, 90, 100):
    cells.append((f'wgrad_eight_phase_cu_percent={v}', dict(wgrad_eight_phase_cu_percent=v)))
cells += [('conv_stream_k=0', dict(conv_stream_k=0)), ('conv_stream_k=2', dict(conv_stream_k=2)),
          ('conv_persistent_1x1=0', dict(conv_persistent_1x1=0)), ('conv_persistent_1x1=2', dict(conv_persistent_1x1=2)),
          ('wgrad_eight_phase=0', dict(wgrad_eight_phase=0)), ('wgrad_eight_phase=2', dict(wgrad_eight_phase=2)),
          ('conv_split_k=1', dict(conv_split_k=1))]
base = lib.get_tuning()
defaults = {n: getattr(base, n) for n, _ in lib.Tuning._fields_ if n != 'size'}
orig_timed = bench.timed
state = {}


def timed(step, steps, warmup, world, device):
    # called once by train_bench with the warm model: run every cell `rounds` times, interleaved
    for _ in range(warmup):
        step()
    for r in range(rounds):
        for name, kw in cells:
            lib.set_tuning(**defaults)
            lib.set_tuning(**kw)
            for _ in range(2):
                step()
            dt, st = orig_timed(step, steps, 0, world, device)
            results.setdefault(name, []).append(st['step_ms_median'])
    lib.set_tuning(**defaults)
    return orig_timed(step, steps, 0, world, device)


bench.timed = timed
import io, contextlib
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    bench.main()
ref = sorted(results['default'])[len(results['default']) // 2]
for name, v in results.items():
    med = sorted(v)[len(v) // 2]
    print(f'{name:40s} median {med:6.2f} ms ({med - ref:+.2f})  rounds: {" ".join(f"{x:.2f}" for x in v)}')
