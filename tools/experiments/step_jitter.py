"""per-step device and host times of bench.py's train step, as a list: is the jitter periodic (something in this process) or
random (the box)?   python tools/experiments/step_jitter.py [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
steps = sys.argv[1] if len(sys.argv) > 1 else '60'
sys.argv = ['bench.py', '--mode', 'train', '--steps', steps, '--warmup', '5', '--no-cpu-baseline']
import bench
import torch


def timed(step, steps, warmup, world, device):
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    marks = []
    import gc
    gc.collect(); gc.disable()
    evs[0].record()
    t0 = time.perf_counter()
    for i in range(steps):
        step()
        evs[i + 1].record()
        marks.append(time.perf_counter())
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    gc.enable()
    if steps > 1:
        dev = [evs[i].elapsed_time(evs[i + 1]) for i in range(steps)]
        host = [1e3 * (b - a) for a, b in zip([t0] + marks[:-1], marks)]
        print('device ms:', ' '.join(f'{v:.1f}' for v in dev), file=sys.stderr)
        print('host   ms:', ' '.join(f'{v:.1f}' for v in host), file=sys.stderr)
        try:
            print('loadavg', open('/proc/loadavg').read().strip(), 'cpus', os.cpu_count(), 'affinity', len(os.sched_getaffinity(0)), file=sys.stderr)
        except Exception:
            pass
    return dt, {'step_ms_median': 0, 'step_ms_p90': 0, 'step_ms_min': 0, 'step_ms_max': 0, 'host_enqueue_ms_median': 0, 'host_enqueue_ms_max': 0}


bench.timed = timed
bench.main()
