"""how long the main stream waits for the weight-gradient stream at the end of the backward pass (bf16 train step of bench.py):
HIP events on the main stream right before and right after the end-of-pass join"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.argv = ['bench.py', '--mode', 'train', '--steps', '20', '--warmup', '5', '--no-cpu-baseline']
import torch
import bench
from brcnn import autograd as A

waits = []
orig = A._queue_stream_join


def patched(main, side):
    key = (main.device.type, main.device.index)
    if A._join_queued.get(key) or A._DEFER_JOIN[0]:
        return

    def join():
        A._join_queued[key] = False
        A._side_seen.pop(key, None)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(main)
        main.wait_stream(side)
        e1.record(main)
        waits.append((e0, e1))
    A._join_queued[key] = True
    try:
        torch.autograd.Variable._execution_engine.queue_callback(join)
    except RuntimeError:
        join()


A._queue_stream_join = patched
import io, contextlib
with contextlib.redirect_stdout(io.StringIO()):
    bench.main()
torch.cuda.synchronize()
ms = sorted(a.elapsed_time(b) for a, b in waits[5:])
print(f'{len(ms)} joins: main stream waited for the weight-gradient stream median {ms[len(ms) // 2]:.3f} ms, min {ms[0]:.3f}, max {ms[-1]:.3f}')
