"""How much slack the host has in the train step: BRCNN_TIME_SYNC=1 makes the one host synchronisation of the step
(the sampler counts, roi_heads.sample_device) report how long it blocks.  A wait near zero means the device is waiting
for launches; a long wait means the step is device-bound."""
import os, sys, time
os.environ['BRCNN_TIME_SYNC'] = '1'
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.argv = ['bench.py', '--mode', 'train', '--steps', '20', '--warmup', '5', '--no-cpu-baseline']
import bench
from brcnn import roi_heads
import torch
_orig = bench.timed


def timed(step, steps, warmup, world, device, after_warmup=None):
    def wrapped():
        t0 = time.perf_counter()
        step()
        host[0] += time.perf_counter() - t0
    host = [0.0]
    for _ in range(warmup):
        wrapped()
    if after_warmup is not None:
        after_warmup()
    torch.cuda.synchronize()
    host[0] = 0.0
    roi_heads.SYNC_WAIT[0], roi_heads.SYNC_WAIT[1] = 0.0, 0
    t0 = time.perf_counter()
    for _ in range(steps):
        wrapped()
    t_queued = time.perf_counter() - t0
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f'per step: wall {1e3 * dt / steps:.2f} ms, host inside step() {1e3 * host[0] / steps:.2f} ms '
          f'(of which blocked on the sampler counts {1e3 * roi_heads.SYNC_WAIT[0] / steps:.2f} ms), '
          f'device tail after the last step was queued {1e3 * (dt - t_queued):.2f} ms', file=sys.stderr)
    return dt, {}


bench.timed = timed
bench.main()
