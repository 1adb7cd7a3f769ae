"""write-heavy streaming rates: fill (write only) and 1 read : 4 writes (the byte mix of the 64 -> 256 1x1 layer)"""
import os, sys, torch
sys.path.insert(0, os.getcwd())


def timed(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); torch.cuda.synchronize()
        ts.append(s.elapsed_time(e))
    return sorted(ts)[len(ts) // 2]


M = 537600
for dt, esz in ((torch.bfloat16, 2), (torch.float32, 4)):
    c = torch.empty(M, 256, device='cuda', dtype=dt)
    x = torch.randn(M, 64, device='cuda').to(dt)
    ms = timed(lambda: c.fill_(1.0))
    print(f'{dt}: fill {M * 256 * esz / 1e6:.0f} MB {ms * 1e3:.1f} us {M * 256 * esz / ms / 1e9:.2f} TB/s')
    ms = timed(lambda: c.zero_())
    print(f'{dt}: zero {ms * 1e3:.1f} us {M * 256 * esz / ms / 1e9:.2f} TB/s')
    cv = c.view(M, 4, 64)
    ms = timed(lambda: cv.copy_(x.unsqueeze(1).expand(M, 4, 64)))
    print(f'{dt}: 1R:4W {ms * 1e3:.1f} us {M * 320 * esz / ms / 1e9:.2f} TB/s')
    a = torch.randn(M, 256, device='cuda').to(dt)
    ms = timed(lambda: torch.relu(a, out=c) if False else torch.clamp_min(a, 0, out=c))
    print(f'{dt}: 1R:1W relu {ms * 1e3:.1f} us {2 * M * 256 * esz / ms / 1e9:.2f} TB/s')
