"""strided writes: 256-byte / 128-byte pieces of 512-byte rows (what one column tile of a 256-channel bf16 output writes)"""
import os, sys, torch


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
c = torch.empty(M, 256, device='cuda', dtype=torch.bfloat16)
for w in (256, 128, 64, 32):
    v = c[:, :w]
    ms = timed(lambda: v.fill_(1.0))
    print(f'fill {w * 2} B of every 512 B row: {ms * 1e3:.1f} us {M * w * 2 / ms / 1e9:.2f} TB/s')
a = torch.randn(M, 64, device='cuda').bfloat16()
w4 = torch.randn(64, 256, device='cuda').bfloat16()
ms = timed(lambda: torch.mm(a, w4, out=c))
print(f'torch.mm (hipBLASLt) M x 64 @ 64 x 256 bf16: {ms * 1e3:.1f} us  {(M * 64 + M * 256) * 2 / ms / 1e9:.2f} TB/s')
c64 = torch.empty(M, 64, device='cuda', dtype=torch.bfloat16)
a256 = torch.randn(M, 256, device='cuda').bfloat16(); w2 = torch.randn(256, 64, device='cuda').bfloat16()
ms = timed(lambda: torch.mm(a256, w2, out=c64))
print(f'torch.mm M x 256 @ 256 x 64: {ms * 1e3:.1f} us  {(M * 64 + M * 256) * 2 / ms / 1e9:.2f} TB/s')
w1 = torch.randn(64, 64, device='cuda').bfloat16()
ms = timed(lambda: torch.mm(a, w1, out=c64))
print(f'torch.mm M x 64 @ 64 x 64: {ms * 1e3:.1f} us  {(M * 64 + M * 64) * 2 / ms / 1e9:.2f} TB/s')
