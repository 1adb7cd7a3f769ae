"""one fp32 conv layer (the RPN tower layer at level 0) a few times: the target of tools/pmc_conv.sh"""
import sys, os, torch
sys.path.insert(0, os.getcwd())
import brcnn
from brcnn import ops
N, H, W, Ci, Co, k = 8, 100, 168, 256, 256, 3
x = torch.randn(N, H, W, Ci, device='cuda')
w = torch.randn(Co, k, k, Ci, device='cuda') * 0.05
sc = torch.rand(Co, device='cuda') + 0.5
sh = torch.randn(Co, device='cuda')
for _ in range(6):
    y = ops.conv2d_nhwc(x, w, sc, sh, None, True, 1, 1)
torch.cuda.synchronize()
