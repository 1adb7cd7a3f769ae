"""one conv layer, a few launches (for rocprofv3 --pmc runs): BF16_TILE picks the tile id, CONV_SHAPE =
N,H,W,Cin,Cout,k the layer (default: the 3x3 256->256 layer on the 100x168 map), CONV_DT = bf16 (default) / f16 / f32,
CONV_RES=1 adds a residual operand"""
import sys, os, torch
sys.path.insert(0, os.getcwd())
import brcnn
from brcnn import ops, lib
N,H,W,Ci,Co,k = [int(v) for v in os.environ.get('CONV_SHAPE', '8,100,168,256,256,3').split(',')]
t = int(os.environ.get('BF16_TILE', '0'))
assert lib.load().brcnn_conv_set_tile_bf16(t) == 0
dt = {'bf16': torch.bfloat16, 'f16': torch.float16, 'f32': torch.float32}[os.environ.get('CONV_DT', 'bf16')]
x = torch.randn(N,H,W,Ci,device='cuda').to(dt); w = (torch.randn(Co,k,k,Ci,device='cuda')*0.05).to(dt)
res = torch.randn(N,H,W,Co,device='cuda').to(dt) if os.environ.get('CONV_RES') == '1' else None
sc = torch.rand(Co,device='cuda')+0.5; sh = torch.randn(Co,device='cuda')
for _ in range(3):
    y = ops.conv2d_nhwc(x,w,sc,sh,res,True,1,k//2)
torch.cuda.synchronize()
