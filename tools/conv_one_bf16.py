import sys, os, torch
sys.path.insert(0, os.getcwd())
import brcnn
from brcnn import ops, lib
N,H,W,Ci,Co,k = 8,100,168,256,256,3
t = int(os.environ.get('BF16_TILE', '0'))
lib.load().brcnn_conv_set_tile_bf16(t)
x = torch.randn(N,H,W,Ci,device='cuda').bfloat16(); w = (torch.randn(Co,k,k,Ci,device='cuda')*0.05).bfloat16()
sc = torch.rand(Co,device='cuda')+0.5; sh = torch.randn(Co,device='cuda')
for _ in range(3):
    y = ops.conv2d_nhwc(x,w,sc,sh,None,True,1,1)
torch.cuda.synchronize()
