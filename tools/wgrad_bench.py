import sys, os, torch, ctypes
sys.path.insert(0, os.getcwd())
import brcnn
from brcnn import ops, lib
L = lib.load()
BF = len(sys.argv) > 1 and sys.argv[1] == 'bf16'
if len(sys.argv) > 2: L.brcnn_conv_set_tile_wgrad_bf16(int(sys.argv[2]))
shapes = [('rpn_l0 3x3 256->256', 8,100,168,256,256,3,1,1), ('s3 3x3 256->256', 8,50,84,256,256,3,1,1),
          ('s2 1x1 128->512', 8,100,168,128,512,1,1,0), ('s3 1x1 1024->256', 8,50,84,1024,256,1,1,0),
          ('s4 3x3 512->512', 8,25,42,512,512,3,1,1), ('fc 12544->1024', 4096,1,1,12544,1024,1,1,0),
          ('s2 3x3 128->128', 8,100,168,128,128,3,1,1)]
def bench(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    s,e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e)/n
for name,N,H,W,Ci,Co,k,st,pd in shapes:
    Ho,Wo = ops.conv_out_size(H,W,k,k,st,pd)
    x = torch.randn(N*H*W,Ci,device='cuda'); dy = torch.randn(N*Ho*Wo,Co,device='cuda')
    if BF: x, dy = x.bfloat16(), dy.bfloat16()
    dw = torch.zeros(Co,k,k,Ci,device='cuda')
    hs=(ctypes.c_int*1)(H); ws=(ctypes.c_int*1)(W)
    fl = 2.0*N*Ho*Wo*Co*k*k*Ci
    ms = bench(lambda: L.brcnn_conv2d_wgrad_nhwc_multi(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), N,1,hs,ws,Ci,Co,k,k,st,pd,1 if BF else 0,None))
    print(f'{name:24s} M={N*Ho*Wo:7d} {ms*1000:8.1f} us {fl/ms/1e9:6.1f} TF')
