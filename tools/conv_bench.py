import sys, os, torch, time
sys.path.insert(0, os.getcwd())
import brcnn
from brcnn import ops, lib
L = lib.load()
shapes = [  # name, N,H,W,Cin,Cout,k,stride,pad,res
 ('rpn_l0 3x3 256->256', 8,100,168,256,256,3,1,1,False),
 ('s3 3x3 256->256 M33600', 8,50,84,256,256,3,1,1,False),
 ('s1 1x1 64->256 +res', 8,200,336,64,256,1,1,0,True),
 ('s1 3x3 64->64', 8,200,336,64,64,3,1,1,False),
 ('s1 1x1 256->64', 8,200,336,256,64,1,1,0,False),
 ('s2 1x1 128->512 +res', 8,100,168,128,512,1,1,0,True),
 ('s2 3x3 128->128', 8,100,168,128,128,3,1,1,False),
 ('s3 1x1 1024->256', 8,50,84,1024,256,1,1,0,False),
 ('s3 1x1 256->1024 +res', 8,50,84,256,1024,1,1,0,True),
 ('s4 3x3 512->512', 8,25,42,512,512,3,1,1,False),
 ('fc 12544->1024', 2048,1,1,12544,1024,1,1,0,False),
 ('s4 1x1 512->2048 +res', 8,25,42,512,2048,1,1,0,True),
 ('s4 1x1 2048->512', 8,25,42,2048,512,1,1,0,False),
 ('s3 3x3 s2 256->256', 8,100,168,256,256,3,2,1,False),
]
def bench(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s,e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e)/n
for name,N,H,W,Ci,Co,k,st,pd,res in shapes:
    x = torch.randn(N,H,W,Ci,device='cuda'); w = torch.randn(Co,k,k,Ci,device='cuda')*0.05
    sc = torch.rand(Co,device='cuda')+0.5; sh = torch.randn(Co,device='cuda')
    Ho,Wo = ops.conv_out_size(H,W,k,k,st,pd)
    r = torch.randn(N,Ho,Wo,Co,device='cuda') if res else None
    fl = 2.0*N*Ho*Wo*Co*k*k*Ci
    out = []
    for dma, nt, wm in ((0, 1, 0), (2, 2, 2), (2, 1, 2), (1, 0, 0)):
        L.brcnn_conv_set_tile(-1, dma)
        L.brcnn_conv_set_tile(wm, nt)
        ms = bench(lambda: ops.conv2d_nhwc(x,w,sc,sh,r,True,st,pd))
        out.append(f'{("reg","auto","dma")[dma]}{nt if nt else ""}{"" if wm == 0 else "m%d" % wm}: {ms*1000:7.1f} us {fl/ms/1e9:6.1f} TF')
    L.brcnn_conv_set_tile(-1, 1)
    L.brcnn_conv_set_tile(0, 0)
    print(f'{name:28s} M={N*Ho*Wo:7d} ' + ' | '.join(out))
