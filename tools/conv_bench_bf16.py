"""per-layer-shape timing of the bf16 implicit-GEMM tile variants (run on the GPU box)"""
import sys, os, torch
sys.path.insert(0, os.getcwd())
import brcnn
from brcnn import ops, lib
L = lib.load()
BF = torch.bfloat16
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
 ('s2 1x1 512->128', 8,100,168,512,128,1,1,0,False),
 ('s3 1x1 512->1024 (ds)', 8,50,84,512,1024,1,1,0,False),
 ('p4 3x3 256->256 M8400', 8,25,42,256,256,3,1,1,False),
 ('s4 dgrad 3x3 512->512 M33600', 8,50,84,512,512,3,1,1,False),
 ('ideal 3x3 256->256 M131072 (512 tiles of 256x256)', 8,128,128,256,256,3,1,1,False),
 ('tower-eq 3x3 256->256 M179200', 8,140,160,256,256,3,1,1,False),
 ('ideal 1x1 1024->1024 M65536', 8,64,128,1024,1024,1,1,0,False),
 ('ideal 1x1 256->1024 M65536 +res', 8,64,128,256,1024,1,1,0,True),
]
if os.environ.get('SHAPES'):
    keep = os.environ['SHAPES'].split(',')
    shapes = [s_ for s_ in shapes if any(k_ in s_[0] for k_ in keep)]
# tile ids as brcnn_conv_set_tile_bf16 takes them; suffix 's' = stream-K schedule forced, 'a' = heuristic, none = off;
# 'n' = without the 256 x 128 two-group kernel (8842 forces it, 8844 forces the 256 x 256 eight-phase kernel)
tiles = sys.argv[1].split(',') if len(sys.argv) > 1 else ['11', '21', '22', '0']
if len(sys.argv) > 2:      # 'il0' / 'il1': LDS-DMA pieces in front of / spread between the MFMA groups
    L.brcnn_conv_set_tile_bf16(-1 if sys.argv[2] == 'il1' else -2)
def bench(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s,e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e)/n
for name,N,H,W,Ci,Co,k,st,pd,res in shapes:
    x = torch.randn(N,H,W,Ci,device='cuda').to(BF); w = (torch.randn(Co,k,k,Ci,device='cuda')*0.05).to(BF)
    sc = torch.rand(Co,device='cuda')+0.5; sh = torch.randn(Co,device='cuda')
    Ho,Wo = ops.conv_out_size(H,W,k,k,st,pd)
    r = torch.randn(N,Ho,Wo,Co,device='cuda').to(BF) if res else None
    fl = 2.0*N*Ho*Wo*Co*k*k*Ci
    by = 2.0*(x.numel() + w.numel() + N*Ho*Wo*Co*(2 if res else 1))
    out = []
    for ts in tiles:
        t = int(ts.rstrip('san'))
        if L.brcnn_conv_set_tile_bf16(t) != 0:
            continue
        L.brcnn_conv_set_tile_bf16(-18 if 'n' in ts else -19)          # 'n': the 256 x 128 two-group kernel off (round 5's heuristic)
        L.brcnn_conv_set_tile_bf16(-5 if 's' in ts else -4 if 'a' in ts else -3)
        ms = bench(lambda: ops.conv2d_nhwc(x,w,sc,sh,r,True,st,pd))
        out.append(f'{ts:>5s}: {ms*1000:6.1f} us {fl/ms/1e9:6.1f} TF')
    L.brcnn_conv_set_tile_bf16(0)
    L.brcnn_conv_set_tile_bf16(-4)
    L.brcnn_conv_set_tile_bf16(-19)
    print(f'{name:24s} M={N*Ho*Wo:7d} hbm-floor {by/8e12*1e6:5.1f} us | ' + ' | '.join(out))
