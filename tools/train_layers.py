"""per-launch table of the conv kernels (forward / data gradient / weight gradient) inside ONE train step:
shape, time (HIP events on the launch stream), TFLOP/s.  BRCNN_DTYPE=bf16|f32|f16 (default bf16)."""
import collections, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from brcnn import lib as L
from brcnn import autograd as _A
_A.WGRAD_SIDE_STREAM = False      # per-launch HIP events on ONE stream: keep the weight-gradient launches on it

dt = os.environ.get('BRCNN_DTYPE', 'bf16')
dev = torch.device('cuda', 0)
model, cfg = bench.build_model('boosting_rcnn_r50_pafpn_1x_coco.py', dev)
model = model.train()
model.set_compute_dtype(dt)
img, metas = bench.synthetic_batch(8, dev)
gtb, gtl = bench.synthetic_gt(8, dev, 80)
lib = L.load()
recs = []


def wrap(name, shape_fn):
    orig = getattr(lib, name)

    def f(*a):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        r = orig(*a)
        e.record()
        recs.append((name.replace('brcnn_conv2d_', '').replace('_nhwc', ''), shape_fn(a), s, e))
        return r
    setattr(lib, name, f)


def multi_shape(a):      # (x,w,scale,shift,res,y,batch,L,hs,ws,cin,cout,kh,kw,stride,pad,relu,dt,stream)
    batch, nl, hs, ws, cin, cout, kh, kw, stride, pad = a[6], a[7], a[8], a[9], a[10], a[11], a[12], a[13], a[14], a[15]
    m = sum(batch * ((hs[i] + 2 * pad - kh) // stride + 1) * ((ws[i] + 2 * pad - kw) // stride + 1) for i in range(nl))
    return (m, cout, kh * kw * cin)


def single_shape(a):     # (x,w,scale,shift,res,y,n,h,w,cin,cout,kh,kw,stride,pad,relu,dt,stream)
    n, h, w, cin, cout, kh, kw, stride, pad = a[6:15]
    return (n * ((h + 2 * pad - kh) // stride + 1) * ((w + 2 * pad - kw) // stride + 1), cout, kh * kw * cin)


def dgrad_shape(a):      # (dy,wt,dx,batch,L,hs,ws,ohs,ows,cin,cout,kh,kw,stride,pad,dt,stream): output rows = input pixels
    batch, nl, hs, ws, cin, cout, kh, kw = a[3], a[4], a[5], a[6], a[9], a[10], a[11], a[12]
    return (sum(batch * hs[i] * ws[i] for i in range(nl)), cin, kh * kw * cout)


def wgrad_shape(a):      # (x,dy,dw,batch,L,hs,ws,cin,cout,kh,kw,stride,pad,dt,stream)
    batch, nl, hs, ws, cin, cout, kh, kw, stride, pad = a[3], a[4], a[5], a[6], a[7], a[8], a[9], a[10], a[11], a[12]
    m = sum(batch * ((hs[i] + 2 * pad - kh) // stride + 1) * ((ws[i] + 2 * pad - kw) // stride + 1) for i in range(nl))
    return (m, cout, kh * kw * cin)


def fused_fwd_shape(a):  # (x,w,gamma,beta,mean,var,eps,res,z,y,batch,L,hs,ws,cin,cout,kh,kw,stride,pad,relu,dt,stream)
    batch, nl, hs, ws, cin, cout, kh, kw, stride, pad = a[10], a[11], a[12], a[13], a[14], a[15], a[16], a[17], a[18], a[19]
    m = sum(batch * ((hs[i] + 2 * pad - kh) // stride + 1) * ((ws[i] + 2 * pad - kw) // stride + 1) for i in range(nl))
    return (m, cout, kh * kw * cin)


def fused_dgrad_shape(a):  # (dy,w_t,z,g,b,m,v,eps,relu,dskip,prev_out,dres,dz,dg,db,ws,nb,batch,ih,iw,oh,ow,cin,cout,kh,kw,...)
    batch, ih, iw, cin, cout, kh, kw = a[17], a[18], a[19], a[22], a[23], a[24], a[25]
    return (batch * ih * iw, cin, kh * kw * cout)


wrap('brcnn_conv2d_bn_act_nhwc_multi', fused_fwd_shape)
wrap('brcnn_conv2d_dgrad_bn_backward_nhwc', fused_dgrad_shape)
wrap('brcnn_conv2d_nhwc_multi', multi_shape)
wrap('brcnn_conv2d_nhwc', single_shape)
wrap('brcnn_conv2d_dgrad_nhwc_multi', dgrad_shape)
wrap('brcnn_conv2d_wgrad_nhwc_multi', wgrad_shape)


def step():
    model.zero_grad(set_to_none=True)
    losses = model(img=img, img_metas=metas, return_loss=True, gt_bboxes=gtb, gt_labels=gtl)
    loss, _ = model._parse_losses(losses)
    loss.backward()


for _ in range(3):
    recs.clear()
    step()
torch.cuda.synchronize()
agg = collections.OrderedDict()
for name, shp, s, e in recs:
    k = (name, shp)
    t = s.elapsed_time(e)
    a = agg.setdefault(k, [0, 0.0])
    a[0] += 1
    a[1] += t
tot = collections.defaultdict(float)
flops = collections.defaultdict(float)
rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
print(f'{dt}: conv launches of one train step (batch 8), sorted by time')
for (name, (m, n, k)), (cnt, ms) in rows:
    fl = 2.0 * m * n * k * cnt
    tot[name] += ms
    flops[name] += fl
    print(f'{name:12s} M={m:8d} N={n:5d} K={k:6d} x{cnt:2d} {ms:8.3f} ms {fl / ms / 1e9:8.1f} TF/s')
for name in tot:
    print(f'TOTAL {name:12s} {tot[name]:8.3f} ms  {flops[name] / tot[name] / 1e9:8.1f} TF/s')
print(f'ALL {sum(tot.values()):.3f} ms {sum(flops.values()) / sum(tot.values()) / 1e9:.1f} TF/s')
