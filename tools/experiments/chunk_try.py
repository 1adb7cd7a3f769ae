"""Experiment: do the HBM-bound early stages (stem, max-pool, layer1, layer2) run faster when the batch goes through
them in image chunks whose activations fit the 256 MiB Infinity Cache?  BRCNN_DTYPE=f32|bf16."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from brcnn import ops
from brcnn.blocks import conv_bn_act_nhwc, to_nhwc

dt = os.environ.get('BRCNN_DTYPE', 'f32')
dev = torch.device('cuda', 0)
model, cfg = bench.build_model('boosting_rcnn_r50_pafpn_1x_utdac.py', dev)
model = model.eval()
model.set_compute_dtype(dt)
bb = model.backbone
img, metas = bench.synthetic_batch(8, dev)
x = to_nhwc(img)
if dt != 'f32':
    x = x.to(torch.bfloat16 if dt == 'bf16' else torch.float16)


def front(xc, upto):
    y = conv_bn_act_nhwc(xc, bb.conv1, bb.bn1, bb._stem_cache, True)
    y = ops.maxpool3x3s2_nhwc(y)
    for name in bb.res_layers[:upto]:
        y = getattr(bb, name).forward_nhwc(y)
    return y


def timed(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n


with torch.no_grad():
    for upto in (1, 2, 3):
        ref = front(x, upto)
        for chunk in (8, 4, 2, 1):
            def run():
                return torch.cat([front(x[i:i + chunk], upto) for i in range(0, 8, chunk)], 0) if chunk < 8 else front(x, upto)
            out = run()
            same = torch.equal(out, ref)
            print(f'{dt} stem..layer{upto}  chunk {chunk}: {timed(run):7.3f} ms  identical={same}')
