"""per-launch table of the conv kernels (forward / data gradient / weight gradient) inside ONE train step:
shape, time (HIP events on the launch stream), TFLOP/s.  BRCNN_DTYPE=bf16|f32|f16 (default bf16)."""
import collections, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from brcnn import profiling

dt = os.environ.get('BRCNN_DTYPE', 'bf16')
dev = torch.device('cuda', 0)
model, cfg = bench.build_model('boosting_rcnn_r50_pafpn_1x_coco.py', dev)
model = model.train()
model.set_compute_dtype(dt)
img, metas = bench.synthetic_batch(8, dev)
gtb, gtl = bench.synthetic_gt(8, dev, 80)


def step():
    model.zero_grad(set_to_none=True)
    losses = model(img=img, img_metas=metas, return_loss=True, gt_bboxes=gtb, gt_labels=gtl)
    loss, _ = model._parse_losses(losses)
    loss.backward()


recs = []
for _ in range(3):
    recs.clear()
    with profiling.record_train_conv_launches(recs):
        step()
torch.cuda.synchronize()
agg = collections.OrderedDict()
for name, kind, shp, s, e, nbytes in recs:
    k = (name.replace('brcnn_conv2d_', '').replace('_nhwc', ''), shp)
    a = agg.setdefault(k, [0, 0.0, 0.0])
    a[0] += 1
    a[1] += s.elapsed_time(e)
    a[2] += nbytes
tot = collections.defaultdict(float)
flops = collections.defaultdict(float)
rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
print(f'{dt}: conv launches of one train step (batch 8), sorted by time')
peak = profiling.FP32_MFMA_PEAK_TFLOPS if dt == 'f32' else profiling.BF16_MFMA_PEAK_TFLOPS
floor_all = 0.0
for (name, (m, n, k)), (cnt, ms, nbytes) in rows:
    fl = 2.0 * m * n * k * cnt
    tot[name] += ms
    flops[name] += fl
    # the launch's floor: MFMA peak of the dtype or the 6.3 TB/s stream rate over its algorithmic bytes, whichever is longer
    t_mfma, t_hbm = fl / (peak * 1e12) * 1e3, nbytes / (profiling.HBM_STREAM_TBS * 1e12) * 1e3
    floor_all += max(t_mfma, t_hbm)
    print(f'{name:12s} M={m:8d} N={n:5d} K={k:6d} x{cnt:2d} {ms:8.3f} ms {fl / ms / 1e9:8.1f} TF/s {nbytes / ms / 1e9:6.2f} TB/s '
          f'bound={"hbm " if t_hbm > t_mfma else "mfma"} {max(t_mfma, t_hbm) / ms:5.2f} of it')
for name in tot:
    print(f'TOTAL {name:12s} {tot[name]:8.3f} ms  {flops[name] / tot[name] / 1e9:8.1f} TF/s')
print(f'ALL {sum(tot.values()):.3f} ms {sum(flops.values()) / sum(tot.values()) / 1e9:.1f} TF/s; sum of the per-launch floors '
      f'{floor_all:.3f} ms = {floor_all / sum(tot.values()):.3f} of the measured time')
