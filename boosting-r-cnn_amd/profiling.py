"""Live timing of the dominant kernel (the MFMA implicit-GEMM conv) with HIP events.

`profile_time`-style helper (the reference's mmdet/utils/profiling.py:10-40 uses CUDA events
around a block); here every `ops.conv2d_nhwc` launch of one inference pass is bracketed by a
pair of events on the stream the kernel is launched on (torch's current stream), and the
algorithmic FLOPs of each launch (2 * M * K * Cout) are summed -- achieved = sum(flops) /
sum(kernel time), the figure bench.py reports against the fp32 MFMA peak.
"""
import contextlib

import torch

from . import ops

FP32_MFMA_PEAK_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 chip peak
BF16_MFMA_PEAK_TFLOPS = 2500.0    # same guide: dense bf16 v_mfma_f32_32x32x16_bf16 (16x the f32 rate)
HBM_PEAK_GBS = 8000.0


@contextlib.contextmanager
def record_conv_launches(records):
    orig = ops.conv2d_nhwc

    def wrapped(x, w, scale=None, shift=None, residual=None, relu=False, stride=1, pad=0, out_f32=False):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        y = orig(x, w, scale, shift, residual, relu, stride, pad, out_f32)
        e.record()
        m = y.shape[0] * y.shape[1] * y.shape[2]
        k = w.shape[1] * w.shape[2] * w.shape[3]
        nbytes = x.element_size() * (x.numel() + w.numel() + (residual.numel() if residual is not None else 0)) + \
            y.element_size() * y.numel()
        records.append((s, e, 2.0 * m * k * w.shape[0], nbytes, (m, w.shape[0], k)))
        return y
    orig_multi = ops.conv2d_nhwc_multi

    def wrapped_multi(x_cat, w, batch, sizes, scale=None, shift=None, residual=None, relu=False,
                      stride=1, pad=0, out_f32=False):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        y, osz = orig_multi(x_cat, w, batch, sizes, scale, shift, residual, relu, stride, pad, out_f32)
        e.record()
        k = w.shape[1] * w.shape[2] * w.shape[3]
        nbytes = x_cat.element_size() * (x_cat.numel() + w.numel()) + y.element_size() * y.numel()
        records.append((s, e, 2.0 * y.shape[0] * k * w.shape[0], nbytes, (y.shape[0], w.shape[0], k)))
        return y, osz
    ops.conv2d_nhwc = wrapped
    ops.conv2d_nhwc_multi = wrapped_multi
    try:
        yield
    finally:
        ops.conv2d_nhwc = orig
        ops.conv2d_nhwc_multi = orig_multi


def conv_stack_roofline(model, img, metas, iters=3, dtype='f32'):
    """returns the `roofline` object of the bench line for the conv/FC stack"""
    best = None
    for _ in range(iters):
        recs = []
        with record_conv_launches(recs), torch.no_grad():
            model.simple_test_device(img, metas, rescale=True)
        torch.cuda.synchronize()
        ms = sum(s.elapsed_time(e) for s, e, *_ in recs)
        if best is None or ms < best[0]:
            best = (ms, recs)
    ms, recs = best
    flops = sum(r[2] for r in recs)
    achieved = flops / (ms * 1e-3) / 1e12
    traffic = None
    try:   # HBM bytes per launch from the committed PMC summary of the same workload
        import json
        import os
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        cands = sorted(f for f in os.listdir(os.path.join(root, 'profiles')) if f.endswith('_conv_traffic.json'))
        t = json.load(open(os.path.join(root, 'profiles', cands[-1])))
        traffic = t['kernels']['conv_igemm_f32_kernel' if dtype == 'f32' else
                               'conv_igemm_bf16_dma_kernel']['hbm_bytes_per_launch']
    except Exception:
        pass
    peak = FP32_MFMA_PEAK_TFLOPS if dtype == 'f32' else BF16_MFMA_PEAK_TFLOPS
    return {
        'bound': 'mfma', 'kernel': f'conv_igemm_{dtype}*_kernel (all conv/FC launches of one pass)',
        'achieved': achieved, 'peak': peak, 'unit': 'TFLOP/s',
        'frac': achieved / peak, 'traffic': traffic,
        'traffic_unit': 'HBM bytes per launch',
        'traffic_source': 'STORED value: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of the same workload, '
                          'committed as profiles/*_conv_traffic.json (not re-measured in this run)',
        'algorithmic_bytes_per_launch': sum(r[3] for r in recs) / max(len(recs), 1),
        'launches': len(recs), 'avg_launch_us': 1000.0 * ms / max(len(recs), 1),
        'algorithmic_gflop_per_pass': flops / 1e9, 'kernel_ms_per_pass': ms,
    }


def per_layer_table(model, img, metas):
    """(shape, ms, TFLOP/s) per conv launch, for DESIGN.md / tuning"""
    recs = []
    with record_conv_launches(recs), torch.no_grad():
        model.simple_test_device(img, metas, rescale=True)
    torch.cuda.synchronize()
    return [(r[4], r[0].elapsed_time(r[1]), r[2] / (r[0].elapsed_time(r[1]) * 1e-3) / 1e12,
             r[3] / (r[0].elapsed_time(r[1]) * 1e-3) / 1e9) for r in recs]
