"""Live timing of the dominant kernel (the MFMA implicit-GEMM conv) with HIP events.

`profile_time`-style helper (the reference's mmdet/utils/profiling.py:10-40 uses CUDA events
around a block); here every `ops.conv2d_nhwc` launch of one inference pass is bracketed by a
pair of events on the stream the kernel is launched on (torch's current stream), and the
algorithmic FLOPs of each launch (2 * M * K * Cout) are summed -- achieved = sum(flops) /
sum(kernel time), the figure bench.py reports against the fp32 MFMA peak.
"""
import contextlib
import time

import torch

from . import ops

FP32_MFMA_PEAK_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 chip peak
BF16_MFMA_PEAK_TFLOPS = 2500.0    # same guide: dense bf16 v_mfma_f32_32x32x16_bf16 (16x the f32 rate)
HBM_PEAK_GBS = 8000.0
HBM_STREAM_TBS = 6.3              # what a streaming kernel sustains on this part (the guide's measured copy rate)


# ---------------------------------------------------------------------------- per-stage breakdown
# tools/analysis_tools/benchmark.py:98-131 times whole iterations only; SURVEY 8d asks for the split backbone / neck /
# RPN tower / RPN post-process / RoIAlign / FC head / NMS beside it.  The detector calls `stage_mark(name)` when a
# stage has been QUEUED; with a recorder attached a mark is one event on the current stream (or a host clock reading
# for CPU tensors), without one it is a global load and a compare.
_STAGE_REC = [None]
STAGES = ('backbone', 'neck', 'rpn_tower', 'rpn_postprocess', 'roi_align', 'fc_head', 'rcnn_decode_nms')


def stage_mark(name):
    rec = _STAGE_REC[0]
    if rec is not None:
        rec.mark(name)


class StageRecorder:
    """`with StageRecorder(cuda=True) as r: model.simple_test_device(...)`; `r.ms()` -> {stage: milliseconds}"""

    def __init__(self, cuda):
        self.cuda = bool(cuda)
        self.marks = []

    def _now(self):
        if self.cuda:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            return e
        return time.perf_counter()

    def mark(self, name):
        self.marks.append((name, self._now()))

    def __enter__(self):
        self.marks = [(None, self._now())]
        self._prev = _STAGE_REC[0]
        _STAGE_REC[0] = self
        return self

    def __exit__(self, *exc):
        _STAGE_REC[0] = self._prev
        return False

    def ms(self):
        if self.cuda:
            torch.cuda.synchronize()
        out = {}
        for (_, a), (name, b) in zip(self.marks, self.marks[1:]):
            dt = a.elapsed_time(b) if self.cuda else (b - a) * 1000.0
            out[name] = out.get(name, 0.0) + dt
        return out


def stage_breakdown(run, cuda, iters=3, tail='copy_out'):
    """per-stage milliseconds of `run()` (one pass), the minimum-total of `iters` passes; `tail`: name of a last mark set
    after `run()` returns (None: the run sets its own last mark)"""
    best = None
    for _ in range(iters):
        with StageRecorder(cuda) as r:
            run()
            if tail:
                r.mark(tail)
        ms = r.ms()
        if best is None or sum(ms.values()) < sum(best.values()):
            best = ms
    return {k: round(v, 4) for k, v in best.items()}


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
    orig_stem, orig_stem_pool = ops.stem7x7s2_nchw, ops.stem7x7s2_pool_nchw

    def stem_record(s, e, img, y, conv_rows):
        # the 7x7 stem as the GEMM it is: 147 real products per output (the padded K the kernels run is their business);
        # bytes: the image and what the launch writes (the conv output, or the pooled map of the fused form)
        records.append((s, e, 2.0 * conv_rows * 147 * y.shape[-1], img.element_size() * img.numel() + y.element_size() * y.numel(),
                        (conv_rows, y.shape[-1], 147)))

    def wrapped_stem(img, w_packed, scale=None, shift=None, relu=True):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        y = orig_stem(img, w_packed, scale, shift, relu)
        e.record()
        stem_record(s, e, img, y, y.shape[0] * y.shape[1] * y.shape[2])
        return y

    def wrapped_stem_pool(img, w_packed, scale=None, shift=None):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        y = orig_stem_pool(img, w_packed, scale, shift)
        e.record()
        ho, wo = ops.conv_out_size(img.shape[2], img.shape[3], 7, 7, 2, 3)
        stem_record(s, e, img, y, img.shape[0] * ho * wo)
        return y
    orig_tail = ops.bottleneck_tail_nhwc

    def wrapped_tail(x, w2, scale2, shift2, w3, scale3, shift3, identity):
        # the fused 3x3 (64 -> 64) + 1x1 (64 -> 256) launch: both GEMMs' products; bytes = x, weights, identity, y (the
        # 64-channel intermediate does not exist).  Shape label: the 256-column GEMM with the same product count (K = 208)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        y = orig_tail(x, w2, scale2, shift2, w3, scale3, shift3, identity)
        e.record()
        m = x.shape[0] * x.shape[1] * x.shape[2]
        nbytes = 4 * (x.numel() + w2.numel() + w3.numel() + identity.numel() + y.numel())
        records.append((s, e, 2.0 * m * (w2.numel() + w3.numel()), nbytes, (m, 256, (w2.numel() + w3.numel()) // 256)))
        return y
    ops.conv2d_nhwc = wrapped
    ops.conv2d_nhwc_multi = wrapped_multi
    ops.stem7x7s2_nchw, ops.stem7x7s2_pool_nchw = wrapped_stem, wrapped_stem_pool
    ops.bottleneck_tail_nhwc = wrapped_tail
    try:
        yield
    finally:
        ops.conv2d_nhwc = orig
        ops.conv2d_nhwc_multi = orig_multi
        ops.stem7x7s2_nchw, ops.stem7x7s2_pool_nchw = orig_stem, orig_stem_pool
        ops.bottleneck_tail_nhwc = orig_tail


def conv_stack_roofline(model, img, metas, iters=5, dtype='f32'):
    """returns the `roofline` object of the bench line for the conv/FC stack: HIP-event time of every conv / FC launch
    of one pass, the MEDIAN pass of `iters` (round 4 reported the best of three, which sat 3 % above what the rocprofv3
    kernel statistics of the same command average to; the best pass stays in the line as `frac_best`)"""
    runs = []
    for _ in range(iters):
        recs = []
        with record_conv_launches(recs), torch.no_grad():
            model.simple_test_device(img, metas, rescale=True)
        torch.cuda.synchronize()
        runs.append((sum(s.elapsed_time(e) for s, e, *_ in recs), recs))
    runs.sort(key=lambda t: t[0])
    ms_best = runs[0][0]
    ms, recs = runs[len(runs) // 2]
    flops = sum(r[2] for r in recs)
    achieved = flops / (ms * 1e-3) / 1e12
    traffic = None
    try:   # HBM bytes per launch from the committed PMC summary of the same workload
        import json
        import os
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        cands = sorted(f for f in os.listdir(os.path.join(root, 'profiles')) if f.endswith('_conv_traffic.json'))
        t = json.load(open(os.path.join(root, 'profiles', cands[-1])))
        ks = t['kernels']
        # r05 on: the aggregate over every conv / FC launch of the pass under its own key (the stored summaries of
        # earlier rounds lumped the two fp32 kernels under the name of one of them)
        traffic = (ks.get('conv_stack_all_launches') or ks['conv_igemm_f32_kernel'])['hbm_bytes_per_launch'] \
            if dtype == 'f32' else None
    except Exception:
        pass
    peak = FP32_MFMA_PEAK_TFLOPS if dtype == 'f32' else BF16_MFMA_PEAK_TFLOPS
    # two roofs per launch (as in train_conv_roofline): a launch cannot finish before max(flops / MFMA peak, algorithmic
    # bytes / 6.3 TB/s) -- x, weights, residual operand and output, each once.  The short-K 1x1 layers of stages 1 / 2 are
    # byte-bound by that measure (K = 64: 25 flop per byte against the 25 of peak / stream rate in fp32)
    bound_ms, hbm_n, hbm_ms, hbm_floor = 0.0, 0, 0.0, 0.0
    for s_, e_, fl, nb, _ in recs:
        t_mfma, t_hbm = fl / (peak * 1e12) * 1e3, nb / (HBM_STREAM_TBS * 1e12) * 1e3
        bound_ms += max(t_mfma, t_hbm)
        if t_hbm > t_mfma:
            hbm_n += 1
            hbm_ms += s_.elapsed_time(e_)
            hbm_floor += t_hbm
    return {
        'bound': 'mfma',
        'frac_of_bound': bound_ms / ms if ms else 0.0, 'bound_ms_per_pass': bound_ms, 'hbm_stream_rate_TBs': HBM_STREAM_TBS,
        'hbm_bound_launches': {'count': hbm_n, 'ms': hbm_ms, 'floor_ms': hbm_floor},
        'kernel': ('conv_pp_f32_kernel (eight-phase 256x256 / 128x256 tiles) + conv_igemm_f32_dma_kernel (64x64 tiles)'
                   if dtype == 'f32' else 'conv_pp_bf16_kernel + conv_igemm_bf16_dma_kernel') +
                  ': every conv / FC launch of one pass',
        'achieved': achieved, 'peak': peak, 'unit': 'TFLOP/s',
        'frac': achieved / peak, 'traffic': traffic,
        'traffic_unit': 'HBM bytes per launch',
        'traffic_source': 'STORED value: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of the same workload, '
                          'committed as profiles/*_conv_traffic.json (not re-measured in this run)',
        'algorithmic_bytes_per_launch': sum(r[3] for r in recs) / max(len(recs), 1),
        'launches': len(recs), 'avg_launch_us': 1000.0 * ms / max(len(recs), 1),
        'algorithmic_gflop_per_pass': flops / 1e9, 'kernel_ms_per_pass': ms,
        'passes': iters, 'kernel_ms_best_pass': ms_best, 'frac_best': flops / (ms_best * 1e-3) / 1e12 / peak,
    }


def per_layer_table(model, img, metas):
    """(shape, ms, TFLOP/s) per conv launch, for DESIGN.md / tuning"""
    recs = []
    with record_conv_launches(recs), torch.no_grad():
        model.simple_test_device(img, metas, rescale=True)
    torch.cuda.synchronize()
    return [(r[4], r[0].elapsed_time(r[1]), r[2] / (r[0].elapsed_time(r[1]) * 1e-3) / 1e12,
             r[3] / (r[0].elapsed_time(r[1]) * 1e-3) / 1e9) for r in recs]


# ---------------------------------------------------------------------------- train step: conv launches
def _out_rows(batch, hs, ws, nl, kh, kw, stride, pad):
    return sum(batch * ((hs[i] + 2 * pad - kh) // stride + 1) * ((ws[i] + 2 * pad - kw) // stride + 1) for i in range(nl))


def _in_rows(batch, hs, ws, nl):
    return sum(batch * hs[i] * ws[i] for i in range(nl))


def _esz(dt):
    return 4 if int(dt) == 0 else 2        # BRCNN_DT_F32 = 0; bf16 / f16 otherwise


# C-ABI entry -> (kind, its argument tuple -> (M, N, K) of the implicit GEMM, its argument tuple -> ALGORITHMIC bytes: every
# operand the launch has to read or write once -- activations in the compute dtype, weight gradients fp32)
_TRAIN_ENTRIES = {
    # (x,w,gamma,beta,mean,var,eps,res,z,y,batch,L,hs,ws,cin,cout,kh,kw,stride,pad,relu,dt,stream)
    'brcnn_conv2d_bn_act_nhwc_multi': (
        'forward', lambda a: (_out_rows(a[10], a[12], a[13], a[11], a[16], a[17], a[18], a[19]), a[15], a[16] * a[17] * a[14]),
        lambda a: _esz(a[21]) * (_in_rows(a[10], a[12], a[13], a[11]) * a[14] + a[15] * a[16] * a[17] * a[14] +
                                 _out_rows(a[10], a[12], a[13], a[11], a[16], a[17], a[18], a[19]) * a[15] * (2 + (1 if a[7] else 0)))),
    # (dy,w_t,z,g,b,m,v,eps,relu,dskip,prev_out,dres,dz,dg,db,ws,nb,batch,ih,iw,oh,ow,cin,cout,kh,kw,stride,pad,dt,stream)
    'brcnn_conv2d_dgrad_bn_backward_nhwc': (
        'dgrad', lambda a: (a[17] * a[18] * a[19], a[22], a[24] * a[25] * a[23]),
        lambda a: _esz(a[28]) * (a[17] * a[20] * a[21] * a[23] + a[22] * a[24] * a[25] * a[23] +
                                 a[17] * a[18] * a[19] * a[22] * (2 + sum(1 for i in (9, 10, 11) if a[i])))),
    # ... the same with the trailing `defer_second_stage` flag (round 6: the entry the train step calls)
    'brcnn_conv2d_dgrad_bn_backward_nhwc_ex': (
        'dgrad', lambda a: (a[17] * a[18] * a[19], a[22], a[24] * a[25] * a[23]),
        lambda a: _esz(a[28]) * (a[17] * a[20] * a[21] * a[23] + a[22] * a[24] * a[25] * a[23] +
                                 a[17] * a[18] * a[19] * a[22] * (2 + sum(1 for i in (9, 10, 11) if a[i])))),
    # (x,w,scale,shift,res,y,batch,L,hs,ws,cin,cout,kh,kw,stride,pad,relu,dt,stream)
    'brcnn_conv2d_nhwc_multi': (
        'forward', lambda a: (_out_rows(a[6], a[8], a[9], a[7], a[12], a[13], a[14], a[15]), a[11], a[12] * a[13] * a[10]),
        lambda a: _esz(a[17]) * (_in_rows(a[6], a[8], a[9], a[7]) * a[10] + a[11] * a[12] * a[13] * a[10] +
                                 _out_rows(a[6], a[8], a[9], a[7], a[12], a[13], a[14], a[15]) * a[11] * (1 + (1 if a[4] else 0)))),
    # (x,w,scale,shift,res,y,n,h,w,cin,cout,kh,kw,stride,pad,relu,dt,stream)
    'brcnn_conv2d_nhwc': (
        'forward', lambda a: (a[6] * ((a[7] + 2 * a[14] - a[11]) // a[13] + 1) * ((a[8] + 2 * a[14] - a[12]) // a[13] + 1),
                              a[10], a[11] * a[12] * a[9]),
        lambda a: _esz(a[16]) * (a[6] * a[7] * a[8] * a[9] + a[10] * a[11] * a[12] * a[9] +
                                 a[6] * ((a[7] + 2 * a[14] - a[11]) // a[13] + 1) * ((a[8] + 2 * a[14] - a[12]) // a[13] + 1) *
                                 a[10] * (1 + (1 if a[4] else 0)))),
    # (dy,wt,dx,batch,L,hs,ws,ohs,ows,cin,cout,kh,kw,stride,pad,dt,stream): output rows = input pixels
    'brcnn_conv2d_dgrad_nhwc_multi': (
        'dgrad', lambda a: (sum(a[3] * a[5][i] * a[6][i] for i in range(a[4])), a[9], a[11] * a[12] * a[10]),
        lambda a: _esz(a[15]) * (sum(a[3] * a[7][i] * a[8][i] for i in range(a[4])) * a[10] + a[9] * a[11] * a[12] * a[10] +
                                 sum(a[3] * a[5][i] * a[6][i] for i in range(a[4])) * a[9])),
    # (img,w,scale,shift,y,batch,h,w,cout,dt,stream): the frozen stem + max-pool launch; the GEMM = the 7x7 conv (147 real
    # products per output), bytes = the fp32 image + the pooled map
    'brcnn_stem7x7s2_pool_nchw': (
        'forward', lambda a: (a[5] * ((a[6] - 1) // 2 + 1) * ((a[7] - 1) // 2 + 1), a[8], 147),
        lambda a: 4 * a[5] * 3 * a[6] * a[7] +
        _esz(a[9]) * a[5] * ((((a[6] - 1) // 2 + 1) - 1) // 2 + 1) * ((((a[7] - 1) // 2 + 1) - 1) // 2 + 1) * a[8]),
    # (x,w2,s2,b2,w3,s3,b3,identity,y,batch,h,w,dt,stream): the fused tail of a frozen stage-1 block = a 3x3 64 -> 64 and a 1x1
    # 64 -> 256 GEMM (the 256-column GEMM with the same product count has K = 208); bytes: x, weights, identity, y
    'brcnn_bottleneck_tail_16': (
        'forward', lambda a: (a[9] * a[10] * a[11], 256, 208),
        lambda a: _esz(a[12]) * (a[9] * a[10] * a[11] * (64 + 2 * 256) + 64 * 576 + 256 * 64)),
    # (x,dy,dw,batch,L,hs,ws,cin,cout,kh,kw,stride,pad,dt,stream)
    'brcnn_conv2d_wgrad_nhwc_multi': (
        'wgrad', lambda a: (_out_rows(a[3], a[5], a[6], a[4], a[9], a[10], a[11], a[12]), a[8], a[9] * a[10] * a[7]),
        lambda a: _esz(a[13]) * (_in_rows(a[3], a[5], a[6], a[4]) * a[7] +
                                 _out_rows(a[3], a[5], a[6], a[4], a[9], a[10], a[11], a[12]) * a[8]) + 4 * a[8] * a[9] * a[10] * a[7]),
}


@contextlib.contextmanager
def record_train_conv_launches(records):
    """every conv / FC launch of the train step (forward incl. the fused conv+BN form, data gradient incl. the fused
    BatchNorm backward, weight gradient) bracketed by HIP events on the stream it is launched on; the weight-gradient
    side stream is switched off meanwhile so that one stream carries, and one event pair times, each launch.
    `records` receives (entry, kind, (M, N, K), start event, end event, algorithmic bytes).  (`brcnn_conv2d_nhwc` calls reach the device
    through `brcnn_conv2d_nhwc_multi` inside the library, not through this table: no double counting.)"""
    from . import autograd as _A
    from . import lib as _L
    lib = _L.load()
    saved_side = _A.WGRAD_SIDE_STREAM
    _A.WGRAD_SIDE_STREAM = False
    originals = {}

    def wrap(name, kind, shape_fn, bytes_fn):
        orig = getattr(lib, name)
        originals[name] = orig

        def f(*a):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            r = orig(*a)
            e.record()
            records.append((name, kind, shape_fn(a), s, e, float(bytes_fn(a))))
            return r
        setattr(lib, name, f)
    for name, (kind, fn, bfn) in _TRAIN_ENTRIES.items():
        wrap(name, kind, fn, bfn)
    try:
        yield
    finally:
        for name, orig in originals.items():
            setattr(lib, name, orig)
        _A.WGRAD_SIDE_STREAM = saved_side


def train_conv_roofline(step, dtype='bf16', iters=2):
    """the `roofline` object of the train half of the bench line: algorithmic FLOPs (2 M N K per launch: forward, data
    gradient, weight gradient of every trainable conv / FC; the frozen stem and stage 1 run forward only and have no
    gradient launches) over the HIP-event time of those launches in one step, against the dense MFMA peak of the
    compute dtype"""
    best = None
    for _ in range(iters):
        recs = []
        with record_train_conv_launches(recs):
            step()
        torch.cuda.synchronize()
        ms = sum(r[3].elapsed_time(r[4]) for r in recs)
        if best is None or ms < best[0]:
            best = (ms, recs)
    ms, recs = best
    peak = FP32_MFMA_PEAK_TFLOPS if dtype == 'f32' else BF16_MFMA_PEAK_TFLOPS
    # two roofs per launch: the MFMA peak of the dtype and the HBM stream rate a kernel can sustain (6.3 TB/s measured
    # copy rate, MI355X_MICROARCH.md; 8 TB/s is the pin rate).  A launch cannot finish before max(flops / peak, bytes /
    # rate); `frac_of_bound` = sum of those floors / measured time says how far the stack is from what the hardware
    # allows for THESE shapes, which the single MFMA fraction cannot (half of the 1x1 launches are byte-bound)
    by_kind, bound_ms, hbm_bound = {}, 0.0, [0, 0.0, 0.0]
    for _, kind, (m, n, k), s, e, nbytes in recs:
        a = by_kind.setdefault(kind, [0.0, 0.0, 0, 0.0, 0.0])
        t = s.elapsed_time(e)
        fl = 2.0 * m * n * k
        t_mfma, t_hbm = fl / (peak * 1e12) * 1e3, nbytes / (HBM_STREAM_TBS * 1e12) * 1e3
        a[0] += fl
        a[1] += t
        a[2] += 1
        a[3] += nbytes
        a[4] += max(t_mfma, t_hbm)
        bound_ms += max(t_mfma, t_hbm)
        if t_hbm > t_mfma:
            hbm_bound[0] += 1
            hbm_bound[1] += t
            hbm_bound[2] += t_hbm
    flops = sum(a[0] for a in by_kind.values())
    achieved = flops / (ms * 1e-3) / 1e12
    traffic, src = None, None
    try:
        import json
        import os
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        cands = sorted(f for f in os.listdir(os.path.join(root, 'profiles')) if f.endswith(f'_conv_traffic_train_{dtype}.json'))
        t = json.load(open(os.path.join(root, 'profiles', cands[-1])))
        traffic = t['hbm_bytes_per_launch']
        src = f'STORED value: profiles/{cands[-1]} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of the same train step; not re-measured in this run)'
    except Exception:
        pass
    return {
        'bound': 'mfma', 'kernel': f'conv_pp_bf16 / conv_igemm_bf16_dma / conv_wgrad_bf16 (every conv / FC launch of one train step, {dtype})'
        if dtype != 'f32' else 'conv_igemm_f32* / conv_wgrad_f32 (every conv / FC launch of one train step)',
        'achieved': achieved, 'peak': peak, 'unit': 'TFLOP/s', 'frac': achieved / peak,
        'traffic': traffic, 'traffic_unit': 'HBM bytes per launch', 'traffic_source': src,
        'launches': len(recs), 'avg_launch_us': 1000.0 * ms / max(len(recs), 1),
        'algorithmic_tflop_per_step': flops / 1e12, 'kernel_ms_per_step': ms,
        'frac_of_bound': bound_ms / ms if ms else 0.0, 'bound_ms_per_step': bound_ms,
        'hbm_stream_rate_TBs': HBM_STREAM_TBS,
        'hbm_bound_launches': {'count': hbm_bound[0], 'ms': hbm_bound[1], 'floor_ms': hbm_bound[2]},
        'algorithmic_gbytes_per_step': sum(a[3] for a in by_kind.values()) / 1e9,
        'by_kind': {k: {'tflop': a[0] / 1e12, 'ms': a[1], 'launches': a[2], 'tflops': a[0] / (a[1] * 1e-3) / 1e12 if a[1] else 0.0,
                        'gbytes': a[3] / 1e9, 'frac_of_bound': a[4] / a[1] if a[1] else 0.0}
                    for k, a in by_kind.items()},
        'note': 'timed with the weight-gradient side stream off (one stream, one HIP-event pair per launch); '
                'ms_per_step of the bench is measured with it on',
    }
