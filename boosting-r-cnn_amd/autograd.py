"""Differentiable wrappers of the HIP convolution / linear / RoI-extraction kernels.

The reference trains through torch autograd over cuDNN/cuBLAS and mmcv's
`RoIAlignFunction`; here the forward, data-gradient and weight-gradient of every trainable
conv / FC are the MFMA implicit-GEMM kernels (`conv_igemm.hip`, `conv_wgrad.hip`) and the RoI
feature gradient is `brcnn_roi_extract_backward`.  Cheap element-wise steps of the training
graph (eval-BN affine, ReLU, adds) stay ordinary differentiable torch ops.
"""
import ctypes

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import lib as _L
from . import ops
from .ops import DT_F32, DT_BF16, _dt, _ptr, _require_gpu, _stream, _conv_stream, conv_out_size


def _ints(vals):
    return (ctypes.c_int * len(vals))(*[int(v) for v in vals])


import contextlib
import os as _os
_ARENA_OFF = _os.environ.get('BRCNN_GRAD_ARENA', '1') == '0'


class _GradArena:
    """zero-initialised fp32 storage for the weight-gradient kernels (they accumulate with atomics): ONE fill
    launch per train step instead of one per layer.  `take` hands out disjoint views of the current chunk; a
    chunk is sized by what the previous step used (`new_step`, called by optim.FusedSGD.step) and simply
    replaced when it runs out -- the views keep their chunk alive for as long as a gradient refers to it."""

    MIN_CHUNK = 16 << 20        # floats

    def __init__(self):
        self.buf, self.off, self.used, self.hint = None, 0, 0, 0
        self.listener = None        # distributed.GradReducer: all-reduces the arena in place, slice by slice
        self._chunks = []           # [(chunk, elements handed out)] of the running step (listener only)
        self._last = (0, 0)         # element range of the view handed out last

    def take(self, shape, device):
        if _ARENA_OFF:
            return torch.zeros(tuple(shape), dtype=torch.float32, device=device)
        n = 1
        for v in shape:
            n *= int(v)
        n_al = (n + 63) // 64 * 64
        if self.buf is None or self.buf.device != device or self.off + n_al > self.buf.numel():
            self.buf = torch.zeros(max(n_al, self.hint - self.used, self.MIN_CHUNK), dtype=torch.float32, device=device)
            self.off = 0
            if self.listener is not None:
                self._chunks.append([self.buf, 0])
                self.listener.chunk_opened(self.buf)
        v = self.buf[self.off:self.off + n].view(shape)
        self._last = (self.off, self.off + n_al)
        self.off += n_al
        self.used += n_al
        if self.listener is not None:
            self._chunks[-1][1] = self.off
        return v

    def launched(self, stream, in_place=True, param=None):
        """the kernel that writes the view handed out last has been issued on `stream`.  `in_place`: autograd takes
        that view as the parameter's `.grad` as it is (nobody copies it on the main stream afterwards) -- only such
        ranges may be all-reduced in place while the backward pass is still running (distributed.GradReducer)"""
        if self.listener is not None and self.buf is not None:
            self.listener.writers_launched(self.buf, self._last[0], self._last[1], stream, in_place, param)

    def used_of(self, buf):
        for b, n in self._chunks:
            if b is buf:
                return n
        return 0

    def new_step(self):
        self.hint, self.used, self.buf = max(self.hint, self.used), 0, None
        self._chunks = []


grad_arena = _GradArena()


# BRCNN_TEST_WGRAD_STALL_CYCLES=n (tests/ddp_worker.py): with a gradient reducer attached, the main stream idles n
# clock cycles after every weight-gradient launch -- whatever autograd queues there next (its copies of dW included)
# then runs AFTER an overlapped all-reduce of the same arena slice had the time to complete, the order that made the
# round-4 two-rank mismatch (profiles/r05_notes.md)
_TEST_STALL_CYCLES = int(_os.environ.get('BRCNN_TEST_WGRAD_STALL_CYCLES', '0'))

# weight-gradient launches on a second HIP stream (see _conv_backward); BRCNN_WGRAD_STREAM=0 keeps one stream
WGRAD_SIDE_STREAM = _os.environ.get('BRCNN_WGRAD_STREAM', '1') != '0'
# weight gradients of layers without a data gradient (the trunk's entry layers, last in the backward pass) stay on the main
# stream (see _conv_backward); BRCNN_WGRAD_TAIL_MAIN=0 sends them to the second stream like the others
WGRAD_TAIL_ON_MAIN = _os.environ.get('BRCNN_WGRAD_TAIL_MAIN', '1') != '0'
# BRCNN_WGRAD_DEFER=1: the second stages (slab reductions) of the 16-bit weight-gradient launches on the second stream
# are batched into one table-driven launch per WGRAD_DEFER_ITEMS layers and at the end-of-pass join
# (csrc/wgrad_defer.hip): 85 launches less per bf16 step of bench.py, same bits.  OFF by default: on one MI355X the step
# is 0.15-0.2 ms LONGER with it (17.96 -> 18.15 ms; the small per-layer reductions fill gaps of the main stream's kernels,
# a batched one competes with them for HBM and its last instance sits in front of the join), and with every launch made
# dearer for the host (under rocprofv3) it does not gain either (19.8 -> 20.1 ms): profiles/r05_notes.md.  Not used while
# a gradient reducer slices the arena during the backward pass (it would have to hear about a range after its flush).
WGRAD_DEFER = _os.environ.get('BRCNN_WGRAD_DEFER', '0') == '1'
WGRAD_DEFER_BYTES = int(_os.environ.get('BRCNN_WGRAD_DEFER_MB', '1024')) << 20
WGRAD_DEFER_ITEMS = int(_os.environ.get('BRCNN_WGRAD_DEFER_ITEMS', '16'))
_defer_arenas = {}       # side stream handle -> the slab arena handed to brcnn_wgrad_defer_begin (kept alive here)
_side_streams = {}
_join_queued = {}        # (device type, index) -> True while a join callback of the running backward pass is queued
_side_seen = {}          # (device type, index) -> {id: tensor} of the weights whose gradient went to the side stream in this pass
                         # (the tensor is kept alive: the id of a freed per-call view would pass for the next one's)


def _wgrad_side_stream(device):
    """the second stream of the weight-gradient launches, or None: off, or under DistributedDataParallel (its
    bucket all-reduce reads a gradient as soon as autograd has accumulated it, on the main stream)"""
    if not WGRAD_SIDE_STREAM:
        return None
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and not _OWN_REDUCER[0]:   # DDP hooks the gradient accumulators
        return None
    key = (device.type, device.index)
    if key not in _side_streams:
        # BRCNN_WGRAD_PRIO: HIP stream priority of the weight-gradient stream (default: the device's default priority)
        prio = _os.environ.get('BRCNN_WGRAD_PRIO')
        _side_streams[key] = torch.cuda.Stream(device) if prio is None else torch.cuda.Stream(device, priority=int(prio))
    return _side_streams[key]


def _defer_on(side):
    """deferral of the slab reductions on `side` (set up on first use); False: off"""
    h = side.cuda_stream
    if not WGRAD_DEFER or grad_arena.listener is not None:
        if h in _defer_arenas:              # (switched off, or a reducer attached since: back to the per-layer form)
            flush_deferred(side)
            _L.check(_L.load().brcnn_wgrad_defer_begin(h, None, 0, 0), 'brcnn_wgrad_defer_begin')
            del _defer_arenas[h]
        return False
    if h not in _defer_arenas:
        arena = torch.empty(WGRAD_DEFER_BYTES, dtype=torch.uint8, device=side.device)
        with torch.cuda.device(side.device):
            st = _L.load().brcnn_wgrad_defer_begin(h, arena.data_ptr(), arena.numel(), WGRAD_DEFER_ITEMS)
        _L.check(st, 'brcnn_wgrad_defer_begin')
        _defer_arenas[h] = arena
    return True


def set_wgrad_defer(on):
    """switch the deferral at run time (tests, A/B measurements); off: pending reductions are launched, the arenas released"""
    global WGRAD_DEFER
    WGRAD_DEFER = bool(on)
    if not on:
        for side in list(_side_streams.values()):
            _defer_on(side)


def flush_deferred(side):
    """launch the batched slab reduction of the weight-gradient launches pending on `side`"""
    h = side.cuda_stream
    if h not in _defer_arenas:
        return
    with torch.cuda.device(side.device):
        st = _L.load().brcnn_wgrad_defer_flush(h)
    if st < 0:
        _L.check(st, 'brcnn_wgrad_defer_flush')


# parallel.GradReducer reduces the gradients after the end-of-backward join (no accumulator hooks): the side stream
# stays on under torch.distributed while one is active
_OWN_REDUCER = [False]


def _side_stream_for(param, device):
    """the side stream for `param`'s gradient launch, or None.  A parameter that feeds several Function nodes of one
    backward pass (a conv shared across pyramid levels in the per-level fallback) gets the side stream for none but
    its first use: autograd sums the uses on the main stream as soon as the second one returns, so the second use
    first makes the main stream wait for the side stream and then runs on the main stream itself."""
    side = _wgrad_side_stream(device)
    if side is None:
        return None
    key = (device.type, device.index)
    seen = _side_seen.setdefault(key, {})
    if id(param) in seen:
        if grad_arena.listener is not None and hasattr(grad_arena.listener, 'shared_parameter'):
            grad_arena.listener.shared_parameter(param)
        flush_deferred(side)
        torch.cuda.current_stream(device).wait_stream(side)
        return None
    seen[id(param)] = param
    return side


def join_side_streams(device=None):
    """make the current stream(s) wait for the side stream(s) and forget the per-pass state: the end-of-backward
    callback, and a backstop for the readers of .grad (optim.FusedSGD.step, the gradient reducer) in case a
    backward pass raised before its callbacks ran"""
    flush_bn_deferred()
    release_held_weight_gradients()
    for key, side in list(_side_streams.items()):
        if device is not None and key != (device.type, device.index):
            continue
        flush_deferred(side)
        torch.cuda.current_stream(torch.device(*key)).wait_stream(side)
        _join_queued[key] = False
        _side_seen.pop(key, None)


_DEFER_JOIN = [False]
# weight-gradient launches held back until the end of a partial backward pass (held_weight_gradients)
_HELD_WGRADS = [None]


_HELD_PENDING = []      # held launches whose context ended with release=False (release_held_weight_gradients)


@contextlib.contextmanager
def held_weight_gradients(enabled=True, release=True):
    """weight-gradient launches of the backward passes inside this context are issued on the second stream only when the
    context ends (`release`) or at the caller's `release_held_weight_gradients()`, behind everything the main stream has
    queued by then (detectors.py, early_rpn_backward: the four tower launches -- M = 179 200, 250 us each -- otherwise run
    under the tower's own data-gradient / GroupNorm chain, which is bandwidth-bound and takes twice its stand-alone time
    beside them).  Not with a gradient reducer attached (it hears about an arena range when its writer is ISSUED)."""
    if not enabled or _HELD_WGRADS[0] is not None or grad_arena.listener is not None:
        yield
        return
    _HELD_WGRADS[0] = []
    try:
        yield
    finally:
        held, _HELD_WGRADS[0] = _HELD_WGRADS[0], None
        _HELD_PENDING.extend(held)
        if release:
            release_held_weight_gradients()


def release_held_weight_gradients():
    """issue the launches collected by held_weight_gradients(release=False); also part of every side-stream join"""
    if not _HELD_PENDING:
        return
    held = list(_HELD_PENDING)
    del _HELD_PENDING[:]
    main = torch.cuda.current_stream(held[0][2])
    ev = main.record_event()
    for side, launch, _ in held:
        side.wait_event(ev)
        launch()


@contextlib.contextmanager
def deferred_side_stream_join():
    """backward passes inside this context do not queue the end-of-pass join: for a partial backward pass in the
    middle of a step (detectors.py, early_rpn_backward) whose results nobody reads before the step's final backward
    pass has joined the side stream (or the readers' backstop join has: FusedSGD.step, GradReducer.reduce)"""
    prev = _DEFER_JOIN[0]
    _DEFER_JOIN[0] = True
    try:
        yield
    finally:
        _DEFER_JOIN[0] = prev


def _queue_stream_join(main, side):
    """main waits for the side stream once, when the running backward pass ends (whoever reads .grad afterwards --
    optimizer, gradient clipping, GradScaler -- is on the main stream)"""
    key = (main.device.type, main.device.index)
    if _join_queued.get(key) or _DEFER_JOIN[0]:
        return

    def join():
        _join_queued[key] = False
        _side_seen.pop(key, None)
        release_held_weight_gradients()
        flush_deferred(side)
        main.wait_stream(side)
    _join_queued[key] = True
    try:
        torch.autograd.Variable._execution_engine.queue_callback(join)
    except RuntimeError:        # not inside a backward pass (a direct call of the Function's backward)
        join()


# ---- deferred BatchNorm second stages -----------------------------------------------------------------------------
# The backward of every trainable eval-mode BatchNorm ends with a small launch that reduces per-strip partial sums to
# dgamma / dbeta (~40 per bf16 train step, 16 - 128 workgroups each, serial on the main stream: 0.29 ms).  Nothing reads
# the results before the optimizer, so the library records them (`*_ex(..., defer_second_stage=1)`) and runs all of them in
# one launch (`brcnn_bn_reduce_flush`) when the backward pass ends -- and again, as a backstop, wherever the side streams
# are joined (FusedSGD.step, GradReducer.reduce).  Only where autograd takes dgamma / dbeta as `.grad` AS THEY ARE (leaf,
# no gradient yet, no hooks, fp32, no DistributedDataParallel): anything else reads them right away.
BN_REDUCE_DEFER = _os.environ.get('BRCNN_BN_REDUCE_DEFER', '1') != '0'
_bn_keep = {}           # stream handle -> tensors the recorded stages read / write (alive until the flush)
_bn_flush_queued = {}


def _bn_defer_ok(*params):
    if not BN_REDUCE_DEFER:
        return False
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and not _OWN_REDUCER[0]:
        return False
    for p in params:
        if p is None or not p.is_leaf or p.grad is not None or p.dtype != torch.float32 or p._backward_hooks or \
                getattr(p, '_post_accumulate_grad_hooks', None):
            return False
    return True


def _bn_deferred(handle, device, *keep):
    """a second stage was recorded on stream `handle`: keep its operands alive and flush when the backward pass ends"""
    _bn_keep.setdefault(handle, []).extend(k for k in keep if k is not None)
    if _bn_flush_queued.get(handle):
        return

    def flush():
        _bn_flush_queued[handle] = False
        flush_bn_deferred(handle, device)
    _bn_flush_queued[handle] = True
    try:
        torch.autograd.Variable._execution_engine.queue_callback(flush)
    except RuntimeError:        # not inside a backward pass (a direct call of the Function's backward)
        flush()


def flush_bn_deferred(handle=None, device=None):
    """run the recorded BatchNorm second stages (of stream `handle`, default: of every stream that has any)"""
    for h in ([handle] if handle is not None else list(_bn_keep.keys())):
        if h not in _bn_keep:
            continue
        st = _L.load().brcnn_bn_reduce_flush(h)
        if st < 0:
            _L.check(st, 'brcnn_bn_reduce_flush')
        del _bn_keep[h]


def _conv_operands(weight, x_cat):
    """(forward operand (Cout,KH,KW,Cin), data-gradient operand (Cin,KH,KW,Cout) or None) of `weight` in the
    activation dtype: the ones the fused optimizer step already wrote for this version of the weight
    (optim.FusedSGD), else one packing launch"""
    pk = getattr(weight, '_brcnn_pack', None)
    if pk is not None and pk[0] == weight._version and pk[1] == x_cat.dtype and pk[2].device == x_cat.device:
        return pk[2], pk[3]
    wsrc = weight.detach()
    if wsrc.dtype != torch.float32 or not wsrc.is_contiguous():
        wsrc = wsrc.float().contiguous()
    cout_, cin_, kh_, kw_ = wsrc.shape
    w_p = torch.empty((cout_, kh_, kw_, cin_), dtype=x_cat.dtype, device=x_cat.device)
    w_t = torch.empty((cin_, kh_, kw_, cout_), dtype=x_cat.dtype, device=x_cat.device) \
        if x_cat.requires_grad else None
    st = _L.load().brcnn_pack_conv_weights(_ptr(wsrc), _ptr(w_p), _ptr(w_t), cout_, cin_, kh_, kw_,
                                           _dt(x_cat), _stream())
    _L.check(st, 'brcnn_pack_conv_weights')
    return w_p, w_t


_GEOM = {}


def _conv_backward(x_cat, weight, w_t_saved, dy, cfg, dskip, need_dx, need_dw, has_dgrad=None):
    """data / weight gradient of ConvNHWCFunction's forward (dy already in the activation dtype, contiguous);
    returns (dx, dw, dskip not yet added)"""
    batch, sizes, out_sizes, stride, pad = cfg
    cout, cin, kh, kw = weight.shape
    dt = _dt(x_cat)
    lib = _L.load()
    L = len(sizes)
    if has_dgrad is None:           # (the caller that computes the data gradient in a launch of its own says so)
        has_dgrad = need_dx
    geo = _GEOM.get((sizes, out_sizes))            # (the ctypes arrays of a geometry are built once: ~60 layers x 4 per step)
    if geo is None:
        if len(_GEOM) > 512:
            _GEOM.clear()
        geo = _GEOM[(sizes, out_sizes)] = (_ints([h for h, _ in sizes]), _ints([w for _, w in sizes]),
                                           _ints([h for h, _ in out_sizes]), _ints([w for _, w in out_sizes]))
    hs, ws, ohs, ows = geo
    dx = dw = None
    # (fp32 only: in bf16 the zero-stuffed MFMA pass is cheaper than four more launches and
    # their weight slices -- measured 54.1 vs 57.6 ms per train step)
    if need_dx and stride == 2 and L == 1 and x_cat.dtype == torch.float32 and \
            (kh, kw, pad) in ((3, 3, 1), (1, 1, 0)) and cout % 32 == 0:
        dx = _dgrad_stride2(dy, weight, batch, sizes[0], out_sizes[0], kh, pad, x_cat.dtype)
    elif need_dx and dskip is not None and stride == 1 and L == 1 and kh == kw and \
            2 * pad == kh - 1 and w_t_saved is not None and dskip.dtype == x_cat.dtype:
        # data gradient + the identity branch's gradient in one epilogue (the forward kernel on
        # the flipped / transposed weights with a residual operand)
        (h, w_), = sizes
        dx = ops.conv2d_nhwc(dy.view(batch, h, w_, cout), w_t_saved, None, None,
                             dskip.contiguous().view(batch, h, w_, cin), False, 1, pad).view(batch * h * w_, cin)
        dskip = None
    elif need_dx:
        w_t = w_t_saved
        if w_t is None:
            w_t = weight.detach().float().flip(2, 3).permute(1, 2, 3, 0).to(x_cat.dtype).contiguous()   # (Cin,KH,KW,Cout)
        dx = torch.empty_like(x_cat)
        st = lib.brcnn_conv2d_dgrad_nhwc_multi(_ptr(dy), _ptr(w_t), _ptr(dx), batch, L, hs, ws, ohs,
                                               ows, cin, cout, kh, kw, stride, pad, dt, _conv_stream())
        _L.check(st, 'brcnn_conv2d_dgrad_nhwc_multi')
    if need_dw:
        dwp = grad_arena.take((cout, kh, kw, cin), dy.device)
        # a second stream only when autograd will take the result as `weight.grad` as it is (a leaf parameter
        # without a gradient yet, in the layout the kernel writes): anything else -- a slice / cat / permute backward,
        # an accumulation into an existing .grad -- is a main-stream kernel that would read dW before it is complete
        takes = weight.is_leaf and weight.grad is None and not weight._backward_hooks and \
            not getattr(weight, '_post_accumulate_grad_hooks', None) and \
            (weight.is_contiguous() if kh * kw == 1 else weight.is_contiguous(memory_format=torch.channels_last))
        # ... or when the ONE consumer of dW is a node that does its work on the second stream itself
        # (PermutedWeightFunction: the first FC's re-layout): the launch may leave the main stream, but the arena range is
        # not `.grad` -- the gradient reducer must not slice it in place (`takes` stays False for it)
        side_ok = takes or bool(getattr(weight, '_brcnn_dw_consumer_on_side', False))
        # ... and only when this layer HAS a data gradient for the launch to hide behind: the first trainable layers of the
        # trunk (their input comes from the frozen stage: no dx) are the last nodes of the backward pass -- the main
        # stream has nothing left to do, the second stream still has its backlog, and the launch would only lengthen the
        # tail the main stream waits for at the end-of-pass join (0.34 ms per bf16 step of bench.py: tools/experiments/
        # join_wait.py); on the main stream it runs beside that backlog
        # (the weight is recorded as seen in EVERY case: a later use of the same parameter in this pass must find it --
        # a tail launch kept on the main stream and never recorded let a second use with a data gradient pass for the
        # first one and go to the second stream, unordered with the sum autograd forms on the main stream: ADVICE r05)
        side = _side_stream_for(weight, dy.device) if side_ok else None
        if side is not None and not (has_dgrad or not WGRAD_TAIL_ON_MAIN):
            side = None
        deferred = False
        if side is None:
            st = lib.brcnn_conv2d_wgrad_nhwc_multi(_ptr(x_cat), _ptr(dy), _ptr(dwp), batch, L, hs, ws,
                                                   cin, cout, kh, kw, stride, pad, dt, _conv_stream())
        else:
            # nothing in the backward pass waits for dW: the weight-gradient launch goes to a second HIP stream and
            # overlaps the data-gradient chain of the layers above (its atomics tail and the other kernel's
            # LDS-DMA / MFMA phases fill each other's gaps); the streams join at the end of the backward pass
            main = torch.cuda.current_stream(dy.device)
            dy.record_stream(side)                          # the allocator must not recycle them under the launch
            x_cat.record_stream(side)
            if _HELD_WGRADS[0] is not None and grad_arena.listener is None and \
                    (takes or getattr(weight, '_brcnn_dw_consumer_on_side', None) == 'views'):
                # issued when the enclosing held_weight_gradients() context ends (operands kept alive by the closure)
                def launch(x_cat=x_cat, dy=dy, dwp=dwp, side=side):
                    st_ = lib.brcnn_conv2d_wgrad_nhwc_multi(_ptr(x_cat), _ptr(dy), _ptr(dwp), batch, L, hs, ws,
                                                            cin, cout, kh, kw, stride, pad, dt, _L.stream_handle(side))
                    _L.check(st_, 'brcnn_conv2d_wgrad_nhwc_multi')
                _HELD_WGRADS[0].append((side, launch, dy.device))
                st = 0
            else:
                side.wait_event(main.record_event())        # dy, x and the zero fill of dW are complete here
                deferred = x_cat.dtype != torch.float32 and _defer_on(side)
                st = lib.brcnn_conv2d_wgrad_nhwc_multi(_ptr(x_cat), _ptr(dy), _ptr(dwp), batch, L, hs, ws,
                                                       cin, cout, kh, kw, stride, pad, dt, _L.stream_handle(side))
            _queue_stream_join(main, side)
        _L.check(st, 'brcnn_conv2d_wgrad_nhwc_multi')
        # `deferred`: the slab reduction of this launch may be pending (wgrad_defer.hip).  A consumer on the second
        # stream (`takes` False: PermutedWeightFunction) reads dW right behind this call: reduce now.  Otherwise dW is
        # `.grad`, read after the end-of-pass join, which flushes.
        if deferred and not takes:
            flush_deferred(side)
        if grad_arena.listener is not None and dy.is_cuda:
            # `takes`: dW is `.grad` itself.  Otherwise autograd (or the caller) COPIES it on the main stream after
            # this function returns (AccumulateGrad's clone of a gradient whose strides are not the parameter's, the
            # backward of a cat / pad / permute in front of the weight) and the copy is what gets reduced: such a
            # range must not be all-reduced in place under that copy
            grad_arena.launched(side if side is not None else torch.cuda.current_stream(dy.device), bool(takes), weight)
            if _TEST_STALL_CYCLES:      # test switch: hold the main stream back behind every weight-gradient launch
                torch.cuda._sleep(_TEST_STALL_CYCLES)
        # (Cout,KH,KW,Cin) -> the parameter's (Cout,Cin,KH,KW): for 1x1 filters the two coincide in
        # memory (a plain view with the parameter's own strides, what DDP's bucket views expect)
        dw = dwp.view(cout, cin, 1, 1) if kh == 1 and kw == 1 else dwp.permute(0, 3, 1, 2)
    return dx, dw, dskip


class ConvNHWCFunction(Function):
    """y = conv(x, w) + b over one or several NHWC segments.

    x_cat (rows, Cin) [rows = sum_l batch*H_l*W_l], weight (Cout,Cin,KH,KW) in the reference's
    parameter layout, bias (Cout) or None.  Returns y_cat (rows_out, Cout)."""

    @staticmethod
    def forward(ctx, x_cat, weight, bias, batch, sizes, stride, pad, with_skip=False, relu=False, out_f32=False):
        """`with_skip`: also return an alias of the input (the identity branch of a residual block);
        its gradient then arrives in THIS backward and is added in the data-gradient kernel's
        epilogue instead of by a separate autograd add over the whole tensor.  `relu`: ReLU in the kernel's epilogue (the
        backward masks dy with the saved output: one launch instead of clamp + threshold_backward around the node).
        `out_f32`: a 16-bit conv writes its result in fp32 (head outputs: no widening copy behind the launch)."""
        _require_gpu(x_cat, weight, bias)
        # fp32 activations: exact-fp32 MFMA; bf16 activations: bf16 MFMA with fp32 accumulation
        # (weights are cast per step from the fp32 master copy, gradients of weights stay fp32)
        # one launch writes the forward operand and (when the input needs a gradient) the flipped /
        # transposed data-gradient operand
        w_p, w_t = _conv_operands(weight, x_cat)
        ctx.w_t = w_t
        x_cat = x_cat.contiguous()
        y, out_sizes = ops.conv2d_nhwc_multi(x_cat, w_p, batch, sizes, None,
                                             bias.detach().float().contiguous() if bias is not None else None,
                                             None, bool(relu), stride, pad,
                                             out_f32=bool(out_f32) and x_cat.dtype != torch.float32)
        if relu:
            ctx.save_for_backward(x_cat, weight, y)
        else:
            ctx.save_for_backward(x_cat, weight)
        ctx.relu = bool(relu)
        ctx.cfg = (batch, tuple(sizes), tuple(out_sizes), stride, pad, bias is not None)
        ctx.bias_ref = bias
        ctx.with_skip = bool(with_skip)
        if with_skip:
            return y, x_cat.view_as(x_cat)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy, dskip=None):
        if ctx.relu:
            x_cat, weight, y = ctx.saved_tensors
            dy = torch.ops.aten.threshold_backward(dy.to(y.dtype), y, 0)
        else:
            x_cat, weight = ctx.saved_tensors
        batch, sizes, out_sizes, stride, pad, has_bias = ctx.cfg
        dy = dy.to(x_cat.dtype).contiguous()
        dx, dw, dskip = _conv_backward(x_cat, weight, ctx.w_t, dy, (batch, sizes, out_sizes, stride, pad), dskip,
                                       ctx.needs_input_grad[0], ctx.needs_input_grad[1])
        db = None
        if has_bias and ctx.needs_input_grad[2]:
            b = ctx.bias_ref
            # the bias gradient's two small launches leave the main chain like the weight gradient (same rule:
            # only where autograd takes the result as `bias.grad` unchanged)
            takes = b is not None and b.is_leaf and b.grad is None and not b._backward_hooks and \
                not getattr(b, '_post_accumulate_grad_hooks', None) and b.dtype == torch.float32
            side = _side_stream_for(b, dy.device) if takes else None
            if side is None:
                db = ops.colsum(dy)
            else:
                main = torch.cuda.current_stream(dy.device)
                side.wait_event(main.record_event())
                with torch.cuda.stream(side):
                    db = ops.colsum(dy)
                dy.record_stream(side)
                db.record_stream(main)
                _queue_stream_join(main, side)
        if dskip is not None:
            dx = dskip if dx is None else dx + dskip
        return dx, dw, db, None, None, None, None, None, None, None


class GroupNormNHWCFunction(Function):
    """GroupNorm(+ReLU) over one or several NHWC segments on the HIP kernels, forward and backward
    (torch's native group norm wants NCHW: two layout copies per layer and direction)."""

    @staticmethod
    def forward(ctx, x_cat, gamma, beta, groups, batch, sizes, eps, relu):
        _require_gpu(x_cat, gamma, beta)
        x_cat = x_cat.contiguous()
        g32, b32 = gamma.detach().float().contiguous(), beta.detach().float().contiguous()
        y, stats = ops.groupnorm_nhwc_multi(x_cat, g32, b32, groups, batch, sizes, eps, relu, return_stats=True)
        ctx.save_for_backward(x_cat, stats, g32, b32)
        ctx.cfg = (groups, batch, tuple(sizes), bool(relu), gamma.dtype, beta.dtype)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x_cat, stats, g32, b32 = ctx.saved_tensors
        groups, batch, sizes, relu, gdt, bdt = ctx.cfg
        dx, dg, db = ops.groupnorm_nhwc_multi_backward(dy.to(x_cat.dtype).contiguous(), x_cat, stats, g32, b32,
                                                       groups, batch, sizes, relu)
        return dx, dg.to(gdt), db.to(bdt), None, None, None, None, None


def groupnorm_nhwc_autograd(x, gamma, beta, groups, eps, relu):
    """x (N,H,W,C) -> GroupNorm(+ReLU), differentiable"""
    n, h, w, c = x.shape
    y = GroupNormNHWCFunction.apply(x.reshape(n * h * w, c), gamma, beta, groups, n, ((h, w),), eps, relu)
    return y.view(n, h, w, c)


def _dgrad_stride2(dy, weight, batch, in_size, out_size, k, pad, dtype):
    """Data gradient of a stride-2 convolution (3x3 / pad 1 or 1x1 / pad 0) by output parity.

    dx[2a+ph, 2b+pw] only receives the taps with kh = ph+1 (mod 2), kw likewise, so each of the
    four parity classes is a small stride-1 convolution over dy (1x1, 1x2, 2x1, 2x2 taps; the 1x1
    stride-2 layer has the single class (0,0)): 9 taps in total instead of the 36 of the same
    kernel run over the zero-stuffed dy.  Each class runs on the forward MFMA kernel, whose
    epilogue scatters the rows straight into the class's pixels of dx
    (brcnn_conv2d_nhwc_scatter2)."""
    H, W = in_size
    Ho, Wo = out_size
    cout, cin = weight.shape[0], weight.shape[1]
    w = weight.detach().float()
    lib = _L.load()
    dt = {torch.float32: DT_F32, torch.bfloat16: DT_BF16, torch.float16: ops.DT_F16}[dtype]
    if k == 1:
        dx = torch.zeros((batch * H * W, cin), dtype=dtype, device=dy.device)
        classes = [(0, 0, w.permute(1, 2, 3, 0), 0, 0)]                       # (ph, pw, taps, pad, origin)
    else:
        dx = torch.empty((batch * H * W, cin), dtype=dtype, device=dy.device)
        # kernel rows feeding even / odd input rows, in dy-offset order: [1] and [2, 0]
        # (slices + flip: list indexing would upload an index tensor, i.e. a host sync, per use)
        wf = w.flip(2, 3)
        rows = (w[:, :, 1:2], wf[:, :, 0::2])
        rows_f = (wf[:, :, 1:2], wf[:, :, 0::2])     # same rows of the column-flipped kernel
        classes = []
        for ph in range(2):
            for pw in range(2):
                src = rows[ph] if pw == 0 else rows_f[ph]
                sub = src[:, :, :, 1:2] if pw == 0 else src[:, :, :, 0::2]
                classes.append((ph, pw, sub.permute(1, 2, 3, 0), 1, 1))
    for ph, pw, wt, cpad, origin in classes:
        wt = wt.to(dtype).contiguous()                                         # (Cin, KH', KW', Cout)
        st = lib.brcnn_conv2d_nhwc_scatter2(_ptr(dy), _ptr(wt), _ptr(dx), batch, Ho, Wo, cout, cin,
                                            wt.shape[1], wt.shape[2], cpad, H, W, ph, pw, origin, dt, _conv_stream())
        _L.check(st, 'brcnn_conv2d_nhwc_scatter2')
    return dx


def _pad_cout(weight, bias, mult=32):
    """zero-pad the output channels to a multiple of `mult` (differentiable): keeps the
    gradient kernels on their vector-load paths (dY rows 16-byte aligned, dgrad Cin % 32 == 0)"""
    cout = weight.shape[0]
    extra = (-cout) % mult
    if extra == 0:
        return weight, bias, cout
    weight = torch.cat([weight, weight.new_zeros((extra,) + tuple(weight.shape[1:]))], 0)
    if bias is not None:
        bias = torch.cat([bias, bias.new_zeros(extra)], 0)
    return weight, bias, cout


def conv2d_nhwc_autograd(x, weight, bias, stride, pad, with_skip=False, out_f32=False):
    """x (N,H,W,Cin) -> (N,Ho,Wo,Cout), differentiable.  `with_skip`: returns (y, x_alias); gradients
    reaching x_alias (a residual block's identity branch) are added inside the conv's data-gradient
    kernel.  `out_f32`: fp32 result from 16-bit operands (head outputs)"""
    n, h, w, cin = x.shape
    kh, kw = weight.shape[2], weight.shape[3]
    ho, wo = conv_out_size(h, w, kh, kw, stride, pad)
    weight, bias, cout = _pad_cout(weight, bias, 32 if x.dtype == torch.float32 else 64)
    y = ConvNHWCFunction.apply(x.reshape(n * h * w, cin), weight, bias, n, ((h, w),), stride, pad, with_skip, False, bool(out_f32))
    skip = None
    if with_skip:
        y, skip = y
        skip = skip.view(n, h, w, cin)
    y = y.view(n, ho, wo, weight.shape[0])
    y = y if cout == weight.shape[0] else y[..., :cout]
    return (y, skip) if with_skip else y


def conv2d_nhwc_multi_autograd(x_cat, weight, bias, batch, sizes, stride, pad, out_f32=False):
    """several NHWC maps that share one set of weights (pyramid levels), concatenated as
    (rows, Cin) -> (rows_out, Cout): one forward / dgrad / wgrad launch for all of them; `out_f32`: fp32 result from
    16-bit operands (head outputs)"""
    weight, bias, cout = _pad_cout(weight, bias, 32 if x_cat.dtype == torch.float32 else 64)
    y = ConvNHWCFunction.apply(x_cat, weight, bias, batch, tuple(sizes), stride, pad, False, False, bool(out_f32))
    return y if cout == weight.shape[0] else y[:, :cout]


class CatRowsAliased(Function):
    """`torch.cat([f.reshape(-1, C) for f in feats], 0)` for maps that already LIE back to back in one buffer (the neck
    wrote them there: ops.output_into): the result is a view of that memory, no launch; backward hands each map its rows
    of the gradient (views)."""

    @staticmethod
    def forward(ctx, *feats):
        f0 = feats[0]
        c = f0.shape[-1]
        rows = [f.numel() // c for f in feats]
        ctx.shapes = [tuple(f.shape) for f in feats]
        ctx.rows = rows
        out = f0.new_empty(0).set_(f0.untyped_storage(), f0.storage_offset(), (sum(rows), c), (c, 1))
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        outs, r0 = [], 0
        for shp, n in zip(ctx.shapes, ctx.rows):
            outs.append(g[r0:r0 + n].view(shp))
            r0 += n
        return tuple(outs)


def cat_rows(feats):
    """(sum rows, C) of NHWC maps, level-major: a view when the maps are adjacent slices of one buffer, else a cat"""
    f0 = feats[0]
    c = f0.shape[-1]
    adjacent = f0.is_cuda and all(f.is_contiguous() and f.dtype == f0.dtype and f.shape[-1] == c for f in feats)
    if adjacent:
        base, off = f0.untyped_storage().data_ptr(), f0.storage_offset()
        for f in feats:
            if f.untyped_storage().data_ptr() != base or f.storage_offset() != off:
                adjacent = False
                break
            off += f.numel()
    if not adjacent or len(feats) == 1:
        return torch.cat([f.reshape(-1, c) for f in feats], 0)
    if torch.is_grad_enabled() and any(f.requires_grad for f in feats):
        return CatRowsAliased.apply(*feats)
    rows = sum(f.numel() // c for f in feats)
    return f0.new_empty(0).set_(f0.untyped_storage(), f0.storage_offset(), (rows, c), (c, 1))


class SplitColumns(Function):
    """y (rows, padded) -> the column ranges [0, n0), [n0, n0 + n1), ... as contiguous fp32 tensors (the fused head
    output: cls | reg [| pad]); backward writes the parts' gradients into ONE (rows, padded) tensor of y's dtype -- where
    slicing + `.float()` + `.contiguous()` cost a copy per part forward and a zero fill, a copy and an add per part backward"""

    @staticmethod
    def forward(ctx, y, *widths):
        ctx.meta = (tuple(y.shape), y.dtype, tuple(int(w) for w in widths))
        outs, c0 = [], 0
        for w in widths:
            outs.append(y[:, c0:c0 + w].float().contiguous() if y.dtype != torch.float32 else y[:, c0:c0 + w].contiguous())
            c0 += w
        return tuple(outs)

    @staticmethod
    @once_differentiable
    def backward(ctx, *grads):
        shape, dtype, widths = ctx.meta
        dy = torch.empty(shape, dtype=dtype, device=next(g for g in grads if g is not None).device)
        c0 = 0
        for w, g in zip(widths, grads):
            if g is None:
                dy[:, c0:c0 + w].zero_()
            else:
                dy[:, c0:c0 + w].copy_(g)
            c0 += w
        if c0 < shape[1]:
            dy[:, c0:].zero_()
        return (dy,) + (None,) * len(widths)


class FusedHeadWeights(Function):
    """Several conv heads that read the same input as ONE conv: the weights (C_i, Cin, KH, KW) and biases (C_i,) of the
    heads concatenated along the output channels and zero-padded to a multiple of `mult` (the RPN's cls | reg | iou heads:
    9 + 36 + 9 -> 64 channels).  Forward: two `cat` launches.  Backward: the gradients of the parts are VIEWS of the fused
    gradient -- no kernel -- so the fused conv's weight-gradient launch may run on the second stream like any other
    (`_brcnn_dw_consumer_on_side`; with torch.cat in the graph its slicing backward sat on the main stream and kept the
    launch -- 0.24 ms at 179 200 rows -- on the critical path of the early RPN backward pass).  A part whose parameter is
    not laid out like the kernel's output gets its copy on the second stream, behind the launch."""

    @staticmethod
    def forward(ctx, mult, n, *wb):
        ws, bs = wb[:n], wb[n:]
        couts = [int(w.shape[0]) for w in ws]
        total = sum(couts)
        extra = (-total) % mult
        parts = list(ws)
        if extra:
            parts.append(ws[0].new_zeros((extra,) + tuple(ws[0].shape[1:])))
        w = torch.cat(parts, 0)
        b = None
        if bs and bs[0] is not None:
            bparts = list(bs) + ([bs[0].new_zeros(extra)] if extra else [])
            b = torch.cat(bparts, 0)
        ctx.couts = couts
        ctx.n = n
        ctx.w_strides = [tuple(w_.stride()) for w_ in ws]
        return w, b

    @staticmethod
    @once_differentiable
    def backward(ctx, gw, gb):
        outs_w, outs_b, r0 = [], [], 0
        side = _wgrad_side_stream(gw.device) if (gw is not None and gw.is_cuda) else None
        copies = []
        for c, st in zip(ctx.couts, ctx.w_strides):
            g = gw[r0:r0 + c] if gw is not None else None
            if g is not None and side is not None and tuple(g.stride()) != st:
                # autograd would clone this part on the MAIN stream into the parameter's layout, under the launch that is
                # still writing it on the second stream: make that copy there, behind the launch
                copies.append(len(outs_w))
            outs_w.append(g)
            outs_b.append(gb[r0:r0 + c] if gb is not None else None)
            r0 += c
        if copies:
            main = torch.cuda.current_stream(gw.device)
            side.wait_event(main.record_event())
            with torch.cuda.stream(side):
                for i in copies:
                    src = outs_w[i]
                    dst = torch.empty_strided(src.shape, ctx.w_strides[i], dtype=src.dtype, device=src.device)
                    dst.copy_(src)
                    dst.record_stream(main)
                    outs_w[i] = dst
            gw.record_stream(side)
            _queue_stream_join(main, side)
        return (None, None) + tuple(outs_w) + tuple(outs_b)


def fused_head_weights(heads, mult):
    """(weight, bias) of `heads` (nn.Conv2d modules sharing input and geometry) fused along the output channels, padded
    to a multiple of `mult`, differentiable; feed to ConvNHWCFunction"""
    ws = [h.weight for h in heads]
    bs = [h.bias for h in heads]
    w, b = FusedHeadWeights.apply(mult, len(ws), *(ws + bs))
    if not w.is_cuda:
        return w, b
    if w.requires_grad and _wgrad_side_stream(w.device) is not None and not _HELD_COPIES_UNSAFE(ws, w):
        w._brcnn_dw_consumer_on_side = 'views'        # (no kernel of the consumer reads dW: the launch may also be HELD)
    elif w.requires_grad and _wgrad_side_stream(w.device) is not None:
        w._brcnn_dw_consumer_on_side = True
    return w, b


def _HELD_COPIES_UNSAFE(ws, w):
    """a part that needs a layout copy behind the launch (its parameter is not stored in the kernel's (Cout, KH, KW, Cin)
    order, i.e. not channels-last): the copy is queued when the backward node runs, so the launch must not be held back
    past it (held_weight_gradients)"""
    for p in ws:
        if p.dim() == 2:            # (linear weights: dW rows are the parameter's rows)
            if not p.is_contiguous():
                return True
            continue
        _, cin, kh, kw = p.shape
        if tuple(p.stride()) != (kh * kw * cin, 1, kw * cin, cin) and not (kh == 1 and kw == 1 and p.is_contiguous()):
            return True
    return False


FC0_ON_SIDE = _os.environ.get('BRCNN_FC0_SIDE', '1') != '0'      # A/B switch of PermutedWeightFunction


class PermutedWeightFunction(Function):
    """The first FC of the box head: the parameter (out, C*ph*pw) -- the reference's column order -- as the (out, ph*pw*C)
    matrix the NHWC RoI features multiply.  Forward: one copy.  Backward: the weight-gradient kernel writes dW in the
    (ph, pw, C) order on the second stream; the copy back to the parameter's order runs THERE, behind it, and its result
    is what autograd hands to the parameter as `.grad` (contiguous, the parameter's layout: no further kernel) -- where
    autograd's own view / permute backward put the weight-gradient launch (0.13 ms at 4096 RoIs) and a 51 MB copy on the
    main stream at the very start of the backward pass."""

    @staticmethod
    def forward(ctx, w2d, c, ph, pw, lazy=False):
        """`lazy`: the caller holds the permuted OPERANDS of this version of the weight already (optim.FusedSGD wrote
        them in its step: `_brcnn_pack_perm`) and nothing will read the permuted fp32 values -- the result is a
        zero-stride placeholder of the right shape, no copy (51 MB read + 51 MB written per step at 1024 x 12544)"""
        ctx.cfg = (c, ph, pw)
        out = w2d.shape[0]
        if lazy:
            return w2d.new_empty((1,)).expand(out, c * ph * pw)
        return w2d.view(out, c, ph, pw).permute(0, 2, 3, 1).reshape(out, -1)

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        c, ph, pw = ctx.cfg
        out = g.shape[0]
        side = _wgrad_side_stream(g.device) if g.is_cuda else None
        if side is None:            # no second stream (switched off, or DistributedDataParallel's hooks): the plain copy
            return g.reshape(out, ph, pw, c).permute(0, 3, 1, 2).reshape(out, -1), None, None, None, None
        main = torch.cuda.current_stream(g.device)
        side.wait_event(main.record_event())        # (the launch that wrote g ran on `side` or, failing that, on `main`)
        with torch.cuda.stream(side):
            r = g.reshape(out, ph, pw, c).permute(0, 3, 1, 2).reshape(out, -1)
        g.record_stream(side)
        r.record_stream(main)                       # (read on the main stream after the end-of-pass join)
        _queue_stream_join(main, side)
        return r, None, None, None, None


def permuted_fc_weight(w2d, c, ph, pw, dtype=None):
    """(out, C*ph*pw) -> (out, ph*pw*C), differentiable; the result may be fed to `linear_autograd` only (its
    weight-gradient launch then goes to the second stream).  `dtype`: the dtype of the activations it will multiply --
    when the fused optimizer has already written the permuted operands of this version of the weight in that dtype,
    the fp32 permutation is skipped altogether (PermutedWeightFunction, `lazy`)."""
    pk = getattr(w2d, '_brcnn_pack_perm', None)
    lazy = (pk is not None and dtype is not None and pk[0] == w2d._version and pk[1] == dtype and w2d.requires_grad and
            pk[2].device == w2d.device and w2d.shape[0] % (32 if dtype == torch.float32 else 64) == 0)
    if not FC0_ON_SIDE and not lazy:
        out = w2d.shape[0]
        return w2d.view(out, c, ph, pw).permute(0, 2, 3, 1).reshape(out, -1)
    w = PermutedWeightFunction.apply(w2d, c, ph, pw, lazy)
    if w.requires_grad and FC0_ON_SIDE:
        w._brcnn_dw_consumer_on_side = True
    if lazy:
        w._brcnn_pack = (w._version, pk[1], pk[2], pk[3])
    return w


def linear_autograd(x, weight, bias, relu=False, out_f32=False):
    """[relu](x (M,K) @ weight(N,K)^T + bias), differentiable (the 1x1 case with H=W=1); `out_f32`: fp32 result from
    16-bit operands"""
    on_side = getattr(weight, '_brcnn_dw_consumer_on_side', False)
    pk = getattr(weight, '_brcnn_pack', None)       # operands the fused optimizer (or permuted_fc_weight) holds for it
    src = weight
    weight, bias, cout = _pad_cout(weight, bias, 32 if x.dtype == torch.float32 else 64)
    w4 = weight.view(weight.shape[0], weight.shape[1], 1, 1)
    if on_side and cout == weight.shape[0]:         # (no padding cat in between: dW travels back through views only)
        w4._brcnn_dw_consumer_on_side = True
    if pk is not None and weight is src and pk[0] == src._version and pk[1] == x.dtype and pk[2].device == x.device:
        w4._brcnn_pack = (w4._version, pk[1], pk[2], pk[3])
    y = ConvNHWCFunction.apply(x.contiguous(), w4, bias, x.shape[0], ((1, 1),), 1, 0, False, bool(relu), bool(out_f32))
    return y if cout == weight.shape[0] else y[:, :cout]


ROI_BACKWARD_GATHER = True      # False: the scatter form (fp32 atomics), kept for A/B and as the generic fallback
ROI_GATHER_ADDEND = _os.environ.get('BRCNN_ROI_ADDEND', '1') != '0'     # (A/B switch: see roi_extract_autograd)


class RoIExtractFunction(Function):
    """Fused SingleRoIExtractor on NHWC maps with the RoIAlign feature gradient."""

    @staticmethod
    def forward(ctx, rois, output_size, strides, finest_scale, sampling_ratio, addends, *feats):
        """`addends`: None, or per level a tensor (or None) that the backward adds to that level's gradient inside
        the gather launch (`brcnn_roi_extract_backward_gather_add`; see `inject_gradients`)"""
        out, _ = ops.roi_extract(list(feats), rois, output_size, strides, finest_scale, sampling_ratio)
        ctx.save_for_backward(rois)
        ctx.cfg = (output_size, tuple(strides), finest_scale, sampling_ratio,
                   [tuple(f.shape) for f in feats], feats[0].dtype)
        ctx.addends = addends
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_out):
        (rois,) = ctx.saved_tensors
        output_size, strides, finest_scale, sampling_ratio, shapes, fdt = ctx.cfg
        ph, pw = (output_size, output_size) if isinstance(output_size, int) else output_size
        L = len(shapes)
        hs, ws = _ints([s[1] for s in shapes]), _ints([s[2] for s in shapes])
        sc = (ctypes.c_float * L)(*[1.0 / s for s in strides])
        lib = _L.load()
        if ROI_BACKWARD_GATHER and ph <= 7 and pw <= 7 and shapes[0][3] % 4 == 0:
            # gather form: every gradient pixel written once, no zero fill, deterministic; a 16-bit pyramid gets
            # its gradient in its own dtype (fp32 accumulation, one rounding at the store)
            g = grad_out.to(fdt).contiguous()
            grads = [torch.empty(s, dtype=fdt, device=grad_out.device) for s in shapes]
            ptrs = (ctypes.c_void_p * L)(*[t.data_ptr() for t in grads])
            nb = lib.brcnn_roi_extract_backward_workspace_bytes_ex(rois.size(0), shapes[0][0], shapes[0][3], L, hs, ws)
            wsp = torch.empty((nb + 3) // 4, dtype=torch.int32, device=grad_out.device)
            adds = ctx.addends
            ctx.addends = None
            aptrs = None
            if adds is not None and any(a is not None for a in adds):
                adds = [a.contiguous() if a is not None else None for a in adds]
                assert all(a is None or (a.dtype == fdt and tuple(a.shape) == s) for a, s in zip(adds, shapes))
                aptrs = (ctypes.c_void_p * L)(*[a.data_ptr() if a is not None else None for a in adds])
            st = lib.brcnn_roi_extract_backward_gather_add(ptrs, aptrs, hs, ws, sc, L, _ptr(rois), _ptr(g), shapes[0][0],
                                                           shapes[0][3], rois.size(0), ph, pw, int(sampling_ratio),
                                                           float(finest_scale), _ptr(wsp), nb, _dt(g), _stream())
            _L.check(st, 'brcnn_roi_extract_backward_gather_add')
            return (None,) * 6 + tuple(grads)
        if fdt != torch.float32:
            raise _L.BrcnnHipError('RoI extract backward: the atomic scatter form is fp32 only')
        g = grad_out.float().contiguous()
        grads = [torch.zeros(s, dtype=torch.float32, device=grad_out.device) for s in shapes]
        ptrs = (ctypes.c_void_p * L)(*[t.data_ptr() for t in grads])
        st = lib.brcnn_roi_extract_backward(ptrs, hs, ws, sc, L, _ptr(rois), _ptr(g), shapes[0][0],
                                            shapes[0][3], rois.size(0), ph, pw, int(sampling_ratio),
                                            float(finest_scale), _stream())
        _L.check(st, 'brcnn_roi_extract_backward')
        if ctx.addends is not None:
            grads = [g if a is None else g + a.to(g.dtype) for g, a in zip(grads, ctx.addends)]
            ctx.addends = None
        return (None,) * 6 + tuple(grads)


def roi_extract_autograd(feats_nhwc, rois, output_size, strides, finest_scale=56, sampling_ratio=0):
    # levels that arrive from `inject_gradients` with a gradient still to be added: the gather launch of the backward
    # pass adds it (one launch and one stream of bytes less than the add per level it replaces)
    addends = None
    ph, pw = (output_size, output_size) if isinstance(output_size, int) else output_size
    if rois.shape[0] > 0 and ROI_BACKWARD_GATHER and ROI_GATHER_ADDEND and ph <= 7 and pw <= 7:
        for i, f in enumerate(feats_nhwc):
            pend = getattr(f, '_brcnn_pending_addend', None)
            if pend is not None and pend[0][pend[1]] is not None and pend[0][pend[1]].dtype == f.dtype and \
                    f.requires_grad and f.shape[3] % 4 == 0:
                if addends is None:
                    addends = [None] * len(feats_nhwc)
                addends[i] = pend[0][pend[1]]
                pend[0][pend[1]] = None           # claimed: _InjectGradients passes this level's gradient through
    return RoIExtractFunction.apply(rois.contiguous().float(), output_size, strides, finest_scale,
                                    sampling_ratio, addends, *[f.contiguous() for f in feats_nhwc])


class GroupedConvFunction(Function):
    """grouped conv (ResNeXt conv2) on NHWC fp32 maps: forward, data gradient and weight gradient on
    the block-diagonal 64-channel tiles of the MFMA kernel (`brcnn_conv2d_*_nhwc_grouped`).
    weight: the reference parameter (Cout, Cin/groups, KH, KW)."""

    @staticmethod
    def forward(ctx, x, weight, groups, stride, pad):
        _require_gpu(x, weight)
        x = x.contiguous()
        w_tiles, window = ops.pack_grouped_weight(weight, groups)
        y = ops.conv2d_nhwc_grouped(x, w_tiles.to(x.dtype), window, None, None, None, False, stride, pad)
        ctx.save_for_backward(x, weight)
        ctx.cfg = (groups, stride, pad, window, tuple(y.shape))
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        groups, stride, pad, window, yshape = ctx.cfg
        n, h, w, cin = x.shape
        cout, cg_in, kh, kw = weight.shape
        cg_out = cout // groups
        dy = dy.to(x.dtype).contiguous()
        gdt = _dt(x)
        lib = _L.load()
        dx = dw = None
        if ctx.needs_input_grad[0]:
            # per-group transposed + flipped filters, as a grouped weight from dy channels to dx channels
            wt = weight.detach().float().view(groups, cg_out, cg_in, kh, kw).flip(3, 4).permute(0, 2, 1, 3, 4)
            wt = wt.reshape(groups * cg_in, cg_out, kh, kw)
            wt_tiles, win_t = ops.pack_grouped_weight(wt, groups)
            wt_tiles = wt_tiles.to(x.dtype)
            dx = torch.empty_like(x)
            st = lib.brcnn_conv2d_dgrad_nhwc_grouped(_ptr(dy), _ptr(wt_tiles), _ptr(dx), n, h, w, yshape[1], yshape[2],
                                                     cin, cout, kh, kw, stride, pad, win_t, gdt, _conv_stream())
            _L.check(st, 'brcnn_conv2d_dgrad_nhwc_grouped')
        if ctx.needs_input_grad[1]:
            dwt = torch.zeros((cout, kh, kw, window), dtype=torch.float32, device=x.device)
            st = lib.brcnn_conv2d_wgrad_nhwc_grouped(_ptr(x), _ptr(dy), _ptr(dwt), n, h, w, cin, cout, kh, kw,
                                                     stride, pad, window, gdt, _conv_stream())
            _L.check(st, 'brcnn_conv2d_wgrad_nhwc_grouped')
            co = torch.arange(cout, device=x.device)
            start = ((co // cg_out) * cg_in) - (co // 64) * window
            idx = (start[:, None] + torch.arange(cg_in, device=x.device)[None, :])
            dw = torch.gather(dwt, 3, idx[:, None, None, :].expand(cout, kh, kw, cg_in)).permute(0, 3, 1, 2)
        return dx, dw, None, None, None


def grouped_conv_autograd(x, weight, groups, stride, pad):
    return GroupedConvFunction.apply(x, weight, groups, stride, pad)


class DeformIm2colFunction(Function):
    """modulated deformable im2col (DCNv2, deform_groups 1) with its backward: x (N,H,W,C),
    offset_mask (N,Ho,Wo,27) raw conv_offset output -> columns (N*Ho*Wo, 9*C)"""

    @staticmethod
    def forward(ctx, x, om, stride, pad):
        _require_gpu(x, om)
        x, om = x.contiguous(), om.contiguous()
        assert om.shape[3] == 27
        col, (ho, wo) = ops.deform_im2col_nhwc(x, om, 3, stride, pad, 1)
        ctx.save_for_backward(x, om)
        ctx.cfg = (stride, pad)
        return col

    @staticmethod
    @once_differentiable
    def backward(ctx, dcol):
        x, om = ctx.saved_tensors
        stride, pad = ctx.cfg
        n, h, w, c = x.shape
        dcol = dcol.float().contiguous()
        dx = torch.zeros_like(x)
        dom = torch.empty_like(om)
        st = _L.load().brcnn_deform_col2im_nhwc(_ptr(x), _ptr(om), _ptr(dcol), _ptr(dx), _ptr(dom), n, h, w, c, 3, 3,
                                                int(stride), int(pad), 1, 27, c, _stream())
        _L.check(st, 'brcnn_deform_col2im_nhwc')
        return dx, dom, None, None


def deform_im2col_autograd(x, om, stride, pad):
    return DeformIm2colFunction.apply(x, om, stride, pad)


class BnActFunction(Function):
    """out = [relu](z * scale + shift [+ res]) over NHWC rows, one kernel each way
    (`brcnn_bn_act_forward/backward`); scale / shift are the (C,) fp32 eval-BN affine."""

    @staticmethod
    def forward(ctx, z, scale, shift, res, relu):
        _require_gpu(z, scale, shift, res)
        z = z.contiguous()
        c = z.shape[-1]
        rows = z.numel() // c
        dt = _dt(z)
        sc, sh = scale.detach().float().contiguous(), shift.detach().float().contiguous()
        r = res.contiguous() if res is not None else None
        out = torch.empty_like(z)
        st = _L.load().brcnn_bn_act_forward(_ptr(z), _ptr(sc), _ptr(sh), _ptr(r), _ptr(out), rows, c,
                                            int(relu), dt, _stream())
        _L.check(st, 'brcnn_bn_act_forward')
        ctx.save_for_backward(z, sc, out if relu else None)
        ctx.cfg = (relu, res is not None, dt, rows, c)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, dout):
        z, sc, out = ctx.saved_tensors
        relu, has_res, dt, rows, c = ctx.cfg
        dout = dout.to(z.dtype).contiguous()
        dz = torch.empty_like(z)
        dres = torch.empty_like(z) if has_res and ctx.needs_input_grad[3] else None
        dscale = torch.empty(c, dtype=torch.float32, device=z.device)
        dshift = torch.empty(c, dtype=torch.float32, device=z.device)
        lib = _L.load()
        nb = lib.brcnn_bn_act_backward_workspace_bytes(rows, c, dt)
        ws = torch.empty(max(nb, 4), dtype=torch.uint8, device=z.device)
        st = lib.brcnn_bn_act_backward(_ptr(dout), _ptr(out), _ptr(z), _ptr(sc), _ptr(dz), _ptr(dres),
                                       _ptr(dscale), _ptr(dshift), _ptr(ws), nb, rows, c, int(relu), dt, _stream())
        _L.check(st, 'brcnn_bn_act_backward')
        return dz, dscale, dshift, dres, None


class BnEvalActFunction(Function):
    """out = [relu](bn_eval(z) [+ res]) with the eval-mode BatchNorm parameters themselves: the affine
    scale = gamma / sqrt(var + eps), shift = beta - mean * scale is formed inside the kernels and the
    backward returns dgamma / dbeta directly (`brcnn_bn_eval_act_forward/backward`)"""

    @staticmethod
    def forward(ctx, z, gamma, beta, mean, var, eps, res, relu):
        _require_gpu(z, gamma, beta, mean, var, res)
        z = z.contiguous()
        c = z.shape[-1]
        rows = z.numel() // c
        dt = _dt(z)
        g32, b32 = gamma.detach().float().contiguous(), beta.detach().float().contiguous()
        m32, v32 = mean.detach().float().contiguous(), var.detach().float().contiguous()
        r = res.contiguous() if res is not None else None
        out = torch.empty_like(z)
        st = _L.load().brcnn_bn_eval_act_forward(_ptr(z), _ptr(g32), _ptr(b32), _ptr(m32), _ptr(v32), float(eps),
                                                 _ptr(r), _ptr(out), rows, c, int(relu), dt, _stream())
        _L.check(st, 'brcnn_bn_eval_act_forward')
        # without a residual the backward recomputes the ReLU mask from z and does not read `out`
        ctx.save_for_backward(z, g32, b32, m32, v32, out if relu and res is not None else None)
        ctx.cfg = (relu, res is not None, dt, rows, c, float(eps), gamma.dtype, beta.dtype)
        ctx.bn_params = (gamma, beta)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, dout):
        z, g32, b32, m32, v32, out = ctx.saved_tensors
        relu, has_res, dt, rows, c, eps, gdt, bdt = ctx.cfg
        dout = dout.to(z.dtype).contiguous()
        dz = torch.empty_like(z)
        dres = torch.empty_like(z) if has_res and ctx.needs_input_grad[6] else None
        # (two views of one buffer: a deferred second stage keeps the BUFFER alive, never the views -- autograd takes a
        # gradient as `.grad` without a copy only while nobody else holds a reference to that very tensor)
        gb = torch.empty(2 * c, dtype=torch.float32, device=z.device)
        dgamma, dbeta = gb[:c], gb[c:]
        lib = _L.load()
        nb = lib.brcnn_bn_act_backward_workspace_bytes(rows, c, dt)
        ws = torch.empty(max(nb, 4), dtype=torch.uint8, device=z.device)
        defer = rows > 0 and gdt == torch.float32 and bdt == torch.float32 and _bn_defer_ok(*ctx.bn_params)
        h = _stream()
        st = lib.brcnn_bn_eval_act_backward_ex(_ptr(dout), _ptr(out), _ptr(z), _ptr(g32), _ptr(b32), _ptr(m32), _ptr(v32), eps,
                                               _ptr(dz), _ptr(dres), _ptr(dgamma), _ptr(dbeta), _ptr(ws), nb, rows, c,
                                               int(relu), dt, h, int(defer))
        _L.check(st, 'brcnn_bn_eval_act_backward_ex')
        if defer:
            _bn_deferred(h, z.device, ws, m32, v32, gb)
        return dz, dgamma.to(gdt), dbeta.to(bdt), None, None, None, dres, None


class BnTail:
    """what the consumer of a conv -> eval-BN -> [ReLU] output needs to run that BatchNorm's backward inside
    its own data-gradient launch (`brcnn_conv2d_dgrad_bn_backward_nhwc`), and where it leaves the results for
    the producer's backward.  Only valid when the consumer is the ONLY user of the producer's output (conv2 /
    conv3 of a Bottleneck): the gradient that then travels along the autograd edge is dz, not d(output)."""
    __slots__ = ('z', 'g', 'b', 'm', 'v', 'eps', 'relu', 'out', 'done', 'dgamma', 'dbeta', 'dres', 'params')

    def __init__(self, z, g, b, m, v, eps, relu, out=None, params=None):
        self.z, self.g, self.b, self.m, self.v, self.eps, self.relu = z, g, b, m, v, eps, relu
        self.params = params    # (gamma, beta) parameters: whether their gradients may be written after the fact (_bn_defer_ok)
        self.out = out          # residual producer (bn3): its output -- the ReLU mask and the consumer's own input
        self.done, self.dgamma, self.dbeta, self.dres = False, None, None, None


class ConvBnEvalActFunction(Function):
    """out = [relu](bn_eval(conv(x, w)) [+ res]) for a trainable conv + eval-mode BatchNorm in ONE forward
    launch (`brcnn_conv2d_bn_act_nhwc_multi`: the conv epilogue stores the raw output z and the activation);
    backward = `brcnn_bn_eval_act_backward` followed by ConvNHWCFunction's data / weight gradient kernels.
    16-bit activations, single map.  `with_skip` as in ConvNHWCFunction."""

    @staticmethod
    def forward(ctx, x_cat, weight, gamma, beta, mean, var, eps, res, relu, batch, size, stride, pad, with_skip,
                in_tail=None, out_tail=None):
        """`out_tail` (an empty list): receives this layer's BnTail, for a sole consumer that will run this
        BatchNorm's backward; `in_tail`: the BnTail of the layer that produced x_cat (this layer is its sole
        consumer)"""
        _require_gpu(x_cat, weight, gamma, beta, mean, var, res)
        w_p, w_t = _conv_operands(weight, x_cat)
        x_cat = x_cat.contiguous()
        cout, kh, kw, cin = w_p.shape
        (h, w_) = size
        ho, wo = conv_out_size(h, w_, kh, kw, stride, pad)
        rows = batch * ho * wo
        g32, b32 = gamma.detach().float().contiguous(), beta.detach().float().contiguous()
        m32, v32 = mean.detach().float().contiguous(), var.detach().float().contiguous()
        r = res.contiguous() if res is not None else None
        z = torch.empty((rows, cout), dtype=x_cat.dtype, device=x_cat.device)
        out = torch.empty_like(z)
        assert r is None or (r.shape == out.shape and r.dtype == out.dtype)
        st = _L.load().brcnn_conv2d_bn_act_nhwc_multi(_ptr(x_cat), _ptr(w_p), _ptr(g32), _ptr(b32), _ptr(m32), _ptr(v32),
                                                      float(eps), _ptr(r), _ptr(z), _ptr(out), batch, 1, _ints([h]),
                                                      _ints([w_]), cin, cout, kh, kw, int(stride), int(pad), int(relu),
                                                      _dt(x_cat), _conv_stream())
        _L.check(st, 'brcnn_conv2d_bn_act_nhwc_multi')
        ctx.w_t = w_t
        ctx.save_for_backward(x_cat, weight, z, g32, b32, m32, v32, out if relu and res is not None else None)
        ctx.cfg = (batch, (tuple(size),), ((ho, wo),), stride, pad, bool(relu), res is not None, float(eps),
                   gamma.dtype, beta.dtype)
        ctx.in_tail = in_tail if (in_tail is not None and w_t is not None and x_cat.requires_grad) else None
        ctx.bn_params = (gamma, beta)
        ctx.tail = None
        if out_tail is not None:
            ctx.tail = BnTail(z, g32, b32, m32, v32, float(eps), bool(relu), out if res is not None else None, (gamma, beta))
            out_tail.append(ctx.tail)
        if with_skip:
            return out, x_cat.view_as(x_cat)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, dout, dskip=None):
        x_cat, weight, z, g32, b32, m32, v32, out = ctx.saved_tensors
        batch, sizes, out_sizes, stride, pad, relu, has_res, eps, gdt, bdt = ctx.cfg
        rows, c = z.shape
        dt = _dt(z)
        dout = dout.to(z.dtype).contiguous()
        lib = _L.load()
        dres = None
        if ctx.tail is not None and ctx.tail.done:
            # the sole consumer's data-gradient launch already ran this BatchNorm's backward: `dout` IS dz
            dz, dgamma, dbeta, dres = dout, ctx.tail.dgamma, ctx.tail.dbeta, ctx.tail.dres
            ctx.tail.done, ctx.tail.dgamma, ctx.tail.dbeta, ctx.tail.dres = False, None, None, None
        else:
            dz = torch.empty_like(z)
            dres = torch.empty_like(z) if has_res and ctx.needs_input_grad[7] else None
            gb = torch.empty(2 * c, dtype=torch.float32, device=z.device)       # (see BnEvalActFunction.backward)
            dgamma, dbeta = gb[:c], gb[c:]
            nb = lib.brcnn_bn_act_backward_workspace_bytes(rows, c, dt)
            ws = torch.empty(max(nb, 4), dtype=torch.uint8, device=z.device)
            defer = rows > 0 and gdt == torch.float32 and bdt == torch.float32 and _bn_defer_ok(*ctx.bn_params)
            h = _stream()
            st = lib.brcnn_bn_eval_act_backward_ex(_ptr(dout), _ptr(out), _ptr(z), _ptr(g32), _ptr(b32), _ptr(m32), _ptr(v32),
                                                   eps, _ptr(dz), _ptr(dres), _ptr(dgamma), _ptr(dbeta), _ptr(ws), nb, rows, c,
                                                   int(relu), dt, h, int(defer))
            _L.check(st, 'brcnn_bn_eval_act_backward_ex')
            if defer:
                _bn_deferred(h, z.device, ws, m32, v32, gb)
        t = ctx.in_tail
        # a residual producer (previous block's bn3) needs this conv's identity alias gradient; a plain one none
        if t is not None and ctx.needs_input_grad[0] and (dskip is not None) == (t.out is not None) and \
                (dskip is None or dskip.dtype == x_cat.dtype):
            # data gradient + the producer's BatchNorm backward in one launch: dx leaves as the producer's dz
            cout, cin, kh, kw = weight.shape
            (h, w_), (ho, wo) = sizes[0], out_sizes[0]
            dzp = torch.empty_like(x_cat)
            tgb = torch.empty(2 * cin, dtype=torch.float32, device=z.device)      # (see BnEvalActFunction.backward)
            t.dgamma, t.dbeta = tgb[:cin], tgb[cin:]
            nb = lib.brcnn_conv2d_dgrad_bn_backward_workspace_bytes(batch, h, w_, cin)
            ws = torch.empty(max(nb, 4), dtype=torch.uint8, device=z.device)
            if dskip is not None:
                dskip = dskip.contiguous()
                t.dres = torch.empty_like(x_cat)
            defer = t.params is not None and _bn_defer_ok(*t.params)
            hs_ = _conv_stream()
            st = lib.brcnn_conv2d_dgrad_bn_backward_nhwc_ex(_ptr(dz), _ptr(ctx.w_t), _ptr(t.z), _ptr(t.g), _ptr(t.b), _ptr(t.m),
                                                            _ptr(t.v), t.eps, int(t.relu), _ptr(dskip), _ptr(t.out), _ptr(t.dres),
                                                            _ptr(dzp), _ptr(t.dgamma),
                                                            _ptr(t.dbeta), _ptr(ws), nb, batch, h, w_, ho, wo, cin, cout, kh, kw,
                                                            stride, pad, dt, hs_, int(defer))
            _L.check(st, 'brcnn_conv2d_dgrad_bn_backward_nhwc_ex')
            if defer:
                _bn_deferred(hs_, z.device, ws, t.m, t.v, tgb)
            t.done = True
            _, dw, _ = _conv_backward(x_cat, weight, ctx.w_t, dz, (batch, sizes, out_sizes, stride, pad), None, False,
                                      ctx.needs_input_grad[1], has_dgrad=True)
            dx = dzp
        else:
            dx, dw, dskip = _conv_backward(x_cat, weight, ctx.w_t, dz, (batch, sizes, out_sizes, stride, pad), dskip,
                                           ctx.needs_input_grad[0], ctx.needs_input_grad[1])
            if dskip is not None:
                dx = dskip if dx is None else dx + dskip
        return (dx, dw, dgamma.to(gdt), dbeta.to(bdt), None, None, None, dres) + (None,) * 8


# conv2 / conv3 of a Bottleneck run the BatchNorm backward of bn1 / bn2 inside their data-gradient launches
FUSE_BN_BACKWARD_INTO_DGRAD = _os.environ.get('BRCNN_FUSE_BN_BWD', '1') != '0'
# ... and conv1 of the NEXT block of a stage the backward of bn3 (residual + ReLU) of the block before it.  Round 2 measured
# this 0.2 ms SLOWER per step (23.87 -> 24.07 ms, profiles/r02_notes.md); with round 4's kernels the same-box interleaved
# A/B (tools/experiments/ab_train.py base=bn3:0 bn3=bn3:1) has it 0.35 ms FASTER (20.63 -> 20.28 ms, every round of four),
# so it is on.  BRCNN_FUSE_BN3_BWD=0 disables it.
FUSE_RESIDUAL_BN_BACKWARD = _os.environ.get('BRCNN_FUSE_BN3_BWD', '1') != '0'


def conv_bn_eval_act_fusable(x, conv, bn, residual):
    """the one-launch training forward applies: 16-bit NHWC activations, dense conv without bias whose
    channel counts suit the 16-bit MFMA kernel, eval-mode BatchNorm"""
    cout, cin = conv.weight.shape[:2]
    return (x.is_cuda and x.dtype in (torch.bfloat16, torch.float16) and conv.groups == 1 and conv.bias is None and
            conv.dilation == (1, 1) and cin % 64 == 0 and cout % 64 == 0 and not bn.training and bn.affine and
            bn.track_running_stats and (residual is None or residual.dtype == x.dtype))


def conv_bn_eval_act_autograd(x, conv, bn, residual=None, relu=True, with_skip=False, sole_consumer=False,
                              single_use_output=False):
    """x (N,H,W,Cin) -> [relu](bn(conv(x)) [+ residual]) (N,Ho,Wo,Cout), differentiable; `with_skip` as in
    conv2d_nhwc_autograd.  `single_use_output`: the caller promises that the result feeds exactly one
    conv_bn_eval_act_autograd(..., sole_consumer=True) call and nothing else -- that call then runs this layer's
    BatchNorm backward inside its data-gradient launch (the tag travels as `out._brcnn_tail`)."""
    n, h, w, cin = x.shape
    cout, _, kh, kw = conv.weight.shape
    stride, pad = conv.stride[0], conv.padding[0]
    ho, wo = conv_out_size(h, w, kh, kw, stride, pad)
    res = residual.reshape(n * ho * wo, cout) if residual is not None else None
    in_tail = getattr(x, '_brcnn_tail', None) if sole_consumer and FUSE_BN_BACKWARD_INTO_DGRAD else None
    out_tail = [] if single_use_output and FUSE_BN_BACKWARD_INTO_DGRAD and \
        (residual is None or FUSE_RESIDUAL_BN_BACKWARD) else None
    y = ConvBnEvalActFunction.apply(x.reshape(n * h * w, cin), conv.weight, bn.weight, bn.bias, bn.running_mean,
                                    bn.running_var, bn.eps, res, relu, n, (h, w), stride, pad, with_skip, in_tail, out_tail)
    skip = None
    if with_skip:
        y, skip = y
        skip = skip.view(n, h, w, cin)
    y = y.view(n, ho, wo, cout)
    if out_tail:
        y._brcnn_tail = out_tail[0]
    return (y, skip) if with_skip else y


def bn_eval_act_autograd(z, bn, res=None, relu=True):
    """z through the eval-mode BatchNorm module `bn` (+ residual, + ReLU), one kernel each way"""
    return BnEvalActFunction.apply(z, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps, res, relu)


def bn_act_supported(z):
    """channel-vector count a power of two (every BatchNorm width of the ResNet family)"""
    c = z.shape[-1]
    v = 4 if z.dtype == torch.float32 else 8
    n = c // v
    return c % v == 0 and n > 0 and (n & (n - 1)) == 0 and z.dtype in (torch.float32, torch.bfloat16, torch.float16)


def bn_act_autograd(z, scale, shift, res=None, relu=True):
    return BnActFunction.apply(z, scale, shift, res, relu)


def wants_grad(x, *params):
    """True when the op must be recorded for backward (grad mode on and some input or
    parameter is trainable); frozen layers keep using the fused inference kernels."""
    if not torch.is_grad_enabled():
        return False
    if x is not None and x.requires_grad:
        return True
    return any(p is not None and p.requires_grad for p in params)


class _InjectGradients(Function):
    """identity on the pyramid levels whose backward adds a stored gradient per level: the gradient the RPN branch
    produced in its own, earlier backward pass (detectors.py, early_rpn_backward) joins the second stage's here --
    the sum autograd forms itself when both branches hang off the same tensors in one backward pass"""

    @staticmethod
    def forward(ctx, holder, *feats):
        ctx.holder = holder
        ctx.set_materialize_grads(False)
        return tuple(f.view_as(f) for f in feats)

    @staticmethod
    @once_differentiable
    def backward(ctx, *grads):
        out = []
        for g, extra in zip(grads, ctx.holder):
            if extra is None:
                out.append(g)
            elif g is None:
                out.append(extra)
            else:
                out.append(g + extra.to(g.dtype))
        ctx.holder = None
        return (None,) + tuple(out)


def inject_gradients(feats, extra):
    """`feats` with `extra[i]` (or None) added to the gradient of level i on the way back.  A level that goes to
    `roi_extract_autograd` hands its addend over to that node (tagged through `_brcnn_pending_addend`): the RoIAlign
    gradient gather adds it while it writes the level's gradient."""
    if not extra or all(e is None for e in extra):
        return tuple(feats)
    holder = list(extra)
    outs = _InjectGradients.apply(holder, *feats)
    for i, o in enumerate(outs):
        if holder[i] is not None:
            o._brcnn_pending_addend = (holder, i)
    return outs
