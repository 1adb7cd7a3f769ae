"""The trunk of the train step (backbone + neck, forward AND backward) as two HIP graphs.

Why: the device-resident train step issues ~770 launches from Python (ctypes calls into libbrcnn_hip.so, a few aten
ops, autograd's own bookkeeping), ~14 ms of host work for a ~17.5 ms device step on the builder's box -- device-bound
there, with two or three milliseconds to spare; on a host that is a fifth slower the second half of the step (everything
behind the sampler's host synchronisation: ~450 launches for ~11 ms of device work) becomes LAUNCH-bound and the step
time follows the host (round 4: 17.9 ms on the builder's box, 20.7 ms in the driver's run, same kernels).  The trunk is
the part whose launch sequence never changes from step to step -- fixed shapes, no host read, no data-dependent control
flow -- and it holds ~60 % of the launches: ~100 of the forward pass and ~250 of the backward pass (data gradients,
fused BatchNorm backward, weight gradients on the second stream).  Captured once, they cost two `hipGraphLaunch` calls
per step.  The proposal stage, the sampler (its one host synchronisation), the RoI head, the losses and the optimizer
stay eager: their shapes or their control flow depend on the data.

How (no tracing compiler: plain HIP stream capture of the launches the eager code issues anyway):

* graph F: `detector.extract_feat_nhwc(static_img)` with autograd recording -- the saved activations land in the
  graphs' private memory pool, at fixed addresses;
* graph B: `torch.autograd.backward(feats, static_grad_outputs)` over that recording.  The weight-gradient kernels
  write into an arena chunk allocated (and zero-filled: a memset node) inside the capture; the gradients
  `torch.autograd.grad` returns for the parameters during the capture are the STATIC gradient tensors of every later step;
* per step: `TrunkFunction.forward` copies the image into the static input (skipped when it already is that tensor),
  replays F and returns detached views of the static pyramid; `TrunkFunction.backward` copies the incoming pyramid
  gradients into the static buffers, replays B and assigns the static gradients to `param.grad` (accumulating where a
  gradient already exists) -- the trunk's parameters never pass through AccumulateGrad nodes;
* the weight-gradient side stream of the capture is a stream of its own: its convolution scratch (keyed by stream,
  `lib.stream_handle`) is not shared with the eager side stream, whose RoI-head / RPN weight gradients may still be
  running when graph B starts;
* the chained stream-K schedule is off under capture (conv_igemm_bf16.hip, `sk_plan`): its epoch flags are not
  replay-safe.  Same bits either way.

MEASURED (round 5, one MI355X, bf16 step of bench.py, `profiles/r05_notes.md`): the replayed step is bit-identical to
the eager one and the host runs 12.4 ms ahead of the device at the sampler's synchronisation instead of 6.0 -- but the
step takes 20.0 ms instead of 18.1: ROCm 7.2's graph executor runs the captured weight-gradient branch serially with
the main chain (the replay costs what the eager step costs WITHOUT the second stream, 19.4 ms, plus ~6 us per node;
DEBUG_HIP_FORCE_GRAPH_QUEUES / DEBUG_HIP_GRAPH_BATCH_SIZE / DEBUG_CLR_GRAPH_PACKET_CAPTURE change nothing).  It
therefore pays only where the host cannot keep up with ~770 launches per step, and is OFF by default everywhere.

The capture is keyed by everything its addresses and branches depend on -- input shape and dtype, compute dtype,
the fusion switches of `autograd`, the storage addresses of every parameter / buffer / packed conv operand of the
trunk and each module's `training` flag -- and is redone when any of it changes (a handful of keys are kept: multi-scale
training re-captures per shape).  Off by default; `detector.graph_trunk = True` (`graph_trunk = True` in a config for the
runner, BRCNN_BENCH_GRAPH_TRUNK=1 for bench.py's train leg) switches it on, BRCNN_GRAPH_TRUNK=0 vetoes it.
"""
import contextlib
import os

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import autograd as _A

ENABLED = os.environ.get('BRCNN_GRAPH_TRUNK', '1') != '0'
MAX_KEYS = 4            # captures kept at one time (the oldest goes first)
MIN_SEEN = 2            # a key is captured when it comes up for the second time: a run whose input shape changes from
                        # batch to batch (aspect-ratio grouped padding) never pays for a capture it would use once
MAX_CAPTURES = 16       # ... and a run that keeps cycling through more shapes than MAX_KEYS stops capturing


@contextlib.contextmanager
def _fresh_leaves(modules):
    """every parameter of `modules` replaced, for the duration, by a NEW leaf that shares its storage (and its packed
    conv operands).  Why: autograd caches a leaf's gradient accumulator node, and a node remembers the stream it was
    created on.  A parameter that took part in any eager step has its accumulator on the DEFAULT stream (kept alive by
    whatever still holds that step's graph: log scalars, outputs); `torch.autograd.grad(..., inputs=params)` inside a
    capture then makes the engine synchronise the default stream with the capturing one -- which drags the NULL stream
    into the capture (`hipStreamWaitEvent(stream:<null>, ...)` in the runtime's trace) and `hipStreamEndCapture`
    segfaults.  Leaves born inside the capture's own forward pass get their accumulators on the capture stream.
    Yields {id(original): alias}."""
    alias, undo = {}, []
    for root in modules:
        for m in root.modules():
            for name, p in list(m._parameters.items()):
                if p is None:
                    continue
                a = alias.get(id(p))
                if a is None:
                    a = torch.nn.Parameter(p.detach(), requires_grad=p.requires_grad)
                    pk = getattr(p, '_brcnn_pack', None)
                    if pk is not None:
                        a._brcnn_pack = pk
                    alias[id(p)] = a
                undo.append((m, name, p))
                m._parameters[name] = a
    try:
        yield alias
    finally:
        for m, name, p in undo:
            m._parameters[name] = p


class _Captured:
    __slots__ = ('key', 'fwd', 'bwd', 'static_img', 'feats', 'gouts', 'grads', 'anchor', 'replays')


class TrunkFunction(Function):
    @staticmethod
    def forward(ctx, anchor, cap, img):
        if img.data_ptr() != cap.static_img.data_ptr():
            cap.static_img.copy_(img)
        cap.fwd.replay()
        cap.replays += 1
        ctx.cap = cap
        outs = tuple(f.detach() for f in cap.feats)
        return outs

    @staticmethod
    @once_differentiable
    def backward(ctx, *gouts):
        cap = ctx.cap
        for dst, g in zip(cap.gouts, gouts):
            if g is None:
                dst.zero_()
            elif g.data_ptr() != dst.data_ptr():
                dst.copy_(g)
        # gradient accumulation (a second backward pass without zero_grad(set_to_none=True) in between): a parameter's
        # `.grad` may already BE the capture's static gradient tensor, holding the previous pass's result -- the replay
        # overwrites it.  Such gradients are set aside first and added back afterwards (they were lost).
        pending = [(p, p.grad.clone()) for p, g in cap.grads if p.grad is not None and p.grad.data_ptr() == g.data_ptr()]
        cap.bwd.replay()
        for p, g in cap.grads:
            if p.grad is None:
                p.grad = g
            elif p.grad.data_ptr() != g.data_ptr():
                p.grad = p.grad + g
        for p, old in pending:
            p.grad = old.add_(p.grad)       # (out of place with respect to the static tensor: the next replay may overwrite it)
        return None, None, None


class GraphedTrunk:
    """`trunk(img)` -> the pyramid (tuple of NHWC tensors that require grad), backbone + neck replayed from HIP graphs"""

    def __init__(self, detector):
        self.det = detector
        self.caps = {}
        self.seen = {}
        self.captures = 0
        self._side = None
        self._capture_stream = None
        self.disabled_reason = None

    # ---- what the captured addresses / branches depend on ---------------------------------------
    def _modules(self):
        mods = [self.det.backbone]
        if self.det.with_neck:
            mods.append(self.det.neck)
        return mods

    def _params(self):
        return [p for m in self._modules() for p in m.parameters()]

    def _key(self, img):
        from . import blocks
        ptrs = []
        for m in self._modules():
            for p in m.parameters():
                pk = getattr(p, '_brcnn_pack', None)
                # (a packed operand that is current is READ by the captured launches; a stale one is re-packed by a
                # launch inside the capture: which of the two is part of the key)
                ptrs.append((p.data_ptr(), p.requires_grad, p.stride(),
                             None if pk is None else (pk[0] == p._version, pk[1], pk[2].data_ptr(),
                                                      None if pk[3] is None else pk[3].data_ptr())))
            for b in m.buffers():
                ptrs.append(b.data_ptr())
            ptrs.append(tuple(s.training for s in m.modules()))
        return (tuple(img.shape), img.dtype, img.device.index, str(blocks.compute_dtype()), _A.FUSE_BN_BACKWARD_INTO_DGRAD,
                _A.FUSE_RESIDUAL_BN_BACKWARD, _A.WGRAD_SIDE_STREAM, blocks.FUSE_CONV_BN_TRAIN, hash(tuple(ptrs)))

    def usable(self, img):
        """graph replay applies: switched on, a HIP tensor, gradients wanted, no DistributedDataParallel hooks on the
        trunk's parameters (they are handed their gradients directly) and nothing asked for a second-order graph"""
        if not ENABLED or not img.is_cuda or not torch.is_grad_enabled() or self.disabled_reason:
            return False
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and not _A._OWN_REDUCER[0]:
            return False            # (possibly) DistributedDataParallel: its reducer wants every gradient through its hooks
        return any(p.requires_grad for p in self._params())

    # ---- capture ---------------------------------------------------------------------------------
    def _capture(self, img, key):
        det = self.det
        dev = img.device
        skey = (dev.type, dev.index)
        if self._side is None:
            self._side = torch.cuda.Stream(dev)
            self._capture_stream = torch.cuda.Stream(dev)
        cap = _Captured()
        cap.key, cap.replays = key, 0
        cap.static_img = img.detach().clone()
        params = [p for p in self._params() if p.requires_grad]
        saved_grads = [p.grad for p in params]
        eager_side = _A._side_streams.get(skey)
        listener, arena_state = _A.grad_arena.listener, (_A.grad_arena.hint,)
        s = self._capture_stream
        # Python's cyclic collector stays off from here to the end of the captures: a collection that starts inside a capture
        # frees device tensors and events from whatever cycle it finds (autograd contexts and their BnTail tags form some), and
        # a free that has to record an event on a capturing stream aborts the process (seen once in ~10 suite runs, in
        # `record_event` of a weight-gradient launch; torch.cuda.graph collects before a capture for the same reason)
        import gc
        gc_was_on = gc.isenabled()
        gc.collect()
        gc.disable()
        try:
            _A.grad_arena.listener = None               # (a gradient reducer must not see, let alone slice, the capture)
            _A._side_streams[skey] = self._side         # the capture's own weight-gradient stream (own conv scratch)
            with _fresh_leaves(self._modules()) as alias:
                leaves = [alias[id(p)] for p in params]
                # warm-up on the capture stream: lazy one-off work (conv scratch registration, kernel attributes, folded
                # BatchNorm caches, schedule tables) must not fall into the capture
                s.wait_stream(torch.cuda.current_stream(dev))
                with torch.cuda.stream(s):
                    for _ in range(2):
                        _A.grad_arena.new_step()
                        feats = det.extract_feat_nhwc(cap.static_img)
                        torch.autograd.grad(feats, leaves, [torch.zeros_like(f) for f in feats], allow_unused=True)
                        _A.join_side_streams(dev)
                    del feats
                torch.cuda.current_stream(dev).wait_stream(s)
                torch.cuda.synchronize(dev)
                cap.fwd, cap.bwd = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
                with torch.cuda.graph(cap.fwd, stream=s):
                    feats = det.extract_feat_nhwc(cap.static_img)
                cap.feats = tuple(feats)
                cap.gouts = [torch.zeros_like(f) for f in cap.feats]
                with torch.cuda.graph(cap.bwd, pool=cap.fwd.pool(), stream=s):
                    _A.grad_arena.new_step()             # a chunk of its own, zero-filled by a memset node of the graph
                    # (autograd.grad over the fresh leaves: no AccumulateGrad node of an earlier eager step, see
                    # _fresh_leaves.  The side stream is joined by the end-of-pass callback of the weight-gradient launches
                    # that forked it; an explicit join here would make the capture wait for a stream that never entered it
                    # when no launch did)
                    grads = torch.autograd.grad(cap.feats, leaves, cap.gouts, allow_unused=True)
            cap.grads = [(p, g if g.dtype == p.dtype else g.to(p.dtype)) for p, g in zip(params, grads) if g is not None]
            cap.feats = tuple(f.detach() for f in cap.feats)     # (drop the capture-time autograd graph)
            del feats, grads, leaves
        finally:
            if eager_side is None:
                _A._side_streams.pop(skey, None)
            else:
                _A._side_streams[skey] = eager_side
            _A._side_seen.pop(skey, None)
            _A._join_queued[skey] = False
            _A.grad_arena.new_step()                     # eager launches never carve from the captured chunk
            _A.grad_arena.hint = 0                       # (its size hint counted the trunk: the eager steps re-learn theirs)
            _A.grad_arena.listener = listener
            del arena_state
            for p, g in zip(params, saved_grads):
                p.grad = g
            if gc_was_on:
                gc.enable()
        cap.anchor = torch.zeros((), device=dev, requires_grad=True)
        self.captures += 1
        return cap

    def __call__(self, img):
        """the pyramid, or None: this input's key has not come up often enough to be worth a capture"""
        key = self._key(img)
        cap = self.caps.get(key)
        if cap is None:
            if len(self.seen) > 256:
                self.seen.clear()
            n = self.seen[key] = self.seen.get(key, 0) + 1
            if n < MIN_SEEN or self.captures >= MAX_CAPTURES:
                return None
            if len(self.caps) >= MAX_KEYS:
                self.caps.pop(next(iter(self.caps)))
            try:
                cap = self._capture(img.contiguous(), key)
            except Exception as e:              # a capture that fails (an op that cannot be captured) is not retried
                self.disabled_reason = f'{type(e).__name__}: {e}'
                raise
            self.caps[key] = cap
        return TrunkFunction.apply(cap.anchor, cap, img)


def trunk_features(detector, img):
    """`detector.extract_feat_nhwc(img)` for the train step: through the HIP graphs when `detector.graph_trunk` is set
    and they apply, eagerly otherwise"""
    if getattr(detector, 'graph_trunk', False):
        gt = detector.__dict__.get('_graphed_trunk')
        if gt is None:
            gt = detector.__dict__['_graphed_trunk'] = GraphedTrunk(detector)
        if gt.usable(img):
            feats = gt(img)
            if feats is not None:
                return feats
    return detector.extract_feat_nhwc(img)
