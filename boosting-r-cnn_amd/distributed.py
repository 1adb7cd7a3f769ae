"""Data-parallel gradient averaging over RCCL without autograd hooks.

The reference wraps the detector in MMDistributedDataParallel (mmdet/apis/train.py:75-83, tools/dist_train.sh:8-9):
torch's reducer hooks every gradient accumulator, copies gradients into its buckets and all-reduces bucket by
bucket.  Here the weight-gradient kernels already write into ONE zero-filled arena per step
(`autograd.grad_arena`: the result of a launch IS `weight.grad`), in the order of the backward pass, which is the
same on every rank.  `GradReducer` therefore all-reduces slices of that arena in place -- no bucket copies, no
accumulator hooks, so the N = 1 optimisations stay on under torch.distributed (weight-gradient launches on the
second stream, dW handed to `.grad` unchanged):

* at the end of the backward pass (`reduce()`): the arena in ONE all-reduce, and the few hundred small gradients that
  do not live in it (BatchNorm / GroupNorm affine, biases, Scale) as one flattened bucket;
* `overlap=True` (off by default): as the backward pass fills the arena, every finished slice of `slice_mb` MiB is
  all-reduced IN PLACE on a communication stream behind an event of the stream that launched its last writer -- but
  only slices cut from runs of ranges that autograd takes as `.grad` unchanged (`writers_launched(..., in_place)`):
  anything autograd copies on the main stream after the launch (a gradient whose strides are not the parameter's, the
  first FC's re-layout, weights behind a cat / pad) is reduced after the pass from the copy.  Round 4 sliced by offset
  alone; the in-place all-reduce raced with those copies and some gradients were reduced twice (profiles/r05_notes.md).
  Measured at
  world size 1 on MI355X (profiles/r03_notes.md): with three streams already busy (main, weight gradients, proposals)
  a fourth stream that WAITS on a weight-gradient event costs the step 1.4-4 ms (19.7 -> 21.6-24 ms, depending on
  which pooled stream it is; the wait alone, without any RCCL call, 3.3 ms) -- more than the ~1.3 ms of ring
  all-reduce it could hide at 8 GPUs.  The single all-reduce after the backward pass costs 0.45 ms at world size 1;
* the average is RCCL's `ReduceOp.AVG` (sum, then one multiply, on backends without it);
* the small gradients travel through a PERSISTENT flat fp32 bucket: one multi-tensor copy in, one all-reduce, and the
  parameters' `.grad` become views of the bucket (no per-step `cat`, no copy back);
* `compress='bf16'` (off by default): the arena is cast to a persistent bf16 buffer, all-reduced as bf16 (half the
  xGMI bytes: 2 x 7/8 x 92 MB instead of 184 MB per GPU and step) and cast back into the fp32 arena; the master
  weights and the optimizer step stay fp32 (tests/test_distributed_cpu.py pins the update against an independent
  computation from the bf16-rounded local gradients);
* every rank must post the same collectives: the layout (chunk sizes in use, small-bucket length and count) is
  compared across the ranks through a fixed-size 2-word collective AHEAD of the payload collectives and read on the
  host by EVERY rank on EVERY step (`strict=True`, the default).  The signature is host data, so under RCCL the two
  words travel on a control stream of their own and the host waits for that stream only (~0.1 ms; read on the main
  stream the same comparison cost the host its whole lead over the device: slack 2.6 instead of 5-6 ms) -- a
  mismatch raises on all ranks at the same point, before anything unpaired has been posted and before the optimizer
  can consume an invalid average (DistributedDataParallel's reducer raises in the same case).  `strict=False` reads
  the comparison one step late in the steady state (no host synchronisation): detection is then ASYMMETRIC -- the rank
  whose layout changed raises at once, a rank whose own layout did not change has already posted payload collectives
  that no longer pair and may block in RCCL until the failing rank's exit tears the job down (torchrun does); it
  raises at its next reduce() / finish() if it gets there.  Use it only where a hang-then-abort on a layout bug is
  acceptable.

One process per GPU; the image batch is sharded by the sampler, weights are replicated (`broadcast_parameters`).
"""
import torch
import torch.distributed as dist

from . import autograd as _A


def replicas_identical(module, process_group=None):
    """True iff every rank holds bit-identical parameters and buffers (one int64 checksum per rank -- the wrapping sum of
    the values' bit patterns -- compared through a 2-word MAX all-reduce; a host synchronisation).  What
    DistributedDataParallel verifies once at construction; the runner can ask for it after every optimizer step
    (`check_replicas = True` in the config / BRCNN_CHECK_REPLICAS=1): data-parallel training is only correct while the
    replicas stay identical (mmdet/apis/train.py:75-83)."""
    total = None
    for t in list(module.parameters()) + list(module.buffers()):
        d = t.detach().contiguous()
        if d.numel() == 0:
            continue
        bits = d.view(torch.uint8).view(-1).to(torch.int64).sum() if d.element_size() not in (2, 4, 8) else \
            d.view({2: torch.int16, 4: torch.int32, 8: torch.int64}[d.element_size()]).view(-1).to(torch.int64).sum()
        total = bits if total is None else total * 31 + bits       # (order-dependent mix; int64 wraps)
    if total is None:
        return True
    t = torch.stack([total, -total])
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=process_group)
    hi, neg_lo = t.tolist()
    return hi == -neg_lo


def _subtract(live, issued):
    """the parts of the sorted ranges `live` that no range of `issued` (sorted, disjoint) covers"""
    out = []
    for lo, hi in live:
        for a, b in issued:
            if b <= lo or a >= hi:
                continue
            if a > lo:
                out.append((lo, a))
            lo = max(lo, b)
            if lo >= hi:
                break
        if lo < hi:
            out.append((lo, hi))
    return out


class GradReducer:
    def __init__(self, params, process_group=None, slice_mb=64, overlap=False, compress=None, strict=True, names=None):
        if not (dist.is_available() and dist.is_initialized()):
            raise RuntimeError('GradReducer needs an initialised torch.distributed process group')
        if compress not in (None, 'bf16'):
            raise ValueError(f"GradReducer: compress={compress!r} (None or 'bf16')")
        params = list(params)
        if params and isinstance(params[0], tuple):        # named_parameters()
            names = {id(p): n for n, p in params}
            params = [p for _, p in params]
        self.names = dict(names or {})                      # id(parameter) -> name, for error messages only
        self.params = [p for p in params if p.requires_grad]
        self.group = process_group
        self.world = dist.get_world_size(process_group)
        self.rank = dist.get_rank(process_group)
        self.slice_elems = max(1, int(slice_mb * (1 << 20) // 4))
        self.overlap = bool(overlap)
        self.compress = compress
        self.strict = bool(strict)   # every rank reads the layout comparison on the host on EVERY step (see above)
        if self.overlap and self.compress:
            raise ValueError('GradReducer: compress needs overlap=False (the slices are all-reduced in place)')
        self._avg = dist.get_backend(process_group) == 'nccl'
        self._works = []
        self._comm = {}
        # chunk -> [elements already handed to a collective, elements whose writers have all been launched]
        self._progress = []
        # the tensors that had a weight-gradient launch in the running step, by id.  The tensor itself is kept: many of
        # them are per-call temporaries (the 4-D view of an FC weight, a padded weight), and the id of a freed one is
        # handed to the next temporary -- which then looked like a second launch for the same parameter (one two-rank
        # bench run in ten stopped with the shared-parameter error, profiles/r05_notes.md)
        self._seen = {}
        self._bucket = None          # persistent flat fp32 storage of the gradients outside the arena
        self._cbuf = None            # persistent bf16 staging of the arena (compress='bf16')
        self._layout = None          # the layout signature the ranks last agreed on
        self._lazy = None            # (host tensor, event, signature) of the previous step's non-blocking comparison
        self.last_bytes = 0          # bytes this rank handed to all-reduce in the last reduce() (payload, one direction)
        _A._OWN_REDUCER[0] = True
        _A.grad_arena.listener = self

    def finish(self):
        """end of training: the last step's deferred layout comparison (strict=False) is read and checked"""
        self._check_lazy()

    def close(self):
        self._lazy = None
        if _A.grad_arena.listener is self:
            _A.grad_arena.listener = None
        _A._OWN_REDUCER[0] = False

    def describe(self):
        """what bench.py prints beside `reduce_ms`"""
        return {'arena': 'in place' if not self.compress else f'{self.compress} staging buffer',
                'overlap': self.overlap, 'bytes_last_step': int(self.last_bytes), 'world': self.world,
                'op': 'AVG' if self._avg else 'SUM + scale'}

    # ---- parameters -----------------------------------------------------------------------------
    def broadcast_parameters(self, module, src=0):
        """rank `src`'s parameters and buffers to every rank (what DistributedDataParallel does at construction)"""
        tensors = [p.data for p in module.parameters()] + [b.data for b in module.buffers()]
        for t in tensors:
            dist.broadcast(t, src, group=self.group)

    # ---- arena listener (called by autograd._GradArena / _conv_backward) ---------------------------
    def chunk_opened(self, buf):
        if not self._progress:
            self.last_bytes = 0         # first chunk of a new step: the overlapped slices of THIS step count from here
            self._seen = {}             # ... and so does the shared-parameter bookkeeping (a backward pass without a
            #                             reduce() behind it must not make every parameter look shared in the next one)
        # [chunk, ranges already handed to a collective, the running in-place run [lo, hi) or None,
        #  the in-place ranges that joined a run: (lo, hi, tensor) -- reduce() checks them against the live gradients]
        self._progress.append([buf, [], None, []])

    def shared_parameter(self, param):
        """a parameter reached a second weight-gradient launch in one backward pass (autograd._side_stream_for, or
        the listener's own bookkeeping below): autograd will ADD that result to the first one's, whose arena slice the
        overlapped form may already have handed to the communication stream"""
        if self.overlap:
            raise RuntimeError('GradReducer(overlap=True): a parameter is used twice in one backward pass (its second '
                               'gradient is accumulated into an arena slice that may already be in flight); use '
                               'overlap=False for this model')

    def writers_launched(self, buf, lo, hi, stream, in_place=True, param=None):
        """the weight-gradient launch writing arena elements [lo, hi) of `buf` has been issued on `stream`.

        Overlapped form: a range joins the running RUN of ranges that may be all-reduced in place during the backward
        pass only if `in_place` -- autograd hands that very storage to the parameter as `.grad`.  Anything else is
        copied by autograd on the main stream after the launch returns (AccumulateGrad clones a gradient whose strides
        are not the parameter's; a cat / pad / permute in front of the weight has a copying backward), and the COPY is
        the gradient reduce() all-reduces.  Round 4 sliced the arena by offset alone: the in-place all-reduce of a
        slice raced with those main-stream copies, and a copy that read already-reduced values was reduced a second
        time (tests/ddp_worker.py; profiles/r05_notes.md).  A range that is not `in_place` ends the run; what is left
        of the run, and the range itself if a gradient still refers to it, is picked up by reduce()."""
        if param is not None:
            if id(param) in self._seen:
                self.shared_parameter(param)
                in_place = False
            self._seen[id(param)] = param
        if not self.overlap:
            return
        for pr in self._progress:
            if pr[0] is buf:
                run = pr[2]
                if not in_place:
                    pr[2] = None
                    return
                if run is not None and run[1] == lo:
                    run[1] = hi
                else:
                    run = pr[2] = [lo, hi]
                pr[3].append((lo, hi, param))
                while run[1] - run[0] >= self.slice_elems:
                    self._issue(buf, run[0], run[0] + self.slice_elems, stream)
                    pr[1].append((run[0], run[0] + self.slice_elems))
                    run[0] += self.slice_elems
                return

    def _ctrl_stream(self, device):
        key = ('ctrl', device.type, device.index)
        if key not in self._comm:
            self._comm[key] = torch.cuda.Stream(device)
        return self._comm[key]

    def _comm_stream(self, device):
        key = (device.type, device.index)
        if key not in self._comm:
            self._comm[key] = torch.cuda.Stream(device)
        return self._comm[key]

    def _all_reduce(self, t, async_op):
        op = dist.ReduceOp.AVG if self._avg else dist.ReduceOp.SUM
        self.last_bytes += t.numel() * t.element_size()
        return dist.all_reduce(t, op=op, group=self.group, async_op=async_op)

    def _issue(self, buf, lo, hi, stream):
        view = buf[lo:hi]
        if not buf.is_cuda:             # host tensors (the gloo tests): no streams to order, the collective is asynchronous
            self._works.append(self._all_reduce(view, True))
            return
        comm = self._comm_stream(buf.device)
        # the slice's writers ran on the main stream or on the weight-gradient side stream: wait for both
        streams = {stream, torch.cuda.current_stream(buf.device)}
        side = _A._side_streams.get((buf.device.type, buf.device.index))
        if side is not None:
            streams.add(side)
        for s_ in streams:
            ev = torch.cuda.Event()
            ev.record(s_)
            comm.wait_event(ev)
        with torch.cuda.stream(comm):
            self._works.append(self._all_reduce(view, True))

    # ---- the ranks agree on what they are about to all-reduce ------------------------------------------
    @staticmethod
    def _signature(values):
        h = 1469598103934665603
        for v in values:
            h = ((h ^ (int(v) & 0xffffffffffff)) * 1099511628211) & 0x3fffffffffffffff
        return h

    def _agree(self, sig, values, device, blocking):
        """all-reduce (MAX) of [sig, -sig]: equal on every rank iff max(sig) == -max(-sig).  Fixed size, so it pairs
        across ranks whatever their layouts are."""
        if blocking:
            # The signature is HOST data (which arena ranges and bucket this rank is about to post): its comparison need
            # not wait for the backward pass the main stream is still executing.  RCCL: the two words travel on a
            # control stream of their own (the process group runs its collectives in the order they are issued: this one
            # first, the payload behind the main stream's events), and the host waits for THAT stream only -- ~0.1 ms,
            # the device keeps its queue; reading the result on the main stream instead cost the step ~0.5 ms (round 5,
            # world size 1: the host sat out the whole backward pass before it could enqueue the optimizer).  Other
            # backends keep the words on the tensors' device and the main stream: gloo stages device tensors through the
            # host, and the one-GPU two-rank test harness crawls (minutes per step) when those staged copies run under
            # both processes' backward kernels -- reading the comparison there drains the device first, as it always did.
            if self._avg and device.type == 'cuda':
                ctrl = self._ctrl_stream(device)
                with torch.cuda.stream(ctrl):
                    t = torch.tensor([sig, -sig], dtype=torch.int64, device=device)
                    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
                    hi, neg_lo = t.tolist()
            else:
                t = torch.tensor([sig, -sig], dtype=torch.int64, device=device)
                dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
                hi, neg_lo = t.tolist()
            if hi != -neg_lo:
                raise RuntimeError(f'GradReducer: rank {self.rank} is about to all-reduce a gradient layout that differs '
                                   f'from another rank\'s (arena chunks in use / small-gradient bucket: {values}); every '
                                   'rank must run the same backward graph (no rank-dependent fallback paths, no unused '
                                   'parameters on some ranks only)')
            self._layout = sig
            return
        t = torch.tensor([sig, -sig], dtype=torch.int64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        if t.is_cuda:
            host = torch.empty(2, dtype=torch.int64).pin_memory()
            host.copy_(t, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            self._lazy = (host, ev, values)
        else:
            self._lazy = (t, None, values)

    def _check_lazy(self):
        if self._lazy is None:
            return
        host, ev, values = self._lazy
        self._lazy = None
        if ev is not None:
            ev.synchronize()            # one step old: completed long ago
        hi, neg_lo = host.tolist()
        if hi != -neg_lo:
            raise RuntimeError(f'GradReducer: the previous step all-reduced different gradient layouts on different ranks '
                               f'(this rank {self.rank}: {values}); the averaged gradients of that step are invalid')

    def _verify_issued(self, chunks):
        """every range that was all-reduced IN PLACE during the backward pass must be a parameter's `.grad` now.

        `writers_launched(..., in_place=True)` is a PREDICTION made at launch time (autograd._conv_backward's `takes`:
        leaf, no gradient yet, no hooks, the kernel's layout -- the conditions under which AccumulateGrad steals the
        arena view).  If it is wrong for a range (a torch release with another steal rule, a hook added later in the
        pass), the parameter's gradient is a COPY of the range made on the main stream, possibly after the slice's
        all-reduce wrote its result back; reduce() would then all-reduce that copy a second time through the small
        bucket -- round 4's `g0 + g1/2` (profiles/r05_notes.md), silently.  Checked on the host from the bookkeeping
        (no device work): raises before any further collective is posted."""
        for (buf, issued, used, live), pr in zip(chunks, self._progress):
            if not issued:
                continue
            lv = sorted(live)
            for lo, hi, param in pr[3]:
                # the part of this in-place range that went out during the pass ...
                sent = [(max(lo, a), min(hi, b)) for a, b in issued if a < hi and b > lo]
                # ... minus what a gradient refers to (the range's tail is alignment padding of < 64 elements)
                n = param.numel() if param is not None and 0 < param.numel() <= hi - lo else hi - lo
                left = [(a, b) for a, b in _subtract(sent, lv) if a < lo + n]
                if left:
                    name = None
                    if param is not None:
                        for i, q in enumerate(self.params):
                            if q is param or (q.numel() == param.numel() and
                                              q.untyped_storage().data_ptr() == param.untyped_storage().data_ptr()):
                                name = self.names.get(id(q), f'parameter #{i}')
                                break
                    shape = tuple(param.shape) if param is not None else '?'
                    raise RuntimeError(
                        f'GradReducer(overlap=True): arena elements {left[0]} of the weight gradient of '
                        f'{name or "a weight"} (shape {shape}, arena range [{lo}, {hi})) were all-reduced in place '
                        'during the backward pass, but no parameter\'s .grad refers to them: autograd did not take the '
                        'kernel\'s output as the gradient (the `takes` prediction of autograd._conv_backward was '
                        'wrong), so the gradient it kept instead would be reduced a second time.  Use overlap=False.')

    # ---- end of the backward pass -----------------------------------------------------------------
    @torch.no_grad()
    def reduce(self):
        """average every gradient over the ranks; call after backward(), before the optimizer step"""
        cuda = any(pr[0].is_cuda for pr in self._progress) or any(p.is_cuda for p in self.params)
        if cuda:
            _A.join_side_streams()              # the last weight-gradient launches, on whatever stream they ran
        if not self._progress:
            self.last_bytes = 0                 # (a step without an arena chunk: nothing was counted during backward)
        self._check_lazy()
        arena_ptrs, chunks = [], []
        for buf, issued, _, _ in self._progress:
            used = _A.grad_arena.used_of(buf)
            arena_ptrs.append((buf.untyped_storage().data_ptr(), buf.numel() * 4))
            chunks.append((buf, issued, used, []))
        # the gradients outside the arena (BatchNorm / GroupNorm affine, biases, Scale, FC weights of odd shapes), and --
        # per chunk -- the element ranges that a parameter's gradient actually refers to: a weight gradient that autograd
        # re-laid out on its way to the parameter (the first FC: (7,7,C) columns back to the reference's (C,7,7)) left a
        # DEAD slice behind in the arena, 51 MB that round 3 all-reduced for nothing
        small = []
        for p in self.params:
            g = p.grad
            if g is None:
                continue
            ptr = g.untyped_storage().data_ptr()
            hit = None
            for (a, n), ch in zip(arena_ptrs, chunks):
                if a <= ptr < a + n:
                    hit = ch
                    break
            if hit is None:
                small.append((p, g))
            else:
                lo = (g.data_ptr() - hit[0].data_ptr()) // 4
                hit[3].append((lo, lo + g.numel()))
        self._verify_issued(chunks)
        pending = []            # (chunk, first element, end element) to all-reduce
        for buf, issued, used, live in chunks:
            live.sort()
            cur = None
            for lo, hi in _subtract(live, issued):      # (the overlapped slices cut gradients wherever they fall)
                lo, hi = max(lo, 0), min(hi, used)
                if hi <= lo:
                    continue
                # alignment padding / tiny gaps: one collective -- unless the gap holds an issued slice's elements
                if cur is not None and lo - cur[1] <= 4096 and not any(a < lo and b > cur[1] for a, b in issued):
                    cur[1] = max(cur[1], hi)
                else:
                    if cur is not None:
                        pending.append((buf, cur[0], cur[1]))
                    cur = [lo, hi]
            if cur is not None:
                pending.append((buf, cur[0], cur[1]))
        n_small = sum(g.numel() for _, g in small)
        values = [v for _, lo, hi in pending for v in (lo, hi)] + [n_small, len(small)]
        sig = self._signature(values)
        device = (pending[0][0] if pending else small[0][1] if small else self.params[0]).device
        self._agree(sig, values, device, blocking=sig != self._layout or self.strict or device.type != 'cuda')
        # ---- arena
        if self.compress == 'bf16':
            total = sum(hi - lo for _, lo, hi in pending)
            if total:
                if self._cbuf is None or self._cbuf.numel() < total or self._cbuf.device != device:
                    self._cbuf = torch.empty(total, dtype=torch.bfloat16, device=device)
                off = 0
                for buf, lo, hi in pending:
                    self._cbuf[off:off + hi - lo].copy_(buf[lo:hi])     # fp32 -> bf16, round to nearest even
                    off += hi - lo
                self._works.append(self._all_reduce(self._cbuf[:total], True))
        else:
            for buf, lo, hi in pending:
                self._works.append(self._all_reduce(buf[lo:hi], True))
        # ---- small gradients: persistent flat bucket, `.grad` becomes a view of it
        if small:
            if self._bucket is None or self._bucket.numel() < n_small or self._bucket.device != device:
                self._bucket = torch.empty(n_small, dtype=torch.float32, device=device)
            flat = self._bucket[:n_small]
            views, off = [], 0
            for _, g in small:
                n = g.numel()
                views.append(flat[off:off + n].view(g.shape))
                off += n
            # gradients that already ARE the bucket's views (a second reduce without a new backward) stay put
            todo = [(v, g) for v, (_, g) in zip(views, small) if v.data_ptr() != g.data_ptr() or g.dtype != torch.float32]
            if todo:
                torch._foreach_copy_([v for v, _ in todo], [g for _, g in todo])   # multi-tensor copy (casts if needed)
            self._all_reduce(flat, False)
            if not self._avg:
                flat.mul_(1.0 / self.world)
            for v, (p, g) in zip(views, small):
                p.grad = v if g.dtype == torch.float32 else v.to(g.dtype)
        for w in self._works:
            w.wait()                            # the current stream waits for the collective (no host block on RCCL)
        if self.compress == 'bf16':
            off = 0
            for buf, lo, hi in pending:
                buf[lo:hi].copy_(self._cbuf[off:off + hi - lo])         # bf16 -> fp32 (exact)
                off += hi - lo
        if not self._avg:
            for buf, issued, used, _ in chunks:
                if used:
                    buf[:used].mul_(1.0 / self.world)
        self._works = []
        self._progress = []
        self._seen = {}
