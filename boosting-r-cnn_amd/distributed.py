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
  all-reduced on a communication stream behind an event of the stream that launched its last writer.  Measured at
  world size 1 on MI355X (profiles/r03_notes.md): with three streams already busy (main, weight gradients, proposals)
  a fourth stream that WAITS on a weight-gradient event costs the step 1.4-4 ms (19.7 -> 21.6-24 ms, depending on
  which pooled stream it is; the wait alone, without any RCCL call, 3.3 ms) -- more than the ~1.3 ms of ring
  all-reduce it could hide at 8 GPUs.  The single all-reduce after the backward pass costs 0.45 ms at world size 1;
* the average is RCCL's `ReduceOp.AVG` (sum, then one multiply, on backends without it).

One process per GPU; the image batch is sharded by the sampler, weights are replicated (`broadcast_parameters`).
"""
import torch
import torch.distributed as dist

from . import autograd as _A


class GradReducer:
    def __init__(self, params, process_group=None, slice_mb=64, overlap=False):
        if not (dist.is_available() and dist.is_initialized()):
            raise RuntimeError('GradReducer needs an initialised torch.distributed process group')
        self.params = [p for p in params if p.requires_grad]
        self.group = process_group
        self.world = dist.get_world_size(process_group)
        self.slice_elems = max(1, int(slice_mb * (1 << 20) // 4))
        self.overlap = bool(overlap)
        self._avg = dist.get_backend(process_group) == 'nccl'
        self._works = []
        self._comm = {}
        # chunk -> [elements already handed to a collective, elements whose writers have all been launched]
        self._progress = []
        _A._OWN_REDUCER[0] = True
        _A.grad_arena.listener = self

    def close(self):
        if _A.grad_arena.listener is self:
            _A.grad_arena.listener = None
        _A._OWN_REDUCER[0] = False

    # ---- parameters -----------------------------------------------------------------------------
    def broadcast_parameters(self, module, src=0):
        """rank `src`'s parameters and buffers to every rank (what DistributedDataParallel does at construction)"""
        tensors = [p.data for p in module.parameters()] + [b.data for b in module.buffers()]
        for t in tensors:
            dist.broadcast(t, src, group=self.group)

    # ---- arena listener (called by autograd._GradArena / _conv_backward) ---------------------------
    def chunk_opened(self, buf):
        self._progress.append([buf, 0, 0])

    def writers_launched(self, buf, upto, stream):
        """every weight-gradient launch writing arena elements [0, upto) of `buf` has been issued, the last one on
        `stream`: hand the finished slices to the communication stream"""
        for pr in self._progress:
            if pr[0] is buf:
                pr[2] = max(pr[2], upto)
                if self.overlap and buf.is_cuda:
                    while pr[2] - pr[1] >= self.slice_elems:
                        self._issue(buf, pr[1], pr[1] + self.slice_elems, stream)
                        pr[1] += self.slice_elems
                return

    def _comm_stream(self, device):
        key = (device.type, device.index)
        if key not in self._comm:
            self._comm[key] = torch.cuda.Stream(device)
        return self._comm[key]

    def _all_reduce(self, t, async_op):
        op = dist.ReduceOp.AVG if self._avg else dist.ReduceOp.SUM
        return dist.all_reduce(t, op=op, group=self.group, async_op=async_op)

    def _issue(self, buf, lo, hi, stream):
        view = buf[lo:hi]
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

    # ---- end of the backward pass -----------------------------------------------------------------
    @torch.no_grad()
    def reduce(self):
        """average every gradient over the ranks; call after backward(), before the optimizer step"""
        cuda = any(pr[0].is_cuda for pr in self._progress) or any(p.is_cuda for p in self.params)
        if cuda:
            _A.join_side_streams()              # the last weight-gradient launches, on whatever stream they ran
        arena_ptrs = []
        for buf, done, _ in self._progress:
            used = _A.grad_arena.used_of(buf)
            arena_ptrs.append((buf.untyped_storage().data_ptr(), buf.numel() * 4))
            if used > done:
                self._works.append(self._all_reduce(buf[done:used], True))
        # the gradients outside the arena: one flattened bucket
        small = []
        for p in self.params:
            g = p.grad
            if g is None:
                continue
            ptr = g.untyped_storage().data_ptr()
            if any(a <= ptr < a + n for a, n in arena_ptrs):
                continue
            small.append(g)
        if small:
            flat = torch.cat([g.reshape(-1).float() for g in small])
            self._all_reduce(flat, False)
            if not self._avg:
                flat.mul_(1.0 / self.world)
            views, off = [], 0
            for g in small:
                n = g.numel()
                views.append(flat[off:off + n].view_as(g))
                off += n
            torch._foreach_copy_(small, views)      # multi-tensor copy: a handful of launches for ~300 tensors
        for w in self._works:
            w.wait()                            # the current stream waits for the collective (no host block on RCCL)
        if not self._avg:
            for buf, _, _ in self._progress:
                used = _A.grad_arena.used_of(buf)
                if used:
                    buf[:used].mul_(1.0 / self.world)
        self._works = []
        self._progress = []
