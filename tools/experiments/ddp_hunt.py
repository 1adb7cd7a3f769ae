"""Localise the two-rank gradient mismatch of round 4: `neck.fpn_convs.4.conv.weight` of the
GradReducer leg off by 48 % of its largest entry against DistributedDataParallel, intermittently.

One rank of a `torch.distributed.run` launch (two ranks on one GPU through gloo: tests/test_ddp_gpu.py TWO_ON_ONE, or
world size 1 over nccl).  For every cell of

    reducer in {none, own (one all-reduce), own (overlapped slices), ddp} x early RPN backward {0,1} x
    weight-gradient side stream {0,1}

runs REPS train steps on the same seeded inputs and compares every parameter gradient with the reference the cell
must reproduce: the rank's own unwrapped gradients (`none`) or their mean over the ranks (reducers).  A gradient off by
more than TOL of the tensor's largest entry counts as bad; for the worst tensor of a bad repetition the least-squares
coefficients (a, b) of  bad ~ a * g_rank0 + b * g_rank1  are printed ((0.5, 0.5) is the right answer; (0.5, 0) = rank 1's
contribution missing; (1, 0.5) = rank 0 unscaled ...).

`own_overlap_r04` is the overlapped reducer with round 4's slicing (by arena offset alone) restored by a subclass in
this file; HUNT_STALL_MS=t holds the main stream back t ms behind every weight-gradient launch (the interleaving that
makes the r04 form fail on every step instead of once in ~20); HUNT_CHANNELS_LAST=1 runs the reducer cells with
channels-last conv weights (the layout of train_detector / bench.py).

    python -m torch.distributed.run --nproc-per-node 2 ... tools/experiments/ddp_hunt.py [reps] [cell prefixes]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import brcnn  # noqa: E402,F401
from brcnn import Config, build_detector  # noqa: E402
from brcnn import autograd as A  # noqa: E402
from brcnn.distributed import GradReducer  # noqa: E402
from tests import util  # noqa: E402

TOL = float(os.environ.get('HUNT_TOL', '2e-4'))


def main():
    # a hang (a collective one rank never posts) ends with every thread's stack on stderr instead of the launcher's kill
    import faulthandler
    faulthandler.dump_traceback_later(int(os.environ.get('DDP_WATCHDOG_S', '900')), exit=True)
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if os.environ.get('BRCNN_DIST_ONE_DEVICE', '0') == '1':
        local = 0
    backend = os.environ.get('BRCNN_DIST_BACKEND', 'nccl')
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    if backend == 'nccl':
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group(backend, rank=rank, world_size=world)
    # HUNT_TRACE_S=t: every collective this rank posts is logged (thread, size, async); after t seconds without progress a
    # daemon thread prints the tail of the log and the totals -- for a hang: do the two ranks stand at the same collective?
    trace_s = float(os.environ.get('HUNT_TRACE_S', '0'))
    if trace_s > 0:
        import threading
        import time
        log, last = [], [time.time()]
        orig_ar = dist.all_reduce

        def traced(t, *a, **kw):
            log.append((threading.current_thread().name, tuple(t.shape), str(t.dtype), bool(kw.get('async_op', False)), t.is_cuda))
            last[0] = time.time()
            return orig_ar(t, *a, **kw)
        dist.all_reduce = traced

        def watch():
            while True:
                time.sleep(5)
                if time.time() - last[0] > trace_s:
                    print(f'TRACE rank {rank}: {len(log)} collectives posted; last 14: {log[-14:]}', flush=True)
                    return
        threading.Thread(target=watch, daemon=True).start()
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    want = sys.argv[2].split(',') if len(sys.argv) > 2 else None
    dtype = os.environ.get('HUNT_DTYPE', 'f32')
    cfg = Config.fromfile(os.path.join(ROOT, 'configs', 'boosting_rcnn', 'boosting_rcnn_r50_pafpn_1x_utdac.py'))
    img, metas, gts, gls = util.demo_inputs(2, 128, 192, seed=10 + rank)
    data = dict(img=img.to(dev), img_metas=metas, gt_bboxes=[b.to(dev) for b in gts], gt_labels=[l.to(dev) for l in gls])

    def fresh():
        m = build_detector(cfg.model)
        m.load_state_dict(util.seeded_state_dict(m, seed=10))
        m = m.to(dev).train()
        m.set_compute_dtype(dtype)
        return m

    def step(m, net, red, early):
        m.early_rpn_backward = early
        m.zero_grad(set_to_none=True)
        A.grad_arena.new_step()
        torch.manual_seed(77)
        losses = net(return_loss=True, **data)
        loss, _ = m._parse_losses(losses)
        loss.backward()
        if red is not None:
            red.reduce()
        torch.cuda.synchronize()
        return {k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None}

    # ---- references: the unwrapped step, a few times (its own repeatability = the noise floor of the atomics)
    m = fresh()
    A.WGRAD_SIDE_STREAM = False
    local_ref = step(m, m, None, False)
    noise = 0.0
    for _ in range(3):
        g = step(m, m, None, False)
        for k, v in g.items():
            noise = max(noise, (v - local_ref[k]).abs().max().item() / (local_ref[k].abs().max().item() + 1e-12))
    both = {}
    for k, v in local_ref.items():
        parts = [torch.empty_like(v) for _ in range(world)]
        dist.all_gather(parts, v.contiguous())
        both[k] = parts
    mean_ref = {k: sum(p) / world for k, p in both.items()}
    if rank == 0:
        print(f'HUNT world={world} backend={backend} dtype={dtype} reps={reps} tol={TOL} unwrapped-step noise={noise:.2e}', flush=True)
    del m

    stall_ms = float(os.environ.get('HUNT_STALL_MS', '0'))
    stall_cycles = 0
    if stall_ms > 0:        # main stream held back behind every weight-gradient launch (autograd._TEST_STALL_CYCLES)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda._sleep(1000)
        e0.record()
        torch.cuda._sleep(2_000_000)
        e1.record()
        e1.synchronize()
        stall_cycles = max(1000, int(stall_ms * 2_000_000 / max(e0.elapsed_time(e1), 1e-3)))
        if rank == 0:
            print(f'HUNT stall {stall_ms} ms = {stall_cycles} ticks behind every weight-gradient launch of a reducer cell', flush=True)
    cl = os.environ.get('HUNT_CHANNELS_LAST', '0') == '1'

    class R04Reducer(GradReducer):
        """round 4's slicing: by arena offset alone, whether autograd takes the range as `.grad` or copies it"""
        def writers_launched(self, buf, lo, hi, stream, in_place=True, param=None):
            return super().writers_launched(buf, lo, hi, stream, True, param)

    cells = []
    for red in ('none', 'own', 'own_overlap', 'own_overlap_r04', 'ddp'):
        for early in (0, 1):
            for side in (0, 1):
                if red == 'ddp' and (early or side):
                    continue
                cells.append((red, early, side))
    for red_kind, early, side in cells:
        name = f'{red_kind}:early{early}:side{side}'
        if want is not None and not any(name.startswith(w) for w in want):
            continue
        A.WGRAD_SIDE_STREAM = bool(side)
        m = fresh()
        net, red = m, None
        if red_kind == 'ddp':
            net = torch.nn.parallel.DistributedDataParallel(m, device_ids=[local], broadcast_buffers=False)
        elif red_kind != 'none':
            cls = R04Reducer if red_kind == 'own_overlap_r04' else GradReducer
            red = cls([p for p in m.parameters() if p.requires_grad], slice_mb=float(os.environ.get('HUNT_SLICE_MB', '8')), overlap=red_kind != 'own')
            red.broadcast_parameters(m)
            if cl:
                from brcnn.blocks import conv_weights_channels_last
                conv_weights_channels_last(m)
            A._TEST_STALL_CYCLES = stall_cycles
        ref = local_ref if red_kind == 'none' else mean_ref
        bad_reps, worst, reports = 0, 0.0, []
        import time as _time
        t_cell = _time.time()
        for r in range(reps):
            t_rep = _time.time()
            g = step(m, net, red, bool(early))
            if os.environ.get('HUNT_PROGRESS') == '1' and rank == 0 and _time.time() - t_rep > 5:
                print(f'  SLOW rep {r} of {name}: {_time.time() - t_rep:.1f} s (cell so far {_time.time() - t_cell:.0f} s)', flush=True)
            bad = []
            for k, v in g.items():
                d = (v - ref[k]).abs().max().item() / (ref[k].abs().max().item() + 1e-12)
                worst = max(worst, d)
                if d > TOL:
                    bad.append((d, k))
            flag = torch.tensor([len(bad)], device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
            if flag.item():
                bad_reps += 1
            if bad and len(reports) < 3:
                bad.sort(reverse=True)
                lines = []
                for d, k in bad[:6]:
                    v = g[k].flatten().double()
                    basis = torch.stack([p.flatten().double() for p in both[k]], 1)       # (n, world)
                    coef = torch.linalg.lstsq(basis, v[:, None]).solution.flatten().tolist()
                    resid = (basis @ torch.tensor(coef, device=dev, dtype=torch.float64) - v).abs().max().item() / \
                        (ref[k].abs().max().item() + 1e-12)
                    nz = (g[k] != 0).float().mean().item()
                    lines.append(f'    {k} {tuple(g[k].shape)} dev={d:.3e} coef={[round(c, 4) for c in coef]} resid={resid:.2e} nonzero={nz:.3f}')
                reports.append(f'  rank {rank} rep {r}: {len(bad)} bad tensors\n' + '\n'.join(lines))
        if red is not None:
            red.close()
        A._TEST_STALL_CYCLES = 0
        del m, net, red
        torch.cuda.synchronize()
        out = [None] * world
        dist.all_gather_object(out, (worst, reports))
        if rank == 0:
            print(f'CELL {name:28s} bad_reps={bad_reps}/{reps} worst={max(o[0] for o in out):.3e}', flush=True)
            for o in out:
                for rep in o[1]:
                    print(rep, flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
