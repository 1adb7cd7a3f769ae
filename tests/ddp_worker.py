"""One rank of the data-parallel gradient check (launched by tests/test_ddp_gpu.py through
`python -m torch.distributed.run`, backend nccl = RCCL, or gloo with both ranks on one GPU): the device-resident train
step -- custom autograd Functions over the HIP kernels, the fused RPN loss with its in-forward all-reduce of the two
normalisers, the host-synchronised sampler -- under DistributedDataParallel's hooks and under the repo's own
GradReducer, each compared FIRST with the mean over the ranks of the unwrapped step's gradients (so that a failure
names the leg that deviates) and then with each other.

    ddp_worker.py <f32|bf16> [legs]        legs: comma list of ddp, own, overlap (default: all three)

The reducer legs run four passes: {early RPN backward off, on} x {weights as loaded, spatial conv weights
channels-last} -- as loaded, autograd COPIES most weight gradients out of the arena (their strides are not the
parameters'), channels-last is the layout train_detector / bench.py run, where the arena slices themselves are `.grad`
and the overlapped form all-reduces them in place during the backward pass.
DDP_WORKER_STALL_MS=t holds the main stream back for t ms behind every weight-gradient launch of the reducer legs
(autograd._TEST_STALL_CYCLES): whatever autograd queues there next -- its copies of dW -- then runs after an
overlapped all-reduce of the same arena slice has completed: the interleaving behind round 4's intermittent mismatch,
forced on every launch (test_two_rank_overlap_with_stalled_main_stream).
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import brcnn  # noqa: E402,F401
from brcnn import Config, build_detector  # noqa: E402
from tests import util  # noqa: E402

# relative to the largest entry of the tensor.  fp32: the unwrapped step repeats itself to 2e-7 (fp32 atomics order,
# tools/experiments/ddp_hunt.py on MI355X: every healthy cell of the matrix 2.0e-7), a wrong average is off by > 1e-1
TOL = {'f32': 2e-5, 'bf16': 3e-2}


def main():
    # a hang (a collective one rank never posts) ends with every thread's stack on stderr instead of the launcher's kill
    import faulthandler
    faulthandler.dump_traceback_later(int(os.environ.get('DDP_WATCHDOG_S', '900')), exit=True)
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    local = int(os.environ.get('LOCAL_RANK', '0'))
    # BRCNN_DIST_ONE_DEVICE=1 + BRCNN_DIST_BACKEND=gloo: every rank on cuda:0, collectives through gloo (RCCL refuses two
    # ranks on one device) -- the world-size-2 control flow on a one-GPU box
    if os.environ.get('BRCNN_DIST_ONE_DEVICE', '0') == '1':
        local = 0
    backend = os.environ.get('BRCNN_DIST_BACKEND', 'nccl')
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    if backend == 'nccl':
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group(backend, rank=rank, world_size=world)
    cfg = Config.fromfile(os.path.join(ROOT, 'configs', 'boosting_rcnn', 'boosting_rcnn_r50_pafpn_1x_utdac.py'))
    dtype = sys.argv[1] if len(sys.argv) > 1 else 'f32'
    legs = sys.argv[2].split(',') if len(sys.argv) > 2 else ['ddp', 'own', 'overlap']
    tol_rel = TOL[dtype]
    img, metas, gts, gls = util.demo_inputs(2, 128, 192, seed=10 + rank)
    data = dict(img=img.to(dev), img_metas=metas, gt_bboxes=[b.to(dev) for b in gts], gt_labels=[l.to(dev) for l in gls])
    from brcnn import autograd as A
    from brcnn.blocks import conv_weights_channels_last
    from brcnn.distributed import GradReducer

    def fresh():
        m = build_detector(cfg.model)
        m.load_state_dict(util.seeded_state_dict(m, seed=10))
        m = m.to(dev).train()
        m.set_compute_dtype(dtype)
        return m

    def step(m, net, red, early):
        m.early_rpn_backward = early        # the RPN branch back-propagated inside the forward pass (detectors.py)
        m.zero_grad(set_to_none=True)
        A.grad_arena.new_step()
        torch.manual_seed(77)
        losses = net(return_loss=True, **data)
        loss, log_vars = m._parse_losses(losses)
        loss.backward()
        if red is not None:
            assert A._wgrad_side_stream(dev) is not None or not A.WGRAD_SIDE_STREAM     # the second stream stays on
            red.reduce()
        torch.cuda.synchronize()
        return {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}, dict(log_vars)

    def close(what, got, want, tol_scale=1.0):
        assert got.keys() == want.keys(), what
        worst = (-1.0, None)
        for k, g in want.items():
            d = (got[k].float() - g.float()).abs().max().item() / (g.abs().max().item() + 1e-12)
            if d > worst[0]:
                worst = (d, k)
        assert worst[0] <= tol_rel * tol_scale, (what, 'worst tensor', worst[1], 'relative deviation', worst[0], 'tolerance', tol_rel * tol_scale)

    stall_ms = float(os.environ.get('DDP_WORKER_STALL_MS', '0'))
    stall_cycles = 0
    if stall_ms > 0:            # torch.cuda._sleep counts device clock ticks: calibrate
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda._sleep(1000)
        e0.record()
        torch.cuda._sleep(2_000_000)
        e1.record()
        e1.synchronize()
        stall_cycles = max(1000, int(stall_ms * 2_000_000 / max(e0.elapsed_time(e1), 1e-3)))

    # ---- the unwrapped step, and the mean of its gradients over the ranks = what every leg has to deliver
    m = fresh()
    plain, plain_logs = step(m, m, None, False)
    del m
    mean = {}
    for k, g in plain.items():
        t = g.clone()
        dist.all_reduce(t)
        mean[k] = t / world
    results = {}
    if 'ddp' in legs:
        m = fresh()
        net = torch.nn.parallel.DistributedDataParallel(m, device_ids=[local], broadcast_buffers=False)
        results['ddp'], ddp_logs = step(m, net, None, False)
        close('DistributedDataParallel vs mean of the unwrapped gradients', results['ddp'], mean)
        for k, v in plain_logs.items():         # the loss values themselves do not depend on the wrapper
            assert abs(v - ddp_logs[k]) <= (1e-5 if dtype == 'f32' else 1e-3) * max(1.0, abs(v)), k
        del m, net
    for leg in ('own', 'overlap'):
        if leg not in legs:
            continue
        m = fresh()
        # `own`: the N > 1 path of bench.py / train_detector (one all-reduce of the arena after the backward pass);
        # `overlap`: finished arena slices all-reduced in place behind the backward pass
        # (64 MiB slices, the default: two or three in flight per step.  With 8 MiB slices -- ~20 staged gloo copies in
        # flight per rank, both ranks on one GPU -- a step of this HARNESS takes 15-60 s: measured, profiles/r05_notes.md;
        # RCCL does not stage through the host)
        red = GradReducer([p for p in m.parameters() if p.requires_grad], slice_mb=64, overlap=leg == 'overlap')
        red.broadcast_parameters(m)
        A._TEST_STALL_CYCLES = stall_cycles
        for cl in (False, True):
            if cl:
                assert conv_weights_channels_last(m) > 0
            for early in (False, True):
                g, _ = step(m, m, red, early)
                close(f'GradReducer({leg}) early_rpn_backward={early} channels_last={cl} vs mean of the unwrapped gradients', g, mean)
                results[leg] = g
        if leg == 'overlap':
            d = red.describe()
            assert d['overlap'] and d['bytes_last_step'] > 100e6, d
        red.close()
        A._TEST_STALL_CYCLES = 0
        del m, red
    if 'ddp' in results:
        for leg in ('own', 'overlap'):
            if leg in results:
                close(f'GradReducer({leg}) vs DistributedDataParallel', results[leg], results['ddp'], 2.0)
    # every rank ends with identical (averaged) gradients
    for leg, grads in results.items():
        for k, g in sorted(grads.items())[:8]:
            t = g.clone()
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            assert torch.equal(t, g), (leg, k)
    if rank == 0:
        print('DDP_OK', len(mean), sorted(results), plain_logs['loss'], flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
