"""One rank of the DistributedDataParallel check (launched by tests/test_ddp_gpu.py through
`python -m torch.distributed.run`, backend nccl = RCCL): the device-resident train step -- custom autograd
Functions over the HIP kernels, the fused RPN loss with its in-forward all-reduce of the two normalisers,
the host-synchronised sampler -- under DDP's gradient hooks, against the same step without the wrapper."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import brcnn  # noqa: E402,F401
from brcnn import Config, build_detector  # noqa: E402
from tests import util  # noqa: E402


def main():
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
    img, metas, gts, gls = util.demo_inputs(2, 128, 192, seed=10 + rank)
    data = dict(img=img.to(dev), img_metas=metas, gt_bboxes=[b.to(dev) for b in gts], gt_labels=[l.to(dev) for l in gls])
    grads = {}
    from brcnn import autograd as A
    from brcnn.distributed import GradReducer
    for mode in ('plain', 'ddp', 'own'):
        m = build_detector(cfg.model)
        m.load_state_dict(util.seeded_state_dict(m, seed=10))
        m = m.to(dev).train()
        m.set_compute_dtype(dtype)
        net, red = m, None
        if mode == 'ddp':
            net = torch.nn.parallel.DistributedDataParallel(m, device_ids=[local], broadcast_buffers=False)
        if mode == 'own':       # the N > 1 path of bench.py / train_detector: arena all-reduced in place, no hooks
            red = GradReducer([p for p in m.parameters() if p.requires_grad], slice_mb=8, overlap=True)   # (the sliced form; the default reduces once at the end)
            red.broadcast_parameters(m)
        for rep in range(2 if mode == 'own' else 1):     # twice: the second pass runs on a fresh arena chunk
            # ... and with the RPN branch back-propagated inside the forward pass (detectors.py): its weight gradients
            # reach the arena, and the reducer, before the second stage's
            m.early_rpn_backward = mode == 'own' and rep == 1
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
        grads[mode] = ({k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}, dict(log_vars))
        if red is not None:
            red.close()
    assert grads['plain'][0].keys() == grads['ddp'][0].keys() == grads['own'][0].keys()
    if world == 1:          # one rank: the wrapper must not change a single value
        for k, g in grads['plain'][0].items():      # (weight gradients accumulate with fp32 atomics: order varies)
            tol = (1e-4 if dtype == 'f32' else 2e-2) * (g.abs().max().item() + 1e-12)
            assert (g - grads['ddp'][0][k]).abs().max().item() <= tol, (k, (g - grads['ddp'][0][k]).abs().max().item(), tol)
        for k, v in grads['plain'][1].items():
            assert abs(v - grads['ddp'][1][k]) <= (1e-5 if dtype == 'f32' else 1e-3) * max(1.0, abs(v)), k
        for k, g in grads['plain'][0].items():
            tol = (1e-4 if dtype == 'f32' else 2e-2) * (g.abs().max().item() + 1e-12)
            assert (g - grads['own'][0][k]).abs().max().item() <= tol, ('own', k, (g - grads['own'][0][k]).abs().max().item(), tol)
    else:
        # several ranks, each on its own images: DDP's and the reducer's averaged gradients agree, and both equal the
        # mean over the ranks of the unwrapped step's gradients (the second `own` pass also back-propagates the RPN
        # branch inside the forward pass)
        for k, g in grads['ddp'][0].items():
            # (fp32 weight gradients accumulate with atomics and the three passes run on two processes sharing the device:
            # the order of the sums varies from pass to pass; 2e-4 of the largest entry was exceeded in 2 of ~10 suite runs,
            # 1e-3 is the tolerance the north star states for fp32 tensors)
            tol = (1e-3 if dtype == 'f32' else 3e-2) * (g.abs().max().item() + 1e-12)
            assert (g - grads['own'][0][k]).abs().max().item() <= tol, ('own vs ddp', k, (g - grads['own'][0][k]).abs().max().item(), tol)
            t = grads['plain'][0][k].clone()
            dist.all_reduce(t)
            assert (g - t / world).abs().max().item() <= tol, ('ddp vs mean of plain', k, (g - t / world).abs().max().item(), tol)
    # every rank ends with identical (averaged) gradients
    for mode in ('ddp', 'own'):
        for k, g in sorted(grads[mode][0].items())[:8]:
            t = g.clone()
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            assert torch.equal(t, g), (mode, k)
    if rank == 0:
        print('DDP_OK', len(grads['ddp'][0]), grads['ddp'][1]['loss'], flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
