"""Benchmark of the Boosting R-CNN hot path on MI355X.

A "step" is one inference pass (`model(return_loss=False, rescale=True, ...)`, the protocol of
tools/analysis_tools/benchmark.py:98-131) of the UTDAC R50-PAFPN Boosting R-CNN over a batch
of 8 synthetic 1333x800 images (BASELINE.json configs[1]), inputs resident in HBM.  Prints
ONE JSON line (see the contract in the task statement) with `roofline` (dominant kernel: the
fp32 MFMA implicit-GEMM conv) and `cpu_baseline` (the oracle pipeline on the host cores).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def synthetic_batch(batch, device, seed=0):
    g = torch.Generator().manual_seed(seed)
    img = torch.randn(batch, 3, 800, 1344, generator=g).to(device)
    metas = [dict(img_shape=(800, 1333, 3), pad_shape=(800, 1344, 3), ori_shape=(800, 1333, 3),
                  scale_factor=np.array([1., 1., 1., 1.], dtype=np.float32), flip=False)
             for _ in range(batch)]
    return img, metas


def build_model(device, seed=0):
    import brcnn  # noqa: F401
    from brcnn import Config, build_detector
    from tests import util
    cfg = Config.fromfile(os.path.join(ROOT, 'configs', 'boosting_rcnn',
                                       'boosting_rcnn_r50_pafpn_1x_utdac.py'))
    m = build_detector(cfg.model)
    m.load_state_dict(util.seeded_state_dict(m, seed=seed))
    return m.to(device).eval().freeze_for_inference(), cfg


def synthetic_gt(batch, device, num_classes, seed=0, num_gt=20):
    """20 boxes / image in the style of tests/test_models/test_forward.py:481-492"""
    rng = np.random.RandomState(seed)
    boxes, labels = [], []
    for _ in range(batch):
        cx, cy = rng.rand(num_gt) * 1333, rng.rand(num_gt) * 800
        bw, bh = rng.rand(num_gt) * 1333 * 0.5 + 8, rng.rand(num_gt) * 800 * 0.5 + 8
        b = np.stack([(cx - bw / 2).clip(0, 1333), (cy - bh / 2).clip(0, 800),
                      (cx + bw / 2).clip(0, 1333), (cy + bh / 2).clip(0, 800)], 1).astype(np.float32)
        boxes.append(torch.from_numpy(b).to(device))
        labels.append(torch.from_numpy(rng.randint(0, num_classes, num_gt)).long().to(device))
    return boxes, labels


def train_bench(args, world, rank, device):
    """full train step of boosting_rcnn_r50_pafpn_1x_coco.py (80 classes), fp32, SGD + clip"""
    import brcnn  # noqa: F401
    from brcnn import Config, build_detector
    from tests import util
    cfg = Config.fromfile(os.path.join(ROOT, 'configs', 'boosting_rcnn', 'boosting_rcnn_r50_pafpn_1x_coco.py'))
    model = build_detector(cfg.model)
    model.load_state_dict(util.seeded_state_dict(model, seed=0))
    model = model.to(device).train()
    model.set_compute_dtype(args.dtype)     # bf16: conv stack fwd/dgrad/wgrad on bf16 MFMA, fp32 master weights
    params = [p for p in model.parameters() if p.requires_grad]
    opt = torch.optim.SGD(params, lr=cfg.optimizer.lr * 1e-3, momentum=cfg.optimizer.momentum,
                          weight_decay=cfg.optimizer.weight_decay)
    net = model
    if world > 1:
        net = torch.nn.parallel.DistributedDataParallel(model, device_ids=[device.index],
                                                        broadcast_buffers=False)
    img, metas = synthetic_batch(args.batch, device, seed=rank)
    gtb, gtl = synthetic_gt(args.batch, device, 80, seed=rank)

    def step():
        opt.zero_grad(set_to_none=True)
        losses = net(img=img, img_metas=metas, return_loss=True, gt_bboxes=gtb, gt_labels=gtl)
        loss, log_vars = model._parse_losses(losses)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(params, max_norm=35, norm_type=2)
        opt.step()
        return log_vars

    for _ in range(args.warmup):
        lv = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        lv = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if rank == 0:
        print(json.dumps({
            'metric': 'images/sec (1333x800) Boosting R-CNN R50-PAFPN train step',
            'value': world * args.batch * args.steps / dt, 'unit': 'images/sec', 'n_gpus': world,
            'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': 1000.0 * dt / args.steps,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': args.dtype,
            'data': 'synthetic',
            'config': {'workload': f'boosting_rcnn_r50_pafpn_1x_coco.py full train step, batch {args.batch} x '
                                   f'3x800x1344 per GPU, 20 GT/img, 512 RoIs/img, SGD+clip, {args.dtype} conv stack',
                       'global_batch': world * args.batch, 'parallelism': f'dp{world}'},
            'loss': lv['loss']}))
    if world > 1:
        dist.destroy_process_group()


def conv_flops_per_image():
    """algorithmic MACs of the conv/FC stack per image (170.0 GMAC at 256 RoIs, C=4)"""
    return 2 * 170.0e9


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=8)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--dtype', choices=['f32', 'bf16'], default='f32',
                    help="arithmetic type of the conv stack at inference: 'f32' (BASELINE configs[1], the "
                         "parity path, default) or 'bf16' (bf16 MFMA, fp32 accumulate; proposal stage, "
                         'heads outputs and NMS stay fp32)')
    ap.add_argument('--mode', choices=['inference', 'train'], default='inference',
                    help="'train' times the full train step (BASELINE configs[2]/[3]): forward_train, "
                         'backward through the HIP dgrad/wgrad kernels, gradient all-reduce (DDP over '
                         'RCCL when N>1), grad-clip, SGD')
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    torch.cuda.set_device(local_rank)
    device = torch.device('cuda', local_rank)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('nccl', rank=rank, world_size=world)

    if args.mode == 'train':
        return train_bench(args, world, rank, device)
    model, cfg = build_model(device)
    model.set_compute_dtype(args.dtype)
    img, metas = synthetic_batch(args.batch, device, seed=rank)

    def step():
        with torch.no_grad():
            return model.simple_test_device(img, metas, rescale=True)

    for _ in range(args.warmup):
        out = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
        nd = out[2].cpu()        # results leave the device every step, as in benchmark.py
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # ---- roofline of the dominant kernel: the conv stack, timed live with HIP events --------
    from brcnn import profiling
    roof = profiling.conv_stack_roofline(model, img, metas, iters=3, dtype=args.dtype)

    line = {
        'metric': 'images/sec (1333x800) Boosting R-CNN R50-PAFPN inference',
        'value': world * args.batch * args.steps / dt,
        'unit': 'images/sec',
        'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': 1000.0 * dt / args.steps,
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': args.dtype, 'data': 'synthetic',
        'config': {'workload': 'boosting_rcnn_r50_pafpn_1x_utdac.py inference (simple_test, rescale), '
                               f'batch {args.batch} x 3x800x1344 per GPU, {args.dtype} MFMA conv stack, '
                               '1000 pre-NMS / 256 proposals per image, seeded synthetic weights',
                   'global_batch': world * args.batch, 'parallelism': f'dp{world}'},
        'roofline': roof,
        'detections_last_step': int(nd.sum()),
    }
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            from oracle import cpu_pipeline
            line['cpu_baseline'] = cpu_pipeline.timed_baseline(cfg, seed=0)
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
