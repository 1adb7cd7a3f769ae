"""Benchmark of the Boosting R-CNN hot path on MI355X.

`python bench.py --gpus N --steps K --warmup W` prints ONE JSON line.

* headline (`value`): inference, one "step" = one pass (`model(return_loss=False, rescale=True)`, the
  protocol of tools/analysis_tools/benchmark.py:98-131) of the UTDAC R50-PAFPN Boosting R-CNN over a
  batch of 8 synthetic 1333x800 images per GPU (BASELINE.json configs[1]), inputs resident in HBM,
  results (boxes, labels, counts) copied to the host every step;
* `train`: the other half of the metric ("inference+train"), timed in the same run: the full train step
  of boosting_rcnn_r50_pafpn_1x_coco.py (BASELINE configs[2]/[3]: bf16 MFMA conv stack, fp32 master
  weights; forward_train with device-resident targets and losses, backward through the HIP dgrad /
  wgrad kernels, in-place all-reduce of the gradient arena over RCCL when N > 1 (brcnn/distributed.py), grad-clip, SGD);
* `roofline` (dominant kernel: the MFMA implicit-GEMM conv, HIP-event timed) and `cpu_baseline` (the
  oracle pipeline on the host cores, rank 0 at N=1 only).

N > 1: launched by the driver as `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`
(RANK / LOCAL_RANK / WORLD_SIZE in the environment); run plainly with `--gpus N`, this file starts that
launcher itself as a child process BEFORE anything touches the GPU and exits with its code.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# HIP streams share the hardware queues round-robin (4 by default): with RCCL's own streams next to the main stream,
# the weight-gradient stream and the reducer's communication stream, two of them would land on one queue and
# serialise.  Read by the HIP runtime when it initialises, i.e. before the first device call below.
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')


def _host_cpus():
    """host cores this process may actually use: min(affinity mask, cgroup cpu quota)"""
    import math
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            n = min(n, max(1, math.ceil(int(quota) / int(period))))
    except Exception:
        pass
    return n


# torch's intra-op pool defaults to one thread per core of the HOST (256 on the MI355X boxes) whatever CPU quota the
# container has (16 there: /sys/fs/cgroup/cpu.max 1600000 100000).  A parallel region on 256 threads burns the quota of a
# 100 ms scheduling period at once, and the whole process -- the thread that launches kernels included -- is throttled
# until the period ends (cpu.stat of one bench call: 171 of 1252 periods throttled).  Read by OpenMP when torch loads.
os.environ.setdefault('OMP_NUM_THREADS', str(_host_cpus()))


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=8)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-bs1', action='store_true',
                    help='skip the batch-1 latency loop (profiling runs: tools/profile_round.sh takes the per-pass kernel '
                         'statistics of the batch-8 workload from a run without it)')
    ap.add_argument('--dtype', choices=['f32', 'bf16', 'f16'], default='f32',
                    help="arithmetic type of the conv stack at inference: 'f32' (BASELINE configs[1], the parity "
                         "path, default), 'bf16' / 'f16' (16-bit MFMA, fp32 accumulate; proposal stage, head outputs "
                         'and NMS stay fp32)')
    ap.add_argument('--train-dtype', choices=['f32', 'bf16', 'f16'], default='bf16',
                    help='conv-stack arithmetic of the train step (configs[2] names bf16)')
    ap.add_argument('--mode', choices=['both', 'inference', 'train'], default='both',
                    help="'both' (default): inference headline + the train step in the same JSON line")
    ap.add_argument('--train-steps', type=int, default=None, help='timed train steps (default: --steps)')
    ap.add_argument('--force-reducer', action='store_true',
                    help='N = 1 only: initialise a world-size-1 RCCL group and run the train step through the gradient '
                         'reducer of the N > 1 path (its overhead against the plain step)')
    return ap.parse_args()


def launch_ranks(args):
    """`--gpus N` without a launcher: start N fresh rank processes (one per GPU) through
    torch.distributed.run -- tools/dist_train.sh:8-9 in the reference -- before any HIP call"""
    import socket
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}',
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    env['OMP_NUM_THREADS'] = str(max(1, _host_cpus() // max(1, args.gpus)))     # the ranks share the container's cores
    return subprocess.call(cmd, env=env)


def synthetic_batch(batch, device, seed=0):
    import numpy as np
    import torch
    g = torch.Generator().manual_seed(seed)
    img = torch.randn(batch, 3, 800, 1344, generator=g).to(device)
    metas = [dict(img_shape=(800, 1333, 3), pad_shape=(800, 1344, 3), ori_shape=(800, 1333, 3),
                  scale_factor=np.array([1., 1., 1., 1.], dtype=np.float32), flip=False)
             for _ in range(batch)]
    return img, metas


def build_model(cfg_name, device, seed=0):
    import brcnn  # noqa: F401
    from brcnn import Config, build_detector
    from brcnn.synth import seeded_state_dict
    cfg = Config.fromfile(os.path.join(ROOT, 'configs', 'boosting_rcnn', cfg_name))
    m = build_detector(cfg.model)
    m.load_state_dict(seeded_state_dict(m, seed=seed))
    return m.to(device), cfg


def synthetic_gt(batch, device, num_classes, seed=0, num_gt=20):
    """20 boxes / image in the style of tests/test_models/test_forward.py:481-492"""
    import numpy as np
    import torch
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


def timed(step, steps, warmup, world, device, after_warmup=None):
    """W untimed steps, then exactly K steps bracketed by barrier + synchronize; max over ranks.  Returns (seconds,
    per-step statistics): a HIP event on the main stream after every step gives each step's DEVICE time (median / p90 /
    max say whether the mean is the steady state or carries a hiccup), the host marks give the enqueue time per step"""
    import torch
    import torch.distributed as dist
    # Python's cyclic collector is held off over the timed steps (BRCNN_BENCH_GC=1 leaves it on): a step creates a few
    # thousand short-lived objects, every few steps a generation-1/2 pass walks the whole heap (model, configs, caches) for
    # 2-5 ms of HOST time -- which a launch-bound second half of the step turns into device time (profiles/r05_notes.md).
    # The runner does the same (`apis.EpochBasedRunner.train`: collector off inside the steps, one collection at every log
    # interval -- the default of 50 steps is longer than this timed region), so the line is measured under the policy real
    # training runs with; `gc_in_timed_region` in the line says which it was.
    # The full collection (~80 ms of host time) runs BEFORE the warm-up steps: between the warm-up and the timed region it
    # left the device idle long enough to drop its clocks, and the first timed step paid 2.7 ms for the ramp
    # (BRCNN_BENCH_DUMP_STEPS=1: 27.3 ms against 24.6 for every other step, three runs of three)
    import gc
    manual_gc = os.environ.get('BRCNN_BENCH_GC', '0') != '1'
    manual_gc = manual_gc and gc.isenabled()
    if manual_gc:
        gc.collect()
        gc.disable()
    try:
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
        for _ in range(warmup):
            step()
        if after_warmup is not None:    # (cheap host-side bookkeeping of the caller: no device idle time in front of the timed steps)
            after_warmup()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        evs[0].record()
        marks = []
        for i in range(steps):
            step()
            evs[i + 1].record()
            marks.append(time.perf_counter())      # host time after the step's enqueue (no synchronisation)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    finally:
        if manual_gc:
            gc.enable()
    dev_ms = sorted(evs[i].elapsed_time(evs[i + 1]) for i in range(steps))
    gaps = [b - a for a, b in zip([t0] + marks[:-1], marks)]
    host_ms = sorted(1e3 * g for g in gaps)
    stats = {'step_ms_median': dev_ms[steps // 2], 'step_ms_p90': dev_ms[min(steps - 1, (9 * steps) // 10)],
             'step_ms_min': dev_ms[0], 'step_ms_max': dev_ms[-1],
             'host_enqueue_ms_median': host_ms[steps // 2], 'host_enqueue_ms_max': host_ms[-1],
             # 'runner policy': collector off inside the steps, as apis.EpochBasedRunner.train runs them; 'on': BRCNN_BENCH_GC=1
             'gc_in_timed_region': 'runner policy (off inside steps)' if manual_gc else 'on'}
    # a one-off stall inside the timed region (seen on fresh boxes: up to ~2 s in one step) is reported, never removed:
    # the line still times exactly K steps
    med = dev_ms[steps // 2]
    if os.environ.get('BRCNN_BENCH_DUMP_STEPS') == '1':      # the steps in order (device ms / host ms), to stderr
        print('bench: steps ' + ' '.join(f'{evs[i].elapsed_time(evs[i + 1]):.2f}/{1e3 * gaps[i]:.2f}' for i in range(steps)),
              file=sys.stderr, flush=True)
    for i in range(steps):
        d = evs[i].elapsed_time(evs[i + 1])
        if d > 5 * med or 1e3 * gaps[i] > max(5 * host_ms[steps // 2], 5 * med):
            print(f'bench: step {i} of {steps}: {d:.1f} ms on the device, {1e3 * gaps[i]:.1f} ms on the host (medians {med:.1f} / '
                  f'{host_ms[steps // 2]:.1f} ms): a one-off stall is inside the timed region', file=sys.stderr, flush=True)
    if world > 1:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt, stats


def train_bench(args, world, rank, device):
    """full train step of boosting_rcnn_r50_pafpn_1x_coco.py (80 classes): SGD + clip, DDP when N > 1"""
    import torch
    model, cfg = build_model('boosting_rcnn_r50_pafpn_1x_coco.py', device)
    model = model.train()
    model.set_compute_dtype(args.train_dtype)
    from brcnn import blocks
    from brcnn.optim import FusedSGD
    blocks.conv_weights_channels_last(model)        # the weight-gradient kernels' layout: no per-layer grad copies
    params = [p for p in model.parameters() if p.requires_grad]
    # the recipes' optimizer (SGD, momentum 0.9, weight decay 1e-4, grad-clip 35) on the fused HIP step, at the
    # learning rate the recipe's schedule gives its FIRST iterations: lr x warmup_ratio (linear warm-up over 500
    # iterations from ratio 0.001, configs/_base_/schedules/schedule_1x.py:5-10) -- randomly initialised weights
    # (no checkpoint can be fetched) diverge within the timed steps at the post-warm-up rate
    warm = float(cfg.get('lr_config', {}).get('warmup_ratio', 1e-3))
    opt = FusedSGD(params, lr=cfg.optimizer.lr * warm, momentum=cfg.optimizer.momentum,
                   weight_decay=cfg.optimizer.weight_decay)
    opt.register_conv_weights(model, blocks.compute_dtype())
    net, reducer = model, None
    if world > 1 and os.environ.get('BRCNN_DDP', 'own') == 'torch':     # the reference's wrapper (apis/train.py:75-83)
        net = torch.nn.parallel.DistributedDataParallel(model, device_ids=[device.index], broadcast_buffers=False,
                                                        bucket_cap_mb=64, gradient_as_bucket_view=True)
    elif world > 1 or args.force_reducer:
        # in-place all-reduce of the weight-gradient arena, slice by slice behind the backward pass; no accumulator
        # hooks, so the second-stream weight gradients and "dW is weight.grad" stay on (brcnn/distributed.py)
        from brcnn.distributed import GradReducer
        reducer = GradReducer(params, slice_mb=float(os.environ.get('BRCNN_REDUCER_SLICE_MB', '64')),
                              overlap=os.environ.get('BRCNN_REDUCER_OVERLAP', '0') == '1',
                              compress=os.environ.get('BRCNN_REDUCER_COMPRESS') or None)    # 'bf16': half the xGMI bytes
        reducer.broadcast_parameters(model)
    img, metas = synthetic_batch(args.batch, device, seed=rank)
    gtb, gtl = synthetic_gt(args.batch, device, 80, seed=rank)
    # fp16: the recipes' static loss scaling (fp16 = dict(loss_scale=512.), mmcv Fp16OptimizerHook)
    scale = 512.0 if args.train_dtype == 'f16' else 1.0
    last = {}
    # the RandomSampler draws torch.randperm on the host (random_sampler.py:58): seeded, so that the loss of the last
    # timed step reproduces from box to box
    torch.manual_seed(1234 + rank)
    reduce_ev = []

    # the RPN branch's backward pass inside the forward pass, beside the proposal stage (detectors.py): gradients are
    # cleared before the forward pass below, the backward seed is the static loss scale; off under DDP
    if os.environ.get('BRCNN_EARLY_RPN_BWD', '1') != '0' and net is model:
        model.early_rpn_backward = True
        model.early_backward_scale = scale

    # BRCNN_BENCH_GRAPH_TRUNK=1: backbone + neck, forward and backward, replayed from two HIP graphs from the second step on
    # (brcnn/graphs.py): ~350 of the step's ~770 launches become two graph launches and the host runs 12 ms ahead instead of
    # 6 -- but ROCm 7.2 executes the captured weight-gradient branch serially with the main chain (20.0 ms per step against
    # 18.1 eager, 19.4 eager without the second stream; profiles/r05_notes.md), so it stays OFF in the measured line
    if net is model and os.environ.get('BRCNN_BENCH_GRAPH_TRUNK', '0') == '1':
        model.graph_trunk = True

    from brcnn.profiling import stage_mark

    def step():
        opt.zero_grad(set_to_none=True)
        losses = net(img=img, img_metas=metas, return_loss=True, gt_bboxes=gtb, gt_labels=gtl)
        loss, log_vars = model._parse_losses(losses)
        (loss * scale if scale != 1.0 else loss).backward()
        stage_mark('backward')
        if reducer is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            reducer.reduce()
            e1.record()
            reduce_ev.append((e0, e1))
        opt.step(max_norm=35, loss_scale=scale)        # clip + unscale + skip-on-inf + SGD + next step's conv operands
        stage_mark('reduce_and_optimizer')
        last['log_vars'] = log_vars

    steps = args.train_steps or args.steps
    # host slack at the step's one synchronisation (roi_heads.sample_device: the sampler's candidate counts): how long
    # the host WAITED there = how far it runs ahead of the device; ~0 means the step is launch-bound from there on
    from brcnn import roi_heads as _rh
    def start_slack_clock():
        _rh.SYNC_WAIT = [0.0, 0]
    dt, step_stats = timed(step, steps, args.warmup, world, device, after_warmup=start_slack_clock)
    sync_wait = _rh.SYNC_WAIT
    _rh.SYNC_WAIT = None
    loss_last_timed = float(last['log_vars']['loss'])       # (the passes below run further steps)
    step_stats['host_slack_at_sync_ms'] = 1e3 * sync_wait[0] / max(1, sync_wait[1])
    from brcnn import lib as _lib
    _lib.handover_status()          # a lost stream-K hand-over inside the timed steps is an error, not a number
    # HIP-event time of GradReducer.reduce() on the main stream over the timed steps: join of the weight-gradient
    # stream + the all-reduce of the arena and of the small-gradient bucket (what the N > 1 runs add to the step)
    reduce_ms = None
    if reduce_ev:
        tail = reduce_ev[-steps:]
        reduce_ms = sum(a.elapsed_time(b) for a, b in tail) / len(tail)
    grad_bytes = sum(p.numel() for p in params) * 4
    from brcnn import profiling
    # every rank runs the roofline pass: step() contains collectives (gradient all-reduce, the fused RPN normaliser,
    # the log scalars), so a rank-0-only pass would pair them with nothing and hang at N > 1
    graphed = model.__dict__.get('_graphed_trunk')
    graph_info = None if graphed is None else {'captures': graphed.captures,
                                               'replays': sum(c.replays for c in graphed.caps.values()),
                                               'disabled': graphed.disabled_reason}
    # where the MAIN stream's time goes (HIP events between the stages as they are queued; the weight-gradient stream and
    # the proposal stream run beside it): backbone, neck, RPN tower, RPN loss + its backward pass (inside the forward pass),
    # sampler (= the wait for proposals / assignment / the host's draw), RoIAlign, FC head, boosting loss, backward,
    # reduce + optimizer
    stages = profiling.stage_breakdown(step, cuda=True, iters=3, tail=None)
    model.graph_trunk = False       # the roofline pass times every conv launch by its own pair of events: eager launches
    roof = profiling.train_conv_roofline(step, dtype=args.train_dtype)
    return {
        'metric': 'images/sec (1333x800) Boosting R-CNN R50-PAFPN train step',
        'value': world * args.batch * steps / dt, 'unit': 'images/sec', 'ms_per_step': 1000.0 * dt / steps,
        'steps': steps, 'warmup': args.warmup, 'dtype': args.train_dtype, 'n_gpus': world, **step_stats,
        'config': {'workload': f'boosting_rcnn_r50_pafpn_1x_coco.py full train step, batch {args.batch} x 3x800x1344 '
                               f'per GPU, 20 GT/img, 512 RoIs/img, SGD+clip, {args.train_dtype} conv stack, '
                               'device-resident targets / losses' +
                               (', gradient arena all-reduced in place over RCCL' if reducer is not None else
                                ', DDP over RCCL' if world > 1 else ''),
                   'global_batch': world * args.batch, 'parallelism': f'dp{world}'},
        'loss': loss_last_timed,
        'lr': cfg.optimizer.lr * warm,
        'grad_bytes': grad_bytes, 'reduce_ms': reduce_ms,
        'grad_allreduce': None if reducer is None else reducer.describe(),
        'graph_trunk': graph_info,
        'stages_ms': stages,
        'roofline': roof,
    }


def inference_bench(args, world, rank, device):
    import torch
    model, cfg = build_model('boosting_rcnn_r50_pafpn_1x_utdac.py', device)
    model = model.eval().freeze_for_inference()
    model.set_compute_dtype(args.dtype)
    img, metas = synthetic_batch(args.batch, device, seed=rank)
    last = {}

    def step():
        # benchmark.py:113-114: `model(return_loss=False, rescale=True, **data)` -- forward_test, the device-resident
        # pass, ONE device->host copy of the padded results, bbox2result to per-class numpy arrays
        with torch.no_grad():
            last['out'] = model(return_loss=False, rescale=True, img=[img], img_metas=[metas])

    dt, step_stats = timed(step, args.steps, args.warmup, world, device)
    from brcnn import lib as _lib
    _lib.handover_status()          # a lost stream-K hand-over inside the timed steps is an error, not a number
    from brcnn import profiling
    roof = profiling.conv_stack_roofline(model, img, metas, iters=5, dtype=args.dtype)

    def device_pass():
        with torch.no_grad():
            det, lab, nd = model.simple_test_device(img, metas, rescale=True)
        return det.cpu(), lab.cpu(), nd.cpu()
    stages = profiling.stage_breakdown(device_pass, cuda=True)
    # batch-1 latency (benchmark.py runs samples_per_gpu=1; the only published neighbours, BASELINE.md, are batch 1):
    # synchronised passes over one image, mean of `steps`
    img1, metas1 = img[:1].contiguous(), metas[:1]
    lat1 = None
    if not args.no_bs1:
        with torch.no_grad():
            for _ in range(max(2, args.warmup)):
                model(return_loss=False, rescale=True, img=[img1], img_metas=[metas1])
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                model(return_loss=False, rescale=True, img=[img1], img_metas=[metas1])
            torch.cuda.synchronize()
            lat1 = (time.perf_counter() - t0) / args.steps * 1000.0
    line = {
        'metric': 'images/sec (1333x800) Boosting R-CNN R50-PAFPN inference',
        'value': world * args.batch * args.steps / dt, 'unit': 'images/sec',
        'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': 1000.0 * dt / args.steps,
        **step_stats, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': args.dtype, 'data': 'synthetic',
        'config': {'workload': 'boosting_rcnn_r50_pafpn_1x_utdac.py inference (model(return_loss=False, rescale=True): '
                               'device pass + result copy + bbox2result), '
                               f'batch {args.batch} x 3x800x1344 per GPU, {args.dtype} MFMA conv stack, '
                               '1000 pre-NMS / 256 proposals per image, seeded synthetic weights',
                   'global_batch': world * args.batch, 'parallelism': f'dp{world}'},
        'roofline': roof,
        'stages_ms': stages,
        'latency_bs1_ms': lat1, 'fps_bs1': None if lat1 is None else 1000.0 / lat1,
        'detections_last_step': int(sum(len(c) for im in last['out'] for c in im)),
    }
    return line, cfg, model


def main():
    args = parse_args()
    if os.environ.get('BRCNN_WATCHDOG_S'):          # a hang (an unpaired collective) ends with every thread's stack on stderr
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ['BRCNN_WATCHDOG_S']), exit=True)
    env_world = os.environ.get('WORLD_SIZE')
    if env_world is None and args.gpus > 1:
        sys.exit(launch_ranks(args))            # nothing has touched the GPU yet
    world = int(env_world or '1')
    if world != args.gpus:
        sys.exit(f'bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with `python bench.py --gpus N` or '
                 f'`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`')
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    # test hook (tests/test_ddp_gpu.py): every rank on cuda:0 and collectives over gloo, to run the N > 1 control flow
    # on a one-GPU box (RCCL refuses two ranks per device).  Not a measurement configuration.
    one_device = os.environ.get('BRCNN_DIST_ONE_DEVICE', '0') == '1'
    backend = os.environ.get('BRCNN_DIST_BACKEND', 'nccl')
    if one_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device('cuda', local_rank)
    if world > 1 or args.force_reducer:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if 'MASTER_PORT' not in os.environ:
            import socket
            with socket.socket() as s_:
                s_.bind(('127.0.0.1', 0))
                os.environ['MASTER_PORT'] = str(s_.getsockname()[1])
        if backend == 'nccl':
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    # Memory pool up front (the part has 288 GB): one 64 GiB block taken and handed back to PyTorch's caching allocator,
    # which then serves the models, activations and workspaces of both legs out of it.  Without it the allocator goes back
    # to hipMalloc whenever a step needs a size it has not cached yet, and on a freshly booted box such a call inside the
    # timed region took 1-2 s in 4 of ~16 first runs (profiles/r04_notes.md) -- the runner can afford that once per
    # training, a 20-step measurement cannot.
    try:
        pool = torch.empty(int(os.environ.get('BRCNN_BENCH_POOL_GIB', '64')) << 30, dtype=torch.uint8, device=device)
        del pool
    except RuntimeError:
        pass
    line, cfg = None, None
    if args.mode in ('both', 'inference'):
        line, cfg, model = inference_bench(args, world, rank, device)
        del model
    if args.mode in ('both', 'train'):
        tr = train_bench(args, world, rank, device)
        if line is None:
            line = dict(tr, higher_is_better=True, scaling='weak', vs_baseline=None, data='synthetic')
        else:
            line['train'] = tr
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline and cfg is not None:
            from oracle import cpu_pipeline
            line['cpu_baseline'] = cpu_pipeline.timed_baseline(cfg, seed=0)
        # RCCL prints its version banner through C stdio, which would otherwise be flushed at exit -- AFTER the line
        # the driver reads as the last one
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(json.dumps(line), flush=True)
    if world > 1 or args.force_reducer:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
