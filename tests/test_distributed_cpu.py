"""N>1 path on CPU: world_size-2 `gloo` processes.  The hot path shards by image with no
data-path collective at inference; training couples ranks only through `reduce_mean` of the
RPN normalisers (atss_rpn_head.py:441-444,458-460) and the fused log-scalar all-reduce of
`_parse_losses` (base.py:202-207).  Both are exercised here."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, ret):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        import brcnn  # noqa: F401
        from brcnn import Config
        from brcnn.dense_heads import reduce_mean
        from brcnn.detectors import BaseDetector
        from tests import util
        from tests.test_host_cpu import CFG
        # reduce_mean: mean over ranks
        t = torch.tensor(float(rank + 1))
        assert reduce_mean(t).item() == pytest.approx((1 + world) / 2)
        # the RPN loss on rank-local images with the cross-rank normaliser
        cfg = Config.fromfile(CFG)
        c = cfg.model.rpn_head.copy()
        c.update(train_cfg=cfg.model.train_cfg.rpn, test_cfg=cfg.model.test_cfg.rpn)
        head = brcnn.build_head(c)
        sizes = [(16, 24), (8, 12), (4, 6), (2, 3), (1, 2)]
        g = torch.Generator().manual_seed(100 + rank)
        cls = [torch.randn(1, 9, h, w, generator=g) for h, w in sizes]
        reg = [torch.randn(1, 36, h, w, generator=g) * 0.3 for h, w in sizes]
        iou = [torch.randn(1, 9, h, w, generator=g) for h, w in sizes]
        _, metas, gts, _ = util.demo_inputs(1, 128, 192, seed=100 + rank, num_gt=2 + 5 * rank)
        losses = head.loss(cls, reg, iou, gts, metas)
        # _parse_losses: one fused all-reduce, values averaged over ranks
        det = BaseDetector()
        loss, log_vars = det._parse_losses(losses)
        local = sum(sum(v) for v in losses.values()).item()
        gathered = [torch.zeros(1) for _ in range(world)]
        dist.all_gather(gathered, torch.tensor([local]))
        assert log_vars['loss'] == pytest.approx(sum(x.item() for x in gathered) / world, rel=1e-5)
        assert loss.item() == pytest.approx(local, rel=1e-6)
        # image sharding: rank r owns images [r*b, (r+1)*b) of the global batch
        b, glob = 2, list(range(world * 2))
        mine = glob[rank * b:(rank + 1) * b]
        allm = [None] * world
        dist.all_gather_object(allm, mine)
        assert sorted(sum(allm, [])) == glob
        ret[rank] = float(log_vars['loss'])
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_training_couplings():
    world, port = 2, _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    assert len(ret) == 2 and ret[0] == pytest.approx(ret[1], rel=1e-6)   # same averaged scalar


def _train_worker(rank, world, port, tmp, ret):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    os.environ['RANK'], os.environ['WORLD_SIZE'], os.environ['LOCAL_RANK'] = str(rank), str(world), str(rank)
    import pathlib
    import brcnn  # noqa: F401
    from brcnn import apis, build_detector
    from brcnn.datasets import build_dataloader, build_dataset
    from oracle import cpu_pipeline
    from tests.test_drivers_cpu import _tiny_cfg
    apis.init_dist('pytorch', backend='gloo')
    try:
        torch.set_num_threads(max(1, cpu_pipeline.available_cpus() // world))
        cfg = _tiny_cfg(pathlib.Path(tmp) / f'r{rank}', max_epochs=1)       # same synthetic dataset on every rank
        cfg.work_dir = os.path.join(tmp, 'work')
        apis.set_random_seed(0)
        with cpu_pipeline.patched():
            model = build_detector(cfg.model)
            ds = build_dataset(cfg.data.train)
            model.CLASSES = ds.CLASSES
            runner = apis.train_detector(model, ds, cfg, distributed=True, validate=False, device=torch.device('cpu'))
            # every rank trained on its own shard: DistributedGroupSampler gives ceil(n / 2 / world) * 2 samples
            assert runner.epoch == 1 and runner.iter == len(runner.history)
            w = torch.cat([p.detach().flatten()[:64] for p in model.parameters() if p.requires_grad])
            gathered = [torch.zeros_like(w) for _ in range(world)]
            dist.all_gather(gathered, w)
            assert torch.equal(gathered[0], gathered[1])           # DDP kept the replicas identical
            # multi_gpu_test: round-robin shards come back in dataset order on rank 0
            dt = build_dataset(cfg.data.test, dict(test_mode=True))
            loader = build_dataloader(dt, 1, 0, dist=True, shuffle=False, rank=rank, world_size=world)
            res = apis.multi_gpu_test(runner.model, loader)
            if rank == 0:
                assert len(res) == len(dt) and all(len(r) == 4 for r in res)
                single = apis.single_gpu_test(runner.model.module,
                                              build_dataloader(dt, 1, 0, dist=False, shuffle=False))
                for a, b in zip(res, single):
                    for x, y in zip(a, b):
                        assert x.shape == y.shape and (x.size == 0 or abs(x - y).max() < 1e-4)
                assert os.path.exists(os.path.join(cfg.work_dir, 'epoch_1.pth'))
            else:
                assert res is None
        ret[rank] = (runner.iter, [row[3]['loss'] for row in runner.history])
    finally:
        dist.destroy_process_group()


def test_two_rank_ddp_train_and_test_drivers(tmp_path):
    """`train_detector(distributed=True)` + `multi_gpu_test` on two gloo ranks over the CPU oracle
    pipeline: sharded sampler, DDP gradient averaging (replicas stay identical), rank-0 checkpoint,
    result gathering in dataset order."""
    world, port = 2, _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_train_worker, args=(world, port, str(tmp_path), ret), nprocs=world, join=True)
    assert len(ret) == 2 and ret[0][0] == ret[1][0] >= 1
    assert ret[0][1] == pytest.approx(ret[1][1], rel=1e-5)        # the logged loss is the all-reduced mean


def _reducer_worker(rank, world, port, ret):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        import brcnn  # noqa: F401
        from brcnn import autograd as A
        from brcnn.distributed import GradReducer
        torch.manual_seed(5)                      # the same "model" on every rank
        shapes = [(64, 3, 3, 32), (128, 1, 1, 64), (300,), (17,), (256, 3, 3, 256), (1,), (40, 1, 1, 8)]
        params = [torch.nn.Parameter(torch.randn(*s)) for s in shapes]
        red = GradReducer(params, slice_mb=0.25)            # 65 536-element slices: the arena spans several
        # rank-dependent start values so that a missing broadcast shows
        with torch.no_grad():
            for p in params:
                p.add_(float(rank))
        mod = torch.nn.ParameterList(params)
        red.broadcast_parameters(mod)
        for p in params:
            t = p.detach().clone()
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            assert torch.equal(t, p.detach())
        for step in range(3):
            g = torch.Generator().manual_seed(1000 * step + rank)
            local = [torch.randn(*s, generator=g) for s in shapes]
            for p, gl in zip(params, local):
                if p.dim() == 4:                  # conv weights: the weight-gradient kernel's result IS .grad (arena view)
                    v = A.grad_arena.take(p.shape, p.device)
                    v.copy_(gl)
                    A.grad_arena.launched(None)
                    p.grad = v
                else:                             # BatchNorm / bias / Scale gradients: ordinary tensors
                    p.grad = gl.clone()
            red.reduce()
            for i, p in enumerate(params):
                allg = [torch.zeros_like(local[i]) for _ in range(world)]
                dist.all_gather(allg, local[i])
                want = sum(allg) / world
                assert torch.allclose(p.grad, want, rtol=1e-6, atol=1e-7), (step, i)
            A.grad_arena.new_step()               # what optim.FusedSGD.step does
        red.close()
        assert A.grad_arena.listener is None and not A._OWN_REDUCER[0]
        ret[rank] = True
    finally:
        dist.destroy_process_group()


def test_two_rank_grad_reducer_averages_arena_and_small_gradients():
    """distributed.GradReducer on two gloo ranks: the weight-gradient arena is all-reduced in place (several slices
    per chunk, a fresh chunk per step), the gradients outside it as one flattened bucket, parameters are broadcast
    from rank 0 -- every gradient ends as the mean over the ranks, on every step"""
    world, port = 2, _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_reducer_worker, args=(world, port, ret), nprocs=world, join=True)
    assert len(ret) == 2


def _reducer_bf16_worker(rank, world, port, ret):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        import brcnn  # noqa: F401
        from brcnn import autograd as A
        from brcnn.distributed import GradReducer
        torch.manual_seed(9)
        shapes = [(64, 3, 3, 32), (300,), (128, 1, 1, 64), (17,)]
        params = [torch.nn.Parameter(torch.randn(*s)) for s in shapes]
        red = GradReducer(params, compress='bf16')
        opt = torch.optim.SGD(params, lr=0.01, momentum=0.9, weight_decay=1e-4)
        for step in range(2):
            g = torch.Generator().manual_seed(77 * step + rank)
            local = [torch.randn(*s, generator=g) for s in shapes]
            for p, gl in zip(params, local):
                if p.dim() == 4:
                    v = A.grad_arena.take(p.shape, p.device)
                    v.copy_(gl)
                    A.grad_arena.launched(None)
                    p.grad = v
                else:
                    p.grad = gl.clone()
            before = [p.detach().clone() for p in params]
            bufs = [opt.state[p].get('momentum_buffer') for p in params]
            bufs = [None if b is None else b.clone() for b in bufs]
            red.reduce()
            for i, p in enumerate(params):
                allg = [torch.zeros_like(local[i]) for _ in range(world)]
                dist.all_gather(allg, local[i])
                if p.dim() == 4:
                    # the wire format: each rank's fp32 gradient rounded to bf16 (nearest even), summed in bf16 by the
                    # collective, widened back to fp32 (exact), divided by the world size (exact for 2)
                    want = (allg[0].bfloat16() + allg[1].bfloat16()).float() / world
                    assert p.grad.dtype == torch.float32 and torch.equal(p.grad, want), (step, i)
                    assert (p.grad - sum(allg) / world).abs().max() <= 2.0 ** -7 * sum(a.abs() for a in allg).max()
                else:                             # the small bucket stays fp32
                    assert torch.allclose(p.grad, sum(allg) / world, rtol=1e-6, atol=1e-7), (step, i)
            # the master update is fp32 arithmetic on fp32 weights: torch's SGD on the reduced gradients equals the
            # closed form evaluated independently in fp32
            opt.step()
            for i, p in enumerate(params):
                d = p.grad + 1e-4 * before[i]
                b = d if bufs[i] is None else 0.9 * bufs[i] + d
                assert torch.allclose(p.detach(), before[i] - 0.01 * b, rtol=1e-6, atol=1e-6), (step, i)
                assert p.dtype == torch.float32 and opt.state[p]['momentum_buffer'].dtype == torch.float32
            A.grad_arena.new_step()
        assert red.describe()['bytes_last_step'] == 2 * (64 * 3 * 3 * 32 + 128 * 64) + 4 * 317
        red.close()
        ret[rank] = True
    finally:
        dist.destroy_process_group()


def test_two_rank_grad_reducer_bf16_wire_format_keeps_fp32_master_update():
    """GradReducer(compress='bf16'): the arena crosses the wire as bf16 (half the bytes), the result is exactly the
    bf16 sum of the bf16-rounded local gradients, and the optimizer step on the fp32 master weights is fp32"""
    world, port = 2, _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_reducer_bf16_worker, args=(world, port, ret), nprocs=world, join=True)
    assert len(ret) == 2


def _reducer_layout_worker(rank, world, port, ret):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        import brcnn  # noqa: F401
        from brcnn import autograd as A
        from brcnn.distributed import GradReducer
        torch.manual_seed(3)
        params = [torch.nn.Parameter(torch.randn(32, 1, 1, 16)), torch.nn.Parameter(torch.randn(24)),
                  torch.nn.Parameter(torch.randn(8))]
        red = GradReducer(params)
        # step 0: identical layouts; the small gradients end up as views of ONE persistent bucket
        v = A.grad_arena.take(params[0].shape, params[0].device)
        v.fill_(float(rank))
        A.grad_arena.launched(None)
        params[0].grad = v
        params[1].grad = torch.full((24,), float(rank))
        params[2].grad = torch.full((8,), 2.0 * rank)
        red.reduce()
        assert torch.equal(params[1].grad, torch.full((24,), 0.5)) and torch.equal(params[2].grad, torch.full((8,), 1.0))
        base = red._bucket.data_ptr()
        assert params[1].grad.data_ptr() == base and params[2].grad.data_ptr() == base + 24 * 4
        A.grad_arena.new_step()
        # step 1: rank 1 "skips" a parameter (an unused branch): the ranks would post all-reduces of different sizes
        v = A.grad_arena.take(params[0].shape, params[0].device)
        v.fill_(1.0)
        A.grad_arena.launched(None)
        params[0].grad = v
        params[1].grad = torch.ones(24)
        params[2].grad = None if rank == 1 else torch.ones(8)
        try:
            red.reduce()
            raised = False
        except RuntimeError as e:
            raised = 'differs from another rank' in str(e) or 'different gradient layouts' in str(e)
        red.close()
        ret[rank] = raised
    finally:
        dist.destroy_process_group()


def test_two_rank_grad_reducer_raises_on_layout_mismatch():
    """a rank whose backward pass produced a different set of gradients must not pair its all-reduces with the other
    rank's: rank 1 raises at once (its own layout changed: blocking comparison); rank 0, whose layout did not change,
    compares through the fixed-size collective as well and raises too"""
    world, port = 2, _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_reducer_layout_worker, args=(world, port, ret), nprocs=world, join=True)
    assert ret[1] is True and ret[0] is True


def _slicing_worker(rank, world, port, ret):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        import brcnn  # noqa: F401
        from brcnn import autograd as A
        from brcnn import distributed as D
        # interval arithmetic of reduce(): what the live gradient ranges leave after the issued slices
        assert D._subtract([(0, 10), (20, 30)], []) == [(0, 10), (20, 30)]
        assert D._subtract([(0, 100)], [(10, 20), (40, 50)]) == [(0, 10), (20, 40), (50, 100)]
        assert D._subtract([(0, 10), (10, 25)], [(0, 16)]) == [(16, 25)]
        assert D._subtract([(5, 9)], [(0, 16)]) == []

        class FakeDeviceBuffer:        # stands in for a HIP arena chunk: only identity and is_cuda are looked at
            is_cuda = True
        issued = []

        class Probe(D.GradReducer):
            def _issue(self, buf, lo, hi, stream):
                issued.append((lo, hi))
        params = [torch.nn.Parameter(torch.zeros(4)) for _ in range(6)]
        red = Probe(params, slice_mb=4e-4, overlap=True)            # ~104-element slices
        n = red.slice_elems
        buf = FakeDeviceBuffer()
        red.chunk_opened(buf)
        # three in-place ranges in a row form ONE run: slices are cut from it wherever they fall
        red.writers_launched(buf, 0, 64, None, True, params[0])
        assert issued == []
        red.writers_launched(buf, 64, 64 + n, None, True, params[1])
        assert issued == [(0, n)]
        # a range autograd will COPY (not in place) ends the run: its leftover and the range itself are not sliced ...
        red.writers_launched(buf, 64 + n, 64 + 4 * n, None, False, params[2])
        assert issued == [(0, n)]
        # ... and the next in-place range starts a new run at its own offset
        red.writers_launched(buf, 64 + 4 * n, 64 + 6 * n, None, True, params[3])
        assert issued == [(0, n), (64 + 4 * n, 64 + 5 * n), (64 + 5 * n, 64 + 6 * n)]
        # a gap (alignment hole) also starts a new run
        red.writers_launched(buf, 64 + 6 * n + 64, 64 + 7 * n + 64, None, True, params[4])
        assert issued[-1] == (64 + 6 * n + 64, 64 + 7 * n + 64)
        assert [tuple(r) for r in red._progress[0][1]] == issued
        # per-call temporaries (the 4-D view of an FC weight, a padded weight) are told apart for the whole step: the
        # reducer keeps the tensor, so the id of a freed one cannot come back on the next temporary (one two-rank bench
        # run in ten stopped here with the shared-parameter error before it did)
        import gc, weakref
        base = torch.zeros(8)
        off = 64 + 8 * n
        for i in range(200):
            t = base.view(2, 4)
            w = weakref.ref(t)
            red.writers_launched(buf, off + 64 * i, off + 64 * (i + 1), None, False, t)     # (never raises)
            del t
            gc.collect()
            assert w() is not None
        # a parameter that reaches a second weight-gradient launch in one pass: refused by the overlapped form
        with pytest.raises(RuntimeError, match='used twice'):
            red.writers_launched(buf, 64 + 8 * n, 64 + 9 * n, None, True, params[0])
        red.close()
        # the plain form (one all-reduce after the pass) tolerates it and slices nothing
        issued.clear()
        red = Probe(params, overlap=False)
        red.chunk_opened(buf)
        red.writers_launched(buf, 0, 10 * n, None, True, params[0])
        red.writers_launched(buf, 10 * n, 20 * n, None, True, params[0])
        assert issued == []
        red.close()
        assert A.grad_arena.listener is None
        ret[rank] = 1.0
    finally:
        dist.destroy_process_group()


def test_overlapped_reducer_slices_only_ranges_that_are_grad():
    """round 5 (the two-rank mismatch of round 4): the overlapped reducer cuts its in-place slices out of RUNS of arena
    ranges that autograd takes as `.grad` unchanged; a range that will be copied on the main stream ends the run, a gap
    starts a new one, a parameter with two launches in one pass is refused.  Host logic only (a stub chunk, recorded
    slices): the real collectives run in tests/test_ddp_gpu.py."""
    port = _free_port()
    ret = mp.Manager().dict()
    mp.spawn(_slicing_worker, args=(1, port, ret), nprocs=1, join=True)
    assert ret[0] == 1.0


def _takes_lie_worker(rank, world, port, ret):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        import brcnn  # noqa: F401
        from brcnn import autograd as A
        from brcnn import distributed as D
        dev = torch.device('cpu')

        def step(red, params, lie_for=None):
            """one 'backward pass' by hand: every weight gradient is written into the arena and announced the way
            autograd._conv_backward does; `lie_for`: that parameter's `takes` prediction says "autograd takes the arena
            view as .grad" although the gradient autograd keeps is a COPY (what a changed steal rule would do)"""
            A.grad_arena.new_step()
            for i, p in enumerate(params):
                dw = A.grad_arena.take(tuple(p.shape), dev)
                dw.copy_(torch.full(p.shape, float(rank + 1) * (i + 1)))
                A.grad_arena.launched(None, True, p)                    # the prediction: in place
                p.grad = dw.clone() if p is lie_for else dw             # what autograd really kept
            red.reduce()

        torch.manual_seed(0)
        params = [torch.nn.Parameter(torch.zeros(40, 50)) for _ in range(6)]
        named = [(f'layer{i}.weight', p) for i, p in enumerate(params)]
        red = D.GradReducer(named, slice_mb=3000 * 4 / (1 << 20), overlap=True)       # 3000-element slices
        # honest predictions: the overlapped slices really travel during the "pass" (gloo, host tensors) and the
        # result is the mean over the ranks
        step(red, params)
        assert red.last_bytes >= sum(p.numel() for p in params) * 4        # (ranges are padded to 64 elements)
        for i, p in enumerate(params):
            assert torch.equal(p.grad, torch.full(p.shape, 1.5 * (i + 1))), i
        # a lying prediction: the issued slice holds elements no .grad refers to -> reduce() raises and names the parameter
        for p in params:
            p.grad = None
        try:
            step(red, params, lie_for=params[2])
            ret[rank] = 'no error'
        except RuntimeError as e:
            msg = str(e)
            ret[rank] = msg if ('layer2.weight' in msg and 'all-reduced in place' in msg) else 'wrong message: ' + msg
        red.close()
        A.grad_arena.new_step()
    finally:
        dist.destroy_process_group()


def test_overlapped_reducer_refuses_a_wrong_takes_prediction():
    """GradReducer(overlap=True) all-reduces, during the backward pass, the arena ranges that
    autograd is PREDICTED to take as `.grad` unchanged.  reduce() now verifies the prediction: an issued range that no
    parameter's gradient refers to afterwards would be reduced twice (round 4's bug) -- it raises, naming the parameter,
    on every rank.  The honest case runs the real overlapped collectives over gloo on host tensors (world size 2)."""
    port = _free_port()
    ret = mp.Manager().dict()
    mp.spawn(_takes_lie_worker, args=(2, port, ret), nprocs=2, join=True)
    for r in (0, 1):
        assert 'layer2.weight' in ret[r] and 'all-reduced in place' in ret[r], ret[r]


def _replica_worker(rank, world, port, ret):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        import brcnn  # noqa: F401
        from brcnn.distributed import replicas_identical
        torch.manual_seed(3)
        m = torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3), torch.nn.BatchNorm2d(8), torch.nn.Linear(4, 2))
        m[0].weight.data = m[0].weight.data.contiguous(memory_format=torch.channels_last)      # a non-default layout too
        assert replicas_identical(m)                       # same seed on both ranks
        if rank == 1:
            with torch.no_grad():
                m[2].bias[1] += 1e-7                       # one bit pattern in one parameter of one rank
        assert not replicas_identical(m)                   # ... and EVERY rank sees it
        if rank == 1:
            with torch.no_grad():
                m[2].bias[1] -= 1e-7
        same_again = replicas_identical(m)                 # (1e-7 on a bias of ~0.3 does not round-trip exactly in general)
        with torch.no_grad():
            for p in m.parameters():
                dist.broadcast(p.data, 0)
            if rank == 1:
                m[1].running_mean[0] = 5.0                 # buffers count as well
        assert not replicas_identical(m)
        ret[rank] = float(same_again) + 1.0
    finally:
        dist.destroy_process_group()


def test_replicas_identical_sees_one_differing_bit_on_every_rank():
    """`distributed.replicas_identical` (the runner's `check_replicas`): bit-identical parameters and buffers on every
    rank or not -- a single changed value on one rank turns the answer on ALL ranks, channels-last parameters and
    BatchNorm buffers included"""
    port = _free_port()
    ret = mp.Manager().dict()
    mp.spawn(_replica_worker, args=(2, port, ret), nprocs=2, join=True)
    assert ret[0] >= 1.0 and ret[1] >= 1.0
