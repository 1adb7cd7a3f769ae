"""Train / test drivers: this repo's counterpart of mmdet/apis/train.py:38-174,
mmdet/apis/test.py:16-169 and the mmcv runner pieces they configure (EpochBasedRunner, SGD
with step LR + linear warm-up, gradient clipping, checkpoint save / resume in mmcv's
checkpoint layout, text logging, per-epoch bbox evaluation, DistSamplerSeedHook).

One process per GPU; `distributed=True` wraps the detector in torch's
DistributedDataParallel over RCCL (the reference's MMDistributedDataParallel), the single-GPU
path runs the bare module.  The data the loaders deliver is already in the "scattered" form
(datasets.collate), so `model.train_step(data, optimizer)` is called directly.
"""
import datetime
import logging
import os
import os.path as osp
import random
import time
from collections import OrderedDict

import numpy as np
import torch
import torch.distributed as dist

from .datasets import build_dataloader, build_dataset

__all__ = ['set_random_seed', 'get_root_logger', 'build_optimizer', 'StepLrUpdater', 'save_checkpoint',
           'load_checkpoint', 'EpochBasedRunner', 'train_detector', 'single_gpu_test', 'multi_gpu_test',
           'init_dist', 'get_dist_info', 'replace_ImageToTensor', 'host_cpus', 'limit_host_threads']


def host_cpus():
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


def limit_host_threads(world_size=1):
    """torch's intra-op pool takes one thread per core of the HOST (256 on the MI355X boxes) whatever CPU quota the
    container has: a parallel region then burns the quota of a scheduling period at once and the whole process -- the
    thread that launches kernels included -- is throttled until the period ends (profiles/r05_notes.md).  The pool is cut
    to the cores this process may use, shared between the ranks of the node.  Called by tools/train.py / tools/test.py;
    an explicit OMP_NUM_THREADS wins."""
    if os.environ.get('OMP_NUM_THREADS'):
        return torch.get_num_threads()
    n = max(1, host_cpus() // max(1, int(world_size)))
    if torch.get_num_threads() > n:
        torch.set_num_threads(n)
    return torch.get_num_threads()


def set_random_seed(seed, deterministic=False):
    """apis/train.py:17-35"""
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
    if deterministic:
        torch.backends.cudnn.deterministic = True
        torch.backends.cudnn.benchmark = False


def get_dist_info():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def init_dist(launcher='pytorch', backend='nccl', **kwargs):
    """`--launcher pytorch`: one process per GPU under torch.distributed.run; RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_* come from the environment (backend 'nccl' is RCCL on ROCm)."""
    assert launcher == 'pytorch', f'launcher {launcher} is not available on this platform'
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    # test hook (tests/test_ddp_gpu.py, as in bench.py): BRCNN_DIST_ONE_DEVICE=1 puts every rank on cuda:0 and
    # BRCNN_DIST_BACKEND=gloo routes the collectives through gloo (RCCL refuses two ranks per device) -- the N > 1 control
    # flow of the drivers on a one-GPU box; not a training configuration
    backend = os.environ.get('BRCNN_DIST_BACKEND', backend)
    if os.environ.get('BRCNN_DIST_ONE_DEVICE', '0') == '1':
        local_rank = 0
    if torch.cuda.is_available():
        torch.cuda.set_device(local_rank)
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29500')
    dist.init_process_group(backend=backend, **kwargs)


def get_root_logger(log_file=None, log_level='INFO', name='mmdet'):
    logger = logging.getLogger(name)
    if getattr(logger, '_brcnn_ready', False):
        return logger
    rank, _ = get_dist_info()
    handlers = [logging.StreamHandler()]
    if rank == 0 and log_file is not None:
        handlers.append(logging.FileHandler(log_file, 'w'))
    fmt = logging.Formatter('%(asctime)s - %(name)s - %(levelname)s - %(message)s')
    for h in handlers:
        h.setFormatter(fmt)
        h.setLevel(log_level if rank == 0 else logging.ERROR)
        logger.addHandler(h)
    logger.setLevel(log_level if rank == 0 else logging.ERROR)
    logger.propagate = False
    logger._brcnn_ready = True
    return logger


# --------------------------------------------------------------------------- optimizer / lr
def build_optimizer(model, cfg):
    """mmcv DefaultOptimizerConstructor for the recipes' `optimizer = dict(type='SGD', ...)`,
    with `paramwise_cfg` (bias_lr_mult, bias_decay_mult, norm_decay_mult, custom_keys)"""
    cfg = dict(cfg)
    paramwise = cfg.pop('paramwise_cfg', None)
    typ = cfg.pop('type')
    cls = getattr(torch.optim, typ)
    module = model.module if hasattr(model, 'module') else model
    if typ == 'SGD' and not cfg.get('nesterov', False) and not cfg.get('dampening', 0) and \
            all(p.is_cuda for p in module.parameters()) and os.environ.get('BRCNN_TORCH_SGD') != '1':
        from .optim import FusedSGD       # same param_groups / state_dict layout, one pass on the HIP kernels
        cls = FusedSGD
    if not paramwise:
        params = [p for p in module.parameters() if p.requires_grad]
        return cls(params, **cfg)
    base_lr, base_wd = cfg.get('lr'), cfg.get('weight_decay', 0.0)
    custom = paramwise.get('custom_keys', {})
    groups = []
    norm_types = (torch.nn.modules.batchnorm._BatchNorm, torch.nn.GroupNorm, torch.nn.LayerNorm)
    norm_params = {id(p) for m in module.modules() if isinstance(m, norm_types) for p in m.parameters(recurse=False)}
    for name, p in module.named_parameters():
        if not p.requires_grad:
            continue
        g = {'params': [p]}
        hit = [k for k in sorted(custom, key=len, reverse=True) if k in name]
        if hit:
            g['lr'] = base_lr * custom[hit[0]].get('lr_mult', 1.)
            if base_wd is not None:
                g['weight_decay'] = base_wd * custom[hit[0]].get('decay_mult', 1.)
        else:
            if name.endswith('.bias') and id(p) not in norm_params:
                if 'bias_lr_mult' in paramwise:
                    g['lr'] = base_lr * paramwise['bias_lr_mult']
                if 'bias_decay_mult' in paramwise and base_wd is not None:
                    g['weight_decay'] = base_wd * paramwise['bias_decay_mult']
            if id(p) in norm_params and 'norm_decay_mult' in paramwise and base_wd is not None:
                g['weight_decay'] = base_wd * paramwise['norm_decay_mult']
        groups.append(g)
    return cls(groups, **cfg)


class StepLrUpdater:
    """mmcv StepLrUpdaterHook + LrUpdaterHook warm-up (by_epoch=True, warmup_by_epoch=False):
    lr = base * gamma^(#steps <= epoch), overridden for iter < warmup_iters by
    constant / linear / exp warm-up of the REGULAR lr of that epoch"""

    def __init__(self, policy='step', step=(8, 11), gamma=0.1, min_lr=None, warmup=None, warmup_iters=0,
                 warmup_ratio=0.1, by_epoch=True, **kwargs):
        assert policy == 'step' and by_epoch
        assert warmup in (None, 'constant', 'linear', 'exp')
        self.step = [step] if isinstance(step, int) else list(step)
        self.gamma, self.min_lr = gamma, min_lr
        self.warmup, self.warmup_iters, self.warmup_ratio = warmup, warmup_iters, warmup_ratio
        self.base_lr = None

    def before_run(self, optimizer):
        for g in optimizer.param_groups:
            g.setdefault('initial_lr', g['lr'])
        self.base_lr = [g['initial_lr'] for g in optimizer.param_groups]

    def regular_lr(self, epoch):
        if isinstance(self.step, int):
            exp = epoch // self.step
        else:
            exp = len(self.step)
            for i, s in enumerate(self.step):
                if epoch < s:
                    exp = i
                    break
        out = []
        for b in self.base_lr:
            lr = b * (self.gamma ** exp)
            out.append(max(lr, self.min_lr) if self.min_lr is not None else lr)
        return out

    def warmup_lr(self, cur_iter, regular):
        if self.warmup == 'constant':
            return [lr * self.warmup_ratio for lr in regular]
        if self.warmup == 'linear':
            k = (1 - cur_iter / self.warmup_iters) * (1 - self.warmup_ratio)
            return [lr * (1 - k) for lr in regular]
        k = self.warmup_ratio ** (1 - cur_iter / self.warmup_iters)
        return [lr * k for lr in regular]

    def lr_at(self, epoch, cur_iter):
        regular = self.regular_lr(epoch)
        if self.warmup is not None and cur_iter < self.warmup_iters:
            return self.warmup_lr(cur_iter, regular)
        return regular

    def apply(self, optimizer, epoch, cur_iter):
        for g, lr in zip(optimizer.param_groups, self.lr_at(epoch, cur_iter)):
            g['lr'] = lr


# --------------------------------------------------------------------------- checkpoints
def _cpu_state(sd):
    return OrderedDict((k, v.detach().cpu()) for k, v in sd.items())


def save_checkpoint(model, filename, optimizer=None, meta=None):
    """mmcv.runner.save_checkpoint layout: {'meta', 'state_dict', 'optimizer'}; parameter names
    are the reference's, so its published checkpoints and ours are interchangeable"""
    module = model.module if hasattr(model, 'module') else model
    meta = dict(meta or {})
    meta.update(time=time.asctime())
    if getattr(module, 'CLASSES', None) is not None:
        meta.update(CLASSES=module.CLASSES)
    ckpt = {'meta': meta, 'state_dict': _cpu_state(module.state_dict())}
    if optimizer is not None:
        ckpt['optimizer'] = optimizer.state_dict()
    os.makedirs(osp.dirname(osp.abspath(filename)), exist_ok=True)
    tmp = filename + '.tmp'
    torch.save(ckpt, tmp)
    os.replace(tmp, filename)


def load_checkpoint(model, filename, map_location='cpu', strict=False, logger=None):
    ckpt = torch.load(filename, map_location=map_location, weights_only=False)
    if not isinstance(ckpt, dict):
        raise RuntimeError(f'No state_dict found in checkpoint file {filename}')
    sd = ckpt.get('state_dict', ckpt)
    sd = OrderedDict((k[7:] if k.startswith('module.') else k, v) for k, v in sd.items())
    module = model.module if hasattr(model, 'module') else model
    missing, unexpected = module.load_state_dict(sd, strict=strict)
    if logger is not None and (missing or unexpected):
        logger.warning(f'load_checkpoint: missing keys {list(missing)[:8]}{"..." if len(missing) > 8 else ""}; '
                       f'unexpected keys {list(unexpected)[:8]}{"..." if len(unexpected) > 8 else ""}')
    return ckpt


# --------------------------------------------------------------------------- runner
class EpochBasedRunner:
    """the slice of mmcv.runner.EpochBasedRunner the recipes configure: workflow [('train', 1)],
    lr_config, optimizer_config(grad_clip), checkpoint_config(interval), log_config(interval),
    evaluation(interval, metric), DistSamplerSeedHook, resume / load_from"""

    def __init__(self, model, optimizer, work_dir, logger, max_epochs, meta=None):
        self.model, self.optimizer, self.work_dir, self.logger = model, optimizer, work_dir, logger
        self.max_epochs, self.meta = max_epochs, meta or {}
        self.epoch = self.iter = self.inner_iter = 0
        self.lr_updater = None
        self.grad_clip = None
        self.ckpt_interval = 1
        self.log_interval = 50
        self.eval_fn, self.eval_interval = None, 1
        self.loss_scaler, self.loss_scale = None, None      # fp16 recipes: static loss scaling (train_detector)
        self.reducer = None                                  # distributed.GradReducer (data parallel without DDP)
        self.early_rpn_backward = True                       # see _early_rpn_backward (cfg.early_rpn_backward)
        self.graph_trunk = False                             # cfg.graph_trunk (brcnn/graphs.py): opt-in, see there
        # data parallel: compare the replicas' parameters bit for bit after every optimizer step (a host sync per
        # step: a debugging / test switch -- `check_replicas = True` in the config or BRCNN_CHECK_REPLICAS=1)
        self.check_replicas = os.environ.get('BRCNN_CHECK_REPLICAS', '0') == '1'
        # cyclic garbage collection at the log interval instead of inside the steps (see train)
        self.manual_gc = os.environ.get('BRCNN_RUNNER_GC', '0') != '1'
        self.log_buffer = OrderedDict()
        self.history = []          # (epoch, iter, lr, {name: value}) rows the text logger printed
        self.eval_history = []
        os.makedirs(work_dir, exist_ok=True)

    def register_training_hooks(self, lr_config, optimizer_config=None, checkpoint_config=None, log_config=None):
        self.lr_updater = StepLrUpdater(**dict(lr_config))
        if optimizer_config and optimizer_config.get('grad_clip'):
            self.grad_clip = dict(optimizer_config['grad_clip'])
        if checkpoint_config is not None:
            self.ckpt_interval = checkpoint_config.get('interval', 1)
        if log_config is not None:
            self.log_interval = log_config.get('interval', 50)

    def register_eval(self, fn, interval=1):
        self.eval_fn, self.eval_interval = fn, interval

    def current_lr(self):
        return [g['lr'] for g in self.optimizer.param_groups]

    def resume(self, checkpoint, resume_optimizer=True):
        ckpt = load_checkpoint(self.model, checkpoint, logger=self.logger)
        self.epoch = ckpt['meta']['epoch']
        self.iter = ckpt['meta']['iter']
        if 'optimizer' in ckpt and resume_optimizer:
            self.optimizer.load_state_dict(ckpt['optimizer'])
        self.logger.info(f'resumed epoch {self.epoch}, iter {self.iter}')

    def load_checkpoint(self, filename):
        self.logger.info(f'load checkpoint from {filename}')
        return load_checkpoint(self.model, filename, logger=self.logger)

    def save_checkpoint(self, out_dir, filename_tmpl='epoch_{}.pth'):
        meta = dict(self.meta, epoch=self.epoch + 1, iter=self.iter)
        path = osp.join(out_dir, filename_tmpl.format(self.epoch + 1))
        save_checkpoint(self.model, path, optimizer=self.optimizer, meta=meta)
        latest = osp.join(out_dir, 'latest.pth')
        try:
            if osp.lexists(latest):
                os.remove(latest)
            os.symlink(osp.basename(path), latest)
        except OSError:
            import shutil
            shutil.copy(path, latest)
        return path

    def _clip(self):
        params = [p for p in self.model.parameters() if p.requires_grad and p.grad is not None]
        if params:
            return torch.nn.utils.clip_grad_norm_(params, **self.grad_clip)

    def _log(self, data_loader, t_iter):
        rank, world = get_dist_info()
        vals = OrderedDict()
        for k, v in self.log_buffer.items():
            vals[k] = float(np.mean([float(x) for x in v]))       # (device scalars, e.g. grad_norm, are read here)
        self.log_buffer.clear()
        lr = self.current_lr()[0]
        self.history.append((self.epoch + 1, self.inner_iter + 1, lr, dict(vals)))
        if rank != 0:
            return
        remaining = (self.max_epochs - self.epoch) * len(data_loader) - (self.inner_iter + 1)
        eta = str(datetime.timedelta(seconds=int(t_iter * remaining)))
        items = ', '.join(f'{k}: {v:.4f}' for k, v in vals.items())
        mem = f', memory: {torch.cuda.max_memory_allocated() // (1024 * 1024)}' if torch.cuda.is_available() else ''
        self.logger.info(f'Epoch [{self.epoch + 1}][{self.inner_iter + 1}/{len(data_loader)}]\tlr: {lr:.3e}, '
                         f'eta: {eta}, time: {t_iter:.3f}{mem}, {items}')

    def _early_rpn_backward(self):
        """switch the detector's early RPN backward pass on where this loop can honour its contract: the fused
        optimizer path (zero_grad before the forward pass, the backward seed = the static loss scale), no
        DistributedDataParallel wrapper (its reducer brackets ONE backward pass); `early_rpn_backward = False` in the
        config or BRCNN_EARLY_RPN_BWD=0 keeps the plain order"""
        from .optim import FusedSGD
        m = self.model
        on = (hasattr(type(m), 'early_rpn_backward') and not hasattr(m, 'module') and isinstance(self.optimizer, FusedSGD)
              and (self.grad_clip is None or self.grad_clip.get('norm_type', 2) == 2)
              and self.early_rpn_backward and os.environ.get('BRCNN_EARLY_RPN_BWD', '1') != '0')
        if hasattr(type(m), 'early_rpn_backward'):
            m.early_rpn_backward = bool(on)
            m.early_backward_scale = float(self.loss_scale or 1.0)
        # `graph_trunk = True` in the config: backbone + neck replayed from HIP graphs once an input shape repeats
        # (brcnn/graphs.py) wherever the trunk's parameter gradients may be assigned directly (no DistributedDataParallel
        # wrapper).  Off by default: on ROCm 7.2 the replay is slower than the eager launches on a host that keeps up
        if hasattr(type(m), 'graph_trunk'):
            m.graph_trunk = bool(self.graph_trunk and not hasattr(m, 'module'))
        return on

    def train(self, data_loader):
        """one epoch.  Python's cyclic collector is held off INSIDE the steps and run at the log interval instead
        (`manual_gc`, BRCNN_RUNNER_GC=1 / `manual_gc = False` in the config leave it alone): a step creates a few thousand
        short-lived objects, every few steps a generation-1/2 pass walks the whole heap (model, config, caches) for 2-5 ms
        of host time, which the launch-bound second half of a step turns into device idle time (profiles/r05_notes.md).
        Reference counting still frees everything acyclic at once; what the collector would have found waits for the
        next log line (<= `log_interval` steps).  bench.py's timed loop measures under this same policy."""
        import gc
        manual_gc = self.manual_gc and gc.isenabled()
        if manual_gc:
            gc.collect()
            gc.freeze()                 # the model, the optimizer state and the config never need another scan
            gc.disable()
        try:
            self._train_epoch(data_loader, gc if manual_gc else None)
        finally:
            if manual_gc:
                gc.enable()
                gc.unfreeze()

    def _train_epoch(self, data_loader, gc_):
        self.model.train()
        sampler = getattr(data_loader, 'sampler', None)
        if hasattr(sampler, 'set_epoch'):
            sampler.set_epoch(self.epoch)           # DistSamplerSeedHook
        t_last, n_since = time.time(), 0
        for i, data in enumerate(data_loader):
            self.inner_iter = i
            self.lr_updater.apply(self.optimizer, self.epoch, self.iter)
            early = self._early_rpn_backward()
            if early:
                # the RPN branch is back-propagated inside the forward pass (detectors.py): gradients are cleared first
                self.optimizer.zero_grad()
            outputs = self.model.train_step(data, self.optimizer) if not hasattr(self.model, 'module') else \
                self._ddp_step(data)
            if not early:
                self.optimizer.zero_grad()
            from .optim import FusedSGD
            if isinstance(self.optimizer, FusedSGD) and (self.grad_clip is None or self.grad_clip.get('norm_type', 2) == 2):
                # clip + (unscale) + SGD + next step's conv operands in one call; the norm stays on the device
                scale = self.loss_scale or 1.0
                (outputs['loss'] * scale if scale != 1.0 else outputs['loss']).backward()
                if self.reducer is not None:
                    self.reducer.reduce()
                ctl = self.optimizer.step(max_norm=self.grad_clip['max_norm'] if self.grad_clip else None, loss_scale=scale,
                                          skip_nonfinite=self.loss_scale is not None)
                if self.grad_clip is not None and ctl is not None:
                    outputs['log_vars']['grad_norm'] = ctl[0]
            elif self.loss_scaler is not None:
                # Fp16OptimizerHook (mmcv/runner/hooks/optimizer.py, the torch >= 1.6 form): scaled backward,
                # unscale, clip, a step that is skipped on inf / nan gradients, and -- static mode -- the
                # scale reset to `loss_scale` every iteration
                self.loss_scaler.scale(outputs['loss']).backward()
                if self.reducer is not None:
                    self.reducer.reduce()
                self.loss_scaler.unscale_(self.optimizer)
                if self.grad_clip is not None:
                    gn = self._clip()
                    if gn is not None:
                        outputs['log_vars']['grad_norm'] = float(gn)
                self.loss_scaler.step(self.optimizer)
                self.loss_scaler.update(self.loss_scale)
            else:
                outputs['loss'].backward()
                if self.reducer is not None:
                    self.reducer.reduce()
                if self.grad_clip is not None:
                    gn = self._clip()
                    if gn is not None:
                        outputs['log_vars']['grad_norm'] = float(gn)
                self.optimizer.step()
            if self.check_replicas and torch.distributed.is_available() and torch.distributed.is_initialized():
                from .distributed import replicas_identical
                if not replicas_identical(self.model):
                    raise RuntimeError(f'iteration {self.iter}: the data-parallel replicas no longer hold identical '
                                       f'parameters (rank {get_dist_info()[0]})')
                self.replica_checks = getattr(self, 'replica_checks', 0) + 1
            for k, v in outputs['log_vars'].items():
                self.log_buffer.setdefault(k, []).append(v)
            self.iter += 1
            n_since += 1
            if (i + 1) % self.log_interval == 0 or i + 1 == len(data_loader):
                now = time.time()
                self._log(data_loader, (now - t_last) / max(n_since, 1))        # reads the log scalars: a sync point
                if torch.cuda.is_available():
                    from . import lib as _lib
                    _lib.handover_status()      # a lost stream-K hand-over since the last log line raises here
                if gc_ is not None:
                    gc_.collect()               # the cyclic garbage of the last `log_interval` steps, between two steps
                    now = time.time()           # (not charged to the next interval's time per iteration)
                t_last, n_since = now, 0
        if getattr(self, 'replica_checks', 0):
            self.logger.info(f'replica check: parameters and buffers bit-identical on every rank after each of '
                             f'{self.replica_checks} optimizer steps')
        self.epoch += 1

    def _ddp_step(self, data):
        """DistributedDataParallel hooks fire on forward(): run the module's train_step body
        through the wrapper (MMDistributedDataParallel.train_step does the same)"""
        losses = self.model(**data)
        loss, log_vars = self.model.module._parse_losses(losses)
        return dict(loss=loss, log_vars=log_vars, num_samples=len(data['img_metas']))

    def run(self, data_loaders, workflow=(('train', 1),)):
        assert len(data_loaders) == 1 and workflow[0][0] == 'train'
        loader = data_loaders[0]
        self.lr_updater.before_run(self.optimizer)
        self.logger.info(f'Start running, work_dir: {self.work_dir}, max: {self.max_epochs} epochs')
        while self.epoch < self.max_epochs:
            self.train(loader)
            rank, _ = get_dist_info()
            if rank == 0 and self.ckpt_interval > 0 and self.epoch % self.ckpt_interval == 0:
                self.epoch -= 1
                self.save_checkpoint(self.work_dir)
                self.epoch += 1
            if self.eval_fn is not None and self.epoch % self.eval_interval == 0:
                res = self.eval_fn(self)
                if res is not None:
                    self.eval_history.append((self.epoch, res))
                    self.logger.info(f'Epoch(val) [{self.epoch}]\t' + ', '.join(f'{k}: {v}' for k, v in res.items()))


def replace_ImageToTensor(pipelines):
    """datasets/utils.py: for test batches > 1 the images must be padded-stacked, i.e. formatted
    by DefaultFormatBundle instead of ImageToTensor"""
    import copy
    pipelines = copy.deepcopy(pipelines)
    for i, p in enumerate(pipelines):
        if p['type'] == 'MultiScaleFlipAug':
            p['transforms'] = replace_ImageToTensor(p['transforms'])
        elif p['type'] == 'ImageToTensor':
            pipelines[i] = {'type': 'DefaultFormatBundle'}
    return pipelines


def _to_device(data, device):
    if isinstance(data, torch.Tensor):
        return data.to(device, non_blocking=True)
    if isinstance(data, dict):
        return {k: (v if k == 'img_metas' else _to_device(v, device)) for k, v in data.items()}
    if isinstance(data, (list, tuple)):
        return [_to_device(v, device) for v in data]
    return data


class _DeviceLoader:
    """moves every batch to the model's device (the scatter step of MMDataParallel)"""

    def __init__(self, loader, device):
        self.loader, self.device = loader, device
        self.sampler = getattr(loader, 'sampler', None)
        self.dataset = loader.dataset

    def __len__(self):
        return len(self.loader)

    def __iter__(self):
        for data in self.loader:
            yield _to_device(data, self.device)


def train_detector(model, dataset, cfg, distributed=False, validate=False, timestamp=None, meta=None,
                   device=None):
    """apis/train.py:38-174"""
    logger = get_root_logger(log_level=cfg.get('log_level', 'INFO'))
    dataset = dataset if isinstance(dataset, (list, tuple)) else [dataset]
    rank, world = get_dist_info()
    if device is None:
        device = torch.device('cuda', torch.cuda.current_device()) if torch.cuda.is_available() else torch.device('cpu')
    loaders = [_DeviceLoader(build_dataloader(ds, cfg.data.samples_per_gpu, cfg.data.workers_per_gpu,
                                              len(cfg.get('gpu_ids', [0])), dist=distributed, seed=cfg.get('seed'),
                                              rank=rank, world_size=world), device) for ds in dataset]
    model = model.to(device)
    if device.type == 'cuda':
        from .blocks import conv_weights_channels_last
        conv_weights_channels_last(model)       # before DDP takes the parameters' strides for its bucket views
    reducer = None
    if distributed and (cfg.get('ddp', 'own') == 'torch' or device.type != 'cuda'):
        from torch.nn.parallel import DistributedDataParallel       # the reference's wrapper (apis/train.py:75-83)
        model = DistributedDataParallel(
            model, device_ids=[device.index] if device.type == 'cuda' else None, broadcast_buffers=False,
            find_unused_parameters=cfg.get('find_unused_parameters', False))
    elif distributed:
        # default on the HIP device: the weight-gradient arena all-reduced in place (distributed.GradReducer)
        from .distributed import GradReducer
        # `grad_allreduce_dtype = 'bf16'` in the config: the weight-gradient arena crosses xGMI as bf16 (fp32 master
        # weights and optimizer step unchanged)
        reducer = GradReducer([(n, p) for n, p in model.named_parameters() if p.requires_grad],
                              compress=cfg.get('grad_allreduce_dtype', None) or os.environ.get('BRCNN_REDUCER_COMPRESS') or None)
        reducer.broadcast_parameters(model)
    optimizer = build_optimizer(model, cfg.optimizer)
    runner_cfg = cfg.get('runner', None) or dict(type='EpochBasedRunner', max_epochs=cfg.total_epochs)
    assert runner_cfg['type'] == 'EpochBasedRunner'
    runner = EpochBasedRunner(model, optimizer, cfg.work_dir, logger, runner_cfg['max_epochs'], meta)
    runner.timestamp = timestamp
    runner.reducer = reducer
    runner.early_rpn_backward = bool(cfg.get('early_rpn_backward', True))
    runner.graph_trunk = bool(cfg.get('graph_trunk', False))
    runner.check_replicas = bool(cfg.get('check_replicas', runner.check_replicas))
    runner.manual_gc = bool(cfg.get('manual_gc', runner.manual_gc))
    if cfg.get('fp16', None) is not None:
        # `fp16 = dict(loss_scale=512.)` (configs/boosting_rcnn/boosting_rcnn_x101_pafpn_mstrain_3x_coco.py:2 ->
        # mmdet/apis/train.py:115-119, mmcv Fp16OptimizerHook): fp16 MFMA conv stack (fp32 accumulation, fp32
        # master weights, fp32 heads / losses) with static loss scaling; BRCNN_FP16_AS_BF16=1 keeps the
        # round-1 behaviour (bf16 conv stack, no scaling needed)
        module = model.module if hasattr(model, 'module') else model
        if os.environ.get('BRCNN_FP16_AS_BF16') == '1':
            logger.info(f'fp16={dict(cfg.fp16)} in the config: BRCNN_FP16_AS_BF16=1 -> bf16 conv stack, no loss scaling')
            module.set_compute_dtype('bf16')
        else:
            scale = cfg.fp16.get('loss_scale', 512.)
            if isinstance(scale, str) or isinstance(scale, dict):
                raise NotImplementedError(f'fp16 loss_scale={scale!r}: only the static float form of the recipes is built')
            module.set_compute_dtype('f16')
            runner.loss_scale = float(scale)
            runner.loss_scaler = torch.amp.GradScaler('cuda', init_scale=float(scale), enabled=device.type == 'cuda')
            logger.info(f'fp16={dict(cfg.fp16)}: fp16 MFMA conv stack, static loss scale {float(scale)}')
    _register_packed(optimizer, model.module if hasattr(model, 'module') else model)
    runner.register_training_hooks(cfg.lr_config, cfg.get('optimizer_config', None),
                                   cfg.get('checkpoint_config', None), cfg.get('log_config', None))
    for hook in cfg.get('custom_hooks', None) or []:
        if hook['type'] == 'NumClassCheckHook':
            _check_num_classes(model, dataset[0], logger)
        else:
            raise KeyError(f'custom hook {hook["type"]} is not available')
    if validate:
        val_spg = cfg.data.val.pop('samples_per_gpu', 1)
        if val_spg > 1:
            cfg.data.val.pipeline = replace_ImageToTensor(cfg.data.val.pipeline)
        val_dataset = build_dataset(cfg.data.val, dict(test_mode=True))
        val_loader = _DeviceLoader(build_dataloader(val_dataset, val_spg, cfg.data.workers_per_gpu, dist=distributed,
                                                    shuffle=False, rank=rank, world_size=world), device)
        eval_cfg = dict(cfg.get('evaluation', {}))
        interval = eval_cfg.pop('interval', 1)

        def do_eval(r):
            if distributed:
                results = multi_gpu_test(r.model, val_loader)
                if get_dist_info()[0] != 0:
                    return None
            else:
                results = single_gpu_test(r.model, val_loader)
            kw = {k: v for k, v in eval_cfg.items() if k not in ('by_epoch', 'save_best', 'rule', 'start')}
            return val_dataset.evaluate(results, logger=logger, **kw)
        runner.register_eval(do_eval, interval)
    if cfg.get('resume_from', None):
        runner.resume(cfg.resume_from)
    elif cfg.get('load_from', None):
        runner.load_checkpoint(cfg.load_from)
    runner.run(loaders, cfg.get('workflow', [('train', 1)]))
    if reducer is not None:
        reducer.finish()            # (strict=False only: the last step's deferred layout comparison is read and checked)
    return runner


def _register_packed(optimizer, module):
    """FusedSGD keeps the conv weights' packed operands of the next step current in the compute dtype"""
    from . import blocks
    from .optim import FusedSGD
    if isinstance(optimizer, FusedSGD):
        optimizer.register_conv_weights(module, blocks.compute_dtype())


def _check_num_classes(model, dataset, logger):
    """core/hook/checkloss... NumClassCheckHook: every head's num_classes must equal len(CLASSES)"""
    module = model.module if hasattr(model, 'module') else model
    classes = dataset.CLASSES
    if classes is None:
        logger.warning(f'Please set `CLASSES` in the {dataset.__class__.__name__}')
        return
    assert not isinstance(classes, str), f'`CLASSES` in {dataset.__class__.__name__} should be a tuple of str'
    for name, m in module.named_modules():
        if hasattr(m, 'num_classes') and type(m).__name__ not in ('RPNHead', 'ATSSRPNHead', 'VGG', 'FusedSemanticHead'):
            assert m.num_classes == len(classes), (
                f'The `num_classes` ({m.num_classes}) in {type(m).__name__} of {type(module).__name__} does not '
                f'match the length of `CLASSES` {len(classes)} in {dataset.__class__.__name__}')


# --------------------------------------------------------------------------- testing
def single_gpu_test(model, data_loader, show=False, out_dir=None, show_score_thr=0.3):
    """apis/test.py:16-79 (no visualisation on this path)"""
    model.eval()
    results = []
    for data in data_loader:
        with torch.no_grad():
            result = model(return_loss=False, rescale=True, **data)
        results.extend(result)
    return results


def multi_gpu_test(model, data_loader, tmpdir=None, gpu_collect=True):
    """apis/test.py:82-169: every rank tests its round-robin shard, rank 0 gets the results in
    dataset order (interleave the shards, drop the sampler's padding)"""
    model.eval()
    results = []
    for data in data_loader:
        with torch.no_grad():
            result = model(return_loss=False, rescale=True, **data)
        results.extend(result)
    rank, world = get_dist_info()
    if world == 1:
        return results
    parts = [None] * world
    dist.all_gather_object(parts, results)
    if rank != 0:
        return None
    ordered = []
    for res in zip(*parts):
        ordered.extend(list(res))
    return ordered[:len(data_loader.dataset)]
