#!/usr/bin/env python
"""Train a detector from a config (counterpart of the reference's tools/train.py:1-190).

    python tools/train.py configs/boosting_rcnn/boosting_rcnn_r50_pafpn_1x_utdac.py --work-dir work_dirs/x
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 tools/train.py CONFIG --launcher pytorch
"""
import argparse
import copy
import os
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')     # before the HIP runtime starts (see brcnn/__init__.py)
import os.path as osp
import sys
import time

sys.path.insert(0, osp.dirname(osp.dirname(osp.abspath(__file__))))
import torch  # noqa: E402

import brcnn  # noqa: E402,F401
from brcnn import Config, build_detector  # noqa: E402
from brcnn.apis import (get_dist_info, get_root_logger, init_dist, limit_host_threads, set_random_seed,  # noqa: E402
                        train_detector)
from brcnn.config import DictAction  # noqa: E402
from brcnn.datasets import build_dataset  # noqa: E402


def parse_args(argv=None):
    p = argparse.ArgumentParser(description='Train a detector')
    p.add_argument('config', help='train config file path')
    p.add_argument('--work-dir', help='the dir to save logs and models')
    p.add_argument('--resume-from', help='the checkpoint file to resume from')
    p.add_argument('--no-validate', action='store_true', help='do not evaluate the checkpoint during training')
    g = p.add_mutually_exclusive_group()
    g.add_argument('--gpus', type=int, help='number of gpus to use (non-distributed: 1)')
    g.add_argument('--gpu-ids', type=int, nargs='+', help='ids of gpus to use (non-distributed: one id)')
    p.add_argument('--seed', type=int, default=None, help='random seed')
    p.add_argument('--deterministic', action='store_true')
    p.add_argument('--cfg-options', nargs='+', action=DictAction,
                   help='override config settings, key=value pairs (xxx=yyy, lists as a,b or "[a,b]")')
    p.add_argument('--launcher', choices=['none', 'pytorch'], default='none', help='job launcher')
    p.add_argument('--allow-random-init', action='store_true',
                   help="train from random weights when the backbone's pretrained checkpoint "
                        '(init_cfg, resolved under $BRCNN_PRETRAINED_DIR) is not available')
    p.add_argument('--dist-backend', default=None, help="override dist_params.backend ('gloo' for CPU runs)")
    p.add_argument('--device-preprocess', action='store_true',
                   help='run Resize/Flip/Normalize/Pad as the fused HIP kernel on the uploaded uint8 image')
    p.add_argument('--local_rank', type=int, default=0)
    args = p.parse_args(argv)
    os.environ.setdefault('LOCAL_RANK', str(args.local_rank))
    return args


def main(argv=None):
    args = parse_args(argv)
    cfg = Config.fromfile(args.config)
    if args.cfg_options is not None:
        cfg.merge_from_dict(args.cfg_options)
    if args.work_dir is not None:
        cfg.work_dir = args.work_dir
    elif cfg.get('work_dir', None) is None:
        cfg.work_dir = osp.join('./work_dirs', osp.splitext(osp.basename(args.config))[0])
    if args.resume_from is not None:
        cfg.resume_from = args.resume_from
    cfg.gpu_ids = args.gpu_ids if args.gpu_ids is not None else list(range(1 if args.gpus is None else args.gpus))
    if args.launcher == 'none':
        distributed = False
    else:
        distributed = True
        params = dict(cfg.get('dist_params', dict(backend='nccl')))
        if args.dist_backend:
            params['backend'] = args.dist_backend
        init_dist(args.launcher, **params)
        _, world_size = get_dist_info()
        cfg.gpu_ids = list(range(world_size))
    limit_host_threads(int(os.environ.get('LOCAL_WORLD_SIZE', 1)))
    os.makedirs(osp.abspath(cfg.work_dir), exist_ok=True)
    cfg.dump(osp.join(cfg.work_dir, osp.basename(args.config)))
    timestamp = time.strftime('%Y%m%d_%H%M%S', time.localtime())
    logger = get_root_logger(log_file=osp.join(cfg.work_dir, f'{timestamp}.log'), log_level=cfg.get('log_level', 'INFO'))
    meta = dict(config=cfg.pretty_text, seed=args.seed, exp_name=osp.basename(args.config))
    logger.info(f'Distributed training: {distributed}')
    if args.seed is not None:
        logger.info(f'Set random seed to {args.seed}, deterministic: {args.deterministic}')
        set_random_seed(args.seed, deterministic=args.deterministic)
    cfg.seed = args.seed
    if args.device_preprocess:
        from brcnn.pipelines import fuse_device_pipeline
        cfg.data.train.pipeline = fuse_device_pipeline(cfg.data.train.pipeline)
    model = build_detector(cfg.model, train_cfg=cfg.get('train_cfg'), test_cfg=cfg.get('test_cfg'))
    if args.allow_random_init:
        os.environ['BRCNN_ALLOW_RANDOM_INIT'] = '1'
    model.init_weights()
    datasets = [build_dataset(cfg.data.train)]
    assert len(cfg.get('workflow', [('train', 1)])) == 1, 'val workflow is not part of the recipes'
    model.CLASSES = datasets[0].CLASSES
    meta['CLASSES'] = datasets[0].CLASSES
    device = None
    if not torch.cuda.is_available():
        raise RuntimeError('tools/train.py needs a GPU: the hot path has no CPU fallback')
    return train_detector(model, datasets, cfg, distributed=distributed, validate=not args.no_validate,
                          timestamp=timestamp, meta=meta, device=device)


if __name__ == '__main__':
    main()
