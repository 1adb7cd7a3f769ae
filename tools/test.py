#!/usr/bin/env python
"""Test (and evaluate) a detector (counterpart of the reference's tools/test.py:1-240).

    python tools/test.py CONFIG CHECKPOINT --eval bbox [--out results.pkl] [--format-only]
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 tools/test.py CONFIG CKPT --launcher pytorch --eval bbox
"""
import argparse
import os
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')     # before the HIP runtime starts (see brcnn/__init__.py)
import os.path as osp
import pickle
import sys

sys.path.insert(0, osp.dirname(osp.dirname(osp.abspath(__file__))))
import torch  # noqa: E402

import brcnn  # noqa: E402,F401
from brcnn import Config, build_detector  # noqa: E402
from brcnn.apis import (_DeviceLoader, get_dist_info, init_dist, limit_host_threads, load_checkpoint,  # noqa: E402
                        multi_gpu_test,
                        replace_ImageToTensor, single_gpu_test)
from brcnn.config import DictAction  # noqa: E402
from brcnn.datasets import build_dataloader, build_dataset  # noqa: E402


def parse_args(argv=None):
    p = argparse.ArgumentParser(description='test (and eval) a model')
    p.add_argument('config', help='test config file path')
    p.add_argument('checkpoint', help='checkpoint file')
    p.add_argument('--out', help='output result file in pickle format')
    p.add_argument('--format-only', action='store_true', help='write the result json without evaluating')
    p.add_argument('--eval', type=str, nargs='+', help='evaluation metrics ("bbox")')
    p.add_argument('--cfg-options', nargs='+', action=DictAction)
    p.add_argument('--eval-options', nargs='+', action=DictAction,
                   help='kwargs of dataset.evaluate() / format_results(), key=value')
    p.add_argument('--launcher', choices=['none', 'pytorch'], default='none')
    p.add_argument('--dist-backend', default=None)
    p.add_argument('--dtype', choices=['f32', 'bf16', 'f16'], default='f32', help='arithmetic type of the conv stack')
    p.add_argument('--device-preprocess', action='store_true')
    p.add_argument('--local_rank', type=int, default=0)
    args = p.parse_args(argv)
    os.environ.setdefault('LOCAL_RANK', str(args.local_rank))
    return args


def main(argv=None):
    args = parse_args(argv)
    assert args.out or args.eval or args.format_only, \
        'Please specify at least one operation (save/eval/format the results) with "--out", "--eval", "--format-only"'
    if args.eval and args.format_only:
        raise ValueError('--eval and --format_only cannot be both specified')
    if args.out is not None and not args.out.endswith(('.pkl', '.pickle')):
        raise ValueError('The output file must be a pkl file.')
    cfg = Config.fromfile(args.config)
    if args.cfg_options is not None:
        cfg.merge_from_dict(args.cfg_options)
    cfg.model.pretrained = None
    if cfg.model.get('backbone', None) and cfg.model.backbone.get('init_cfg', None):
        cfg.model.backbone.init_cfg = None
    cfg.data.test.test_mode = True
    spg = cfg.data.test.pop('samples_per_gpu', 1)
    if spg > 1:
        cfg.data.test.pipeline = replace_ImageToTensor(cfg.data.test.pipeline)
    if args.device_preprocess:
        from brcnn.pipelines import fuse_device_pipeline
        cfg.data.test.pipeline = fuse_device_pipeline(cfg.data.test.pipeline)
    distributed = args.launcher != 'none'
    if distributed:
        params = dict(cfg.get('dist_params', dict(backend='nccl')))
        if args.dist_backend:
            params['backend'] = args.dist_backend
        init_dist(args.launcher, **params)
    rank, world = get_dist_info()
    limit_host_threads(int(os.environ.get('LOCAL_WORLD_SIZE', 1)))
    if not torch.cuda.is_available():
        raise RuntimeError('tools/test.py needs a GPU: the hot path has no CPU fallback')
    device = torch.device('cuda', torch.cuda.current_device())
    dataset = build_dataset(cfg.data.test)
    loader = _DeviceLoader(build_dataloader(dataset, spg, cfg.data.workers_per_gpu, dist=distributed, shuffle=False,
                                            rank=rank, world_size=world), device)
    cfg.model.train_cfg = None
    model = build_detector(cfg.model, test_cfg=cfg.get('test_cfg'))
    ckpt = load_checkpoint(model, args.checkpoint, map_location='cpu')
    model.CLASSES = ckpt.get('meta', {}).get('CLASSES', dataset.CLASSES)
    model = model.to(device).eval()
    model.set_compute_dtype(args.dtype)
    outputs = multi_gpu_test(model, loader) if distributed else single_gpu_test(model, loader)
    if rank != 0:
        return None
    if args.out:
        print(f'\nwriting results to {args.out}')
        with open(args.out, 'wb') as f:
            pickle.dump(outputs, f)
    kwargs = {} if args.eval_options is None else args.eval_options
    if args.format_only:
        dataset.format_results(outputs, **kwargs)
        return outputs
    if args.eval:
        eval_kwargs = dict(cfg.get('evaluation', {}))
        for key in ['interval', 'tmpdir', 'start', 'gpu_collect', 'save_best', 'rule', 'by_epoch']:
            eval_kwargs.pop(key, None)
        eval_kwargs.update(dict(metric=args.eval, **kwargs))
        metric = dataset.evaluate(outputs, **eval_kwargs)
        print(metric)
        return metric
    return outputs


if __name__ == '__main__':
    main()
