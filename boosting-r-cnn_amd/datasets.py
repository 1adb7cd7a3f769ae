"""COCO-format dataset, samplers, collate and dataloader builder without
pycocotools / mmcv: the data side of `configs/_base_/datasets/*_coco.py`.

Mirrors mmdet/datasets/{custom.py,coco.py,builder.py,samplers/group_sampler.py,
samplers/distributed_sampler.py} and mmcv.parallel.collate for the fields the Boosting R-CNN
path consumes (img, img_metas, gt_bboxes, gt_labels, gt_bboxes_ignore).
"""
import copy
import json
import math
import os.path as osp
from collections import defaultdict
from functools import partial

import numpy as np
import torch
from torch.utils.data import DataLoader, Dataset, Sampler

from .pipelines import Compose, DataContainer
from .registry import Registry, build_from_cfg

DATASETS = Registry('dataset')


class COCO:
    """the part of pycocotools.coco.COCO the reference's CocoDataset calls
    (datasets/api_wrappers/coco_api.py:11-46): index of images / annotations / categories"""

    def __init__(self, annotation_file=None):
        self.dataset, self.anns, self.cats, self.imgs = {}, {}, {}, {}
        self.imgToAnns, self.catToImgs = defaultdict(list), defaultdict(list)
        if annotation_file is not None:
            with open(annotation_file) as f:
                self.dataset = json.load(f)
            assert isinstance(self.dataset, dict), 'annotation file format not supported'
            self.createIndex()

    def createIndex(self):
        for ann in self.dataset.get('annotations', []):
            self.imgToAnns[ann['image_id']].append(ann)
            self.anns[ann['id']] = ann
        for img in self.dataset.get('images', []):
            self.imgs[img['id']] = img
        for cat in self.dataset.get('categories', []):
            self.cats[cat['id']] = cat
        for ann in self.dataset.get('annotations', []):
            self.catToImgs[ann['category_id']].append(ann['image_id'])

    def get_cat_ids(self, cat_names=()):
        cats = self.dataset.get('categories', [])
        if len(cat_names):
            cats = [c for c in cats if c['name'] in cat_names]
        return [c['id'] for c in cats]

    def get_img_ids(self):
        return list(self.imgs.keys())

    def get_ann_ids(self, img_ids=()):
        img_ids = img_ids if isinstance(img_ids, (list, tuple)) else [img_ids]
        return [a['id'] for i in img_ids for a in self.imgToAnns.get(i, [])]

    def load_anns(self, ids):
        return [self.anns[i] for i in ids]

    def load_imgs(self, ids):
        return [self.imgs[i] for i in ids]

    def load_cats(self, ids):
        return [self.cats[i] for i in ids]

    def loadRes(self, results):
        """detections (list of dict image_id/bbox/score/category_id) as a COCO index"""
        res = COCO()
        res.dataset['images'] = list(self.dataset['images'])
        res.dataset['categories'] = copy.deepcopy(self.dataset['categories'])
        anns = copy.deepcopy(results)
        for i, ann in enumerate(anns):
            x, y, w, h = ann['bbox']
            ann['area'] = w * h
            ann['id'] = i + 1
            ann['iscrowd'] = 0
        res.dataset['annotations'] = anns
        res.createIndex()
        return res


@DATASETS.register_module()
class CustomDataset(Dataset):
    """datasets/custom.py:15-330 (annotation parsing left to the subclass)"""
    CLASSES = None

    def __init__(self, ann_file, pipeline, classes=None, data_root=None, img_prefix='', seg_prefix=None,
                 proposal_file=None, test_mode=False, filter_empty_gt=True):
        def under_root(path):           # relative paths are taken below data_root (custom.py:78-88)
            if data_root is None or path is None or osp.isabs(path):
                return path
            return osp.join(data_root, path)
        self.data_root, self.test_mode, self.filter_empty_gt = data_root, test_mode, filter_empty_gt
        self.ann_file, self.img_prefix = under_root(ann_file), under_root(img_prefix)
        self.CLASSES = self.get_classes(classes)
        self.proposals = None
        self.data_infos = self.load_annotations(self.ann_file)
        if not test_mode:               # training: drop what _filter_imgs rejects, then the aspect-ratio groups
            self.data_infos = [self.data_infos[i] for i in self._filter_imgs()]
            self._set_group_flag()
        self.pipeline = Compose(pipeline)

    def __len__(self):
        return len(self.data_infos)

    @classmethod
    def get_classes(cls, classes=None):
        """custom.py:171-200: None -> the dataset's own names, a path -> one name per non-empty line, a sequence as it is"""
        if classes is None:
            return cls.CLASSES
        if isinstance(classes, (tuple, list)):
            return classes
        if isinstance(classes, str):
            with open(classes) as f:
                return [name for name in (line.strip() for line in f) if name]
        raise ValueError(f'Unsupported type {type(classes)} of classes.')

    def pre_pipeline(self, results):
        results['img_prefix'] = self.img_prefix
        results['seg_prefix'] = None
        results['proposal_file'] = None
        results['bbox_fields'] = []
        results['mask_fields'] = []
        results['seg_fields'] = []

    def _filter_imgs(self, min_size=32):
        return [i for i, info in enumerate(self.data_infos) if min(info['width'], info['height']) >= min_size]

    def _set_group_flag(self):
        """images with aspect ratio > 1 form group 1, the rest group 0"""
        self.flag = np.zeros(len(self), dtype=np.uint8)
        for i in range(len(self)):
            info = self.data_infos[i]
            if info['width'] / info['height'] > 1:
                self.flag[i] = 1

    def _rand_another(self, idx):
        pool = np.where(self.flag == self.flag[idx])[0]
        return np.random.choice(pool)

    def __getitem__(self, idx):
        if self.test_mode:
            return self.prepare_test_img(idx)
        while True:
            data = self.prepare_train_img(idx)
            if data is None:
                idx = self._rand_another(idx)
                continue
            return data

    def prepare_train_img(self, idx):
        results = dict(img_info=self.data_infos[idx], ann_info=self.get_ann_info(idx))
        self.pre_pipeline(results)
        return self.pipeline(results)

    def prepare_test_img(self, idx):
        results = dict(img_info=self.data_infos[idx])
        self.pre_pipeline(results)
        return self.pipeline(results)


@DATASETS.register_module()
class CocoDataset(CustomDataset):
    """datasets/coco.py:22-560 (bbox annotations, result json, bbox mAP)"""
    CLASSES = ('person', 'bicycle', 'car', 'motorcycle', 'airplane', 'bus', 'train', 'truck', 'boat',
               'traffic light', 'fire hydrant', 'stop sign', 'parking meter', 'bench', 'bird', 'cat', 'dog',
               'horse', 'sheep', 'cow', 'elephant', 'bear', 'zebra', 'giraffe', 'backpack', 'umbrella',
               'handbag', 'tie', 'suitcase', 'frisbee', 'skis', 'snowboard', 'sports ball', 'kite',
               'baseball bat', 'baseball glove', 'skateboard', 'surfboard', 'tennis racket', 'bottle',
               'wine glass', 'cup', 'fork', 'knife', 'spoon', 'bowl', 'banana', 'apple', 'sandwich',
               'orange', 'broccoli', 'carrot', 'hot dog', 'pizza', 'donut', 'cake', 'chair', 'couch',
               'potted plant', 'bed', 'dining table', 'toilet', 'tv', 'laptop', 'mouse', 'remote',
               'keyboard', 'cell phone', 'microwave', 'oven', 'toaster', 'sink', 'refrigerator', 'book',
               'clock', 'vase', 'scissors', 'teddy bear', 'hair drier', 'toothbrush')

    def load_annotations(self, ann_file):
        self.coco = COCO(ann_file)
        self.cat_ids = self.coco.get_cat_ids(cat_names=self.CLASSES)
        self.cat2label = {cat_id: i for i, cat_id in enumerate(self.cat_ids)}
        self.img_ids = self.coco.get_img_ids()
        data_infos, total_ann_ids = [], []
        for i in self.img_ids:
            info = self.coco.load_imgs([i])[0]
            info['filename'] = info['file_name']
            data_infos.append(info)
            total_ann_ids.extend(self.coco.get_ann_ids(img_ids=[i]))
        assert len(set(total_ann_ids)) == len(total_ann_ids), f"Annotation ids in '{ann_file}' are not unique!"
        return data_infos

    def get_ann_info(self, idx):
        img_id = self.data_infos[idx]['id']
        ann_info = self.coco.load_anns(self.coco.get_ann_ids(img_ids=[img_id]))
        return self._parse_ann_info(self.data_infos[idx], ann_info)

    def _filter_imgs(self, min_size=32):
        valid_inds = []
        ids_with_ann = set(a['image_id'] for a in self.coco.anns.values())
        ids_in_cat = set()
        for class_id in self.cat_ids:
            ids_in_cat |= set(self.coco.catToImgs[class_id])
        ids_in_cat &= ids_with_ann
        valid_img_ids = []
        for i, info in enumerate(self.data_infos):
            img_id = self.img_ids[i]
            if self.filter_empty_gt and img_id not in ids_in_cat:
                continue
            if min(info['width'], info['height']) >= min_size:
                valid_inds.append(i)
                valid_img_ids.append(img_id)
        self.img_ids = valid_img_ids
        return valid_inds

    def _parse_ann_info(self, img_info, ann_info):
        gt_bboxes, gt_labels, gt_bboxes_ignore = [], [], []
        for ann in ann_info:
            if ann.get('ignore', False):
                continue
            x1, y1, w, h = ann['bbox']
            inter_w = max(0, min(x1 + w, img_info['width']) - max(x1, 0))
            inter_h = max(0, min(y1 + h, img_info['height']) - max(y1, 0))
            if inter_w * inter_h == 0:
                continue
            if ann['area'] <= 0 or w < 1 or h < 1:
                continue
            if ann['category_id'] not in self.cat_ids:
                continue
            bbox = [x1, y1, x1 + w, y1 + h]
            if ann.get('iscrowd', False):
                gt_bboxes_ignore.append(bbox)
            else:
                gt_bboxes.append(bbox)
                gt_labels.append(self.cat2label[ann['category_id']])
        if gt_bboxes:
            gt_bboxes = np.array(gt_bboxes, dtype=np.float32)
            gt_labels = np.array(gt_labels, dtype=np.int64)
        else:
            gt_bboxes = np.zeros((0, 4), dtype=np.float32)
            gt_labels = np.array([], dtype=np.int64)
        if gt_bboxes_ignore:
            gt_bboxes_ignore = np.array(gt_bboxes_ignore, dtype=np.float32)
        else:
            gt_bboxes_ignore = np.zeros((0, 4), dtype=np.float32)
        return dict(bboxes=gt_bboxes, labels=gt_labels, bboxes_ignore=gt_bboxes_ignore, masks=[],
                    seg_map=img_info['filename'].replace('jpg', 'png'))

    # ---- results ------------------------------------------------------------------------
    @staticmethod
    def xyxy2xywh(bbox):
        b = bbox.tolist()
        return [b[0], b[1], b[2] - b[0], b[3] - b[1]]

    def _det2json(self, results):
        json_results = []
        for idx in range(len(self)):
            img_id = self.img_ids[idx]
            for label, bboxes in enumerate(results[idx]):
                for i in range(bboxes.shape[0]):
                    json_results.append(dict(image_id=img_id, bbox=self.xyxy2xywh(bboxes[i]),
                                             score=float(bboxes[i][4]), category_id=self.cat_ids[label]))
        return json_results

    def _proposal2json(self, results):
        """class-agnostic proposals (k, 5) per image (coco.py:208-221): category 1"""
        json_results = []
        for idx in range(len(self)):
            img_id = self.img_ids[idx]
            bboxes = results[idx]
            for i in range(bboxes.shape[0]):
                json_results.append(dict(image_id=img_id, bbox=self.xyxy2xywh(bboxes[i]), score=float(bboxes[i][4]),
                                         category_id=1))
        return json_results

    def results2json(self, results, outfile_prefix):
        """coco.py:275-309: per-class lists -> '<prefix>.bbox.json' (also the 'proposal' file); one (k, 5) array per
        image -> '<prefix>.proposal.json'.  (Mask tuples are outside the hot path.)"""
        if isinstance(results[0], list):
            out = f'{outfile_prefix}.bbox.json'
            with open(out, 'w') as f:
                json.dump(self._det2json(results), f)
            return dict(bbox=out, proposal=out)
        if isinstance(results[0], np.ndarray):
            out = f'{outfile_prefix}.proposal.json'
            with open(out, 'w') as f:
                json.dump(self._proposal2json(results), f)
            return dict(proposal=out)
        raise TypeError('invalid type of results')

    def fast_eval_recall(self, results, proposal_nums, iou_thrs, logger=None):
        """coco.py:311-333: average recall over `iou_thrs` per proposal number, against the non-crowd, non-ignored
        ground-truth boxes; `results` are per-image proposals (k, 4 | 5)"""
        from .evaluation import eval_recalls
        gt_bboxes = []
        for img_id in self.img_ids:
            anns = self.coco.load_anns(self.coco.get_ann_ids(img_ids=[img_id]))
            boxes = [[a['bbox'][0], a['bbox'][1], a['bbox'][0] + a['bbox'][2], a['bbox'][1] + a['bbox'][3]]
                     for a in anns if not (a.get('ignore', False) or a['iscrowd'])]
            gt_bboxes.append(np.array(boxes, dtype=np.float32) if boxes else np.zeros((0, 4)))
        recalls = eval_recalls(gt_bboxes, results, proposal_nums, iou_thrs, logger=logger)
        return recalls.mean(axis=1)

    def format_results(self, results, jsonfile_prefix=None, **kwargs):
        assert isinstance(results, list), 'results must be a list'
        assert len(results) == len(self), (
            f'The length of results is not equal to the dataset len: {len(results)} != {len(self)}')
        tmp_dir = None
        if jsonfile_prefix is None:
            import tempfile
            tmp_dir = tempfile.TemporaryDirectory()
            jsonfile_prefix = osp.join(tmp_dir.name, 'results')
        return self.results2json(results, jsonfile_prefix), tmp_dir

    def evaluate(self, results, metric='bbox', logger=None, jsonfile_prefix=None, classwise=False,
                 proposal_nums=(100, 300, 1000), iou_thrs=None, metric_items=None):
        """coco.py:362-560: 'bbox' (COCO mAP), 'proposal' (class-agnostic AR: useCats = 0) through this repo's COCOeval
        restatement, 'proposal_fast' (eval_recalls on per-image proposals).  'segm' needs masks: outside the hot path."""
        from .evaluation import COCOeval
        metrics = metric if isinstance(metric, list) else [metric]
        for m in metrics:
            if m not in ('bbox', 'proposal', 'proposal_fast'):
                raise KeyError(f'metric {m} is not supported (bbox / proposal / proposal_fast on this path)')
        if iou_thrs is None:
            iou_thrs = np.linspace(.5, 0.95, int(np.round((0.95 - .5) / .05)) + 1, endpoint=True)
        if metric_items is not None and not isinstance(metric_items, list):
            metric_items = [metric_items]
        eval_results = {}
        for m in metrics:
            if m == 'proposal_fast':
                ar = self.fast_eval_recall(results, proposal_nums, iou_thrs, logger='silent')
                for i, num in enumerate(proposal_nums):
                    eval_results[f'AR@{num}'] = ar[i]
                if logger is not None:
                    logger.info(''.join(f'\nAR@{num}\t{ar[i]:.4f}' for i, num in enumerate(proposal_nums)))
                continue
            part = self._evaluate_coco(results, m, logger, jsonfile_prefix, classwise, proposal_nums, iou_thrs, metric_items)
            if part is None:
                break
            eval_results.update(part)
        return eval_results

    def _evaluate_coco(self, results, metric, logger, jsonfile_prefix, classwise, proposal_nums, iou_thrs, metric_items):
        from .evaluation import COCOeval
        eval_results = {}
        dets = self._det2json(results) if isinstance(results[0], list) else self._proposal2json(results)
        if jsonfile_prefix is not None:
            kind = 'bbox' if isinstance(results[0], list) else 'proposal'
            with open(f'{jsonfile_prefix}.{kind}.json', 'w') as f:
                json.dump(dets, f)
        if metric == 'bbox' and not isinstance(results[0], list):
            raise KeyError('bbox is not in results')
        if len(dets) == 0:
            if logger is not None:
                logger.error('The testing results of the whole dataset is empty.')
            return None
        coco_dt = self.coco.loadRes(dets)
        ev = COCOeval(self.coco, coco_dt, 'bbox')
        ev.params.catIds = self.cat_ids
        ev.params.imgIds = self.img_ids
        ev.params.maxDets = list(proposal_nums)
        ev.params.iouThrs = iou_thrs
        names = {'mAP': 0, 'mAP_50': 1, 'mAP_75': 2, 'mAP_s': 3, 'mAP_m': 4, 'mAP_l': 5,
                 'AR@100': 6, 'AR@300': 7, 'AR@1000': 8, 'AR_s@1000': 9, 'AR_m@1000': 10, 'AR_l@1000': 11}
        if metric_items is not None:
            for item in metric_items:
                if item not in names:
                    raise KeyError(f'metric item {item} is not supported')
        if metric == 'proposal':
            ev.params.useCats = 0
            ev.evaluate()
            ev.accumulate()
            text = ev.summarize()
            if logger is not None:
                logger.info('\n' + text)
            for item in (metric_items or ['AR@100', 'AR@300', 'AR@1000', 'AR_s@1000', 'AR_m@1000', 'AR_l@1000']):
                eval_results[item] = float(f'{ev.stats[names[item]]:.3f}')
            return eval_results
        ev.evaluate()
        ev.accumulate()
        text = ev.summarize()
        if logger is not None:
            logger.info('\n' + text)
        if classwise:
            precisions = ev.eval['precision']      # (T, R, K, A, M)
            per_class = {}
            for idx, cat_id in enumerate(self.cat_ids):
                pr = precisions[:, :, idx, 0, -1]
                pr = pr[pr > -1]
                per_class[self.coco.load_cats([cat_id])[0]['name']] = float(np.mean(pr)) if pr.size else float('nan')
            eval_results['bbox_classwise'] = per_class
        if metric_items is None:
            metric_items = ['mAP', 'mAP_50', 'mAP_75', 'mAP_s', 'mAP_m', 'mAP_l']
        for item in metric_items:
            eval_results[f'bbox_{item}'] = float(f'{ev.stats[names[item]]:.3f}')
        ap = ev.stats[:6]
        eval_results['bbox_mAP_copypaste'] = ' '.join(f'{v:.3f}' for v in ap)
        return eval_results


# --------------------------------------------------------------------------- samplers
def _aspect_groups(flag, quantum):
    """the members of every non-empty aspect-ratio group (`dataset.flag`: 0 tall / 1 wide) with the number of slots
    the group fills once padded to a multiple of `quantum` (a batch, or a batch on every rank)"""
    flag = np.asarray(flag).astype(np.int64)
    out = []
    for g in range(int(flag.max()) + 1 if flag.size else 0):
        members = np.flatnonzero(flag == g)
        if members.size:
            out.append((members, -(-members.size // quantum) * quantum))
    return out


class GroupSampler(Sampler):
    """Single-process training order (samplers/group_sampler.py:11-53): every batch of `samples_per_gpu` holds images of
    ONE aspect-ratio group; groups are padded to whole batches by re-drawing members; batches are visited in random
    order.  The index stream is the reference's for the same `np.random` state (golden g13): per group one in-place
    shuffle and one `choice` for the padding (drawn even when it is empty), then ONE permutation of the batch rows."""

    def __init__(self, dataset, samples_per_gpu=1):
        assert hasattr(dataset, 'flag')
        self.dataset = dataset
        self.samples_per_gpu = int(samples_per_gpu)
        self.flag = np.asarray(dataset.flag).astype(np.int64)
        self.num_samples = sum(slots for _, slots in _aspect_groups(self.flag, self.samples_per_gpu))

    def __iter__(self):
        rows = []
        for members, slots in _aspect_groups(self.flag, self.samples_per_gpu):
            np.random.shuffle(members)
            pad = np.random.choice(members, slots - members.size)
            rows.append(np.concatenate([members, pad]).reshape(-1, self.samples_per_gpu))
        rows = np.concatenate(rows) if rows else np.zeros((0, self.samples_per_gpu), np.int64)
        order = np.random.permutation(range(len(rows)))
        return iter(rows[order].reshape(-1).astype(np.int64).tolist())

    def __len__(self):
        return self.num_samples


class DistributedGroupSampler(Sampler):
    """Data-parallel training order (samplers/group_sampler.py:56-148): the same grouping, padded so that every rank
    receives the same number of whole batches; the order is a function of (seed + epoch) alone -- every rank builds the
    full list from its own `torch.Generator` and keeps its contiguous share.  Index stream = the reference's (golden
    g13): per group one `randperm`, padding by cycling through that permutation, then ONE `randperm` of the batch rows."""

    def __init__(self, dataset, samples_per_gpu=1, num_replicas=None, rank=None, seed=0):
        if num_replicas is None or rank is None:
            import torch.distributed as dist
            live = dist.is_available() and dist.is_initialized()
            num_replicas = (dist.get_world_size() if live else 1) if num_replicas is None else num_replicas
            rank = (dist.get_rank() if live else 0) if rank is None else rank
        assert hasattr(dataset, 'flag')
        self.dataset = dataset
        self.samples_per_gpu = int(samples_per_gpu)
        self.num_replicas, self.rank = int(num_replicas), int(rank)
        self.seed = 0 if seed is None else seed
        self.epoch = 0
        self.flag = dataset.flag
        groups = _aspect_groups(self.flag, self.samples_per_gpu * self.num_replicas)
        self.total_size = sum(slots for _, slots in groups)
        self.num_samples = self.total_size // self.num_replicas

    def __iter__(self):
        gen = torch.Generator()
        gen.manual_seed(self.epoch + self.seed)
        rows = []
        for members, slots in _aspect_groups(self.flag, self.samples_per_gpu * self.num_replicas):
            shuffled = members[torch.randperm(members.size, generator=gen).numpy()]
            rows.append(np.resize(shuffled, slots).reshape(-1, self.samples_per_gpu))      # cyclic padding
        rows = np.concatenate(rows) if rows else np.zeros((0, self.samples_per_gpu), np.int64)
        order = torch.randperm(len(rows), generator=gen).numpy()
        per_rank = len(rows) // self.num_replicas
        mine = rows[order][self.rank * per_rank:(self.rank + 1) * per_rank]
        return iter(mine.reshape(-1).astype(np.int64).tolist())

    def __len__(self):
        return self.num_samples

    def set_epoch(self, epoch):
        self.epoch = epoch


class DistributedSampler(Sampler):
    """samplers/distributed_sampler.py:8-39 (test-time: in-order round-robin shards, padded)"""

    def __init__(self, dataset, num_replicas, rank, shuffle=False, seed=0):
        self.dataset, self.num_replicas, self.rank = dataset, num_replicas, rank
        self.shuffle, self.seed, self.epoch = shuffle, seed if seed is not None else 0, 0
        self.num_samples = int(math.ceil(len(dataset) / num_replicas))
        self.total_size = self.num_samples * num_replicas

    def __iter__(self):
        if self.shuffle:
            g = torch.Generator()
            g.manual_seed(self.epoch + self.seed)
            indices = torch.randperm(len(self.dataset), generator=g).tolist()
        else:
            indices = torch.arange(len(self.dataset)).tolist()
        indices = (indices * math.ceil(self.total_size / max(len(indices), 1)))[:self.total_size]
        assert len(indices) == self.total_size
        indices = indices[self.rank:self.total_size:self.num_replicas]
        assert len(indices) == self.num_samples
        return iter(indices)

    def __len__(self):
        return self.num_samples

    def set_epoch(self, epoch):
        self.epoch = epoch


# --------------------------------------------------------------------------- collate
def _pad_stack(tensors, pad_dims, padding_value):
    ndim = tensors[0].dim()
    max_shape = [0] * pad_dims
    for d in range(1, pad_dims + 1):
        max_shape[d - 1] = max(t.size(-d) for t in tensors)
    out = []
    for t in tensors:
        assert t.dim() == ndim
        pad = [0] * (pad_dims * 2)
        for d in range(1, pad_dims + 1):
            pad[2 * d - 1] = max_shape[d - 1] - t.size(-d)
        out.append(torch.nn.functional.pad(t, pad, value=padding_value) if any(pad) else t)
    return torch.stack(out, 0)


def collate(batch, samples_per_gpu=1):
    """mmcv.parallel.collate followed by the one-device scatter of MMDataParallel: stacked
    DataContainers become one zero-padded tensor, cpu_only ones a list of python objects, the
    others a list of tensors; sequences (test-time aug lists) are collated element-wise."""
    if not isinstance(batch, (list, tuple)):
        raise TypeError(f'{type(batch)} is not supported.')
    first = batch[0]
    if isinstance(first, DataContainer):
        assert len(batch) <= samples_per_gpu or len(batch) % samples_per_gpu == 0
        if first.cpu_only:
            return [s.data for s in batch]
        if first.stack:
            assert isinstance(first.data, torch.Tensor)
            if first.pad_dims is not None:
                return _pad_stack([s.data for s in batch], first.pad_dims, first.padding_value)
            return torch.stack([s.data for s in batch], 0)
        return [s.data for s in batch]
    if isinstance(first, (list, tuple)):
        return [collate(list(samples), samples_per_gpu) for samples in zip(*batch)]
    if isinstance(first, dict):
        return {key: collate([d[key] for d in batch], samples_per_gpu) for key in first}
    if isinstance(first, torch.Tensor):
        return _pad_stack(list(batch), 2, 0) if first.dim() >= 2 else torch.stack(list(batch), 0)
    return torch.utils.data.dataloader.default_collate(batch)


def worker_init_fn(worker_id, num_workers, rank, seed):
    worker_seed = num_workers * rank + worker_id + seed
    np.random.seed(worker_seed)
    import random
    random.seed(worker_seed)


def build_dataset(cfg, default_args=None):
    """datasets/builder.py:62-87"""
    if isinstance(cfg, (list, tuple)):
        return ConcatDataset([build_dataset(c, default_args) for c in cfg])
    if cfg['type'] == 'ConcatDataset':
        return ConcatDataset([build_dataset(c, default_args) for c in cfg['datasets']], cfg.get('separate_eval', True))
    if cfg['type'] == 'RepeatDataset':
        return RepeatDataset(build_dataset(cfg['dataset'], default_args), cfg['times'])
    if isinstance(cfg.get('ann_file'), (list, tuple)):
        import copy
        parts = []
        for i, ann in enumerate(cfg['ann_file']):
            c = copy.deepcopy(dict(cfg))
            c['ann_file'] = ann
            if isinstance(cfg.get('img_prefix'), (list, tuple)):
                c['img_prefix'] = cfg['img_prefix'][i]
            parts.append(build_dataset(c, default_args))
        return ConcatDataset(parts, cfg.get('separate_eval', True))
    return build_from_cfg(cfg, DATASETS, default_args)


def build_dataloader(dataset, samples_per_gpu, workers_per_gpu, num_gpus=1, dist=True, shuffle=True,
                     seed=None, rank=0, world_size=1, **kwargs):
    """datasets/builder.py:90-165"""
    if dist:
        if shuffle:
            sampler = DistributedGroupSampler(dataset, samples_per_gpu, world_size, rank, seed=seed)
        else:
            sampler = DistributedSampler(dataset, world_size, rank, shuffle=False, seed=seed)
        batch_size, num_workers = samples_per_gpu, workers_per_gpu
    else:
        sampler = GroupSampler(dataset, samples_per_gpu) if shuffle else None
        batch_size, num_workers = num_gpus * samples_per_gpu, num_gpus * workers_per_gpu
    init_fn = partial(worker_init_fn, num_workers=num_workers, rank=rank, seed=seed) if seed is not None else None
    from .pipelines import first_device_transform
    leaves = _pipeline_leaves(dataset)
    splits = {first_device_transform(d.pipeline) for d in leaves if isinstance(d.pipeline, Compose)}
    if splits - {None}:
        # the fused device front door (FusedResizeNormalizePad ...): workers are forked after the parent
        # has initialised the GPU and must not touch it, so they run the host part of the pipeline only
        # (file decode, annotations) and hand back raw samples; the device transforms and the collate step
        # run in the main process (MainProcessTail).  The dataset object itself keeps its whole pipeline
        # (`dataset[i]` and a second loader see it unchanged): the loader reads through HostPartView.
        if len(splits) != 1 or any(repr(d.pipeline) != repr(leaves[0].pipeline) for d in leaves):
            raise NotImplementedError('datasets under one wrapper with different pipelines and a device transform')
        split = splits.pop()
        tail = Compose(leaves[0].pipeline.transforms[split:])
        view = HostPartView(dataset, leaves, split)
        loader = DataLoader(view, batch_size=batch_size, sampler=sampler, num_workers=num_workers,
                            collate_fn=identity_collate, pin_memory=False, worker_init_fn=init_fn, **kwargs)
        return MainProcessTail(loader, tail, partial(collate, samples_per_gpu=samples_per_gpu))
    return DataLoader(dataset, batch_size=batch_size, sampler=sampler, num_workers=num_workers,
                      collate_fn=partial(collate, samples_per_gpu=samples_per_gpu), pin_memory=False,
                      worker_init_fn=init_fn, **kwargs)


def identity_collate(batch):
    return batch


def _pipeline_leaves(dataset):
    """the datasets that own a `pipeline` under `dataset` (itself, or below RepeatDataset / ConcatDataset wrappers)"""
    if getattr(dataset, 'pipeline', None) is not None:
        return [dataset]
    if hasattr(dataset, 'datasets'):
        return [d for sub_ in dataset.datasets for d in _pipeline_leaves(sub_)]
    if hasattr(dataset, 'dataset') and not isinstance(dataset.dataset, dict):
        return _pipeline_leaves(dataset.dataset)
    return []


class HostPartView(Dataset):
    """`dataset` as the DataLoader workers see it: every sample stops in front of the first device transform.  The
    underlying datasets are not modified outside `__getitem__` (a worker owns a forked copy; in the main process
    the swap is undone before returning)."""

    def __init__(self, dataset, leaves, split):
        self.dataset, self.leaves, self.split = dataset, leaves, split
        for name in ('flag', 'CLASSES'):
            if hasattr(dataset, name):
                setattr(self, name, getattr(dataset, name))

    def __len__(self):
        return len(self.dataset)

    def __getitem__(self, idx):
        full = [d.pipeline for d in self.leaves]
        for d in self.leaves:
            d.pipeline = Compose(d.pipeline.transforms[:self.split])
        try:
            return self.dataset[idx]
        finally:
            for d, p in zip(self.leaves, full):
                d.pipeline = p


class MainProcessTail:
    """A DataLoader whose workers ran only the host part of the pipeline: applies the remaining (device)
    transforms to every raw sample in the main process, then collates."""

    def __init__(self, loader, tail, collate_fn):
        self.loader, self.tail, self.collate_fn = loader, tail, collate_fn
        ds = loader.dataset
        self.sampler, self.batch_size = loader.sampler, loader.batch_size
        self.dataset = ds.dataset if isinstance(ds, HostPartView) else ds

    def __len__(self):
        return len(self.loader)

    def __iter__(self):
        for raw in self.loader:
            samples = [self.tail(r) for r in raw]
            if any(s_ is None for s_ in samples):
                raise RuntimeError('a device transform dropped a sample; only the host part of a pipeline may '
                                   'reject samples (the dataset re-draws there)')
            yield self.collate_fn(samples)


# --------------------------------------------------------------------------- PASCAL VOC
@DATASETS.register_module()
class XMLDataset(CustomDataset):
    """datasets/xml_style.py:12-170: image ids from a list file, one Annotations/<id>.xml each"""

    def __init__(self, min_size=None, img_subdir='JPEGImages', ann_subdir='Annotations', **kwargs):
        assert self.CLASSES or kwargs.get('classes', None), 'CLASSES in `XMLDataset` can not be None.'
        self.img_subdir, self.ann_subdir = img_subdir, ann_subdir
        super().__init__(**kwargs)
        self.cat2label = {cat: i for i, cat in enumerate(self.CLASSES)}
        self.min_size = min_size

    def _root(self, img_id):
        import xml.etree.ElementTree as ET
        return ET.parse(osp.join(self.img_prefix, self.ann_subdir, f'{img_id}.xml')).getroot()

    def load_annotations(self, ann_file):
        data_infos = []
        with open(ann_file) as f:
            img_ids = [ln.rstrip('\n\r') for ln in f]
        for img_id in img_ids:
            filename = osp.join(self.img_subdir, f'{img_id}.jpg')
            size = self._root(img_id).find('size')
            if size is not None:
                width, height = int(size.find('width').text), int(size.find('height').text)
            else:
                from .pipelines import imread
                height, width = imread(osp.join(self.img_prefix, filename)).shape[:2]
            data_infos.append(dict(id=img_id, filename=filename, width=width, height=height))
        return data_infos

    def _filter_imgs(self, min_size=32):
        valid_inds = []
        for i, info in enumerate(self.data_infos):
            if min(info['width'], info['height']) < min_size:
                continue
            if self.filter_empty_gt:
                for obj in self._root(info['id']).findall('object'):
                    if obj.find('name').text in self.CLASSES:
                        valid_inds.append(i)
                        break
            else:
                valid_inds.append(i)
        return valid_inds

    def get_ann_info(self, idx):
        bboxes, labels, bboxes_ignore, labels_ignore = [], [], [], []
        for obj in self._root(self.data_infos[idx]['id']).findall('object'):
            name = obj.find('name').text
            if name not in self.CLASSES:
                continue
            label = self.cat2label[name]
            difficult = obj.find('difficult')
            difficult = 0 if difficult is None else int(difficult.text)
            bb = obj.find('bndbox')
            bbox = [int(float(bb.find(k).text)) for k in ('xmin', 'ymin', 'xmax', 'ymax')]
            ignore = False
            if self.min_size:
                assert not self.test_mode
                if bbox[2] - bbox[0] < self.min_size or bbox[3] - bbox[1] < self.min_size:
                    ignore = True
            if difficult or ignore:
                bboxes_ignore.append(bbox)
                labels_ignore.append(label)
            else:
                bboxes.append(bbox)
                labels.append(label)
        if not bboxes:
            bboxes, labels = np.zeros((0, 4)), np.zeros((0,))
        else:
            bboxes, labels = np.array(bboxes, ndmin=2) - 1, np.array(labels)
        if not bboxes_ignore:
            bboxes_ignore, labels_ignore = np.zeros((0, 4)), np.zeros((0,))
        else:
            bboxes_ignore, labels_ignore = np.array(bboxes_ignore, ndmin=2) - 1, np.array(labels_ignore)
        return dict(bboxes=bboxes.astype(np.float32), labels=labels.astype(np.int64),
                    bboxes_ignore=bboxes_ignore.astype(np.float32), labels_ignore=labels_ignore.astype(np.int64))


@DATASETS.register_module()
class VOCDataset(XMLDataset):
    """datasets/voc.py:11-93"""
    CLASSES = ('aeroplane', 'bicycle', 'bird', 'boat', 'bottle', 'bus', 'car', 'cat', 'chair', 'cow',
               'diningtable', 'dog', 'horse', 'motorbike', 'person', 'pottedplant', 'sheep', 'sofa', 'train',
               'tvmonitor')

    def __init__(self, **kwargs):
        super().__init__(**kwargs)
        if 'VOC2007' in self.img_prefix:
            self.year = 2007
        elif 'VOC2012' in self.img_prefix:
            self.year = 2012
        else:
            raise ValueError('Cannot infer dataset year from img_prefix')

    def evaluate(self, results, metric='mAP', logger=None, proposal_nums=(100, 300, 1000), iou_thr=0.5,
                 scale_ranges=None):
        from collections import OrderedDict
        from .evaluation import eval_map, eval_recalls
        if not isinstance(metric, str):
            assert len(metric) == 1
            metric = metric[0]
        if metric not in ('mAP', 'recall'):
            raise KeyError(f'metric {metric} is not supported')
        annotations = [self.get_ann_info(i) for i in range(len(self))]
        eval_results = OrderedDict()
        iou_thrs = [iou_thr] if isinstance(iou_thr, float) else iou_thr
        if metric == 'recall':      # voc.py:91-106: proposals (k, 4 | 5) per image, legacy +1 extents
            recalls = eval_recalls([ann['bboxes'] for ann in annotations], results, proposal_nums, iou_thrs, logger=logger,
                                   use_legacy_coordinate=True)
            for i, num in enumerate(proposal_nums):
                for j, thr in enumerate(iou_thrs):
                    eval_results[f'recall@{num}@{thr}'] = recalls[i, j]
            if recalls.shape[1] > 1:
                for i, num in enumerate(proposal_nums):
                    eval_results[f'AR@{num}'] = recalls[i].mean()
            return eval_results
        ds_name = 'voc07' if self.year == 2007 else self.CLASSES
        mean_aps = []
        for thr in iou_thrs:
            mean_ap, _ = eval_map(results, annotations, scale_ranges=None, iou_thr=thr, dataset=ds_name,
                                  logger=logger, use_legacy_coordinate=True)
            mean_aps.append(mean_ap)
            eval_results[f'AP{int(thr * 100):02d}'] = round(mean_ap, 3)
        eval_results['mAP'] = sum(mean_aps) / len(mean_aps)
        return eval_results


@DATASETS.register_module()
class RepeatDataset:
    """datasets/dataset_wrappers.py:173-230: the dataset repeated `times` times per epoch"""

    def __init__(self, dataset, times):
        self.dataset = build_dataset(dataset) if isinstance(dataset, dict) else dataset
        self.times = times
        self.CLASSES = self.dataset.CLASSES
        if hasattr(self.dataset, 'flag'):
            self.flag = np.tile(self.dataset.flag, times)
        self._ori_len = len(self.dataset)

    def __getitem__(self, idx):
        return self.dataset[idx % self._ori_len]

    def __len__(self):
        return self.times * self._ori_len


@DATASETS.register_module()
class ConcatDataset(torch.utils.data.ConcatDataset):
    """datasets/dataset_wrappers.py:13-60 (joint evaluation is not needed by the recipes)"""

    def __init__(self, datasets, separate_eval=True):
        super().__init__([build_dataset(d) if isinstance(d, dict) else d for d in datasets])
        self.CLASSES = self.datasets[0].CLASSES
        self.separate_eval = separate_eval
        if hasattr(self.datasets[0], 'flag'):
            self.flag = np.concatenate([d.flag for d in self.datasets])
