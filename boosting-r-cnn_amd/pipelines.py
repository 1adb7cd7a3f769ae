"""Input pipeline to the path's front door.

Host-side mirror of the reference's data transforms for the Boosting R-CNN configs
(mmdet/datasets/pipelines/{loading,transforms,formating,test_time_aug,compose}.py), same
names / arguments / result-dict keys, so `train_pipeline` / `test_pipeline` of
configs/_base_/datasets/utdac_detection_coco.py build unchanged:

    LoadImageFromFile -> LoadAnnotations -> Resize(1333,800,keep_ratio) -> RandomFlip
    -> Normalize(mean,std,to_rgb) -> Pad(size_divisor=32) -> DefaultFormatBundle -> Collect
    LoadImageFromFile -> MultiScaleFlipAug[Resize, RandomFlip, Normalize, Pad, ImageToTensor, Collect]

The image arithmetic the reference delegates to mmcv -> OpenCV (absent here, like mmcv) is
restated: `imresize_u8` is OpenCV's 8-bit INTER_LINEAR (11-bit fixed-point coefficients,
`cv::resize` resizeGeneric_ / HResizeLinear / VResizeLinear<uchar,int,short>), `imnormalize`
is mmcv.imnormalize (fp32 subtract, multiply by fp32(1/std)).  PARITY UNPINNED for the resize:
neither cv2 nor a golden vector of it exists in this container; the restatement follows the
published algorithm and is cross-checked by its invariants (tests/test_data_cpu.py); the
transform LOGIC around it is pinned by golden fixtures from the imported reference (g12, g16).

`FusedResizeNormalizePad` is the device form of Resize+RandomFlip+Normalize+Pad for one
decoded uint8 image (HIP kernel `brcnn_preprocess_u8`, csrc/preprocess.hip), bit-identical to
the host chain above; `fuse_device_pipeline` rewrites a pipeline config to use it.
"""
import collections
import os.path as osp

import numpy as np
import torch

from .registry import Registry, build_from_cfg

PIPELINES = Registry('pipeline')

INTER_BITS = 11
INTER_SCALE = 1 << INTER_BITS


# --------------------------------------------------------------------------- image arithmetic
def rescale_size(old_size, scale):
    """mmcv.image.geometric.rescale_size: (w, h), scale = factor or (long, short) bound"""
    w, h = old_size
    if isinstance(scale, (float, int)):
        if scale <= 0:
            raise ValueError(f'Invalid scale {scale}, must be positive.')
        scale_factor = scale
    elif isinstance(scale, tuple):
        max_long_edge, max_short_edge = max(scale), min(scale)
        scale_factor = min(max_long_edge / max(h, w), max_short_edge / min(h, w))
    else:
        raise TypeError(f'Scale must be a number or tuple of int, but got {type(scale)}')
    return int(w * float(scale_factor) + 0.5), int(h * float(scale_factor) + 0.5)


def linear_coeffs(src, dst):
    """OpenCV INTER_LINEAR sample positions and 11-bit coefficients of one axis:
    returns (ofs int32[dst], c0 int32[dst], c1 int32[dst], first index that reads only `ofs`)."""
    scale = 1.0 / (float(dst) / float(src))         # cv::resize: scale_x = 1. / inv_scale_x
    d = np.arange(dst, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int32)
    f = (f - s.astype(np.float32)).astype(np.float32)
    neg = s < 0
    f[neg] = 0.0
    s[neg] = 0
    edge = s >= src - 1
    f[edge] = 0.0
    s[edge] = src - 1
    c0 = np.rint((np.float32(1.0) - f) * np.float32(INTER_SCALE)).astype(np.int32)
    c1 = np.rint(f * np.float32(INTER_SCALE)).astype(np.int32)
    return s, c0, c1


def imresize_u8(img, size):
    """cv2.resize(img, (w, h), interpolation=cv2.INTER_LINEAR) for uint8 HxWxC / HxW images"""
    assert img.dtype == np.uint8
    w, h = int(size[0]), int(size[1])
    sh, sw = img.shape[:2]
    if (sw, sh) == (w, h):
        return img.copy()
    squeeze = img.ndim == 2
    src = img[:, :, None] if squeeze else img
    xo, a0, a1 = linear_coeffs(sw, w)
    yo, b0, b1 = linear_coeffs(sh, h)
    x1 = np.minimum(xo + 1, sw - 1)      # a1 == 0 where the right tap would fall outside
    rows = src.astype(np.int32)
    hor = rows[:, xo, :] * a0[None, :, None] + rows[:, x1, :] * a1[None, :, None]    # (sh, w, C)
    y0 = np.clip(yo, 0, sh - 1)
    y1 = np.clip(yo + 1, 0, sh - 1)
    s0 = hor[y0] >> 4
    s1 = hor[y1] >> 4
    out = (((b0[:, None, None] * s0) >> 16) + ((b1[:, None, None] * s1) >> 16) + 2) >> 2
    out = np.clip(out, 0, 255).astype(np.uint8)
    return out[:, :, 0] if squeeze else out


def imresize(img, size, return_scale=False):
    h, w = img.shape[:2]
    if img.dtype == np.uint8:
        out = imresize_u8(img, size)
    else:   # float images: plain bilinear with the same sample positions
        xo, a0, a1 = linear_coeffs(w, size[0])
        yo, b0, b1 = linear_coeffs(h, size[1])
        src = img.astype(np.float32)
        if src.ndim == 2:
            src = src[:, :, None]
        x1 = np.minimum(xo + 1, w - 1)
        fa = (a1 / INTER_SCALE).astype(np.float32)[None, :, None]
        hor = src[:, xo] * (1 - fa) + src[:, x1] * fa
        fb = (b1 / INTER_SCALE).astype(np.float32)[:, None, None]
        out = hor[np.clip(yo, 0, h - 1)] * (1 - fb) + hor[np.clip(yo + 1, 0, h - 1)] * fb
        out = out.reshape((size[1], size[0]) + img.shape[2:]).astype(img.dtype)
    if not return_scale:
        return out
    return out, size[0] / w, size[1] / h


def imrescale(img, scale, return_scale=False):
    h, w = img.shape[:2]
    new_size = rescale_size((w, h), scale)
    out = imresize(img, new_size)
    if return_scale:
        # mmcv returns the single factor it sampled; Resize recomputes w/h factors from shapes
        return out, min(max(scale) / max(h, w), min(scale) / min(h, w)) if isinstance(scale, tuple) else scale
    return out


def imflip(img, direction='horizontal'):
    assert direction in ('horizontal', 'vertical', 'diagonal')
    if direction == 'horizontal':
        return np.flip(img, axis=1)
    if direction == 'vertical':
        return np.flip(img, axis=0)
    return np.flip(img, axis=(0, 1))


def imnormalize(img, mean, std, to_rgb=True):
    """mmcv.imnormalize: float32 copy, BGR->RGB, subtract mean, multiply by 1/std (fp32)"""
    img = img.astype(np.float32)
    mean = np.asarray(mean, dtype=np.float64).reshape(1, -1)
    stdinv = 1.0 / np.asarray(std, dtype=np.float64).reshape(1, -1)
    if to_rgb:
        img = img[..., ::-1]
    img = img - mean.astype(np.float32)
    return (img * stdinv.astype(np.float32)).astype(np.float32)


def impad(img, shape=None, pad_val=0):
    h, w = shape
    out = np.full((h, w) + img.shape[2:], pad_val, dtype=img.dtype)
    out[:img.shape[0], :img.shape[1]] = img
    return out


def impad_to_multiple(img, divisor, pad_val=0):
    pad_h = int(np.ceil(img.shape[0] / divisor)) * divisor
    pad_w = int(np.ceil(img.shape[1] / divisor)) * divisor
    return impad(img, (pad_h, pad_w), pad_val)


def imread(path):
    """decoded BGR uint8 HxWx3 (cv2.imread order, as mmcv.imfrombytes(flag='color')).  `.npy`
    arrays (already BGR) are accepted so synthetic datasets need no image codec."""
    if path.endswith('.npy'):
        arr = np.load(path)
        assert arr.dtype == np.uint8 and arr.ndim == 3
        return arr
    from PIL import Image
    with Image.open(path) as im:
        rgb = np.asarray(im.convert('RGB'))
    return np.ascontiguousarray(rgb[:, :, ::-1])


# --------------------------------------------------------------------------- transforms
class Compose:
    """pipelines/compose.py: sequentially apply transforms; a None result aborts"""

    def __init__(self, transforms):
        assert isinstance(transforms, collections.abc.Sequence)
        self.transforms = []
        for t in transforms:
            if isinstance(t, dict):
                self.transforms.append(build_from_cfg(t, PIPELINES))
            elif callable(t):
                self.transforms.append(t)
            else:
                raise TypeError('transform must be callable or a dict')

    def __call__(self, data):
        for t in self.transforms:
            data = t(data)
            if data is None:
                return None
        return data

    def __repr__(self):
        return self.__class__.__name__ + '(' + ', '.join(repr(t) for t in self.transforms) + ')'


@PIPELINES.register_module()
class LoadImageFromFile:
    """loading.py:13-87"""

    def __init__(self, to_float32=False, color_type='color', file_client_args=None):
        assert color_type == 'color'
        self.to_float32 = to_float32

    def __call__(self, results):
        if results['img_prefix'] is not None:
            filename = osp.join(results['img_prefix'], results['img_info']['filename'])
        else:
            filename = results['img_info']['filename']
        img = imread(filename)
        if self.to_float32:
            img = img.astype(np.float32)
        results['filename'] = filename
        results['ori_filename'] = results['img_info']['filename']
        results['img'] = img
        results['img_shape'] = img.shape
        results['ori_shape'] = img.shape
        results['img_fields'] = ['img']
        return results

    def __repr__(self):
        return f'{self.__class__.__name__}(to_float32={self.to_float32})'


@PIPELINES.register_module()
class LoadAnnotations:
    """loading.py:204-390, bbox / label part (masks and seg maps are outside the path)"""

    def __init__(self, with_bbox=True, with_label=True, with_mask=False, with_seg=False, poly2mask=True,
                 file_client_args=None):
        assert not with_mask and not with_seg, 'mask / seg annotations are outside the Boosting R-CNN path'
        self.with_bbox = with_bbox
        self.with_label = with_label

    def __call__(self, results):
        ann = results['ann_info']
        if self.with_bbox:
            results['gt_bboxes'] = ann['bboxes'].copy()
            ignore = ann.get('bboxes_ignore', None)
            if ignore is not None:
                results['gt_bboxes_ignore'] = ignore.copy()
                results['bbox_fields'].append('gt_bboxes_ignore')
            results['bbox_fields'].append('gt_bboxes')
        if self.with_label:
            results['gt_labels'] = ann['labels'].copy()
        return results

    def __repr__(self):
        return f'{self.__class__.__name__}(with_bbox={self.with_bbox}, with_label={self.with_label})'


@PIPELINES.register_module()
class Resize:
    """transforms.py:31-315 (image + boxes)"""

    def __init__(self, img_scale=None, multiscale_mode='range', ratio_range=None, keep_ratio=True,
                 bbox_clip_border=True, backend='cv2', override=False):
        if img_scale is None:
            self.img_scale = None
        else:
            self.img_scale = img_scale if isinstance(img_scale, list) else [img_scale]
            self.img_scale = [tuple(s) for s in self.img_scale]
        if ratio_range is not None:
            assert len(self.img_scale) == 1
        else:
            assert multiscale_mode in ['value', 'range']
        assert backend == 'cv2'
        self.multiscale_mode = multiscale_mode
        self.ratio_range = ratio_range
        self.keep_ratio = keep_ratio
        self.override = override
        self.bbox_clip_border = bbox_clip_border

    @staticmethod
    def random_select(img_scales):
        idx = np.random.randint(len(img_scales))
        return img_scales[idx], idx

    @staticmethod
    def random_sample(img_scales):
        assert len(img_scales) == 2
        longs = [max(s) for s in img_scales]
        shorts = [min(s) for s in img_scales]
        long_edge = np.random.randint(min(longs), max(longs) + 1)
        short_edge = np.random.randint(min(shorts), max(shorts) + 1)
        return (long_edge, short_edge), None

    @staticmethod
    def random_sample_ratio(img_scale, ratio_range):
        lo, hi = ratio_range
        assert lo <= hi
        ratio = np.random.random_sample() * (hi - lo) + lo
        return (int(img_scale[0] * ratio), int(img_scale[1] * ratio)), None

    def _random_scale(self, results):
        if self.ratio_range is not None:
            scale, idx = self.random_sample_ratio(self.img_scale[0], self.ratio_range)
        elif len(self.img_scale) == 1:
            scale, idx = self.img_scale[0], 0
        elif self.multiscale_mode == 'range':
            scale, idx = self.random_sample(self.img_scale)
        else:
            scale, idx = self.random_select(self.img_scale)
        results['scale'] = scale
        results['scale_idx'] = idx

    def target_size(self, results):
        """(new_w, new_h) this transform gives the image"""
        h, w = results['img'].shape[:2]
        if self.keep_ratio:
            return rescale_size((w, h), results['scale'])
        return int(results['scale'][0]), int(results['scale'][1])

    def _resize_img(self, results):
        for key in results.get('img_fields', ['img']):
            h, w = results[key].shape[:2]
            new_w, new_h = self.target_size(results)
            img = imresize(results[key], (new_w, new_h))
            results[key] = img
            w_scale, h_scale = new_w / w, new_h / h
            results['img_shape'] = img.shape
            results['pad_shape'] = img.shape
            results['scale_factor'] = np.array([w_scale, h_scale, w_scale, h_scale], dtype=np.float32)
            results['keep_ratio'] = self.keep_ratio

    def _resize_bboxes(self, results):
        for key in results.get('bbox_fields', []):
            bboxes = results[key] * results['scale_factor']
            if self.bbox_clip_border:
                img_shape = results['img_shape']
                bboxes[:, 0::2] = np.clip(bboxes[:, 0::2], 0, img_shape[1])
                bboxes[:, 1::2] = np.clip(bboxes[:, 1::2], 0, img_shape[0])
            results[key] = bboxes

    def _pick_scale(self, results):
        if 'scale' not in results:
            if 'scale_factor' in results:
                img_shape = results['img'].shape[:2]
                scale_factor = results['scale_factor']
                assert isinstance(scale_factor, float)
                results['scale'] = tuple([int(x * scale_factor) for x in img_shape][::-1])
            else:
                self._random_scale(results)
        else:
            if not self.override:
                assert 'scale_factor' not in results, 'scale and scale_factor cannot be both set.'
            else:
                results.pop('scale')
                results.pop('scale_factor', None)
                self._random_scale(results)

    def __call__(self, results):
        self._pick_scale(results)
        self._resize_img(results)
        self._resize_bboxes(results)
        return results

    def __repr__(self):
        return (f'{self.__class__.__name__}(img_scale={self.img_scale}, multiscale_mode={self.multiscale_mode}, '
                f'ratio_range={self.ratio_range}, keep_ratio={self.keep_ratio}, '
                f'bbox_clip_border={self.bbox_clip_border})')


@PIPELINES.register_module()
class RandomFlip:
    """transforms.py:318-470"""
    VALID = ['horizontal', 'vertical', 'diagonal']

    def __init__(self, flip_ratio=None, direction='horizontal'):
        if isinstance(flip_ratio, list):
            assert all(isinstance(r, float) for r in flip_ratio) and 0 <= sum(flip_ratio) <= 1
        elif isinstance(flip_ratio, float):
            assert 0 <= flip_ratio <= 1
        elif flip_ratio is not None:
            raise ValueError('flip_ratios must be None, float, or list of float')
        self.flip_ratio = flip_ratio
        if isinstance(direction, str):
            assert direction in self.VALID
        elif isinstance(direction, list):
            assert set(direction).issubset(set(self.VALID))
        else:
            raise ValueError('direction must be either str or list of str')
        self.direction = direction
        if isinstance(flip_ratio, list):
            assert len(self.flip_ratio) == len(self.direction)

    @staticmethod
    def bbox_flip(bboxes, img_shape, direction):
        assert bboxes.shape[-1] % 4 == 0
        flipped = bboxes.copy()
        h, w = img_shape[0], img_shape[1]
        if direction in ('horizontal', 'diagonal'):
            flipped[..., 0::4] = w - bboxes[..., 2::4]
            flipped[..., 2::4] = w - bboxes[..., 0::4]
        if direction in ('vertical', 'diagonal'):
            flipped[..., 1::4] = h - bboxes[..., 3::4]
            flipped[..., 3::4] = h - bboxes[..., 1::4]
        if direction not in RandomFlip.VALID:
            raise ValueError(f"Invalid flipping direction '{direction}'")
        return flipped

    def decide(self, results):
        """sets results['flip'] / ['flip_direction'] exactly as the reference draws them"""
        cur_dir = None
        if 'flip' not in results:
            direction_list = (self.direction if isinstance(self.direction, list) else [self.direction]) + [None]
            if isinstance(self.flip_ratio, list):
                ratios = self.flip_ratio + [1 - sum(self.flip_ratio)]
            else:
                single = self.flip_ratio / (len(direction_list) - 1)
                ratios = [single] * (len(direction_list) - 1) + [1 - self.flip_ratio]
            cur_dir = np.random.choice(direction_list, p=ratios)
            results['flip'] = cur_dir is not None
        if 'flip_direction' not in results:
            results['flip_direction'] = cur_dir

    def __call__(self, results):
        self.decide(results)
        if results['flip']:
            for key in results.get('img_fields', ['img']):
                results[key] = imflip(results[key], direction=results['flip_direction'])
            for key in results.get('bbox_fields', []):
                results[key] = self.bbox_flip(results[key], results['img_shape'], results['flip_direction'])
        return results

    def __repr__(self):
        return f'{self.__class__.__name__}(flip_ratio={self.flip_ratio})'


@PIPELINES.register_module()
class Normalize:
    """transforms.py:700-739"""

    def __init__(self, mean, std, to_rgb=True):
        self.mean = np.array(mean, dtype=np.float32)
        self.std = np.array(std, dtype=np.float32)
        self.to_rgb = to_rgb

    def __call__(self, results):
        for key in results.get('img_fields', ['img']):
            results[key] = imnormalize(results[key], self.mean, self.std, self.to_rgb)
        results['img_norm_cfg'] = dict(mean=self.mean, std=self.std, to_rgb=self.to_rgb)
        return results

    def __repr__(self):
        return f'{self.__class__.__name__}(mean={self.mean}, std={self.std}, to_rgb={self.to_rgb})'


@PIPELINES.register_module()
class Pad:
    """transforms.py:625-697"""

    def __init__(self, size=None, size_divisor=None, pad_val=0):
        self.size = size
        self.size_divisor = size_divisor
        self.pad_val = pad_val
        assert size is not None or size_divisor is not None
        assert size is None or size_divisor is None

    def padded_shape(self, h, w):
        if self.size is not None:
            return tuple(self.size)
        d = self.size_divisor
        return int(np.ceil(h / d)) * d, int(np.ceil(w / d)) * d

    def __call__(self, results):
        for key in results.get('img_fields', ['img']):
            img = results[key]
            results[key] = impad(img, self.padded_shape(img.shape[0], img.shape[1]), self.pad_val)
        results['pad_shape'] = results['img'].shape
        results['pad_fixed_size'] = self.size
        results['pad_size_divisor'] = self.size_divisor
        return results

    def __repr__(self):
        return (f'{self.__class__.__name__}(size={self.size}, size_divisor={self.size_divisor}, '
                f'pad_val={self.pad_val})')


@PIPELINES.register_module()
class RandomCrop:
    """transforms.py:742-896 (image + boxes + labels; masks / seg maps are outside the path)"""

    def __init__(self, crop_size, crop_type='absolute', allow_negative_crop=False, recompute_bbox=False,
                 bbox_clip_border=True):
        if crop_type not in ['relative_range', 'relative', 'absolute', 'absolute_range']:
            raise ValueError(f'Invalid crop_type {crop_type}.')
        if crop_type in ['absolute', 'absolute_range']:
            assert crop_size[0] > 0 and crop_size[1] > 0
            assert isinstance(crop_size[0], int) and isinstance(crop_size[1], int)
        else:
            assert 0 < crop_size[0] <= 1 and 0 < crop_size[1] <= 1
        self.crop_size, self.crop_type = tuple(crop_size), crop_type
        self.allow_negative_crop, self.bbox_clip_border = allow_negative_crop, bbox_clip_border
        self.recompute_bbox = recompute_bbox
        self.bbox2label = {'gt_bboxes': 'gt_labels', 'gt_bboxes_ignore': 'gt_labels_ignore'}

    def _crop_data(self, results, crop_size, allow_negative_crop):
        """transforms.py:720-781.  One window per image field (two draws each, rows first: the reference's RNG order); the
        boxes move with the LAST window drawn, are clipped to it, and empty ones leave together with their labels."""
        ch, cw = crop_size
        assert ch > 0 and cw > 0
        top = left = 0
        shape = None
        for field in results.get('img_fields', ['img']):
            src = results[field]
            top = np.random.randint(0, max(src.shape[0] - ch, 0) + 1)
            left = np.random.randint(0, max(src.shape[1] - cw, 0) + 1)
            window = src[top:top + ch, left:left + cw, ...]
            results[field], shape = window, window.shape
        results['img_shape'] = shape
        shift = np.array([left, top, left, top], dtype=np.float32)
        for field in results.get('bbox_fields', []):
            moved = results[field] - shift
            if self.bbox_clip_border:
                moved[:, 0::2] = moved[:, 0::2].clip(0, shape[1])
                moved[:, 1::2] = moved[:, 1::2].clip(0, shape[0])
            keep = (moved[:, 2] > moved[:, 0]) & (moved[:, 3] > moved[:, 1])
            if field == 'gt_bboxes' and not (allow_negative_crop or keep.any()):
                return None                      # every ground truth fell outside: the caller draws again
            results[field] = moved[keep]
            labels = self.bbox2label.get(field)
            if labels in results:
                results[labels] = results[labels][keep]
        return results

    def _get_crop_size(self, image_size):
        h, w = image_size
        if self.crop_type == 'absolute':
            return min(self.crop_size[0], h), min(self.crop_size[1], w)
        if self.crop_type == 'absolute_range':
            assert self.crop_size[0] <= self.crop_size[1]
            crop_h = np.random.randint(min(h, self.crop_size[0]), min(h, self.crop_size[1]) + 1)
            crop_w = np.random.randint(min(w, self.crop_size[0]), min(w, self.crop_size[1]) + 1)
            return crop_h, crop_w
        if self.crop_type == 'relative':
            crop_h, crop_w = self.crop_size
            return int(h * crop_h + 0.5), int(w * crop_w + 0.5)
        crop_size = np.asarray(self.crop_size, dtype=np.float32)
        crop_h, crop_w = crop_size + np.random.rand(2) * (1 - crop_size)
        return int(h * crop_h + 0.5), int(w * crop_w + 0.5)

    def __call__(self, results):
        crop_size = self._get_crop_size(results['img'].shape[:2])
        return self._crop_data(results, crop_size, self.allow_negative_crop)

    def __repr__(self):
        return (f'{self.__class__.__name__}(crop_size={self.crop_size}, crop_type={self.crop_type}, '
                f'allow_negative_crop={self.allow_negative_crop}, bbox_clip_border={self.bbox_clip_border})')


@PIPELINES.register_module()
class AutoAugment:
    """auto_augment.py:44-109: one of several transform sequences, drawn per sample"""

    def __init__(self, policies):
        import copy
        assert isinstance(policies, list) and len(policies) > 0, 'Policies must be a non-empty list.'
        for policy in policies:
            assert isinstance(policy, list) and len(policy) > 0, 'Each policy in policies must be a non-empty list.'
            for augment in policy:
                assert isinstance(augment, dict) and 'type' in augment
        self.policies = copy.deepcopy(policies)
        self.transforms = [Compose(policy) for policy in self.policies]

    def __call__(self, results):
        transform = np.random.choice(self.transforms)
        return transform(results)

    def __repr__(self):
        return f'{self.__class__.__name__}(policies={self.policies})'


class DataContainer:
    """mmcv.parallel.DataContainer surface used by the formatting transforms / collate:
    `stack` tensors are padded to a common shape and stacked, `cpu_only` data stay python
    objects, everything else becomes a per-sample list."""

    def __init__(self, data, stack=False, padding_value=0, cpu_only=False, pad_dims=2):
        self._data = data
        self.stack = stack
        self.padding_value = padding_value
        self.cpu_only = cpu_only
        self.pad_dims = pad_dims

    @property
    def data(self):
        return self._data

    def __repr__(self):
        return f'DataContainer({self._data!r})'


DC = DataContainer


def to_tensor(data):
    if isinstance(data, torch.Tensor):
        return data
    if isinstance(data, np.ndarray):
        return torch.from_numpy(data)
    if isinstance(data, collections.abc.Sequence) and not isinstance(data, str):
        return torch.tensor(data)
    if isinstance(data, int):
        return torch.LongTensor([data])
    if isinstance(data, float):
        return torch.FloatTensor([data])
    raise TypeError(f'type {type(data)} cannot be converted to tensor.')


@PIPELINES.register_module()
class ImageToTensor:
    """formating.py:59-93"""

    def __init__(self, keys):
        self.keys = keys

    def __call__(self, results):
        for key in self.keys:
            img = results[key]
            if isinstance(img, torch.Tensor):      # already formatted on the device
                continue
            if img.ndim < 3:
                img = np.expand_dims(img, -1)
            results[key] = to_tensor(np.ascontiguousarray(img.transpose(2, 0, 1)))
        return results

    def __repr__(self):
        return f'{self.__class__.__name__}(keys={self.keys})'


@PIPELINES.register_module()
class DefaultFormatBundle:
    """formating.py:172-252 (img, proposals, gt_bboxes, gt_bboxes_ignore, gt_labels)"""

    def __call__(self, results):
        if 'img' in results:
            img = results['img']
            results.setdefault('pad_shape', img.shape)
            results.setdefault('scale_factor', 1.0)
            num_channels = 1 if len(img.shape) < 3 else img.shape[2]
            results.setdefault('img_norm_cfg', dict(mean=np.zeros(num_channels, dtype=np.float32),
                                                    std=np.ones(num_channels, dtype=np.float32), to_rgb=False))
            if len(img.shape) < 3:
                img = np.expand_dims(img, -1)
            img = np.ascontiguousarray(img.transpose(2, 0, 1))
            results['img'] = DC(to_tensor(img), stack=True)
        for key in ['proposals', 'gt_bboxes', 'gt_bboxes_ignore', 'gt_labels']:
            if key in results:
                results[key] = DC(to_tensor(results[key]))
        return results

    def __repr__(self):
        return self.__class__.__name__


@PIPELINES.register_module()
class Collect:
    """formating.py:255-318"""

    def __init__(self, keys, meta_keys=('filename', 'ori_filename', 'ori_shape', 'img_shape', 'pad_shape',
                                        'scale_factor', 'flip', 'flip_direction', 'img_norm_cfg')):
        self.keys = keys
        self.meta_keys = meta_keys

    def __call__(self, results):
        data = {}
        data['img_metas'] = DC({k: results[k] for k in self.meta_keys}, cpu_only=True)
        for key in self.keys:
            data[key] = results[key]
        return data

    def __repr__(self):
        return f'{self.__class__.__name__}(keys={self.keys}, meta_keys={self.meta_keys})'


@PIPELINES.register_module()
class MultiScaleFlipAug:
    """test_time_aug.py:11-121"""

    def __init__(self, transforms, img_scale=None, scale_factor=None, flip=False, flip_direction='horizontal'):
        self.transforms = Compose(transforms)
        assert (img_scale is None) ^ (scale_factor is None), 'Must have but only one variable can be setted'
        if img_scale is not None:
            self.img_scale = img_scale if isinstance(img_scale, list) else [img_scale]
            self.img_scale = [tuple(s) for s in self.img_scale]
            self.scale_key = 'scale'
        else:
            self.img_scale = scale_factor if isinstance(scale_factor, list) else [scale_factor]
            self.scale_key = 'scale_factor'
        self.flip = flip
        self.flip_direction = flip_direction if isinstance(flip_direction, list) else [flip_direction]

    def __call__(self, results):
        aug_data = []
        flip_args = [(False, None)]
        if self.flip:
            flip_args += [(True, d) for d in self.flip_direction]
        for scale in self.img_scale:
            for flip, direction in flip_args:
                _results = results.copy()
                _results[self.scale_key] = scale
                _results['flip'] = flip
                _results['flip_direction'] = direction
                aug_data.append(self.transforms(_results))
        out = {key: [] for key in aug_data[0]}
        for data in aug_data:
            for key, val in data.items():
                out[key].append(val)
        return out

    def __repr__(self):
        return (f'{self.__class__.__name__}(transforms={self.transforms}, img_scale={self.img_scale}, '
                f'flip={self.flip}, flip_direction={self.flip_direction})')


# --------------------------------------------------------------------------- device form
def first_device_transform(compose):
    """index of the first transform of a Compose that touches the GPU (itself, or through nested
    transforms such as MultiScaleFlipAug), or None.  DataLoader workers are forked after the parent has
    initialised the device and must never run those: `datasets.build_dataloader` keeps them in the main
    process."""
    def on_device(t):
        if getattr(t, 'runs_on_device', False):
            return True
        inner = getattr(t, 'transforms', None)
        inner = getattr(inner, 'transforms', inner)
        return bool(inner) and any(on_device(u) for u in inner)
    for i, t in enumerate(compose.transforms):
        if on_device(t):
            return i
    return None


@PIPELINES.register_module()
class FusedResizeNormalizePad:
    """Resize + RandomFlip + Normalize + Pad of one decoded uint8 BGR image as ONE HIP kernel
    (`brcnn_preprocess_u8`): the uint8 image is uploaded once (3 B/pixel instead of the 12 B/pixel
    fp32 tensor the host chain ships) and the normalised, padded fp32 CHW tensor is produced
    on the device, bit-identical to Resize -> RandomFlip -> Normalize -> Pad above.  Takes
    the union of those transforms' arguments; boxes are transformed on the host as before."""

    def __init__(self, mean, std, to_rgb=True, img_scale=None, multiscale_mode='range', ratio_range=None,
                 keep_ratio=True, bbox_clip_border=True, flip_ratio=None, direction='horizontal',
                 size=None, size_divisor=None, pad_val=0, device='cuda'):
        assert pad_val == 0
        self.runs_on_device = True
        self.resize = Resize(img_scale, multiscale_mode, ratio_range, keep_ratio, bbox_clip_border)
        self.flip = RandomFlip(flip_ratio, direction)
        self.pad = Pad(size, size_divisor, pad_val)
        self.mean = np.array(mean, dtype=np.float32)
        self.std = np.array(std, dtype=np.float32)
        self.to_rgb = to_rgb
        self.device = device

    def __call__(self, results):
        from . import ops
        img = results['img']
        assert img.dtype == np.uint8 and img.ndim == 3 and img.shape[2] == 3
        h, w = img.shape[:2]
        self.resize._pick_scale(results)
        new_w, new_h = self.resize.target_size(results)
        results['img_shape'] = (new_h, new_w, 3)
        results['scale_factor'] = np.array([new_w / w, new_h / h, new_w / w, new_h / h], dtype=np.float32)
        results['keep_ratio'] = self.resize.keep_ratio
        self.resize._resize_bboxes(results)
        self.flip.decide(results)
        if results['flip']:
            for key in results.get('bbox_fields', []):
                results[key] = self.flip.bbox_flip(results[key], results['img_shape'], results['flip_direction'])
        ph, pw = self.pad.padded_shape(new_h, new_w)
        src = torch.from_numpy(np.ascontiguousarray(img)).to(self.device, non_blocking=True)
        out = torch.empty((3, ph, pw), dtype=torch.float32, device=self.device)
        ops.preprocess_u8(src, out, new_w, new_h, results['flip_direction'] if results['flip'] else None,
                          self.mean, self.std, self.to_rgb)
        results['img'] = out
        results['img_norm_cfg'] = dict(mean=self.mean, std=self.std, to_rgb=self.to_rgb)
        results['pad_shape'] = (ph, pw, 3)
        results['pad_fixed_size'] = self.pad.size
        results['pad_size_divisor'] = self.pad.size_divisor
        return results

    def __repr__(self):
        return f'{self.__class__.__name__}(resize={self.resize}, flip={self.flip}, pad={self.pad})'


@PIPELINES.register_module()
class DeviceFormatBundle(DefaultFormatBundle):
    """DefaultFormatBundle for an image that FusedResizeNormalizePad already left on the device
    as a (3,H,W) tensor"""
    runs_on_device = True

    def __call__(self, results):
        img = results.pop('img')
        results = super().__call__(results)
        results['img'] = DC(img, stack=True)
        return results


def fuse_device_pipeline(pipeline_cfg, device='cuda'):
    """Rewrite a reference pipeline config so that its [Resize, RandomFlip, Normalize, Pad] run
    is replaced by one FusedResizeNormalizePad (and DefaultFormatBundle by DeviceFormatBundle).
    Anything else is kept as is; pipelines without that exact run are returned unchanged."""
    import copy
    out = []
    cfgs = [dict(c) for c in copy.deepcopy(list(pipeline_cfg))]
    i = 0
    while i < len(cfgs):
        c = cfgs[i]
        if c['type'] == 'MultiScaleFlipAug':
            c = dict(c)
            c['transforms'] = fuse_device_pipeline(c['transforms'], device)
            out.append(c)
            i += 1
            continue
        types = [x['type'] for x in cfgs[i:i + 4]]
        if types == ['Resize', 'RandomFlip', 'Normalize', 'Pad']:
            r, f, n, p = cfgs[i:i + 4]
            fused = dict(type='FusedResizeNormalizePad', device=device)
            for src in (r, f, n, p):
                fused.update({k: v for k, v in src.items() if k != 'type'})
            out.append(fused)
            i += 4
            continue
        if c['type'] == 'DefaultFormatBundle':
            c = dict(type='DeviceFormatBundle')
        out.append(c)
        i += 1
    return out
