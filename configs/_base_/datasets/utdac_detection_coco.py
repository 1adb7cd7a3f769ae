# UTDAC2020 underwater detection set in COCO json format, 1333x800 keep-ratio resize,
# ImageNet mean/std normalisation, pad to a multiple of 32.
dataset_type = 'CocoDataset'
classes = ('echinus', 'starfish', 'holothurian', 'scallop')
data_root = 'data/UTDAC2020/'
img_norm_cfg = dict(mean=[123.675, 116.28, 103.53], std=[58.395, 57.12, 57.375], to_rgb=True)

_resize_scale = (1333, 800)
train_pipeline = [
    dict(type='LoadImageFromFile'),
    dict(type='LoadAnnotations', with_bbox=True),
    dict(type='Resize', img_scale=_resize_scale, keep_ratio=True),
    dict(type='RandomFlip', flip_ratio=0.5),
    dict(type='Normalize', **img_norm_cfg),
    dict(type='Pad', size_divisor=32),
    dict(type='DefaultFormatBundle'),
    dict(type='Collect', keys=['img', 'gt_bboxes', 'gt_labels']),
]
test_pipeline = [
    dict(type='LoadImageFromFile'),
    dict(type='MultiScaleFlipAug', img_scale=_resize_scale, flip=False,
         transforms=[
             dict(type='Resize', keep_ratio=True),
             dict(type='RandomFlip'),
             dict(type='Normalize', **img_norm_cfg),
             dict(type='Pad', size_divisor=32),
             dict(type='ImageToTensor', keys=['img']),
             dict(type='Collect', keys=['img']),
         ]),
]


def _split(split, pipeline):
    return dict(type=dataset_type,
                ann_file=f'data/UTDAC2020/annotations/instances_{split}2017.json',
                img_prefix=data_root + f'{split}2017/',
                pipeline=pipeline)


data = dict(samples_per_gpu=2, workers_per_gpu=2,
            train=_split('train', train_pipeline),
            val=_split('val', test_pipeline),
            test=_split('val', test_pipeline))
evaluation = dict(interval=1, metric='bbox')
del _resize_scale
