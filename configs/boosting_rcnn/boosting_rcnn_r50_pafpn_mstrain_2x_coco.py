# Boosting R-CNN, ResNet-50 + PAFPN on COCO (80 classes) with multi-scale training.
# Child of the UTDAC config: only what differs is stated.
_base_ = 'boosting_rcnn_r50_pafpn_1x_utdac.py'

model = dict(
    rpn_head=dict(gamma=2, loss_bbox=dict(loss_weight=2.0), aug_reg_loss=dict(loss_weight=2.0)),
    roi_head=dict(bbox_head=dict(num_classes=80)),
    train_cfg=dict(rpn=dict(sampler=dict(_delete_=True, type='PseudoSampler'))),
    test_cfg=dict(rcnn=dict(nms=dict(iou_threshold=0.5))))

dataset_type = 'CocoDataset'
data_root = 'data/coco/'
img_norm_cfg = dict(mean=[123.675, 116.28, 103.53], std=[58.395, 57.12, 57.375], to_rgb=True)

_short_sides = [(s, 1333) for s in range(480, 801, 32)]
_plain = [dict(type='Resize', img_scale=_short_sides, multiscale_mode='value', keep_ratio=True)]
_crop = [dict(type='Resize', img_scale=[(400, 4200), (500, 4200), (600, 4200)],
              multiscale_mode='value', keep_ratio=True),
         dict(type='RandomCrop', crop_type='absolute_range', crop_size=(384, 600),
              allow_negative_crop=True),
         dict(type='Resize', img_scale=_short_sides, multiscale_mode='value', override=True,
              keep_ratio=True)]
train_pipeline = [
    dict(type='LoadImageFromFile'),
    dict(type='LoadAnnotations', with_bbox=True),
    dict(type='RandomFlip', flip_ratio=0.5),
    dict(type='AutoAugment', policies=[_plain, _crop]),
    dict(type='Normalize', **img_norm_cfg),
    dict(type='Pad', size_divisor=1),
    dict(type='DefaultFormatBundle'),
    dict(type='Collect', keys=['img', 'gt_bboxes', 'gt_labels']),
]
test_pipeline = [
    dict(type='LoadImageFromFile'),
    dict(type='MultiScaleFlipAug', img_scale=(1333, 800), flip=False,
         transforms=[
             dict(type='Resize', keep_ratio=True),
             dict(type='RandomFlip'),
             dict(type='Normalize', **img_norm_cfg),
             dict(type='Pad', size_divisor=32),
             dict(type='ImageToTensor', keys=['img']),
             dict(type='Collect', keys=['img']),
         ]),
]


def _coco(split, pipeline):
    return dict(type=dataset_type, ann_file=data_root + f'annotations/instances_{split}2017.json',
                img_prefix=data_root + f'{split}2017/', pipeline=pipeline)


data = dict(samples_per_gpu=4, workers_per_gpu=4,
            train=dict(type='RepeatDataset', times=2, dataset=_coco('train', train_pipeline)),
            val=_coco('val', test_pipeline), test=_coco('val', test_pipeline))
evaluation = dict(interval=1, metric='bbox')

optimizer = dict(type='SGD', lr=0.01, momentum=0.9, weight_decay=0.0001)
optimizer_config = dict(grad_clip=None)
# step=[9, 11] measured better than [8, 11] for this recipe
lr_config = dict(policy='step', warmup='linear', warmup_iters=500, warmup_ratio=0.001, step=[9, 11])
runner = dict(type='EpochBasedRunner', max_epochs=12)
del _short_sides, _plain, _crop
