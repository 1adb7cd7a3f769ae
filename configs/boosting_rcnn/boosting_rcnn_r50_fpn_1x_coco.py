# Boosting R-CNN, ResNet-50 + plain FPN on COCO: regression on encoded deltas with a CIoU loss
# (reg_decoded_bbox=False), P6 from C5 ('on_input').  Child of the UTDAC recipe; like the
# reference's file it overrides the pipelines but not the dataset entries of `data`.
_base_ = 'boosting_rcnn_r50_pafpn_1x_utdac.py'

model = dict(
    neck=dict(type='FPN', add_extra_convs='on_input'),
    rpn_head=dict(reg_decoded_bbox=False, gamma=2,
                  loss_bbox=dict(type='CIoULoss', loss_weight=1.0),
                  aug_reg_loss=dict(loss_weight=2.0)),
    roi_head=dict(bbox_head=dict(num_classes=80)),
    train_cfg=dict(rpn=dict(sampler=dict(_delete_=True, type='PseudoSampler'))))

dataset_type = 'CocoDataset'
data_root = 'data/coco/'
img_norm_cfg = dict(mean=[123.675, 116.28, 103.53], std=[58.395, 57.12, 57.375], to_rgb=True)

_short_sides = [(s, 1333) for s in range(480, 801, 32)]
train_pipeline = [
    dict(type='LoadImageFromFile'),
    dict(type='LoadAnnotations', with_bbox=True),
    dict(type='RandomFlip', flip_ratio=0.5),
    dict(type='AutoAugment', policies=[
        [dict(type='Resize', img_scale=_short_sides, multiscale_mode='value', keep_ratio=True)],
        [dict(type='Resize', img_scale=[(400, 4200), (500, 4200), (600, 4200)],
              multiscale_mode='value', keep_ratio=True),
         dict(type='RandomCrop', crop_type='absolute_range', crop_size=(384, 600),
              allow_negative_crop=True),
         dict(type='Resize', img_scale=_short_sides, multiscale_mode='value', override=True,
              keep_ratio=True)]]),
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

optimizer = dict(type='SGD', lr=0.005, momentum=0.9, weight_decay=0.0001)
optimizer_config = dict(_delete_=True, grad_clip=dict(max_norm=35, norm_type=2))
data = dict(samples_per_gpu=6, workers_per_gpu=6)
lr_config = dict(policy='step', warmup='linear', warmup_iters=500, warmup_ratio=0.001, step=[9, 11])
runner = dict(type='EpochBasedRunner', max_epochs=12)
del _short_sides
