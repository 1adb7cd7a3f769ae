# Boosting R-CNN, ResNet-50 + PAFPN, 1x schedule, UTDAC2020 (4 classes).
# RetinaRPN (ATSSRPNHead: 4-conv GN tower, objectness x IoU scoring) -> ProbRoIHead with the
# boosting re-weighted classification loss and prior-fused test-time scores.
_base_ = ['../_base_/datasets/utdac_detection_coco.py', '../_base_/default_runtime.py',
          '../_base_/schedules/schedule_1x.py']

_strides = [8, 16, 32, 64, 128]
_nms07 = dict(type='nms', iou_threshold=0.7)

_rpn_head = dict(
    type='ATSSRPNHead', in_channels=256, feat_channels=256, stacked_convs=4,
    reg_decoded_bbox=True, gamma=0.5, atss=False,
    anchor_generator=dict(type='AnchorGenerator', octave_base_scale=4, scales_per_octave=3,
                          ratios=[0.5, 1.0, 2.0], strides=_strides),
    bbox_coder=dict(type='DeltaXYWHBBoxCoder', target_means=[.0, .0, .0, .0],
                    target_stds=[1.0, 1.0, 1.0, 1.0]),
    loss_cls=dict(type='FocalLoss', use_sigmoid=True, gamma=2.0, alpha=0.25, loss_weight=1.0),
    loss_centerness=dict(type='CrossEntropyLoss', use_sigmoid=True, loss_weight=1.0),
    loss_bbox=dict(type='IoULoss', loss_weight=1.0),
    aug_reg_loss=dict(type='MSELoss', loss_weight=1.0))

_roi_head = dict(
    type='ProbRoIHead', boost=True, gamma=0.5,
    bbox_roi_extractor=dict(type='SingleRoIExtractor',
                            roi_layer=dict(type='RoIAlign', output_size=7, sampling_ratio=0),
                            out_channels=256, featmap_strides=_strides),
    bbox_head=dict(type='ProbConvFCBBoxHead', num_shared_fcs=2, in_channels=256,
                   fc_out_channels=1024, roi_feat_size=7, num_classes=4,
                   bbox_coder=dict(type='DeltaXYWHBBoxCoder', target_means=[0., 0., 0., 0.],
                                   target_stds=[0.1, 0.1, 0.2, 0.2]),
                   reg_class_agnostic=False,
                   loss_cls=dict(type='CrossEntropyLoss', use_sigmoid=False, loss_weight=2.0),
                   loss_bbox=dict(type='L1Loss', loss_weight=2.0)))


def _max_iou(pos, neg, min_pos, low_quality):
    return dict(type='MaxIoUAssigner', pos_iou_thr=pos, neg_iou_thr=neg, min_pos_iou=min_pos,
                match_low_quality=low_quality, ignore_iof_thr=-1)


model = dict(
    type='FasterRCNN',
    backbone=dict(type='ResNet', depth=50, num_stages=4, out_indices=(0, 1, 2, 3), frozen_stages=1,
                  norm_cfg=dict(type='BN', requires_grad=True), norm_eval=True, style='pytorch',
                  init_cfg=dict(type='Pretrained', checkpoint='torchvision://resnet50')),
    neck=dict(type='PAFPN', in_channels=[256, 512, 1024, 2048], out_channels=256, start_level=1,
              add_extra_convs='on_output', num_outs=5),
    rpn_head=_rpn_head,
    roi_head=_roi_head,
    train_cfg=dict(
        rpn=dict(assigner=_max_iou(0.5, 0.5, 0, True),
                 sampler=dict(_delete_=True, type='PseudoSampler'),
                 allowed_border=-1, pos_weight=-1, debug=False),
        rpn_proposal=dict(nms_pre=4000, max_per_img=2000, nms=_nms07, min_bbox_size=0),
        rcnn=dict(assigner=_max_iou(0.6, 0.6, 0.6, False),
                  sampler=dict(type='RandomSampler', num=512, pos_fraction=0.25, neg_pos_ub=-1,
                               add_gt_as_proposals=True),
                  pos_weight=-1, debug=False)),
    test_cfg=dict(
        rpn=dict(nms_pre=1000, max_per_img=256, nms=_nms07, min_bbox_size=0),
        # soft-nms is also supported here, e.g. nms=dict(type='soft_nms', iou_threshold=0.7, min_score=0.0)
        rcnn=dict(score_thr=0.05, nms=_nms07, max_per_img=100)))

optimizer_config = dict(_delete_=True, grad_clip=dict(max_norm=35, norm_type=2))
data = dict(samples_per_gpu=4, workers_per_gpu=8)
del _strides, _nms07, _rpn_head, _roi_head
