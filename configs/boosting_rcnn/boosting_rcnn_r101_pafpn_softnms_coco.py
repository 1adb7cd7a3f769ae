# BASELINE.json run 5: ResNet-101 override of the COCO PAFPN recipe with the soft-NMS test
# settings of boosting_rcnn_r2_101_dcn_pafpn_mstrain_3x_coco.py:24-28 and 2000 proposals/img.
_base_ = 'boosting_rcnn_r50_pafpn_mstrain_2x_coco.py'
model = dict(
    backbone=dict(depth=101, init_cfg=dict(type='Pretrained', checkpoint='torchvision://resnet101')),
    test_cfg=dict(rpn=dict(nms_pre=2000, max_per_img=2000),
                  rcnn=dict(score_thr=0.0001,
                            nms=dict(_delete_=True, type='soft_nms', iou_threshold=0.7, min_score=0.0),
                            max_per_img=200)))
