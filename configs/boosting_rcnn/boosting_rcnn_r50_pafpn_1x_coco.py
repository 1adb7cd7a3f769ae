# The name README.md:63 and BASELINE.json use for the COCO R50-PAFPN recipe.  The reference
# tree does not contain this file; the shipped recipe with these settings is
# boosting_rcnn_r50_pafpn_mstrain_2x_coco.py, so this is a thin wrapper around it.
_base_ = 'boosting_rcnn_r50_pafpn_mstrain_2x_coco.py'
