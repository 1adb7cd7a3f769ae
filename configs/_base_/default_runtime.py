# Runtime defaults shared by the boosting configs (same keys/values as the reference's
# configs/_base_/default_runtime.py so that tooling written against it keeps working).
checkpoint_config = dict(interval=1)
log_config = dict(interval=50, hooks=[dict(type='TextLoggerHook')])
custom_hooks = [dict(type='NumClassCheckHook')]
# 'nccl' is RCCL under PyTorch-ROCm
dist_params = dict(backend='nccl')
log_level = 'INFO'
load_from = None
resume_from = None
workflow = [('train', 1)]
