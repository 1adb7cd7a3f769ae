"""ctypes binding of libbrcnn_hip.so (the C ABI declared in include/brcnn_hip.h).

The HIP library is the product: there is no CPU or eager-PyTorch fallback.  `load()` raises
if the shared object is missing or does not load, `check()` raises on a non-zero status,
and every wrapper in `ops.py` refuses tensors that are not on a HIP device.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# (BRCNN_LIB_PATH: a development switch -- tools/experiments/ab_lib.sh times two builds of the library on one box)
LIB_PATH = os.environ.get('BRCNN_LIB_PATH') or os.path.join(_HERE, 'lib', 'libbrcnn_hip.so')
_lib = None

c_int, c_i64, c_f32, c_f64, c_ptr, c_size = (ctypes.c_int, ctypes.c_int64, ctypes.c_float,
                                             ctypes.c_double, ctypes.c_void_p, ctypes.c_size_t)

# name -> (restype, argtypes); must list every symbol include/brcnn_hip.h declares
SIGNATURES = {
    'brcnn_version': (c_int, []),
    'brcnn_device_count': (c_int, []),
    'brcnn_roi_align_forward': (c_int, [c_ptr] * 5 + [c_int] * 7 + [c_f32] + [c_int] * 4 + [c_ptr]),
    'brcnn_roi_align_backward': (c_int, [c_ptr] * 3 + [c_int] * 7 + [c_f32] + [c_int] * 3 + [c_ptr]),
    'brcnn_roi_extract_forward': (c_int, [c_ptr] * 4 + [c_int] + [c_ptr] * 3 + [c_int] * 6 +
                                  [c_f32, c_int, c_ptr]),
    'brcnn_roi_extract_forward_ordered': (c_int, [c_ptr] * 4 + [c_int] + [c_ptr] * 3 + [c_int] * 6 +
                                          [c_f32, c_int, c_ptr, c_ptr]),
    'brcnn_roi_extract_prep_workspace_bytes': (c_size, [c_int]),
    'brcnn_roi_extract_order_min_rois': (c_int, []),
    'brcnn_roi_extract_forward_prepared': (c_int, [c_ptr] * 4 + [c_int] + [c_ptr] * 3 + [c_int] * 6 +
                                           [c_f32, c_int, c_ptr, c_ptr, c_size, c_ptr]),
    'brcnn_get_tuning': (c_int, [c_ptr]),
    'brcnn_set_tuning': (c_int, [c_ptr]),
    'brcnn_roi_extract_backward': (c_int, [c_ptr] * 4 + [c_int] + [c_ptr] * 2 + [c_int] * 6 +
                                   [c_f32, c_ptr]),
    'brcnn_roi_extract_backward_workspace_bytes': (c_size, [c_int]),
    'brcnn_roi_extract_backward_workspace_bytes_ex': (c_size, [c_int, c_int, c_int, c_int, c_ptr, c_ptr]),
    'brcnn_roi_extract_backward_gather': (c_int, [c_ptr] * 4 + [c_int] + [c_ptr] * 2 + [c_int] * 6 + [c_f32, c_ptr, c_size,
                                                                                                    c_int, c_ptr]),
    'brcnn_roi_extract_backward_gather_add': (c_int, [c_ptr] * 5 + [c_int] + [c_ptr] * 2 + [c_int] * 6 + [c_f32, c_ptr, c_size,
                                                                                                        c_int, c_ptr]),
    'brcnn_nms_workspace_bytes': (c_size, [c_i64, c_int, c_i64]),
    'brcnn_nms': (c_int, [c_ptr] * 4 + [c_int, c_i64, c_i64, c_f32, c_int, c_int, c_ptr, c_ptr,
                                        c_ptr, c_size, c_ptr]),
    'brcnn_softnms_workspace_bytes': (c_size, [c_i64, c_int]),
    'brcnn_softnms': (c_int, [c_ptr] * 4 + [c_int, c_i64, c_f32, c_f32, c_f32, c_int, c_int,
                                            c_ptr, c_ptr, c_ptr, c_ptr, c_size, c_ptr]),
    'brcnn_sigmoid_focal_loss_forward': (c_int, [c_ptr] * 4 + [c_i64, c_i64, c_f32, c_f32, c_ptr]),
    'brcnn_sigmoid_focal_loss_backward': (c_int, [c_ptr] * 4 + [c_i64, c_i64, c_f32, c_f32, c_ptr]),
    'brcnn_conv2d_nhwc': (c_int, [c_ptr] * 6 + [c_int] * 11 + [c_ptr]),
    'brcnn_conv_set_tile': (c_int, [c_int, c_int]),
    'brcnn_clock_probe': (c_int, [c_ptr, c_i64, c_ptr]),
    'brcnn_nms_prepare': (c_int, [c_ptr] * 9 + [c_int, c_int, c_ptr]),
    'brcnn_nms_collect': (c_int, [c_ptr] * 7 + [c_int, c_int, c_int, c_ptr]),
    'brcnn_conv2d_nhwc_scatter2': (c_int, [c_ptr] * 3 + [c_int] * 14 + [c_ptr]),
    'brcnn_conv2d_nhwc_multi': (c_int, [c_ptr] * 6 + [c_int, c_int, c_ptr, c_ptr] + [c_int] * 8 +
                                [c_ptr]),
    'brcnn_conv2d_dgrad_nhwc_multi': (c_int, [c_ptr] * 3 + [c_int, c_int] + [c_ptr] * 4 + [c_int] * 7 +
                                      [c_ptr]),
    'brcnn_conv2d_wgrad_nhwc_multi': (c_int, [c_ptr] * 3 + [c_int, c_int] + [c_ptr] * 2 + [c_int] * 7 +
                                      [c_ptr]),
    'brcnn_wgrad_defer_begin': (c_int, [c_ptr, c_ptr, c_size, c_int]),
    'brcnn_wgrad_defer_flush': (c_int, [c_ptr]),
    'brcnn_wgrad_defer_pending': (c_int, [c_ptr]),
    'brcnn_wgrad_defer_stats': (c_int, [c_ptr, c_ptr, c_ptr]),
    'brcnn_stem_workspace_bytes': (c_size, [c_int, c_int, c_int]),
    'brcnn_stem7x7s2_nchw': (c_int, [c_ptr] * 6 + [c_int] * 6 + [c_ptr]),
    'brcnn_stem7x7s2_pool_nchw': (c_int, [c_ptr] * 5 + [c_int] * 5 + [c_ptr]),
    'brcnn_bottleneck_tail_f32': (c_int, [c_ptr] * 9 + [c_int] * 3 + [c_ptr]),
    'brcnn_bottleneck_tail_16': (c_int, [c_ptr] * 9 + [c_int] * 4 + [c_ptr]),
    'brcnn_conv_set_tile_bf16': (c_int, [c_int]),
    'brcnn_conv_workspace_bytes': (c_size, []),
    'brcnn_conv_set_workspace': (c_int, [c_ptr, c_ptr, c_size]),
    'brcnn_conv_handover_status': (c_int, []),
    'brcnn_maxpool3x3s2_nhwc': (c_int, [c_ptr] * 2 + [c_int] * 5 + [c_ptr]),
    'brcnn_maxpool3x3s2_nhwc_backward': (c_int, [c_ptr] * 4 + [c_int] * 5 + [c_ptr]),
    'brcnn_groupnorm_nhwc': (c_int, [c_ptr] * 5 + [c_int] * 4 + [c_f32, c_int, c_int, c_ptr]),
    'brcnn_groupnorm_nhwc_multi': (c_int, [c_ptr] * 5 + [c_int, c_int, c_ptr, c_int, c_int, c_f32,
                                                          c_int, c_int, c_ptr]),
    'brcnn_groupnorm_nhwc_multi_backward_workspace_bytes': (c_size, [c_int, c_int, c_ptr, c_int, c_int]),
    'brcnn_groupnorm_nhwc_multi_backward': (c_int, [c_ptr] * 9 + [c_size, c_int, c_int, c_ptr] + [c_int] * 4 +
                                            [c_ptr]),
    'brcnn_upsample_nearest_add_nhwc': (c_int, [c_ptr] * 2 + [c_int] * 7 + [c_ptr]),
    'brcnn_upsample_nearest_add_nhwc_out': (c_int, [c_ptr] * 3 + [c_int] * 7 + [c_ptr]),
    'brcnn_upsample_nearest_add_nhwc_backward': (c_int, [c_ptr] * 2 + [c_int] * 7 + [c_ptr]),
    'brcnn_colsum_workspace_bytes': (c_size, [c_i64, c_int]),
    'brcnn_colsum': (c_int, [c_ptr] * 3 + [c_size, c_i64, c_int, c_int, c_ptr]),
    'brcnn_nchw_to_nhwc': (c_int, [c_ptr] * 2 + [c_int] * 4 + [c_ptr]),
    'brcnn_nhwc_to_nchw': (c_int, [c_ptr] * 2 + [c_int] * 4 + [c_ptr]),
    'brcnn_rpn_score': (c_int, [c_ptr] * 3 + [c_i64, c_int, c_int, c_int, c_ptr]),
    'brcnn_rpn_decode': (c_int, [c_ptr] * 2 + [c_int, c_f32, c_ptr] + [c_int] * 7 + [c_ptr, c_ptr, c_f64, c_f32, c_f32,
                                                             c_f32, c_ptr, c_ptr, c_ptr]),
    'brcnn_rcnn_decode': (c_int, [c_ptr] * 6 + [c_int, c_int, c_int, c_f32, c_ptr, c_ptr, c_f64] + [c_ptr] * 5),
    'brcnn_rpn_decode_levels': (c_int, [c_ptr] * 5 + [c_int, c_int, c_ptr, c_ptr, c_ptr, c_int, c_ptr, c_ptr, c_ptr, c_ptr,
                                        c_f64, c_f32, c_f32, c_f32, c_ptr, c_ptr, c_ptr, c_ptr]),
    'brcnn_conv_set_tile_wgrad_bf16': (c_int, [c_int]),
    'brcnn_bn_act_forward': (c_int, [c_ptr] * 5 + [c_i64, c_int, c_int, c_int, c_ptr]),
    'brcnn_bn_act_backward_workspace_bytes': (ctypes.c_size_t, [c_i64, c_int, c_int]),
    'brcnn_bn_act_backward': (c_int, [c_ptr] * 9 + [ctypes.c_size_t, c_i64, c_int, c_int, c_int, c_ptr]),
    'brcnn_bn_eval_act_forward': (c_int, [c_ptr] * 5 + [c_f32, c_ptr, c_ptr, c_i64, c_int, c_int, c_int, c_ptr]),
    'brcnn_conv2d_bn_act_nhwc_multi': (c_int, [c_ptr] * 6 + [c_f32] + [c_ptr] * 3 + [c_int, c_int, c_ptr, c_ptr] + [c_int] * 8 +
                                       [c_ptr]),
    'brcnn_conv2d_dgrad_bn_backward_workspace_bytes': (ctypes.c_size_t, [c_int] * 4),
    'brcnn_conv2d_dgrad_bn_backward_nhwc': (c_int, [c_ptr] * 7 + [c_f32, c_int] + [c_ptr] * 7 + [ctypes.c_size_t] + [c_int] * 12 +
                                            [c_ptr]),
    'brcnn_bn_eval_act_backward': (c_int, [c_ptr] * 7 + [c_f32] + [c_ptr] * 5 + [ctypes.c_size_t, c_i64, c_int, c_int,
                                                                                c_int, c_ptr]),
    'brcnn_conv2d_dgrad_bn_backward_nhwc_ex': (c_int, [c_ptr] * 7 + [c_f32, c_int] + [c_ptr] * 7 + [ctypes.c_size_t] +
                                               [c_int] * 12 + [c_ptr, c_int]),
    'brcnn_bn_eval_act_backward_ex': (c_int, [c_ptr] * 7 + [c_f32] + [c_ptr] * 5 + [ctypes.c_size_t, c_i64, c_int, c_int,
                                                                                   c_int, c_ptr, c_int]),
    'brcnn_bn_reduce_flush': (c_int, [c_ptr]),
    'brcnn_bn_reduce_pending': (c_int, []),
    'brcnn_conv2d_nhwc_grouped': (c_int, [c_ptr] * 6 + [c_int] * 12 + [c_ptr]),
    'brcnn_avgpool_nhwc': (c_int, [c_ptr, c_ptr] + [c_int] * 9 + [c_ptr]),
    'brcnn_deform_im2col_nhwc': (c_int, [c_ptr] * 3 + [c_int] * 11 + [c_ptr]),
    'brcnn_roi_align_set_exact': (c_int, [c_int]),
    'brcnn_conv2d_dgrad_nhwc_grouped': (c_int, [c_ptr] * 3 + [c_int] * 13 + [c_ptr]),
    'brcnn_conv2d_wgrad_nhwc_grouped': (c_int, [c_ptr] * 3 + [c_int] * 11 + [c_ptr]),
    'brcnn_deform_col2im_nhwc': (c_int, [c_ptr] * 5 + [c_int] * 11 + [c_ptr]),
    'brcnn_pack_conv_weights': (c_int, [c_ptr] * 3 + [c_int] * 5 + [c_ptr]),
    'brcnn_rpn_topk_workspace_bytes': (ctypes.c_size_t, [c_ptr, c_int, c_int, c_int]),
    'brcnn_rpn_topk': (c_int, [c_ptr, c_ptr, c_int, c_int, c_int, c_ptr, c_ptr, c_ptr, ctypes.c_size_t, c_ptr]),
    'brcnn_preprocess_u8': (c_int, [c_ptr, c_int, c_int, c_ptr] + [c_int] * 5 + [c_ptr, c_ptr, c_int, c_ptr]),
    'brcnn_nms_prepare_levels': (c_int, [c_ptr] * 9 + [c_int, c_int, c_int, c_ptr, c_ptr]),
    'brcnn_nms_collect_sorted': (c_int, [c_ptr] * 10 + [c_int] * 4 + [c_ptr]),
    'brcnn_rpn_decode_levels_dscale': (c_int, [c_ptr] * 6 + [c_int, c_int, c_ptr, c_ptr, c_ptr, c_int, c_ptr, c_ptr, c_ptr,
                                               c_ptr, c_f64, c_f32, c_f32, c_f32, c_ptr, c_ptr, c_ptr, c_ptr]),
    'brcnn_sgd_workspace_bytes': (c_size, [c_int, c_ptr]),
    'brcnn_sgd_step': (c_int, [c_ptr] * 7 + [c_int, c_f32, c_f32, c_f32, c_int, c_ptr, c_size, c_ptr, c_ptr]),
    'brcnn_pack_conv_weights_batch': (c_int, [c_ptr] * 5 + [c_int, c_int, c_ptr, c_ptr]),
    'brcnn_pack_fc_weight_permuted': (c_int, [c_ptr] * 3 + [c_int] * 4 + [c_ptr, c_ptr]),
    'brcnn_bbox_overlaps': (c_int, [c_ptr, c_int, c_int, c_ptr, c_int, c_int, c_int, c_int, c_f32, c_ptr, c_ptr]),
    'brcnn_assign_max_iou': (c_int, [c_ptr, c_i64, c_int, c_ptr, c_int, c_int, c_ptr, c_ptr, c_int, c_ptr, c_ptr, c_int,
                                     c_ptr, c_ptr] + [c_f32] * 5 + [c_int] + [c_ptr] * 5),
    'brcnn_rcnn_sample': (c_int, [c_ptr, c_ptr, c_int, c_int] + [c_ptr] * 5 + [c_int, c_int, c_int, c_f32, c_ptr, c_ptr,
                                  c_int, c_int, c_ptr, c_ptr, c_ptr, c_int] + [c_ptr] * 7),
    'brcnn_rpn_loss_workspace_bytes': (c_size, [c_int, c_int, c_ptr, c_ptr, c_int]),
    'brcnn_rpn_loss_forward': (c_int, [c_ptr, c_int, c_int, c_int] + [c_ptr] * 5 + [c_int] + [c_ptr] * 6 +
                               [c_size, c_ptr, c_ptr, c_ptr]),
    'brcnn_rpn_loss_finalize': (c_int, [c_ptr, c_ptr, c_int] + [c_ptr] * 5),
    'brcnn_rpn_loss_backward': (c_int, [c_ptr, c_int, c_int, c_int] + [c_ptr] * 5 + [c_int] + [c_ptr] * 8 +
                                [c_size, c_ptr, c_ptr, c_ptr]),
    'brcnn_boost_loss_workspace_bytes': (c_size, [c_int]),
    'brcnn_boost_loss_forward': (c_int, [c_ptr] * 6 + [c_int, c_int, c_int, c_ptr, c_ptr, c_size, c_ptr, c_ptr, c_ptr]),
    'brcnn_boost_loss_backward': (c_int, [c_ptr] * 6 + [c_int, c_int, c_int] + [c_ptr] * 6),
    'brcnn_boost_loss_forward_ex': (c_int, [c_ptr] * 6 + [c_int, c_int, c_int, c_ptr, c_ptr, c_size, c_ptr, c_ptr, c_ptr]),
    'brcnn_boost_loss_backward_ex': (c_int, [c_ptr] * 6 + [c_int, c_int, c_int] + [c_ptr] * 6),
}


class BrcnnHipError(RuntimeError):
    pass


class Tuning(ctypes.Structure):
    """include/brcnn_hip.h: brcnn_tuning -- the library's policy switches as one struct (`get_tuning()` / `set_tuning()`)"""
    _fields_ = [(n, ctypes.c_int) for n in (
        'size', 'conv_stream_k', 'conv_split_k', 'conv_eight_phase_16bit', 'conv_persistent_1x1', 'conv_eight_phase_f32',
        'wgrad_slab_reduction', 'wgrad_eight_phase', 'wgrad_reduce_in_launch', 'wgrad_generation_percent',
        'wgrad_eight_phase_cu_percent', 'roi_exact_order', 'roi_rows_per_wave', 'roi_visit_order', 'roi_prepared_records')]


def get_tuning():
    t = Tuning()
    t.size = ctypes.sizeof(Tuning)
    check(load().brcnn_get_tuning(ctypes.addressof(t)), 'brcnn_get_tuning')
    return t


def set_tuning(**fields):
    """change some of the policy switches (the others keep their current values); returns the struct that was set"""
    t = get_tuning()
    for k, v in fields.items():
        if k == 'size' or not hasattr(t, k):
            raise KeyError(f'brcnn_tuning has no field {k!r}')
        setattr(t, k, int(v))
    check(load().brcnn_set_tuning(ctypes.addressof(t)), 'brcnn_set_tuning')
    return t


def load():
    """Load the HIP library (once).  No fallback: a missing library is an error."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise BrcnnHipError(
            f'{LIB_PATH} is missing: build it with `python __graft_entry__.py` (or '
            f'`python boosting-r-cnn_amd/build.py`); the HIP library is the product path and '
            f'there is no CPU fallback')
    try:
        lib = ctypes.CDLL(LIB_PATH)
    except OSError as e:
        raise BrcnnHipError(f'cannot load {LIB_PATH}: {e}')
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(status, what):
    if status != 0:
        if status == -22:
            raise BrcnnHipError(f'{what}: invalid argument (status -22)')
        if status <= -1000:
            raise BrcnnHipError(f'{what}: HIP error {-status - 1000}')
        if status == -62:
            raise BrcnnHipError(f'{what}: an earlier convolution launch lost a stream-K hand-over between two workgroups '
                                '(BRCNN_EHANDOVER): the results computed since the last check are invalid')
        raise BrcnnHipError(f'{what}: status {status}')


# ---- caller-owned conv scratch (include/brcnn_hip.h: brcnn_conv_set_workspace) ------------------------------------
_workspaces = {}        # (device index, hipStream_t handle) -> the torch tensor registered as that stream's conv workspace


import torch as _torch

# the raw handle of torch's current stream without building a torch.cuda.Stream object (a third of a microsecond instead of
# three to four: the train step asks ~110 times per step)
_raw_current = getattr(_torch._C, '_cuda_getCurrentRawStream', None)
_cur_device = getattr(_torch._C, '_cuda_getDevice', None)


def raw_stream_handle(stream=None):
    """the hipStream_t of `stream` (default: torch's current stream) for a C-ABI call that launches no convolution"""
    if stream is None:
        if _raw_current is not None:
            return _raw_current(_cur_device())
        stream = _torch.cuda.current_stream()
    return stream.cuda_stream


def stream_handle(stream=None):
    """the hipStream_t of `stream` (default: torch's current stream) for a C-ABI call that launches a convolution
    (forward, data gradient, weight gradient); on first sight of a (device, stream) pair its convolution scratch
    (stream-K hand-over slots, weight-gradient slabs: 288 MiB) is allocated HERE, by the caller, and registered -- the
    library allocates nothing for the streams this module drives.  The device is part of the key: the default stream
    is handle 0 on every device, and the scratch lives in one device's memory (the library keys its table the same way).
    Streams that only ever run the other kernels (proposal stage, losses, copies) go through `raw_stream_handle` and
    pin nothing."""
    if stream is None and _raw_current is not None:
        dev = _cur_device()
        h = _raw_current(dev)
        if (dev, h) in _workspaces:
            return h
    if stream is None:
        stream = _torch.cuda.current_stream()
    h = stream.cuda_stream
    key = (stream.device.index, h)
    if key not in _workspaces:
        lib = load()
        nb = int(lib.brcnn_conv_workspace_bytes())
        with _torch.cuda.stream(stream):        # (also makes the stream's device current for the registration call)
            ws = _torch.empty(nb, dtype=_torch.uint8, device=stream.device)
            check(lib.brcnn_conv_set_workspace(h, ws.data_ptr(), nb), 'brcnn_conv_set_workspace')
        _workspaces[key] = ws
    return h


def handover_status():
    """raise if a stream-K hand-over timed out since the last call (cheap: reads one host word)"""
    check(load().brcnn_conv_handover_status(), 'brcnn_conv_handover_status')
