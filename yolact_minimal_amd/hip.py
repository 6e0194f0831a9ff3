"""ctypes binding of the C-ABI in include/yolact_hip.h (libyolact_hip.so, gfx950).

This is the only place Python touches the native library.  Every wrapper takes torch CUDA tensors
(torch is used for device memory and streams only), checks dtype/contiguity, and passes raw device
pointers + sizes + the current HIP stream.  There is deliberately NO fallback: if the shared library
is missing or a call fails, a RuntimeError is raised.
"""
import ctypes
import os

import torch

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('YM_LIB_PATH') or os.path.join(_PKG, 'libyolact_hip.so')   # override: debug builds only

ACT_NONE, ACT_RELU, ACT_TANH, ACT_GELU = 0, 1, 2, 3

# every symbol include/yolact_hip.h declares (checked by tests/test_abi.py without a GPU)
ABI_SYMBOLS = (
    'ym_abi_version', 'ym_last_error', 'ym_nchw_to_nhwc4', 'ym_pack_conv_weight', 'ym_fold_bn',
    'ym_sizeof_conv_desc', 'ym_conv2d_workspace_bytes', 'ym_conv2d_tile_counters', 'ym_conv2d_fwd', 'ym_maxpool3x3s2_fwd', 'ym_stem_conv_bn_relu_maxpool', 'ym_bilinear2x_fwd',
    'ym_softmax_rows', 'ym_nms_workspace_bytes', 'ym_detect_fast_nms', 'ym_detect_greedy_nms',
    'ym_greedy_nms_workspace_bytes', 'ym_greedy_nms', 'ym_mask_assemble', 'ym_mask_resize_binarize',
    'ym_boxes_to_pixels', 'ym_expf_cr', 'ym_nms_batch_workspace_bytes', 'ym_detect_fast_nms_batch', 'ym_after_nms_batch_workspace_bytes',
    'ym_after_nms_batch', 'ym_head_grad_gather', 'ym_scatter3',
    'ym_pack_conv_weight_dgrad', 'ym_pack_conv_weights_batch', 'ym_conv2d_wgrad_workspace_bytes', 'ym_conv2d_wgrad', 'ym_conv2d_wgrad_slabs', 'ym_wgrad_reduce_batch', 'ym_bn_train_fwd',
    'ym_val_preprocess', 'ym_layernorm', 'ym_patch_merge_layernorm', 'ym_swin_window_attention',
    'ym_mask_loss_workspace_bytes', 'ym_mask_loss_fwd_bwd', 'ym_mask_loss_batch_workspace_bytes', 'ym_mask_loss_batch',
    'ym_mask_iou_workspace_bytes', 'ym_mask_iou', 'ym_box_iou', 'ym_match_detections', 'ym_rle_encode', 'ym_ann_to_mask_workspace_bytes', 'ym_poly_to_mask', 'ym_runs_to_mask', 'ym_train_aug_image', 'ym_train_aug_masks',
    'ym_layernorm_bwd_workspace_bytes', 'ym_layernorm_bwd', 'ym_patch_merge_layernorm_bwd', 'ym_gelu_fwd', 'ym_gelu_bwd',
    'ym_swin_window_attention_bwd', 'ym_adamw_step', 'ym_drop_path_add', 'ym_drop_path_bwd', 'ym_select_positives',
    'ym_match_anchors', 'ym_match_anchors_batch', 'ym_loss_workspace_bytes', 'ym_class_box_loss', 'ym_semantic_loss',
    'ym_semantic_loss_batch',
    'ym_bn_train_bwd_workspace_bytes', 'ym_bn_train_bwd', 'ym_bn_train_bwd_apply', 'ym_act_bias_bwd', 'ym_conv2d_fuses_bn_stats', 'ym_bn_train_fwd_stats', 'ym_maxpool3x3s2_bwd', 'ym_maxpool3x3s2_fwd_idx', 'ym_maxpool3x3s2_bwd_idx', 'ym_bilinear2x_bwd', 'ym_sgd_step',
)


class ConvSeg(ctypes.Structure):
    _fields_ = [('n_begin', ctypes.c_int), ('n_end', ctypes.c_int), ('out', ctypes.c_void_p),
                ('batch_stride', ctypes.c_int64), ('pitch', ctypes.c_int32), ('act', ctypes.c_int32)]


class ConvDesc(ctypes.Structure):
    _fields_ = [('inp', ctypes.c_void_p), ('weight', ctypes.c_void_p), ('scale', ctypes.c_void_p),
                ('shift', ctypes.c_void_p), ('residual', ctypes.c_void_p),
                ('B', ctypes.c_int32), ('H', ctypes.c_int32), ('W', ctypes.c_int32), ('Cin', ctypes.c_int32),
                ('Cout', ctypes.c_int32), ('KH', ctypes.c_int32), ('KW', ctypes.c_int32),
                ('stride', ctypes.c_int32), ('pad', ctypes.c_int32), ('Ho', ctypes.c_int32),
                ('Wo', ctypes.c_int32), ('k_pad', ctypes.c_int32), ('nseg', ctypes.c_int32),
                ('seg', ConvSeg * 3), ('tile_m', ctypes.c_int32), ('tile_n', ctypes.c_int32),
                ('ksplit', ctypes.c_int32), ('kwaves', ctypes.c_int32), ('transposed', ctypes.c_int32),
                ('stages', ctypes.c_int32), ('bn_sum', ctypes.c_void_p), ('bn_sumsq', ctypes.c_void_p),
                ('tile_counters', ctypes.c_void_p), ('nlevels', ctypes.c_int32), ('level_h', ctypes.c_int32 * 5),
                ('level_w', ctypes.c_int32 * 5), ('tail_tiles', ctypes.c_int32), ('tail_ksplit', ctypes.c_int32), ('mma', ctypes.c_int32),
                ('bnb_relu', ctypes.c_int32), ('bnb_y', ctypes.c_void_p), ('bnb_out', ctypes.c_void_p),
                ('bnb_mean', ctypes.c_void_p), ('bnb_invstd', ctypes.c_void_p), ('bnb_gamma', ctypes.c_void_p),
                ('bnb_beta', ctypes.c_void_p), ('grid_wgs', ctypes.c_int32)]


class WgradDesc(ctypes.Structure):
    _fields_ = [('x', ctypes.c_void_p), ('dy', ctypes.c_void_p), ('dw', ctypes.c_void_p),
                ('B', ctypes.c_int32), ('H', ctypes.c_int32), ('W', ctypes.c_int32), ('Cin', ctypes.c_int32),
                ('Cin_real', ctypes.c_int32), ('Cout', ctypes.c_int32), ('Cout_real', ctypes.c_int32),
                ('KH', ctypes.c_int32), ('KW', ctypes.c_int32), ('stride', ctypes.c_int32), ('pad', ctypes.c_int32),
                ('Ho', ctypes.c_int32), ('Wo', ctypes.c_int32), ('msplit', ctypes.c_int32), ('accumulate', ctypes.c_int32),
                ('row_end', ctypes.c_int32 * 2), ('dw_seg', ctypes.c_void_p * 2), ('lds_buffers', ctypes.c_int32)]


class WgradReduceItem(ctypes.Structure):
    _fields_ = [('slabs', ctypes.c_void_p), ('dw', ctypes.c_void_p), ('first_block', ctypes.c_uint32), ('blocks', ctypes.c_uint32),
                ('plan', ctypes.c_uint32 * 14)]


class AugPlanC(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in ('H', 'W', 'mirror', 'cx', 'cy', 'cw', 'ch', 'q', 'px', 'py', 'r', 'S', 'final_mode',
                                              'fx', 'fy', 'has_brightness', 'has_contrast')] + \
               [(n, ctypes.c_float) for n in ('brightness', 'contrast', 'saturation', 'hue')] + \
               [('mean', ctypes.c_float * 3), ('std', ctypes.c_float * 3)]


class PackItem(ctypes.Structure):
    _fields_ = [('src', ctypes.c_void_p), ('dst', ctypes.c_void_p)] + \
               [(n, ctypes.c_int32) for n in ('cout', 'cin', 'kh', 'kw', 'pad_a', 'pad_b', 'rows', 'kind')] + \
               [('first_chunk', ctypes.c_uint32), ('reserved', ctypes.c_uint32)]


class MaskLossItem(ctypes.Structure):
    _fields_ = [('proto', ctypes.c_void_p), ('coef_full', ctypes.c_void_p), ('anchor_box', ctypes.c_void_p),
                ('anchor_gt', ctypes.c_void_p), ('gt_masks_ds', ctypes.c_void_p), ('anchor_idx', ctypes.c_void_p),
                ('n', ctypes.c_int32), ('wscale', ctypes.c_float), ('n_dev', ctypes.c_void_p), ('dproto', ctypes.c_void_p),
                ('dcoef_full', ctypes.c_void_p)]


class NmsCfg(ctypes.Structure):
    _fields_ = [('num_anchors', ctypes.c_int32), ('num_classes', ctypes.c_int32), ('coef_dim', ctypes.c_int32),
                ('top_k', ctypes.c_int32), ('max_det', ctypes.c_int32), ('score_thre', ctypes.c_float),
                ('iou_thre', ctypes.c_float), ('img_size', ctypes.c_float)]


_lib = None


def lib():
    """Load libyolact_hip.so once; raise (never fall back) if it is not there."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f'{LIB_PATH} is missing: build it with `python -c "import __graft_entry__ as g; g.build()"` '
                f'(or `make -C yolact_minimal_amd/csrc`). yolact_minimal_amd has no non-HIP fallback.')
        L = ctypes.CDLL(LIB_PATH)
        vp, i32, i64, f32, sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float, ctypes.c_size_t
        L.ym_abi_version.restype = ctypes.c_int
        L.ym_last_error.restype = ctypes.c_char_p
        L.ym_nchw_to_nhwc4.argtypes = [vp, vp, i32, i32, i32, i32, vp]
        L.ym_pack_conv_weight.argtypes = [vp, vp, i32, i32, i32, i32, i32, i32, vp]
        L.ym_fold_bn.argtypes = [vp, vp, vp, vp, f32, vp, vp, i32, vp]
        L.ym_sizeof_conv_desc.restype = sz
        if L.ym_sizeof_conv_desc() != ctypes.sizeof(ConvDesc):
            raise RuntimeError(f'libyolact_hip.so was built from a different ym_conv_desc ({L.ym_sizeof_conv_desc()} B) than '
                               f'yolact_minimal_amd/hip.py mirrors ({ctypes.sizeof(ConvDesc)} B): rebuild the library')
        L.ym_conv2d_workspace_bytes.argtypes = [ctypes.POINTER(ConvDesc)]
        L.ym_conv2d_workspace_bytes.restype = sz
        L.ym_conv2d_fwd.argtypes = [ctypes.POINTER(ConvDesc), vp, sz, vp]
        L.ym_conv2d_tile_counters.argtypes = [ctypes.POINTER(ConvDesc)]
        L.ym_maxpool3x3s2_fwd.argtypes = [vp, vp, i32, i32, i32, i32, vp]
        L.ym_stem_conv_bn_relu_maxpool.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, i32, vp]
        L.ym_bilinear2x_fwd.argtypes = [vp, vp, i32, i32, i32, i32, i32, vp]
        L.ym_softmax_rows.argtypes = [vp, vp, i64, i32, vp]
        L.ym_nms_workspace_bytes.argtypes = [ctypes.POINTER(NmsCfg)]
        L.ym_nms_workspace_bytes.restype = sz
        nms_args = [vp, vp, vp, vp, ctypes.POINTER(NmsCfg), vp, vp, vp, vp, vp, vp, sz, vp]
        L.ym_detect_fast_nms.argtypes = nms_args
        L.ym_detect_greedy_nms.argtypes = nms_args
        L.ym_greedy_nms_workspace_bytes.argtypes = [i32]
        L.ym_greedy_nms_workspace_bytes.restype = sz
        L.ym_greedy_nms.argtypes = [vp, i32, f32, vp, vp, vp, sz, vp]
        L.ym_mask_assemble.argtypes = [vp, vp, vp, i32, i32, i32, i32, i32, vp, vp]
        L.ym_mask_resize_binarize.argtypes = [vp, i32, i32, i32, i32, i32, vp, vp]
        L.ym_boxes_to_pixels.argtypes = [vp, vp, i32, f32, vp]
        L.ym_expf_cr.argtypes = [vp, vp, i64, vp]
        L.ym_nms_batch_workspace_bytes.argtypes = [ctypes.POINTER(NmsCfg), i32]
        L.ym_nms_batch_workspace_bytes.restype = sz
        L.ym_detect_fast_nms_batch.argtypes = [vp, vp, vp, vp, ctypes.POINTER(NmsCfg), i32, vp, vp, vp, vp, vp, vp, sz, vp]
        L.ym_after_nms_batch_workspace_bytes.argtypes = [i32, i32, i32, i32, i32]
        L.ym_after_nms_batch_workspace_bytes.restype = sz
        L.ym_after_nms_batch.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp, vp, vp, sz, vp]
        L.ym_head_grad_gather.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp, vp, i32, vp, vp, vp]
        L.ym_scatter3.argtypes = [vp, vp, i32, vp, i32, vp, i32, i32, vp]
        L.ym_pack_conv_weight_dgrad.argtypes = [vp, vp, i32, i32, i32, i32, i32, vp]
        L.ym_pack_conv_weights_batch.argtypes = [vp, i32, i32, vp]
        L.ym_conv2d_wgrad_workspace_bytes.argtypes = [ctypes.POINTER(WgradDesc)]
        L.ym_conv2d_wgrad_workspace_bytes.restype = sz
        L.ym_conv2d_wgrad.argtypes = [ctypes.POINTER(WgradDesc), vp, sz, vp]
        L.ym_conv2d_wgrad_slabs.argtypes = [ctypes.POINTER(WgradDesc), vp, sz, ctypes.POINTER(WgradReduceItem), vp]
        L.ym_wgrad_reduce_batch.argtypes = [vp, i32, ctypes.c_uint32, vp]
        L.ym_bn_train_fwd.argtypes = [vp, i64, i32, vp, vp, f32, f32, vp, vp, vp, i32, vp, vp, vp, vp, sz, vp]
        L.ym_val_preprocess.argtypes = [vp, i32, i32, i32, i32, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float), vp, vp]
        L.ym_layernorm.argtypes = [vp, vp, vp, f32, vp, i64, i32, vp]
        L.ym_patch_merge_layernorm.argtypes = [vp, i32, i32, i32, i32, vp, vp, f32, vp, vp]
        L.ym_swin_window_attention.argtypes = [vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp, vp]
        L.ym_mask_loss_workspace_bytes.argtypes = []
        L.ym_mask_loss_workspace_bytes.restype = sz
        L.ym_mask_loss_fwd_bwd.argtypes = [vp, vp, vp, vp, vp, vp, i32, i32, i32, f32, f32, vp, vp, vp, vp, sz, vp]
        L.ym_mask_loss_batch_workspace_bytes.argtypes = [i32]
        L.ym_mask_loss_batch_workspace_bytes.restype = sz
        L.ym_mask_loss_batch.argtypes = [ctypes.POINTER(MaskLossItem), i32, i32, i32, f32, vp, vp, vp, sz, vp]
        L.ym_mask_iou_workspace_bytes.argtypes = [i32, i32, i64]
        L.ym_mask_iou_workspace_bytes.restype = sz
        L.ym_mask_iou.argtypes = [vp, i32, vp, i32, i64, vp, vp, sz, vp]
        L.ym_box_iou.argtypes = [vp, i32, vp, i32, vp, vp]
        L.ym_match_detections.argtypes = [vp, vp, vp, vp, i32, i32, vp, i32, i32, vp, vp]
        L.ym_rle_encode.argtypes = [vp, i32, i32, i32, vp, i32, vp, vp, i32, vp, vp, sz, vp]
        L.ym_ann_to_mask_workspace_bytes.argtypes = [i32, i32, i32]
        L.ym_ann_to_mask_workspace_bytes.restype = sz
        L.ym_poly_to_mask.argtypes = [vp, vp, vp, i32, i32, i32, vp, vp, sz, vp]
        L.ym_runs_to_mask.argtypes = [vp, vp, i32, i32, i32, vp, vp, sz, vp]
        L.ym_layernorm_bwd_workspace_bytes.argtypes = [i32]
        L.ym_layernorm_bwd_workspace_bytes.restype = sz
        L.ym_layernorm_bwd.argtypes = [vp, vp, vp, f32, i64, i32, vp, vp, vp, vp, sz, vp]
        L.ym_patch_merge_layernorm_bwd.argtypes = [vp, vp, i32, i32, i32, i32, vp, f32, vp, vp, vp, vp, sz, vp]
        L.ym_gelu_fwd.argtypes = [vp, vp, i64, vp]
        L.ym_gelu_bwd.argtypes = [vp, vp, vp, i64, vp]
        L.ym_swin_window_attention_bwd.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp, vp, vp, vp]
        L.ym_select_positives.argtypes = [vp, vp, i32, i32, i32, vp, vp, vp]
        L.ym_drop_path_add.argtypes = [vp, vp, vp, ctypes.c_float, vp, i32, i64, vp]
        L.ym_drop_path_bwd.argtypes = [vp, vp, ctypes.c_float, vp, i32, i64, vp]
        L.ym_adamw_step.argtypes = [vp, vp, vp, vp, i64, f32, f32, f32, f32, f32, i32, vp]
        L.ym_train_aug_image.argtypes = [vp, i32, ctypes.POINTER(AugPlanC), vp, vp]
        L.ym_train_aug_masks.argtypes = [vp, i32, vp, i32, ctypes.POINTER(AugPlanC), vp, vp]
        L.ym_match_anchors.argtypes = [vp, i32, vp, i32, f32, f32, vp, vp, vp, vp, vp, sz, vp]
        L.ym_match_anchors_batch.argtypes = [vp, vp, i32, vp, i32, f32, f32, vp, vp, vp, vp, vp, sz, vp]
        L.ym_semantic_loss_batch.argtypes = [vp, i32, i32, i32, i32, vp, vp, i32, vp, f32, vp, vp, vp]
        L.ym_loss_workspace_bytes.argtypes = [i32, i32]
        L.ym_loss_workspace_bytes.restype = sz
        L.ym_class_box_loss.argtypes = [vp, vp, vp, vp, i32, i32, i32, f32, f32, i32, vp, vp, vp, vp, vp, vp, sz, vp]
        L.ym_semantic_loss.argtypes = [vp, i32, i32, i32, vp, vp, i32, i32, f32, vp, vp, vp]
        L.ym_conv2d_fuses_bn_stats.argtypes = [ctypes.POINTER(ConvDesc)]
        L.ym_bn_train_fwd_stats.argtypes = [vp, i64, i32, vp, vp, f32, f32, vp, vp, vp, i32, vp, vp, vp, vp, vp]
        L.ym_bn_train_bwd_workspace_bytes.argtypes = [i64, i32]
        L.ym_bn_train_bwd_workspace_bytes.restype = sz
        L.ym_bn_train_bwd.argtypes = [vp, vp, vp, i64, i32, vp, vp, vp, vp, i32, vp, vp, vp, vp, vp, sz, vp]
        L.ym_bn_train_bwd_apply.argtypes = [vp, vp, vp, i64, i32, vp, vp, vp, vp, i32, vp, vp, vp, vp, vp, vp]
        L.ym_act_bias_bwd.argtypes = [vp, vp, i64, i32, i32, vp, vp, vp, sz, vp]
        L.ym_maxpool3x3s2_bwd.argtypes = [vp, vp, vp, i32, i32, i32, i32, vp]
        L.ym_maxpool3x3s2_fwd_idx.argtypes = [vp, vp, vp, i32, i32, i32, i32, vp]
        L.ym_maxpool3x3s2_bwd_idx.argtypes = [vp, vp, vp, i32, i32, i32, i32, vp]
        L.ym_bilinear2x_bwd.argtypes = [vp, vp, i32, i32, i32, i32, i32, vp]
        L.ym_sgd_step.argtypes = [vp, vp, vp, i64, f32, f32, f32, i32, vp]
        for name in ABI_SYMBOLS:
            fn = getattr(L, name)
            if name not in ('ym_last_error', 'ym_conv2d_workspace_bytes', 'ym_nms_workspace_bytes',
                            'ym_greedy_nms_workspace_bytes', 'ym_conv2d_wgrad_workspace_bytes',
                            'ym_sizeof_conv_desc', 'ym_bn_train_bwd_workspace_bytes', 'ym_mask_loss_workspace_bytes', 'ym_mask_loss_batch_workspace_bytes', 'ym_loss_workspace_bytes', 'ym_mask_iou_workspace_bytes', 'ym_layernorm_bwd_workspace_bytes',
                            'ym_ann_to_mask_workspace_bytes'):
                fn.restype = ctypes.c_int
        _lib = L
    return _lib


def check(rc, what):
    if rc != 0:
        raise RuntimeError(f'{what} failed (rc={rc}): {lib().ym_last_error().decode()}')


def stream_ptr():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t, dtype=torch.float32):
    if t is None:
        return None
    if not t.is_cuda or t.dtype != dtype or not t.is_contiguous():
        raise RuntimeError(f'expected a contiguous {dtype} CUDA tensor, got {t.dtype} {t.device} '
                           f'contiguous={t.is_contiguous()}')
    return ctypes.c_void_p(t.data_ptr())


# ---- thin per-op wrappers (used by engine.py, utils/output_utils.py and the tests) ----------------

def nchw_to_nhwc4(img, out):
    b, c, h, w = img.shape
    check(lib().ym_nchw_to_nhwc4(ptr(img), ptr(out), b, c, h, w, stream_ptr()), 'ym_nchw_to_nhwc4')


def pack_conv_weight(w_oihw, cin_pad, k_pad):
    cout, cin, kh, kw = w_oihw.shape
    out = torch.empty(cout, k_pad, device=w_oihw.device, dtype=torch.float32)
    check(lib().ym_pack_conv_weight(ptr(w_oihw.contiguous()), ptr(out), cout, cin, kh, kw, cin_pad, k_pad,
                                    stream_ptr()), 'ym_pack_conv_weight')
    return out


def fold_bn(gamma, beta, mean, var, eps):
    c = gamma.numel()
    scale, shift = torch.empty_like(gamma), torch.empty_like(gamma)
    check(lib().ym_fold_bn(ptr(gamma), ptr(beta), ptr(mean), ptr(var), eps, ptr(scale), ptr(shift), c,
                           stream_ptr()), 'ym_fold_bn')
    return scale, shift


def conv_workspace_bytes(desc):
    """Bytes of scratch ym_conv2d_fwd needs for this descriptor (0 = none); raises if the descriptor is invalid."""
    n = lib().ym_conv2d_workspace_bytes(ctypes.byref(desc))
    if n == 0 and (desc.ksplit > 1 or desc.tail_tiles > 0) and desc.kwaves == 0:
        err = lib().ym_last_error().decode()
        if err and 'conv' in err and desc.tail_tiles > 0:
            raise RuntimeError(f'ym_conv2d_workspace_bytes: {err}')
    return n


TILE_COUNTERS = 16384     # int32 entries the engines allocate for ym_conv_desc.tile_counters


def conv2d_fwd(desc, workspace=None):
    counters = desc.tile_counters
    if counters and lib().ym_conv2d_tile_counters(ctypes.byref(desc)) > TILE_COUNTERS:
        desc.tile_counters = None              # more output tiles than counters: separate reduce launch (this launch only)
    ws_ptr = ctypes.c_void_p(workspace.data_ptr()) if workspace is not None else None
    ws_bytes = workspace.numel() * workspace.element_size() if workspace is not None else 0
    try:
        check(lib().ym_conv2d_fwd(ctypes.byref(desc), ws_ptr, ws_bytes, stream_ptr()), 'ym_conv2d_fwd')
    finally:
        desc.tile_counters = counters          # the caller's descriptor is not ours to edit (tuners re-plan it with other tiles)


def stem_conv_bn_relu_maxpool(img, w_packed, scale, shift, out):
    """img NCHW [B,3,H,W] -> out NHWC [B,Hp,Wp,64]: the eval-mode ResNet stem in one launch (ym_stem_conv_bn_relu_maxpool)."""
    b, c, h, w = img.shape
    if c != 3 or tuple(w_packed.shape) != (64, 224):
        raise RuntimeError(f'stem: expects a 3-channel image and a [64, 224] packed filter, got {tuple(img.shape)} / {tuple(w_packed.shape)}')
    check(lib().ym_stem_conv_bn_relu_maxpool(ptr(img), ptr(w_packed), ptr(scale), ptr(shift), ptr(out), b, h, w, w_packed.shape[1],
                                             stream_ptr()), 'ym_stem_conv_bn_relu_maxpool')


def maxpool3x3s2(x, out):
    b, h, w, c = x.shape
    check(lib().ym_maxpool3x3s2_fwd(ptr(x), ptr(out), b, h, w, c, stream_ptr()), 'ym_maxpool3x3s2_fwd')


def bilinear2x(x, out, align_corners):
    b, h, w, c = x.shape
    check(lib().ym_bilinear2x_fwd(ptr(x), ptr(out), b, h, w, c, int(bool(align_corners)), stream_ptr()),
          'ym_bilinear2x_fwd')


def softmax_rows(x, out):
    c = x.shape[-1]
    rows = x.numel() // c
    check(lib().ym_softmax_rows(ptr(x), ptr(out), rows, c, stream_ptr()), 'ym_softmax_rows')


def layernorm(x, gamma, beta, eps, out):
    c = x.shape[-1]
    check(lib().ym_layernorm(ptr(x), ptr(gamma), ptr(beta), float(eps), ptr(out), x.numel() // c, c, stream_ptr()), 'ym_layernorm')


def patch_merge_layernorm(x, gamma, beta, eps, out):
    b, h, w, c = x.shape
    check(lib().ym_patch_merge_layernorm(ptr(x), b, h, w, c, ptr(gamma), ptr(beta), float(eps), ptr(out), stream_ptr()),
          'ym_patch_merge_layernorm')


def swin_window_attention(qkv, qkv_bias, table, b, h, w, c, heads, window, shift, out):
    check(lib().ym_swin_window_attention(ptr(qkv), ptr(qkv_bias), ptr(table), b, h, w, c, heads, window, shift, ptr(out),
                                         stream_ptr()), 'ym_swin_window_attention')


def mask_assemble(proto, coefs, boxes, out, do_crop=True):
    hp, wp, k = proto.shape
    n = coefs.shape[0]
    check(lib().ym_mask_assemble(ptr(proto), ptr(coefs), ptr(boxes), n, hp, wp, k, int(do_crop), ptr(out),
                                 stream_ptr()), 'ym_mask_assemble')


def mask_resize_binarize(masks, img_h, img_w, out):
    n, hp, wp = masks.shape
    check(lib().ym_mask_resize_binarize(ptr(masks), n, hp, wp, img_h, img_w, ptr(out), stream_ptr()),
          'ym_mask_resize_binarize')


def expf_cr(x):
    """exp(x) rounded to nearest float (the decode's exp; see include/yolact_hip.h)."""
    y = torch.empty_like(x)
    check(lib().ym_expf_cr(ptr(x), ptr(y), x.numel(), stream_ptr()), 'ym_expf_cr')
    return y


def boxes_to_pixels(boxes_f, boxes_px, size):
    n = boxes_f.shape[0]
    check(lib().ym_boxes_to_pixels(ptr(boxes_f), ptr(boxes_px, torch.int32), n, float(size), stream_ptr()),
          'ym_boxes_to_pixels')
