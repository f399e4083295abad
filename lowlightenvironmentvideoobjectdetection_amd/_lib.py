"""ctypes binding of libllv.so (the C ABI declared in include/llv.h).

This is the only place the shared library is opened. There is no fallback: if the library is
missing or an entry point fails, the caller gets an exception.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libllv.so')

LLV_MAX_SEG = 5
LLV_MAX_SRC = 3
LLV_MAX_BATCH = 16

# llv_conv_desc.flags
CONV_RELU = 1
CONV_RES_ADD = 2
CONV_RES_RSUB = 4
CONV_PIXSHUF = 8
CONV_RES_UPS = 16
CONV_OUT_F32 = 32
CONV_LEAKY = 64
CONV_ACT_FIRST = 128
CONV_PS_EMAJOR = 256
CONV_CLSMAX = 512
CONV_MASK_POS = 1024

TILE_AUTO, TILE_128x128, TILE_64x256, TILE_32x256, TILE_16x256, TILE_64x64 = 0, 1, 2, 3, 4, 5
TILE_128x128_REGSTAGE = 6
TILE_128x128_K64, TILE_64x256_K64 = 7, 8
TILE_64x64_K64 = 14
TILE_TAP128, TILE_TAP64 = 9, 10
TILE_TAP64_R4 = 11
TILE_TAP64_G3 = 12
TILE_TAP512 = 13


class LlvError(RuntimeError):
    pass


class Extents(C.Structure):
    """llv_extents: bytes addressable from each pointer of a descriptor (0 = not declared); `below` is an output bit mask."""
    _fields_ = [('x', C.c_uint64 * LLV_MAX_SRC), ('w', C.c_uint64), ('bias', C.c_uint64), ('res', C.c_uint64), ('y', C.c_uint64),
                ('workspace', C.c_uint64), ('w_halo', C.c_uint64), ('w_pre', C.c_uint64), ('bias_pre', C.c_uint64), ('below', C.c_int32)]
    NAMES = ('x0', 'x1', 'x2', 'w', 'bias', 'res', 'y', 'workspace', 'w_halo', 'w_pre', 'bias_pre')

    def as_dict(self):
        d = {f'x{i}': int(self.x[i]) for i in range(LLV_MAX_SRC)}
        d.update({n: int(getattr(self, n)) for n in self.NAMES[3:]})
        return d


class BottleneckDesc(C.Structure):
    """llv_bottleneck_desc (include/llv.h)."""
    _fields_ = [('x', C.c_void_p), ('y', C.c_void_p), ('w1', C.c_void_p), ('w2', C.c_void_p), ('w3', C.c_void_p),
                ('b1', C.c_void_p), ('b2', C.c_void_p), ('b3', C.c_void_p),
                ('batch', C.c_int32), ('h', C.c_int32), ('w', C.c_int32), ('mid', C.c_int32),
                ('x_ld', C.c_int32), ('y_ld', C.c_int32)]


class PackJob(C.Structure):
    """llv_pack_job"""
    _fields_ = [('w', C.c_void_p), ('gamma', C.c_void_p), ('var', C.c_void_p), ('eps', C.c_float), ('dst', C.c_void_p)] + \
               [(n, C.c_int32) for n in ('cout', 'cin', 'kh', 'kw', 'cpad', 'k_pad', 'row0', 'mode', 'pad', 'KH', 'KW', 'padp', 'rows')]


class FoldJob(C.Structure):
    """llv_fold_job"""
    _fields_ = [('b', C.c_void_p), ('gamma', C.c_void_p), ('beta', C.c_void_p), ('mean', C.c_void_p), ('var', C.c_void_p), ('eps', C.c_float),
                ('out', C.c_void_p), ('n', C.c_int32)]


LLV_WGRAD_JOB_ARGS = 368


class WgradJob(C.Structure):
    """llv_wgrad_job"""
    _fields_ = [('kind', C.c_int32), ('wg_begin', C.c_int32), ('n_wg', C.c_int32), ('reserved', C.c_int32),
                ('args', C.c_ubyte * LLV_WGRAD_JOB_ARGS)]


class UnpackJob(C.Structure):
    """llv_unpack_job"""
    _fields_ = [('dw', C.c_void_p), ('db', C.c_void_p), ('w', C.c_void_p), ('gamma', C.c_void_p), ('mean', C.c_void_p), ('var', C.c_void_p),
                ('eps', C.c_float), ('gw', C.c_void_p), ('ggamma', C.c_void_p), ('gbeta', C.c_void_p), ('gbias', C.c_void_p)] + \
               [(n, C.c_int32) for n in ('cout', 'cin', 'kh', 'kw', 'cpad', 'k_pad')]


class RetinaLossDesc(C.Structure):
    """llv_retina_loss_desc (include/llv.h)."""
    _fields_ = [('cls', C.c_void_p), ('reg', C.c_void_p), ('dcls', C.c_void_p), ('dreg', C.c_void_p),
                ('cls_ld', C.c_int32), ('reg_ld', C.c_int32), ('dcls_ld', C.c_int32), ('dreg_ld', C.c_int32),
                ('batch', C.c_int32), ('n_levels', C.c_int32), ('num_anchors', C.c_int32), ('num_classes', C.c_int32),
                ('level_h', C.c_int32 * LLV_MAX_SEG), ('level_w', C.c_int32 * LLV_MAX_SEG),
                ('level_off', C.c_int64 * LLV_MAX_SEG), ('level_anchor0', C.c_int64 * LLV_MAX_SEG),
                ('anchors_per_image', C.c_int64),
                ('anchors', C.c_void_p), ('gt_inds', C.c_void_p), ('gt_boxes', C.c_void_p), ('gt_labels', C.c_void_p), ('max_gt', C.c_int32),
                ('gamma', C.c_float), ('alpha', C.c_float), ('cls_weight', C.c_float), ('reg_weight', C.c_float), ('beta', C.c_float),
                ('pos_weight', C.c_float), ('grad_scale', C.c_float),
                ('means', C.c_float * 4), ('stds', C.c_float * 4),
                ('sums', C.c_void_p), ('avg_factor', C.c_void_p), ('losses', C.c_void_p), ('counts', C.c_void_p),
                ('partials', C.c_void_p), ('partials_cap', C.c_int32)]


class ConvDesc(C.Structure):
    _fields_ = [
        ('x', C.c_void_p * LLV_MAX_SRC),
        ('w', C.c_void_p),
        ('bias', C.c_void_p),
        ('res', C.c_void_p),
        ('y', C.c_void_p),
        ('n_src', C.c_int32), ('src_cvec', C.c_int32),
        ('x_ld', C.c_int32), ('y_ld', C.c_int32), ('res_ld', C.c_int32),
        ('cin', C.c_int32), ('cout', C.c_int32), ('w_rows', C.c_int32), ('k_pad', C.c_int32),
        ('kh', C.c_int32), ('kw', C.c_int32), ('stride', C.c_int32), ('pad', C.c_int32),
        ('groups', C.c_int32), ('batch', C.c_int32),
        ('n_seg', C.c_int32),
        ('seg_h', C.c_int32 * LLV_MAX_SEG), ('seg_w', C.c_int32 * LLV_MAX_SEG),
        ('seg_ho', C.c_int32 * LLV_MAX_SEG), ('seg_wo', C.c_int32 * LLV_MAX_SEG),
        ('seg_xoff', C.c_int64 * LLV_MAX_SEG), ('seg_yoff', C.c_int64 * LLV_MAX_SEG),
        ('flags', C.c_int32),
        ('res_h', C.c_int32), ('res_w', C.c_int32),
        ('tile', C.c_int32),
        ('k_splits', C.c_int32),
        ('workspace', C.c_void_p),
        ('workspace_bytes', C.c_size_t),
        ('w_halo', C.c_void_p),
        ('halo_bn', C.c_int32),
        ('y_group_stride', C.c_int64),
        ('act_slope', C.c_float),
        ('ps_out_h', C.c_int32), ('ps_out_w', C.c_int32),
        ('dilation', C.c_int32),
        ('extents', C.POINTER(Extents)),
        ('offset_dev', C.c_void_p), ('offset_x', C.c_int32), ('offset_res', C.c_int32), ('offset_max', C.c_int64),
        ('clsmax', C.c_void_p), ('seg_written', C.c_void_p), ('cls_c', C.c_int32), ('store_thr', C.c_float),
        ('mask', C.c_void_p), ('mask_ld', C.c_int32),
    ]


class InputCvDesc(C.Structure):
    _fields_ = [('conv', ConvDesc), ('w_pre', C.c_void_p), ('bias_pre', C.c_void_p)]


class DecodeDesc(C.Structure):
    _fields_ = [
        ('n_levels', C.c_int32), ('num_classes', C.c_int32), ('num_anchors', C.c_int32), ('nms_pre', C.c_int32),
        ('level_h', C.c_int32 * LLV_MAX_SEG), ('level_w', C.c_int32 * LLV_MAX_SEG),
        ('stride_w', C.c_int32 * LLV_MAX_SEG), ('stride_h', C.c_int32 * LLV_MAX_SEG),
        ('cls_off', C.c_int64 * LLV_MAX_SEG), ('reg_off', C.c_int64 * LLV_MAX_SEG),
        ('cls_ld', C.c_int32), ('reg_ld', C.c_int32),
        ('logits_f32', C.c_int32),
        ('cls', C.c_void_p), ('reg', C.c_void_p), ('base_anchors', C.c_void_p),
        ('means', C.c_float * 4), ('stds', C.c_float * 4),
        ('max_ratio', C.c_float),
        ('clip', C.c_int32),
        ('img_h', C.c_float), ('img_w', C.c_float),
        ('scale_factor', C.c_float * 4),
        ('rescale', C.c_int32),
        ('boxes', C.c_void_p), ('scores', C.c_void_p), ('cand_anchor', C.c_void_p),
        ('n_cand', C.c_int32),
        ('workspace', C.c_void_p), ('workspace_bytes', C.c_size_t),
        ('batch', C.c_int32),
        ('img_h_b', C.c_float * LLV_MAX_BATCH), ('img_w_b', C.c_float * LLV_MAX_BATCH),
        ('scale_factor_b', (C.c_float * 4) * LLV_MAX_BATCH),
        ('clsmax', C.c_void_p), ('seg_written', C.c_void_p), ('seg_tiles', C.c_int32),
    ]


# name -> (restype, argtypes); every symbol include/llv.h declares
_P, _I, _L, _F, _Z = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_size_t
SIGNATURES = {
    'llv_abi_version': (_I, []),
    'llv_strerror': (C.c_char_p, [_I]),
    'llv_conv2d_nhwc_fwd': (_I, [C.POINTER(ConvDesc), _P]),
    'llv_conv2d_flops': (C.c_double, [C.POINTER(ConvDesc)]),
    'llv_conv2d_workspace_bytes': (_Z, [C.POINTER(ConvDesc)]),
    'llv_conv2d_plan': (_I, [C.POINTER(ConvDesc), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    'llv_conv2d_extents': (_I, [C.POINTER(ConvDesc), C.POINTER(Extents)]),
    'llv_convpair_extents': (_I, [C.POINTER(InputCvDesc), _I, C.POINTER(Extents)]),
    'llv_extents_fit': (_I, [C.POINTER(Extents), C.POINTER(Extents)]),
    'llv_inputcvblock_fwd': (_I, [C.POINTER(InputCvDesc), _P]),
    'llv_outputcvblock_fwd': (_I, [C.POINTER(InputCvDesc), _P]),
    'llv_maxpool3x3s2_nhwc': (_I, [_P, _P, _I, _I, _I, _I, _P]),
    'llv_subsample2_nhwc': (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _P]),
    'llv_maxpool3x3s2_nhwc_bwd': (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    'llv_pack_nchw_f32_to_nhwc_f16': (_I, [_P, _P, _I, _I, _I, _I, _I, _P, _P, _P]),
    'llv_unpack_nhwc_f16_to_nchw_f32': (_I, [_P, _P, _I, _I, _I, _I, _I, _P]),
    'llv_unpack_nhwc_f32_to_nchw_f32': (_I, [_P, _P, _I, _I, _I, _I, _I, _P]),
    'llv_retina_decode_ncand': (_I, [C.POINTER(DecodeDesc)]),
    'llv_retina_decode_workspace_bytes': (_Z, [C.POINTER(DecodeDesc)]),
    'llv_retina_decode_topk': (_I, [C.POINTER(DecodeDesc), _P]),
    'llv_multiclass_nms_workspace_bytes': (_Z, [_I, _I, _I]),
    'llv_multiclass_nms': (_I, [_P, _P, _I, _I, _I, _F, _F, _I, _P, _P, _P, _P, _P, _Z, _P]),
    'llv_multiclass_nms_batched_workspace_bytes': (_Z, [_I, _I, _I, _I]),
    'llv_multiclass_nms_lds_bytes': (_Z, [_I, _I]),
    'llv_multiclass_nms_lds_bytes_ex': (_Z, [_I, _I, _I]),
    'llv_multiclass_nms_lds_limit': (_Z, []),
    'llv_multiclass_nms_batched': (_I, [_P, _P, _I, _I, _I, _I, _F, _F, _I, _P, _P, _P, _P, _P, _Z, _P]),
    'llv_multiclass_nms_batched_sf': (_I, [_P, _P, _P, _I, _I, _I, _I, _F, _F, _I, _P, _P, _P, _P, _P, _Z, _P]),
    'llv_multiclass_nms_batched_ex': (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _F, _F, _I, _P, _P, _P, _P, _P, _Z, _P]),
    'llv_brighten_normalize_pad_u8': (_I, [_P, _I, _I, _I, _I, _I, _F, C.POINTER(C.c_float), C.POINTER(C.c_float), _I, _P, _P, _P]),
    'llv_upsample2x_nhwc': (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _P]),
    'llv_yolo_decode': (_I, [_P, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P]),
    'llv_nms_workspace_bytes': (_Z, [_I]),
    'llv_nms': (_I, [_P, _P, _I, _F, _P, _P, _P, _Z, _P]),
    'llv_embed_aggregate': (_I, [_P, _P, _P, _P, _I, _L, _I, _I, _I, _I, _P]),
    'llv_embed_aggregate_ring': (_I, [_P, _P, _P, _I, _I, _L, _I, _I, _I, _I, _P]),
    'llv_embed_aggregate_bwd': (_I, [_P, _P, _P, _P, _P, _I, _I, _L, _I, _I, _I, _I, _P]),
    'llv_flow_warp_bwd': (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    'llv_add_f32_to_f16': (_I, [_P, _P, _L, _I, _I, _P]),
    'llv_flow_prepare': (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _P, _F, _P]),
    'llv_bilinear_resize_f32': (_I, [_P, _P, _I, _I, _I, _I, _I, _F, _F, _P]),
    'llv_flow_warp': (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    'llv_sigmoid_focal_loss_fwd': (_I, [_P, _P, _P, _L, _I, _F, _F, _P]),
    'llv_sigmoid_focal_loss_bwd': (_I, [_P, _P, _P, _L, _I, _F, _F, _P]),
    'llv_softmax_ce_fwd': (_I, [_P, _P, _P, _P, _L, _I, _I, _P]),
    'llv_softmax_ce_bwd': (_I, [_P, _P, _P, _P, _P, _L, _I, _I, _P]),
    'llv_iou_assign_workspace_bytes': (_Z, [_I, _L]),
    'llv_iou_assign': (_I, [_P, _I, _P, _P, _L, _F, _F, _F, _I, _I, _P, _P, _P, _Z, _P]),
    'llv_modulated_deform_im2col': (_I, [_P, _P, _P] + [_I] * 15 + [_P]),
    'llv_modulated_deform_col2im': (_I, [_P, _P, _P, _P, _P] + [_I] * 15 + [_P]),
    'llv_taf_pair_concat': (_I, [_P, _P, _I, _I, _I, _I, _P]),
    'llv_taf_mul_ref': (_I, [_P, _P, _P, _I, _I, _I, _I, _P]),
    'llv_taf_softmax_fuse': (_I, [_P, _P, _P, _I, _I, _I, _I, _P]),
    'llv_add_f16': (_I, [_P, _P, _P, _L, _I, _I, _I, _I, _P]),
    'llv_coco_match': (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _P, _I, _P, _I, _L, _P, _P, _P, _P]),
    'llv_copy_d2d': (_I, [_P, _P, _Z, _P]),
    'llv_pack_results': (_I, [_P, _P, _P, _I, _I, _P, _P, _P]),
    'llv_guard_trips': (_I, [C.POINTER(C.c_uint64), _I]),
    'llv_guard_trips_to': (_I, [_P, _P]),
    'llv_roi_align': (_I, [_P, _P, _P] + [_I] * 8 + [_F, _I, _I, _I, _P]),
    'llv_selsa_pack_kv': (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    'llv_selsa_softmax': (_I, [_P, _P, _L, _I, _I, _I, _I, _P, _F, _P]),
    'llv_softmax_rows_masked': (_I, [_P, _P, _I, _I, _I, _I, _P, _P]),
    'llv_dets_to_rois': (_I, [_P, _P, _P, _P, _I, _I, _P]),
    'llv_roi_align_bwd': (_I, [_P, _P, _P] + [_I] * 7 + [_F, _I, _I, _P]),
    'llv_l2norm_rows_f16': (_I, [_P, _I, _P, _L, _I, _L, _L, _P, _P]),
    'llv_msra_top2_gather': (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    'llv_troi_attention_fuse': (_I, [_P, _P, _P, _I, _L, _I, _I, _P]),
    'llv_bottleneck_fwd': (_I, [C.POINTER(BottleneckDesc), _P]),
    'llv_delta2bbox': (_I, [_P, _P, _P, _L, _I, _P, _P, _F, _F, _F, _I, _P, _P]),
    'llv_eval_tpfp': (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _P, _I, _F, _I, _L, _P, _P, _P]),
    'llv_conv2d_nhwc_wgrad': (_I, [C.POINTER(ConvDesc), _P, _I, _P, _P, _P]),
    'llv_conv2d_nhwc_wgrad_plan': (_I, [C.POINTER(ConvDesc), _P, _I, _P, _P, _F, C.POINTER(WgradJob), C.POINTER(C.c_int)]),
    'llv_conv2d_nhwc_wgrad_jobs': (_I, [_P, _P, _I, _I, _P]),
    'llv_conv2d_nhwc_dgrad': (_I, [C.POINTER(ConvDesc), _P, _P, _I, _I, _I, _I, _I, _P, _I, _P, _I, _I, _P, _Z, _P]),
    'llv_pack_conv_weights': (_I, [_P, _P, _P, _F, _P] + [_I] * 13 + [_P]),
    'llv_fold_bias': (_I, [_P, _P, _P, _P, _P, _F, _P, _I, _P]),
    'llv_unpack_wgrad': (_I, [_P, _P, _P, _P, _P, _P, _F, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    'llv_relu_bwd': (_I, [_P, _P, _P, _P, _L, _P]),
    'llv_relu_res_bwd': (_I, [_P, _P, _P, _L, _P]),
    'llv_act_bwd': (_I, [_P, _P, _P, _P, _L, _I, _I, _I, _I, _I, _F, _P]),
    'llv_upsample2x_nhwc_bwd': (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _P]),
    'llv_bias_grad': (_I, [_P, _L, _I, _I, _P, _P]),
    'llv_upsample_add_bwd': (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    'llv_sgd_momentum': (_I, [_P, _P, _P, _L, _F, _F, _F, _F, _I, _P]),
    'llv_retina_loss': (_I, [_P, _P]),
    'llv_flow_warp_bwd_flow': (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    'llv_bilinear_resize_f32_bwd': (_I, [_P, _P, _I, _I, _I, _I, _I, _F, _F, _P]),
    'llv_yolo_select_workspace_bytes': (_Z, [_I, _I]),
    'llv_yolo_select': (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _F, _P, _L, _L, _P, _P, _P, _P, _Z, _P]),
    'llv_convlstm_gate': (_I, [_P, _P, _P, _P, _P, _P, _L, _I, _I, _I, _I, _P]),
    'llv_convlstm_gate_bwd': (_I, [_P] * 11 + [_L, _I, _I, _I, _I, _P]),
    'llv_pack_conv_weights_batched': (_I, [_P, _I, _I, _P]),
    'llv_fold_bias_batched': (_I, [_P, _I, _I, _P]),
    'llv_pack_dgrad_weights_batched': (_I, [_P, _I, _I, _P]),
    'llv_unpack_wgrad_batched': (_I, [_P, _I, _I, _P]),
}

_lib = None


def lib():
    """Open libllv.so once. Raises LlvError if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise LlvError(
                f'{LIB_PATH} not found: build it with `python -c "import __graft_entry__ as g; g.build()"` '
                f'or `make -C {os.path.join(_HERE, "csrc")}`. There is no CPU fallback.')
        # torch first: its wheel bundles its own libamdhip64.so, and libllv.so must bind to THAT runtime (the one whose streams
        # and allocations it is handed). Loaded before torch, libllv.so pulls in /opt/rocm's copy and the process ends up with two
        # HIP runtimes: every launch then fails (seen as `kernel launch failed` in build(); smoke() run in one process).
        import torch  # noqa: F401
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError if the ABI lost a symbol
            fn.restype = res
            fn.argtypes = args
        if handle.llv_abi_version() != 3:
            raise LlvError('libllv.so ABI version mismatch')
        _lib = handle
    return _lib


def check(code, what=''):
    if code != 0:
        msg = lib().llv_strerror(code).decode()
        raise LlvError(f'{what or "llv call"} failed: {msg} ({code})')
