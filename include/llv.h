/*
 * llv.h -- C ABI of libllv.so: the MI355X (gfx950) kernels behind the low-light
 * video detection hot path (enhance-CNN -> ResNet/FPN -> RetinaHead -> decode -> NMS).
 *
 * Conventions
 *   - extern "C", plain pointers and sizes; no torch / C++ types cross this boundary.
 *   - Every pointer is a DEVICE pointer unless the name ends in _host. Buffers are
 *     caller-owned; no entry point allocates, frees or synchronises (graph-capture safe).
 *   - `stream` is a hipStream_t passed as void*. Work is stream-ordered.
 *   - Return value: 0 = LLV_OK, negative = LLV_E_* (bad argument, launch failure).
 *     Nothing throws. llv_strerror() maps a code to text.
 *   - Activations are NHWC fp16 ("channels-last half"), channel counts padded to a
 *     multiple of 8 unless stated; accumulation and epilogues are fp32.
 *
 * What each entry point replaces in the reference (paths relative to /root/reference):
 *   llv_conv2d_nhwc_fwd      torch.nn.Conv2d (+ folded BatchNorm2d, ReLU, residual add, nearest
 *                            upsample-add, PixelShuffle) as used by
 *                            mmdetection/mmdet/models/backbones/resnet.py:260-299,622-640,
 *                            mmdetection/mmdet/models/necks/fpn.py:170-221,
 *                            mmdetection/mmdet/models/dense_heads/retina_head.py:93-114,
 *                            mmtracking/mmtrack/models/cleaners/FastDVDnet.py:7-116,
 *                            mmtracking/mmtrack/models/aggregators/embed_aggregator.py:64-72
 *   llv_maxpool3x3s2_nhwc    nn.MaxPool2d(3,2,1), resnet.py:570,632
 *   llv_pack_nchw_f32_to_nhwc_f16 / llv_unpack_nhwc_f16_to_nchw_f32
 *                            layout change at the model boundary (img tensor N,3,H,W fp32:
 *                            mmdetection/mmdet/models/detectors/base.py:169-183)
 *   llv_retina_decode_topk   AnchorHead._get_bboxes per-level sigmoid/max/topk/gather/delta2bbox,
 *                            mmdetection/mmdet/models/dense_heads/anchor_head.py:636-711 and
 *                            mmdetection/mmdet/core/bbox/coder/delta_xywh_bbox_coder.py:134-237,
 *                            anchors from mmdetection/mmdet/core/anchor/anchor_generator.py:207-270
 *   llv_multiclass_nms       multiclass_nms -> mmcv.ops.batched_nms -> mmcv.ops.nms,
 *                            mmdetection/mmdet/core/post_processing/bbox_nms.py:7-93
 *   llv_nms                  mmcv.ops.nms(boxes, scores, iou_threshold, offset=0) (mmcv-full 1.3.0,
 *                            un-vendored; call site bbox_nms.py:2,84)
 *   llv_embed_aggregate      EmbedAggregator.forward tail (L2-norm, cosine, softmax over frames,
 *                            weighted sum), embed_aggregator.py:70-81
 *   llv_flow_warp            flow_warp_feats, mmtracking/mmtrack/core/motion/flow.py:4-41
 *   llv_sigmoid_focal_loss_* mmcv.ops.sigmoid_focal_loss fwd/bwd, call site
 *                            mmdetection/mmdet/models/losses/focal_loss.py:59-98
 *   llv_modulated_deform_im2col  column step of mmcv.ops.modulated_deform_conv2d (mmcv-full 1.3.0, un-vendored; call site
 *                            mmtracking/mmtrack/models/aggregators/denoising_aggregator.py:103-111), SURVEY.md 8f item 3
 *   llv_roi_align            mmcv.ops.roi_align behind SingleRoIExtractor (first piece)
 *   llv_eval_tpfp            tpfp_default / tpfp_imagenet of eval_map,
 *                            mmdetection/mmdet/core/evaluation/mean_ap.py:60-244, bbox_overlaps.py:4-48
 *   llv_coco_match           COCOeval.evaluateImg (pycocotools, un-vendored; call site coco_video_dataset.py:374-461)
 *   llv_iou_assign           MaxIoUAssigner.assign + bbox_overlaps,
 *                            mmdetection/mmdet/core/bbox/assigners/max_iou_assigner.py:60-212,
 *                            mmdetection/mmdet/core/bbox/iou_calculators/iou2d_calculator.py:43-159
 */
#ifndef LLV_H_
#define LLV_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LLV_OK 0
#define LLV_E_BADARG (-1)
#define LLV_E_LAUNCH (-2)
#define LLV_E_UNSUPPORTED (-3)
#define LLV_E_WORKSPACE (-4)
#define LLV_E_EXTENT (-5) /* the launch would touch memory outside a declared buffer extent (llv_extents) */

#define LLV_ABI_VERSION 3

#define LLV_MAX_SEG 5 /* pyramid levels batched into one conv launch */
#define LLV_MAX_SRC 3 /* input tensors concatenated along C by one conv launch */
#define LLV_MAX_BATCH 16 /* images decoded / suppressed by one batched post-processing launch */

/*
 * Buffer extents: bytes addressable from each pointer of a descriptor to the end of the allocation it points into
 * (0 = not declared, not checked). The reference never needed this (ATen owns its tensors' sizes); a C ABI that takes
 * raw pointers does: several kernels fetch through hardware buffer resources whose scalar offsets are outside the
 * range check, read residuals / skips as whole 16-byte lines and start resources before the tensor (padding rows).
 * llv_conv2d_extents / llv_convpair_extents replay, on the host and in the kernels' own 32-bit arithmetic, every address
 * the selected kernel variant forms for its operands (all workgroups, waves, lanes and k-steps; the buffer rule "voffset
 * >= num_records reads zero, soffset is not checked" included) and report 1 + the highest byte touched per pointer, and
 * whether any access falls below a pointer. When a descriptor carries `extents` and the environment variable
 * LLV_CHECK_EXTENTS=1 is set (or the library was built with -DLLV_DEBUG_EXTENTS), llv_conv2d_nhwc_fwd /
 * llv_inputcvblock_fwd / llv_outputcvblock_fwd run that replay before launching and return LLV_E_EXTENT on a violation.
 */
typedef struct llv_extents {
    uint64_t x[LLV_MAX_SRC];
    uint64_t w, bias, res, y, workspace, w_halo, w_pre, bias_pre;
    int32_t below; /* out (llv_*_extents): bit i set = an access below pointer i (order of the fields above, x[0] = bit 0) */
} llv_extents;

/* llv_conv_desc.flags */
#define LLV_CONV_RELU 1        /* y = max(y, 0) last */
#define LLV_CONV_RES_ADD 2     /* y = conv + bias + res */
#define LLV_CONV_RES_RSUB 4    /* y = res - (conv + bias)   (FastDVDnet.py:115 `in1 - x`) */
#define LLV_CONV_PIXSHUF 8     /* store through nn.PixelShuffle(2): out (2Ho,2Wo,cout/4) */
#define LLV_CONV_RES_UPS 16    /* res is (res_h,res_w), nearest-upsampled to (Ho,Wo) (fpn.py:185-191) */
#define LLV_CONV_OUT_F32 32    /* y is fp32 instead of fp16 */
#define LLV_CONV_PS_EMAJOR 256 /* with PIXSHUF: the conv's output channels (weight rows, bias) are packed sub-pixel-major, channel = (2*dy+dx)*(Cout/4) + cc, instead of torch's 4*cc + 2*dy + dx; 3x3/s1 launches on the tap kernel only */
#define LLV_CONV_ACT_FIRST 128 /* with RES_ADD: y = act(conv + bias) + res (Darknet ResBlock, darknet.py:47-53) instead of act(conv + bias + res) */
#define LLV_CONV_LEAKY 64      /* y = y > 0 ? y : act_slope*y last (FlowNetSimple LeakyReLU(0.1), flownet_simple.py:66-68) */
#define LLV_CONV_MASK_POS 1024 /* last: y = mask > 0 ? y : 0, `mask` an fp16 tensor laid out like y (mask_ld channels per pixel). The data
                                * gradient of a convolution whose INPUT is a ReLU output: dX takes that ReLU's mask from the activation
                                * itself in the launch that computes it (llv_conv2d_nhwc_dgrad with the flag in the forward descriptor);
                                * not with PIXSHUF / OUT_F32 / planar groups */
#define LLV_CONV_CLSMAX 512    /* the class convolution of a dense head (retina_head.py:93-114 `retina_cls`) with the first step of
                                * get_bboxes (anchor_head.py:636-711: sigmoid, per-anchor class maximum) in its epilogue: with OUT_F32 on the
                                * 3x3/s1/p1 tap kernel (128-channel tile), channels a*cls_c + c. The launch atomically maximises
                                * clsmax[row*A + a] (A = cout / cls_c) with the order-preserving uint encoding of every logit of anchor a, and
                                * stores a pixel's 16-channel block of y only if one of its logits exceeds store_thr, recording that as bit
                                * (block % 8) of seg_written[row*ceil(cout/128) + block/8]. llv_retina_decode_topk reads both
                                * (llv_decode_desc.clsmax / seg_written): the dense fp32 logits are neither written in full nor read back.
                                * cls_c % 16 == 0. */

/*
 * One convolution launch, implicit GEMM on MFMA (fp16 in, fp32 accumulate).
 *
 *   y[p, co] = epilogue( bias[co] + sum_{r,q,c} w[co][r][q][c] * x[pix(p,r,q), c] )
 *
 * x      : n_src tensors, NHWC fp16, x_ld channels per pixel each; the launch sees them
 *          concatenated along C (src_cvec 8-channel vectors per source). n_src == 1 normally.
 *          With groups == n_src > 1, group g reads source tensor g instead (grouped conv over separate
 *          tensors: FastDVDnet.py:27-28 Conv2d(9, 90, 3, groups=3) over three frames).
 * w      : per group [w_rows][k_pad] fp16, k = (r*kw + q)*cin + c, zero padded to k_pad (mult of 32)
 *          and to w_rows (mult of 16) rows; groups are consecutive.
 * bias   : fp32 [groups*cout] (folded BN shift or conv bias), never NULL.
 * res    : optional NHWC fp16 residual (res_ld channels per pixel), see flags.
 * y      : NHWC fp16 (or fp32 with LLV_CONV_OUT_F32), y_ld channels per pixel.
 * cin    : per-group input channels, multiple of 8 (zero padded by the producer).
 * cout   : per-group output channels stored, multiple of 4.
 * Segments: n_seg >= 1 images-sets of different spatial size that share the weights (the five
 *          FPN levels through one RetinaHead tower conv). Segment s has `batch` images of
 *          seg_h[s] x seg_w[s] pixels starting at pixel offset seg_xoff[s] of x, and writes
 *          batch x seg_ho[s] x seg_wo[s] pixels starting at pixel offset seg_yoff[s] of y.
 * groups : blockIdx.y; group g reads channels [g*cin, (g+1)*cin) of x and writes
 *          channels [g*cout, (g+1)*cout) of y.
 */
typedef struct llv_conv_desc {
    const void* x[LLV_MAX_SRC];
    const void* w;
    const float* bias;
    const void* res;
    void* y;
    int32_t n_src, src_cvec;
    int32_t x_ld, y_ld, res_ld;
    int32_t cin, cout, w_rows, k_pad;
    int32_t kh, kw, stride, pad;
    int32_t groups, batch;
    int32_t n_seg;
    int32_t seg_h[LLV_MAX_SEG], seg_w[LLV_MAX_SEG], seg_ho[LLV_MAX_SEG], seg_wo[LLV_MAX_SEG];
    int64_t seg_xoff[LLV_MAX_SEG], seg_yoff[LLV_MAX_SEG];
    int32_t flags;
    int32_t res_h, res_w;
    int32_t tile;     /* 0 = auto; else LLV_TILE_* to force a tile shape (tests, tuning) */
    int32_t k_splits; /* 0 = auto; else force this many K splits (1 = none) */
    void* workspace;  /* fp32 split-K partial sums; llv_conv2d_workspace_bytes(desc) bytes (0 if no split) */
    size_t workspace_bytes;
    /* Optional second packing of the same weights for the direct (halo-in-LDS) 3x3 stride-1 kernel:
     * [group][chunk][n-tile][halo_bn rows][Kc_pad + 8] fp16 with Kc = 9*CK, CK = min(cin, 32), k = tap*CK + c.
     * NULL (or tile != 0) selects the implicit-GEMM kernel. halo_bn in {16, 32, 64}. */
    const void* w_halo;
    int32_t halo_bn;
    /* != 0: group g writes channels [0, cout) of its own plane y + g*y_group_stride (fp16 elements, multiple of 8)
     * instead of channels [g*cout, (g+1)*cout) of one interleaved tensor; the consumer reads the planes as n_src sources. */
    int64_t y_group_stride;
    float act_slope;               /* negative slope for LLV_CONV_LEAKY */
    int32_t ps_out_h, ps_out_w;    /* LLV_CONV_PIXSHUF: stored extent (<= 2Ho x 2Wo; 0 = full). A 4x4/stride-2/pad-1 transposed conv is
                                      a 3x3 conv to 4*Cout channels + this shuffle; the crop is FlowNetSimple.crop_like (flownet_simple.py:246-251) */
    int32_t dilation;              /* 0 or 1: none; d > 1: tap (r, q) reads input (y*stride - pad + r*d, x*stride - pad + q*d) -- the dilated
                                      last stage of the DC5 backbones of the SELSA configs (resnet.py `dilations=(1, 1, 1, 2)`); forward only,
                                      on the implicit-GEMM kernels */
    const llv_extents* extents;    /* optional declared sizes of the buffers behind the pointers above (NULL: none) */
    /* Sliding window over a resident clip (llv_inputcvblock_fwd / llv_outputcvblock_fwd only; llv_conv2d_nhwc_fwd returns
     * LLV_E_UNSUPPORTED): offset_dev != NULL is a DEVICE pointer to one int64, an offset in fp16 elements that the kernel adds,
     * when it starts, to x[0..n_src-1] (offset_x != 0) and / or to res (offset_res != 0). A captured HIP graph then reads whichever
     * window of the clip the caller last stored there (stream-ordered, e.g. a fill kernel in front of the replay): no staging copy
     * of the window, no re-capture. The reference re-stacks the frames on the host for every key frame
     * (mmtracking/mmtrack/models/vid/selsa_fastdvd_darkfarm.py:294-319). offset_max = the largest value the caller will store
     * (used by the extent check). */
    const int64_t* offset_dev;
    int32_t offset_x, offset_res;
    int64_t offset_max;
    /* LLV_CONV_CLSMAX (see the flag): */
    uint32_t* clsmax;        /* [rows of y][cout / cls_c], zero before the launch (llv_retina_decode_topk leaves it zeroed again) */
    uint8_t* seg_written;    /* [rows of y][ceil(cout / 128)]: one bit per 16-channel block; every entry of a computed pixel is written */
    int32_t cls_c;           /* classes per anchor */
    float store_thr;         /* a segment is stored when one of its logits is > store_thr: pass logit(score_thr) minus a margin */
    /* LLV_CONV_MASK_POS: */
    const void* mask;
    int32_t mask_ld;
} llv_conv_desc;

#define LLV_TILE_AUTO 0
#define LLV_TILE_128x128 1 /* cout tile x pixel tile */
#define LLV_TILE_64x256 2
#define LLV_TILE_32x256 3
#define LLV_TILE_16x256 4
#define LLV_TILE_64x64 5
#define LLV_TILE_128x128_K64 7 /* 128x128 tile, 64-wide k-steps (128-byte rows = full L2 lines per request) */
#define LLV_TILE_64x256_K64 8
#define LLV_TILE_64x64_K64 14 /* 64x64 tile, 64-wide k-steps, four-stage ring (single-image maps: half the barriers per MFMA) */
#define LLV_TILE_TAP128 9 /* 3x3/s1/p1 tap-streamed kernel: 8x32-pixel tile x 128 channels, halo resident in LDS */
#define LLV_TILE_TAP64_R4 11 /* tap kernel, 64 channels, weight ring of 4 (slices requested 3 steps ahead; 2 workgroups per CU) */
#define LLV_TILE_TAP64 10  /* same, 64 output channels per workgroup */
#define LLV_TILE_TAP512 13 /* tap kernel on a 16x32-pixel tile x 64 channels, a whole 32-channel chunk (halo + nine weight slices) per load phase,
                            * two workgroups per CU; bias [+ ReLU] -> fp16 epilogue and one source only (LLV_E_UNSUPPORTED otherwise) */
#define LLV_TILE_TAP64_G3 12 /* TAP64 with the grouped main loop (one barrier per tap row, weights two groups ahead; A/B: not faster) */
#define LLV_TILE_128x128_REGSTAGE 6 /* 128x128 with register-staged loads (the round's first kernel; kept for A/B) */

int llv_abi_version(void);
const char* llv_strerror(int code);

int llv_conv2d_nhwc_fwd(const llv_conv_desc* desc, void* stream);
/* bytes of workspace the launch described by desc needs (split-K partial sums); 0 when it runs unsplit */

/* Which kernel llv_conv2d_nhwc_fwd will run for this descriptor: *tile = the LLV_TILE_* id the planner picks
 * (-1: the direct halo kernel selected by w_halo), *k_splits = the split-K factor. Diagnostic / profiling aid:
 * the reference has no counterpart (cuDNN picks its algorithm behind torch.nn.functional.conv2d). */
int llv_conv2d_plan(const llv_conv_desc* d, int* tile, int* k_splits);
size_t llv_conv2d_workspace_bytes(const llv_conv_desc* desc);
/* algorithmic FLOPs (2*MACs, true channels) the launch described by desc performs */
double llv_conv2d_flops(const llv_conv_desc* desc);

/* need = 1 + highest byte offset the launch llv_conv2d_nhwc_fwd(d) would touch through each pointer (0: pointer unused);
 * pure host arithmetic, no launch. Returns LLV_OK, or the error llv_conv2d_nhwc_fwd would return for d. */
int llv_conv2d_extents(const llv_conv_desc* d, llv_extents* need);
/* LLV_OK when need fits have for every declared (non-zero) field of have and need->below == 0, else LLV_E_EXTENT */
int llv_extents_fit(const llv_extents* need, const llv_extents* have);

/*
 * FastDVDnet InputCvBlock in one launch (mmtracking/mmtrack/models/cleaners/FastDVDnet.py:22-36):
 *   Conv2d(9, 90, 3, groups=3) + BN + ReLU -> Conv2d(90, 32, 3) + BN + ReLU; the 90-channel intermediate stays in LDS.
 * conv  : the SECOND conv as an llv_conv_desc: x[0..2] = the three NHWC8 frame tensors (n_src = 3, x_ld >= 8),
 *         cin = 96 (three groups of 30 + 2 zero channels), cout <= 32 (multiple of 8), w_halo = its halo packing
 *         ([3 chunks][32 rows][296]), bias, y, flags in {0, LLV_CONV_RELU}, one segment, 3x3 / stride 1 / pad 1.
 * w_pre : the FIRST (grouped) conv, halo packing [3 groups][32 rows][104] fp16 (k = tap*8 + c, rows 30,31 zero).
 * bias_pre : fp32 [96] (folded BN shift, 0 for the pad channels).
 */
typedef struct llv_inputcv_desc {
    llv_conv_desc conv;
    const void* w_pre;
    const float* bias_pre;
} llv_inputcv_desc;
int llv_inputcvblock_fwd(const llv_inputcv_desc* desc, void* stream);
/* FastDVDnet OutputCvBlock in one launch (FastDVDnet.py:64-74): Conv2d(32,32,3)+BN+ReLU -> Conv2d(32,3,3), with the
 * DenBlock residual `in1 - x` (FastDVDnet.py:115) as LLV_CONV_RES_RSUB. conv = the second conv (x[0] = the 32-channel
 * input tensor, cin = 32, cout <= 16 multiple of 8, w_halo [16 rows][296], res = in1); w_pre = first conv, halo packing
 * [32 rows][296]; bias_pre fp32 [32]. */
int llv_outputcvblock_fwd(const llv_inputcv_desc* desc, void* stream);
/* llv_conv2d_extents for the two fused blocks (output_block = 0: InputCvBlock, 1: OutputCvBlock); conv.extents declares w_pre / bias_pre too */
int llv_convpair_extents(const llv_inputcv_desc* desc, int output_block, llv_extents* need);

/* 3x3 stride-2 pad-1 max pooling, NHWC fp16, c multiple of 8. */
int llv_maxpool3x3s2_nhwc(const void* x, void* y, int batch, int h, int w, int c, void* stream);
/* y (n, ceil(h/2), ceil(w/2), y_ld) = the even pixels of x (n, h, w, x_ld): what a 1x1 / stride-2 convolution reads, as a dense map. */
int llv_subsample2_nhwc(const void* x, void* y, int n, int h, int w, int c, int x_ld, int y_ld, void* stream);
/* Its backward pass (training with an unfrozen stem, frozen_stages = -1 as in the authors' configs): dx += dy routed to each window's
 * first maximum in scan order (ATen's tie rule); x = the pooling input, *_ld = channels per pixel of each buffer. Gather form, no atomics. */
int llv_maxpool3x3s2_nhwc_bwd(const void* x, const void* dy, void* dx, int batch, int h, int w, int c, int x_ld, int dy_ld, int dx_ld, void* stream);

/* NCHW fp32 -> NHWC fp16 with c_pad >= c channels per pixel (extra channels zeroed).
 * Optional per-channel affine (x*scale[c] + shift[c]) applied first when scale != NULL. */
int llv_pack_nchw_f32_to_nhwc_f16(const float* x, void* y, int batch, int c, int h, int w, int c_pad,
                                  const float* scale, const float* shift, void* stream);
/* NHWC fp16 (ld channels per pixel) -> NCHW fp32, first c channels. */
int llv_unpack_nhwc_f16_to_nchw_f32(const void* x, float* y, int batch, int c, int h, int w, int ld,
                                    void* stream);
/* NHWC fp32 (ld channels per pixel) -> NCHW fp32, first c channels. */
int llv_unpack_nhwc_f32_to_nchw_f32(const float* x, float* y, int batch, int c, int h, int w, int ld,
                                    void* stream);

/*
 * RetinaNet box decode for one image: per level sigmoid -> max over classes -> top-k
 * (score desc, anchor index asc) -> gather -> delta2bbox -> clip -> /scale_factor.
 */
typedef struct llv_decode_desc {
    int32_t n_levels;
    int32_t num_classes;   /* C */
    int32_t num_anchors;   /* A (base anchors per cell) */
    int32_t nms_pre;       /* <=0: keep all */
    int32_t level_h[LLV_MAX_SEG], level_w[LLV_MAX_SEG];
    int32_t stride_w[LLV_MAX_SEG], stride_h[LLV_MAX_SEG];
    int64_t cls_off[LLV_MAX_SEG]; /* pixel offset of the level inside cls / reg buffers */
    int64_t reg_off[LLV_MAX_SEG];
    int32_t cls_ld, reg_ld;       /* channels per pixel (>= A*C, >= A*4) */
    int32_t logits_f32;           /* 1: cls/reg are fp32, 0: fp16 */
    const void* cls;              /* NHWC logits, channel = a*C + c */
    const void* reg;              /* NHWC deltas, channel = a*4 + k */
    const float* base_anchors;    /* [n_levels][A][4] fp32 (anchor_generator.py:142-185) */
    float means[4], stds[4];
    float max_ratio;              /* |log(wh_ratio_clip)| */
    int32_t clip;                 /* clip_border && max_shape given */
    float img_h, img_w;           /* max_shape */
    float scale_factor[4];        /* divide when rescale */
    int32_t rescale;
    /* outputs */
    float* boxes;                 /* [n_cand][4] */
    float* scores;                /* [n_cand][C] */
    int32_t* cand_anchor;         /* [n_cand] global anchor index (level offset + cell*A + a) */
    int32_t n_cand;               /* sum over levels of min(nms_pre, h*w*A), filled by llv_retina_decode_ncand */
    void* workspace; size_t workspace_bytes;
    /* batch > 1: images n = 0..batch-1 in one set of launches. Level l of image n starts n*level_h*level_w pixels
     * after cls_off[l] / reg_off[l] (the pyramid layout of engine.Pyramid); outputs are [batch][n_cand][...];
     * img_h_b / img_w_b / scale_factor_b replace img_h / img_w / scale_factor. batch <= LLV_MAX_BATCH. 0 means 1. */
    int32_t batch;
    float img_h_b[LLV_MAX_BATCH], img_w_b[LLV_MAX_BATCH];
    float scale_factor_b[LLV_MAX_BATCH][4];
    /* cls written by a LLV_CONV_CLSMAX launch (both NULL otherwise): clsmax [rows][A] holds every anchor's maximum logit (read instead of
     * the logits for the per-anchor score, and reset to zero here); seg_written [rows][seg_tiles] says, one bit per 16-channel block,
     * which parts of a pixel's logits were stored -- a candidate's classes in a block that was not are read as score 0 (all of them were
     * <= store_thr, i.e. below score_thr: the NMS that follows drops them either way). logits_f32 must be 1. */
    uint32_t* clsmax;
    const uint8_t* seg_written;
    int32_t seg_tiles;
} llv_decode_desc;

int llv_retina_decode_ncand(const llv_decode_desc* desc);
size_t llv_retina_decode_workspace_bytes(const llv_decode_desc* desc);
int llv_retina_decode_topk(const llv_decode_desc* desc, void* stream);

/*
 * multiclass_nms for one image. boxes [n][4] fp32, scores [n][c_ld] fp32 (first C columns used). n <= 16384 and
 * num_classes * max_num <= 8192 (LLV_E_UNSUPPORTED beyond: the class lists are sorted and merged in LDS).
 * Output: dets [max_num][5], labels [max_num] int64, count[1] int32, keep_flat [max_num] int32
 * (flat index i*C + c of each kept detection, score-descending).
 */
size_t llv_multiclass_nms_workspace_bytes(int n, int num_classes, int max_num);
int llv_multiclass_nms(const float* boxes, const float* scores, int n, int num_classes, int c_ld,
                       float score_thr, float iou_thr, int max_num, float* dets, int64_t* labels,
                       int32_t* count, int32_t* keep_flat, void* workspace, size_t workspace_bytes,
                       void* stream);

/* The same for `batch` images in one set of launches: boxes [batch][n][4], scores [batch][n][c_ld],
 * dets [batch][max_num][5], labels / keep_flat [batch][max_num], count [batch]. */
size_t llv_multiclass_nms_batched_workspace_bytes(int batch, int n, int num_classes, int max_num);
/* LDS one class list needs (candidate keys padded to a power of two + max_num kept boxes) and the limit a launch may use:
 * a geometry above the limit is refused with LLV_E_UNSUPPORTED before anything is launched (a failed launch would poison a
 * stream capture). n <= 8192 always fits; 8192 < n <= 16384 fits for max_num <= 1536.
 * _ex also takes num_classes: a single-class list of >= 2048 boxes (an RPN level: mmdetection/mmdet/models/dense_heads/rpn_head.py:88-234
 * with nms_pre = 12000, max_per_img = 2000) runs on the sort + bit-mask kernels, which hold only the keys in LDS, whatever max_num. */
size_t llv_multiclass_nms_lds_bytes(int n, int max_num);
size_t llv_multiclass_nms_lds_bytes_ex(int n, int max_num, int num_classes);
size_t llv_multiclass_nms_lds_limit(void);
int llv_multiclass_nms_batched(const float* boxes, const float* scores, int batch, int n, int num_classes, int c_ld,
                               float score_thr, float iou_thr, int max_num, float* dets, int64_t* labels,
                               int32_t* count, int32_t* keep_flat, void* workspace, size_t workspace_bytes,
                               void* stream);

/* Plain single-class NMS: keep [n] int64 (score-descending kept indices), count[1] int32. n <= 16384. */
size_t llv_nms_workspace_bytes(int n);
int llv_nms(const float* boxes, const float* scores, int n, float iou_thr, int64_t* keep, int32_t* count,
            void* workspace, size_t workspace_bytes, void* stream);

/*
 * EmbedAggregator tail. x_embed [hw][c], ref_embed [n][hw][c] (fp16 NHWC, ld channels per pixel),
 * ref [n][hw][c] fp16 -> out [hw][c] fp16.
 */
int llv_embed_aggregate(const void* x_embed, const void* ref_embed, const void* ref, void* out, int n_ref,
                        int64_t hw, int c, int embed_ld, int ref_ld, int out_ld, void* stream);

/* Ring-buffer form: embed / feat hold n_slots frames [slot][hw][c]; the key frame is slot key_slot and is excluded
 * from the references (mmtracking/mmtrack/models/vid/llvod.py:201-206). */
int llv_embed_aggregate_ring(const void* embed, const void* feat, void* out, int n_slots, int key_slot, int64_t hw,
                             int c, int embed_ld, int feat_ld, int out_ld, void* stream);

/* Backward of the two temporal ops (training of the flow / aggregation video detectors, mmtracking/mmtrack/models/vid/dff.py:49-182,
 * fgfa.py:52-194, llvod.py:45-154), what torch autograd derives for embed_aggregator.py:70-81 and flow.py:4-41.
 * llv_embed_aggregate_bwd (ring form: embed / feat [n_slots][hw][c], the key frame is slot key_slot): grad_out [hw][c] fp16;
 *   d_embed [n_slots][hw][c] and d_feat [n_slots][hw][c] are fp16 buffers the gradients are ADDED to (slot key_slot of d_feat is
 *   not touched: the key frame's features reach the output only through its embedding).
 * llv_flow_warp_bwd: gradient with respect to the warped features only (the flow network is frozen on this path): dy fp16 NHWC
 *   (n,h,w,c) with ld, flow as in llv_flow_warp, dx fp32 [n][h][w][c] accumulated with atomics (zero it first);
 *   llv_add_f32_to_f16 then adds it into an fp16 gradient buffer [rows][ld]. */
int llv_embed_aggregate_bwd(const void* embed, const void* feat, const void* grad_out, void* d_embed, void* d_feat, int n_slots,
                            int key_slot, int64_t hw, int c, int embed_ld, int feat_ld, int gout_ld, void* stream);
int llv_flow_warp_bwd(const void* dy, const float* flow, float* dx, int n, int h, int w, int c, int ld, int hf, int wf, void* stream);
/* The OTHER gradient of the warp: d(loss)/d(flow) (n, 2, hf, wf) fp32, ADDED (grid gradient of grid_sample -- zero where the sample was
 * clamped to the border -- through grid = (x + u) / W * 2 - 1 and the bilinear resize of the flow field; flow.py:4-41). dy (n, h, w, dy_ld)
 * fp16 = gradient of the warped map, x (n, h, w, x_ld) fp16 = the map that was warped. */
int llv_flow_warp_bwd_flow(const void* dy, const void* x, const float* flow, float* dflow, int n, int h, int w, int c, int dy_ld, int x_ld,
                           int hf, int wf, void* stream);
/* Backward of llv_bilinear_resize_f32 (same arguments): dx (nc, h, w) += resize^T (dy (nc, ho, wo)) * mul. */
int llv_bilinear_resize_f32_bwd(const float* dy, float* dx, int nc, int h, int w, int ho, int wo, float scale_factor, float mul, void* stream);
int llv_add_f32_to_f16(const float* src, void* dst, int64_t rows, int c, int ld, void* stream);

/* FlowNetSimple.prepare_imgs (flownet_simple.py:149-190) for image pairs: x (n,6,h,w) NCHW fp32 normalised images ->
 * y (n, ho, wo, 8) NHWC fp16: v = x*scale[c] + shift[c] (de-normalise, /255, -flow mean folded by the caller), zero for
 * rows >= img_h / cols >= img_w, bilinear resize (align_corners=False) by scale_factor = wo/w; channels 6,7 zero. */
int llv_flow_prepare(const float* x, void* y, int n, int h, int w, int ho, int wo, int img_h, int img_w,
                     const float* scale, const float* shift, float resize_scale, void* stream);
/* F.interpolate(x, scale_factor=s, mode='bilinear', align_corners=False) * mul on NCHW fp32 (flownet_simple.py:236-242). */
int llv_bilinear_resize_f32(const float* x, float* y, int nc, int h, int w, int ho, int wo, float scale_factor, float mul,
                            void* stream);

/* flow_warp_feats: x NHWC fp16 (n,h,w,c), flow NCHW fp32 (n,2,hf,wf) -> y NHWC fp16. */
int llv_flow_warp(const void* x, const float* flow, void* y, int n, int h, int w, int c, int ld, int hf,
                  int wf, void* stream);

/* sigmoid focal loss (mmcv semantics): logits [n][c] fp32, labels [n] int64 (c == background).
 * fwd writes loss [n][c]; bwd writes grad_input [n][c] = dloss/dlogit (unweighted). */
int llv_sigmoid_focal_loss_fwd(const float* logits, const int64_t* labels, float* loss, int64_t n, int c,
                               float gamma, float alpha, void* stream);
int llv_sigmoid_focal_loss_bwd(const float* logits, const int64_t* labels, float* grad, int64_t n, int c,
                               float gamma, float alpha, void* stream);

/* softmax cross entropy of the two-stage heads (RPN with use_sigmoid=False, BBoxHead.loss_cls): cross_entropy
 * (mmdetection/mmdet/models/losses/cross_entropy_loss.py:9-40) = F.cross_entropy(pred, label, weight=class_weight, reduction='none').
 * logits [n][ld] fp32 (c <= ld columns used), labels [n] int64 (outside [0, c): ignored, loss and gradient 0), class_weight [c]
 * or NULL. fwd writes loss [n]; bwd writes grad [n][ld] (columns < c) = grad_loss[r] * dloss[r]/dlogit. */
int llv_softmax_ce_fwd(const float* logits, const int64_t* labels, const float* class_weight, float* loss, int64_t n, int c,
                       int ld, void* stream);
int llv_softmax_ce_bwd(const float* logits, const int64_t* labels, const float* class_weight, const float* grad_loss,
                       float* grad, int64_t n, int c, int ld, void* stream);

/* MaxIoUAssigner on device. gt [k][4], anchors [n][4], valid [n] uint8 (may be NULL) ->
 * assigned_gt_inds [n] int64 (-1 ignore / 0 negative / 1-based gt), max_overlaps [n] fp32. */
size_t llv_iou_assign_workspace_bytes(int k, int64_t n);
int llv_iou_assign(const float* gt, int k, const float* anchors, const uint8_t* valid, int64_t n,
                   float pos_iou_thr, float neg_iou_thr, float min_pos_iou, int match_low_quality,
                   int gt_max_assign_all, int64_t* assigned_gt_inds, float* max_overlaps, void* workspace,
                   size_t workspace_bytes, void* stream);

/* The dense head's loss in one pass (AnchorHead.loss with FocalLoss + L1Loss / SmoothL1Loss and PseudoSampler:
 * mmdetection/mmdet/models/dense_heads/anchor_head.py:175-271,375-493, losses/focal_loss.py:160-181, smooth_l1_loss.py:8-45,
 * core/bbox/coder/delta_xywh_bbox_coder.py:88-130). cls / reg: the fp32 logit / delta buffers of all levels, pixel-major (level l of
 * image n starts at row level_off[l] + n * level_h[l] * level_w[l]; anchor a of a pixel owns cls columns [a*C, (a+1)*C) and reg
 * columns [4a, 4a+4)). gt_inds (batch, anchors_per_image): llv_iou_assign's output per image in mmdet's anchor order (level-major,
 * then y, x, a; level_anchor0[l] = first anchor of level l). Outputs: losses[0..L) = loss_cls per level, losses[L..2L) = loss_bbox per
 * level (already * loss_weight / avg_factor); dcls / dreg (fp16, written, not added) = d(sum of all terms)/d(logit) * grad_scale;
 * avg_factor (device scalar) = sum over images of max(#positives, 1). sums: 2L floats of scratch, zero on the first call. beta = 0: L1, > 0: smooth L1. */
typedef struct llv_retina_loss_desc {
    const float* cls; const float* reg;
    void* dcls; void* dreg;
    int cls_ld, reg_ld, dcls_ld, dreg_ld;
    int batch, n_levels, num_anchors, num_classes;
    int level_h[LLV_MAX_SEG], level_w[LLV_MAX_SEG];
    int64_t level_off[LLV_MAX_SEG], level_anchor0[LLV_MAX_SEG];
    int64_t anchors_per_image;
    const float* anchors;
    const int64_t* gt_inds;
    const float* gt_boxes; const int64_t* gt_labels; int max_gt;
    float gamma, alpha, cls_weight, reg_weight, beta, pos_weight, grad_scale;
    float means[4], stds[4];
    float* sums; float* avg_factor; float* losses;
    int* counts;     /* `batch` ints of scratch, ZERO on the first call (every call leaves them and `sums` zero again) */
    float* partials; /* optional scratch of partials_cap floats, partials_cap >= 2 * n_levels * LLV_RETINA_LOSS_MAX_GROUPS: the workgroups'
                      * partial sums go to slots of their own and are added in a fixed order (deterministic loss values; with NULL they are
                      * added to `sums` by fp32 atomics, ~1,800 to the same address for a 736 x 1280 image: 85 us of serialised atomics) */
    int32_t partials_cap;
} llv_retina_loss_desc;
#define LLV_RETINA_LOSS_MAX_GROUPS 2048
int llv_retina_loss(const llv_retina_loss_desc* desc, void* stream);

/* Modulated deformable convolution, column step (ModulatedDCNPack.forward, denoising_aggregator.py:103-111 ->
 * mmcv.ops.modulated_deform_conv2d): x (n,h,w,x_ld) NHWC fp16 (c channels used, c % 8 == 0, (c/deform_groups) % 8 == 0);
 * om (n,ho,wo,om_ld) fp32 = the raw conv_offset output: channels [0, 2*dg*K) offsets, (g*K + k)*2 = dy, +1 = dx; channels
 * [2*dg*K, 3*dg*K) mask logits (sigmoid applied here), K = kh*kw, k = i*kw + j.
 * col (n,ho,wo,col_ld) fp16, channel k*c + ch = mask * bilinear(x[ch]; y*stride - pad + i*dilation + dy, ...) with mmcv's
 * border rule (zero outside (-1, size); corners outside the image contribute zero). The convolution itself is then
 * llv_conv2d_nhwc_fwd as a 1x1 conv over col with the kh x kw weight packed tap-major. */
int llv_modulated_deform_im2col(const void* x, const float* om, void* col, int n, int h, int w, int c, int x_ld, int om_ld,
                                int col_ld, int kh, int kw, int stride, int pad, int dilation, int deform_groups, int ho,
                                int wo, void* stream);
/* Its backward (mmcv modulated_deformable_col2im + col2im_coord, the gradient path of ModulatedDCNPack in training): from the column
 * gradient dcol (n,ho,wo,col_ld) fp16, ACCUMULATE into dx (n,h,w,c) fp32 the input gradient and into dom (n,ho,wo,om_ld) fp32 the
 * gradient of the raw conv_offset output: offsets (dy, dx of every tap) and mask LOGITS (the sigmoid's derivative included). Same
 * geometry arguments and border rule as the forward; both outputs must be zeroed (or hold what to add to) by the caller. */
int llv_modulated_deform_col2im(const void* dcol, const void* x, const float* om, float* dx, float* dom, int n, int h, int w, int c,
                                int x_ld, int om_ld, int col_ld, int kh, int kw, int stride, int pad, int dilation, int deform_groups,
                                int ho, int wo, void* stream);

/* Element-wise steps of TemporalAttentionFusion.forward (denoising_aggregator.py:135-152) on contiguous NHWC fp16 tensors with
 * m channels (m % 8 == 0). The reference's loop over the reference frame i is batched: pair (i, j) = image i*n + j.
 *   llv_taf_pair_concat   x (n,h,w,m) -> y (n*n,h,w,2m): y[i*n+j] = cat(x[j], x[i])           (:141-143)
 *   llv_taf_mul_ref       xd (n*n,h,w,m), x (n,h,w,m) -> y (n*n,h,w,m): y[i*n+j] = xd[i*n+j] * x[i]   (:146)
 *   llv_taf_softmax_fuse  xcor (n*n,h,w,m), x (n,h,w,m) -> out (n,h,w,m): out[i] = sum_j softmax_j(xcor[i*n+j]) * x[j]   (:147-148)
 *   llv_add_f16           y = a + b over `rows` rows of c halfs with row strides a_ld / b_ld / y_ld (all % 8 == 0)   (:193,197) */
int llv_taf_pair_concat(const void* x, void* y, int n, int h, int w, int m, void* stream);
int llv_taf_mul_ref(const void* xd, const void* x, void* y, int n, int h, int w, int m, void* stream);
int llv_taf_softmax_fuse(const void* xcor, const void* x, void* out, int n, int h, int w, int m, void* stream);
int llv_add_f16(const void* a, const void* b, void* y, int64_t rows, int c, int a_ld, int b_ld, int y_ld, void* stream);

/* RoIAlign (mmcv.ops.roi_align, pool_mode 'avg'; mmcv-full 1.3.0 un-vendored; call site: the `roi_layer=dict(type='RoIAlign',
 * output_size=7, sampling_ratio=2)` of mmtracking/configs/vid/llvod/ (every .py there) through mmdet's SingleRoIExtractor) -- the first piece of the
 * two-stage path of SURVEY.md section 8f item 4. feat (n,h,w,ld) NHWC fp16 (c channels, c % 8 == 0); rois [k][5] fp32 = (batch index,
 * x1, y1, x2, y2); out (k,ph,pw,c) NHWC fp16, or fp32 with out_f32. sampling_ratio 0 = adaptive (ceil(bin size)). */
int llv_roi_align(const void* feat, const float* rois, void* out, int n, int h, int w, int c, int ld, int k, int ph, int pw,
                  float spatial_scale, int sampling_ratio, int aligned, int out_f32, void* stream);
/* Its backward with respect to the feature map (two-stage training, selsa_roi_head.py:94-111 -> mmcv roi_align backward): grad_out
 * (k,ph,pw,c) fp32 NHWC -> dfeat (n,h,w,c) fp32, accumulated with atomics (the caller zeroes it). */
int llv_roi_align_bwd(const float* grad_out, const float* rois, float* dfeat, int n, int h, int w, int c, int k, int ph, int pw,
                      float spatial_scale, int sampling_ratio, int aligned, void* stream);

/* ---- SELSA RoI head: selsa_aggregator.py:29-73, selsa_bbox_head.py:60-84, bbox_head.py:270-371.
 * The matrix products run on llv_conv2d_nhwc_fwd (fully connected layer = 1x1 / 7x7-valid convolution over the RoI batch; the attention
 * of an aggregator = two grouped 1x1 convolutions, one group per attention block, whose weights are the packed reference rows).
 * llv_selsa_pack_kv: rows [row0, row0 + rows) of the packed weights from src (rows, blocks*d) fp16 (row stride src_ld):
 *   transposed == 0: dst [blocks][dst_rows][dst_kpad], dst[g][row0 + r][k] = src[r][g*d + k]   (scores = x_embed . ref_embed^T)
 *   transposed != 0: dst [blocks][dst_rows][dst_kpad], dst[g][k][row0 + r] = src[r][g*d + k]   (x_new = weights . ref_new)
 * llv_selsa_softmax: scores fp32 [n_rows][blocks][m_ld] -> weights fp16, same shape: softmax(scale * s) over the valid reference rows
 *   [0, m_static + min(*live_count, m_live_max)) (live_count NULL: all m_live_max), 0 elsewhere.
 * llv_softmax_rows_masked: y[r] = softmax(x[r][0..c)) for r < *count (count NULL: every row), 0 for the padding rows behind.
 * llv_dets_to_rois: rois[i] = (batch_index, dets[i][0..3]) for i < *count, zeros behind (max_rows rows); boxes4 (may be NULL) gets the
 *   same boxes as [max_rows][4] (the `rois[:, 1:]` of bbox_head.py:340 for llv_delta2bbox). */
int llv_selsa_pack_kv(const void* src, void* dst, int rows, int row0, int blocks, int d, int src_ld, int dst_rows, int dst_kpad, int transposed,
                      void* stream);
int llv_selsa_softmax(const float* scores, void* weights, int64_t n_rows, int blocks, int m_ld, int m_static, int m_live_max,
                      const int32_t* live_count, float scale, void* stream);
int llv_softmax_rows_masked(const float* x, float* y, int n, int c, int x_ld, int y_ld, const int32_t* count, void* stream);
int llv_dets_to_rois(const float* dets, const int32_t* count, float* rois, float* boxes4, int max_rows, int batch_index, void* stream);

/* ---- TemporalRoIAlign (the RoI extractor of the authors' own model: mmtracking/configs/vid/llvod/llvod_fastdvdnet_darkfarm.py:49-59,
 * mmtracking/mmtrack/models/roi_heads/roi_extractors/temporal_roi_align.py). Its two dense products run on llv_conv2d_nhwc_fwd (the
 * cosine similarity maps as a 1x1 convolution of the normalised reference maps with the normalised RoI points as weights, fp32 out;
 * the 3x3 embed convolution over the 7x7 RoI maps); these are the remaining steps.
 *   llv_l2norm_rows_f16     y[r] = x[r] / ||x[r]||_2 over c channels (:123-126); x fp32 (x_is_f32) or fp16, y fp16; row strides in
 *                           elements; inv_norm (optional) receives 1 / ||x[r]||.
 *   llv_msra_top2_gather    most_similar_roi_align :141-168 with num_most_similar_points = 2: sim [img_n*hw][sim_ld] fp32 (row = pixel p
 *                           of reference image i, column = RoI point r), ref [img_n*hw][ref_ld] fp16 (un-normalised maps) ->
 *                           out [img_n][rows][c] fp16 = softmax-weighted sum of the two most similar pixels' features; idx_out / w_out
 *                           (optional, [img_n][rows][2]) receive the picks and weights. Ties: lower pixel index first.
 *   llv_troi_attention_fuse temporal_attentional_feature_aggregation :76-96 behind the embed conv: x, emb [n_img][rows][c] fp16 (image 0
 *                           = the key frame's RoI features), out [rows][c] fp32; `blocks` attention blocks (c / blocks channels each,
 *                           a power-of-two multiple of 8), n_img <= 16. */
int llv_l2norm_rows_f16(const void* x, int x_is_f32, void* y, int64_t rows, int c, int64_t x_ld, int64_t y_ld, float* inv_norm, void* stream);
int llv_msra_top2_gather(const float* sim, const void* ref, void* out, int32_t* idx_out, float* w_out, int img_n, int hw, int rows,
                         int sim_ld, int c, int ref_ld, void* stream);
int llv_troi_attention_fuse(const void* x, const void* emb, float* out, int n_img, int64_t rows, int c, int blocks, void* stream);

/* DeltaXYWHBBoxCoder.decode = delta2bbox (mmdet/core/bbox/coder/delta_xywh_bbox_coder.py:134-237) for (n, 4) boxes and (n, 4*k) deltas
 * (k = 1: class-agnostic, k = #classes: the class-specific decode of BBoxHead.get_bboxes, bbox_head.py:337-340): out (n, 4*k) fp32.
 * means / stds: 4 host floats each; clip != 0: clamp to [0, img_w] x [0, img_h] (max_shape); scale_factor (4 host floats) or NULL:
 * divide the decoded corners by it (rescale=True). Same fp32 operation order as the reference (no FMA contraction). */
int llv_delta2bbox(const float* rois, const float* deltas, float* out, int64_t n, int k, const float* means, const float* stds,
                   float wh_ratio_clip, float img_h, float img_w, int clip, const float* scale_factor, void* stream);

/* One whole stride-1 Bottleneck (mmdet/models/backbones/resnet.py:260-299 with no downsample, dilation 1, plugins off; eval-mode
 * BatchNorm folded into the weights as llv_conv2d_nhwc_fwd's callers do): y = relu(conv3(relu(conv2(relu(conv1(x))))) + x) in a single
 * launch that keeps both narrow intermediates in LDS. x, y: (batch, h, w, 4*mid) NHWC fp16 with row strides x_ld / y_ld (elements,
 * multiples of 8); x and y must not alias. w1 [mid][4*mid], w2 [mid][9*mid] (k = (ky*3 + kx)*mid + c), w3 [4*mid][mid] fp16;
 * b1, b2 [mid], b3 [4*mid] fp32. mid = 64 or 128 (ResNet-50/101/152 layer1 and layer2); LLV_E_UNSUPPORTED otherwise. */
typedef struct llv_bottleneck_desc {
    const void* x;
    void* y;
    const void* w1;
    const void* w2;
    const void* w3;
    const float* b1;
    const float* b2;
    const float* b3;
    int32_t batch, h, w, mid;
    int32_t x_ld, y_ld;
} llv_bottleneck_desc;
int llv_bottleneck_fwd(const llv_bottleneck_desc* desc, void* stream);

/* True/false-positive marking of eval_map (mean_ap.py:268-400 calls tpfp_default :154-244 or, for datasets 'det'/'vid',
 * tpfp_imagenet :60-151 once per class and image): all (class, image) cells in one launch, one wavefront per cell.
 * dets [total_dets][5] fp32 (x1,y1,x2,y2,score); cell c owns dets[det_off[c] .. det_off[c+1]) and gts[gt_off[c] .. gt_off[c+1])
 * (gts [total_gts][4] fp32: the cell's `bboxes` followed by its `bboxes_ignore`, gt_ignore[j] = 1 for the latter; may be NULL).
 * area_ranges [n_ranges][2] double (min_area, max_area), n_ranges = 0: no area limits. mode 0 = default, 1 = imagenet.
 * tp, fp: [max(n_ranges,1)][total_dets] uint8. Score ties are taken lower index first. At most 2048 detections and 4096
 * ground-truth boxes per cell (LLV_E_UNSUPPORTED beyond; the caller passes the maxima of its offsets). */
int llv_eval_tpfp(const float* dets, const int* det_off, const float* gts, const int* gt_off, const uint8_t* gt_ignore,
                  int n_cells, int max_det_per_cell, int max_gt_per_cell, const double* area_ranges, int n_ranges,
                  float iou_thr, int mode, int64_t total_dets, uint8_t* tp, uint8_t* fp, void* stream);

/* COCO-protocol matching (CocoVideoDataset.evaluate, mmtracking/mmtrack/datasets/coco_video_dataset.py:374-461 ->
 * mmdet CocoDataset.evaluate -> pycocotools COCOeval.evaluateImg; pycocotools is un-vendored): one wavefront per (category, image)
 * cell, one lane per (area range, IoU threshold) pair (n_ranges * n_thrs <= 64). dets [total_dets][4] double (x, y, w, h), per cell
 * already in score-descending order and truncated to the largest maxDets; gts [total_gts][4] double with gt_area (the annotation's
 * `area`), gt_crowd (`iscrowd`) and gt_ignore flags; area_ranges [n_ranges][2], iou_thrs [n_thrs] double.
 * Out: dt_matched, dt_ignore [n_ranges][n_thrs][total_dets] uint8; gt_valid [n_ranges][n_cells] int32 = ground truths of the cell
 * that count for the range. At most 256 ground truths per cell. */
int llv_coco_match(const double* dets, const int* det_off, const double* gts, const double* gt_area, const uint8_t* gt_crowd,
                   const uint8_t* gt_ignore, const int* gt_off, int n_cells, int max_gt_per_cell, const double* area_ranges,
                   int n_ranges, const double* iou_thrs, int n_thrs, int64_t total_dets, uint8_t* dt_matched, uint8_t* dt_ignore,
                   int32_t* gt_valid, void* stream);

/* Stream-ordered copy done by a KERNEL (a launch-plan step: the streaming FastDVDnet plan carries the last two first-stage
 * DenBlock outputs of a step to the front of the next step's buffer). Regions must not overlap; dst may be device-mapped host
 * memory. Inside a captured graph it is a kernel node, not a memcpy node. */
int llv_copy_d2d(void* dst, const void* src, size_t bytes, void* stream);

/* One frame of the ConvLSTM inside DarkResNet's DenoisingBlock (mmtracking/mmtrack/models/backbones/dark_resnet.py:226-246,267-269):
 * gates (rows, gates_ld >= 4p) fp16 = conv2([x_t | h_{t-1}]) with the channels in i | f | o | g order, every gate through a sigmoid;
 * c (rows, p) fp32 is the cell state, updated in place (c = f*c + i*g); h (rows, h_ld) fp16 = o * tanh(c) is the next frame's recurrent
 * input; act (rows, act_ld) fp16 = relu(h * scale + shift) is the block's norm2 + ReLU (eval-mode BatchNorm as a per-channel affine). */
int llv_convlstm_gate(const void* gates, float* c, const float* scale, const float* shift, void* h, void* act, int64_t rows, int p,
                      int gates_ld, int h_ld, int act_ld, void* stream);
/* Its backward for one frame: gates as above, c_prev (rows, p) fp32 the cell state BEFORE the frame; dh (rows, h_ld) fp16 the gradient
 * the recurrent consumer (the next frame's conv2) sent to h, or null; dact (rows, act_ld) fp16 the gradient of act; dc_in (rows, p) fp32
 * the gradient the next frame sent to this frame's new cell state, or null -> dgates (rows, gates_ld) fp16, dc_prev (rows, p) fp32 (may
 * alias dc_in), and -- unless null -- ADDS into dscale / dshift [p] fp32 the gradients of the affine (norm2's gamma / sqrt(var + eps), beta). */
int llv_convlstm_gate_bwd(const void* gates, const float* c_prev, const float* scale, const float* shift, const void* dh, const void* dact,
                          const float* dc_in, void* dgates, float* dc_prev, float* dscale, float* dshift, int64_t rows, int p,
                          int gates_ld, int h_ld, int act_ld, void* stream);

/* bbox2result's device-to-host step (mmdetection/mmdet/core/bbox/transforms.py:99-116) as one kernel: the NMS outputs of `batch`
 * images (dets [batch][max_num][5] fp32, labels [batch][max_num] int64, count [batch] int32) -> rows [batch][max_num][6] fp32
 * (x1, y1, x2, y2, score, label) and count_out [batch] int32; rows / count_out may be device-mapped pinned host memory, which
 * the host reads after synchronising the stream. */
int llv_pack_results(const float* dets, const int64_t* labels, const int32_t* count, int batch, int max_num, float* rows,
                     int32_t* count_out, void* stream);

/* Capacity guards of the post-processing kernels: every data-dependent index (list positions drawn from atomic counters, anchor /
 * box indices read back from memory) is clamped to its buffer, and clamps that fire are counted. *count_host = guards fired since
 * the last reset (0 with consistent inputs; > 0 means launches of a step ran on counters that were not zeroed or on lists that
 * changed between passes, which without the guards is an out-of-bounds store). Synchronises the device; never call while capturing. */
int llv_guard_trips(uint64_t* count_host, int reset);
/* The same count written by one kernel at this point of the stream: *dst (device-visible: device memory or device-mapped pinned host
 * memory) = guards fired since the last reset. For the end of a captured step, next to llv_pack_results; the first call must be
 * made outside a stream capture (it looks the counters' addresses up). */
int llv_guard_trips_to(uint32_t* dst, void* stream);

/* Test-time image pipeline of the reference in one pass:
 * SeqBrighten (mmtracking/mmtrack/datasets/pipelines/transforms.py:376-432: one gain amp = m*h*w*3 / sum(frame0/255) for the
 * clip, np.clip(img*amp, 0, 255).astype(uint8)), Normalize (mmdetection/mmdet/datasets/pipelines/transforms.py:547-575:
 * BGR->RGB, (x - mean) * (1/std)) and Pad (:476-524, zeros) -> fp16 NHWC8 (3 real + 5 zero channels).
 * frames: uint8 (t,h,w,3) on the device; out: fp16 (t,hp,wp,8); workspace: 8 bytes. The gain's denominator is summed exactly in
 * integers (the reference sums float32 pairwise: its gain can differ in the 7th digit). */
int llv_brighten_normalize_pad_u8(const void* frames, int t, int h, int w, int hp, int wp, float m, const float* mean3,
                                  const float* std3, int to_rgb, void* out, void* workspace, void* stream);

/* ---- YOLOv3 stand-in for config #5 ------------------------------------------------------------
 * F.interpolate(scale_factor=2) of yolo_neck.py:123 written into a channel slice of the concat buffer (torch.cat of :124):
 * dst[n][2y+dy][2x+dx][c] = src[n][y][x][c], fp16 NHWC, row strides src_ld / dst_ld, c multiple of 8. */
int llv_upsample2x_nhwc(const void* src, void* dst, int n, int h, int w, int c, int src_ld, int dst_ld, void* stream);
/* YOLOV3Head._get_bboxes per level before the top-k (yolo_head.py:262-279) + YOLOBBoxCoder.decode (yolo_bbox_coder.py:61-89)
 * + YOLOAnchorGenerator grid (anchor_generator.py:639-665): pred fp32 [n][h][w][ld] with channel a*num_attrib + attr ->
 * boxes [n][h*w*A][4] (xyxy), conf [n][h*w*A] = sigmoid(attr 4). anchors_wh: A (w, h) pairs (host floats). */
int llv_yolo_decode(const float* pred, int n, int h, int w, int num_anchors, int num_attrib, int ld, int stride,
                    const float* anchors_wh, float* boxes, float* conf, void* stream);

/* YOLOV3Head._get_bboxes between the per-anchor decode and the NMS (yolo_head.py:281-335) for one level of `batch` images: the nms_pre
 * anchors with the highest objectness (torch.topk order: best first; all n anchors in anchor order when n <= nms_pre or nms_pre <= 0),
 * their boxes (divided by inv_scale[img][0..4) when given: the rescale of :316-318), objectness and class scores sigmoid(logit) -- zero
 * for rows whose objectness is below conf_thr (:320-327) -- written to rows [out_row0, out_row0 + m) of the per-image candidate lists
 * out_boxes [batch][n_total][4], out_conf [batch][n_total], out_scores [batch][n_total][num_classes]. pred / boxes / conf as for
 * llv_yolo_decode. workspace: llv_yolo_select_workspace_bytes(batch, n). */
size_t llv_yolo_select_workspace_bytes(int batch, int n);
int llv_yolo_select(const float* pred, const float* boxes, const float* conf, int batch, int n, int num_anchors, int num_attrib,
                    int num_classes, int ld, int nms_pre, float conf_thr, const float* inv_scale, int64_t out_row0, int64_t n_total,
                    float* out_boxes, float* out_conf, float* out_scores, void* workspace, size_t workspace_bytes, void* stream);
/* llv_multiclass_nms_batched with multiclass_nms's score_factors (bbox_nms.py:55-62): the threshold test uses the raw score,
 * the ranking / reported score is score * score_factors[i] (YOLOv3: objectness). score_factors [batch][n], may be NULL. */
int llv_multiclass_nms_batched_sf(const float* boxes, const float* scores, const float* score_factors, int batch, int n,
                                  int num_classes, int c_ld, float score_thr, float iou_thr, int max_num, float* dets,
                                  int64_t* labels, int32_t* count, int32_t* keep_flat, void* workspace, size_t workspace_bytes,
                                  void* stream);
/* The same with class-specific boxes (multiclass_nms with multi_bboxes of shape (n, #class*4), bbox_nms.py:31-35: the two-stage bbox
 * heads with reg_class_agnostic=False): class_specific != 0 reads boxes as [batch][n][num_classes][4]. */
int llv_multiclass_nms_batched_ex(const float* boxes, const float* scores, const float* score_factors, int class_specific, int batch,
                                  int n, int num_classes, int c_ld, float score_thr, float iou_thr, int max_num, float* dets,
                                  int64_t* labels, int32_t* count, int32_t* keep_flat, void* workspace, size_t workspace_bytes,
                                  void* stream);

/* ---- backward pass of the convolution engine (llv_conv2d_nhwc_{dgrad,wgrad}) ----------------
 * Replaces torch autograd's cudnn_convolution_backward_{input,weight} under mmdet's ResNet / FPN / RetinaHead
 * (resnet.py:260-299, fpn.py:170-221, retina_head.py:93-114) in BaseDetector.train_step (base.py:220-253).
 *
 * wgrad: dw[g][co][tap*cin + ci] += sum over output pixels of dy[p][g*cout + co] * x[p @ tap][g*cin + ci]
 *   d      geometry and the INPUT tensor of the forward conv (x[0], x_ld, cin (padded), cout, w_rows, k_pad, kh, kw,
 *          stride, pad, groups, batch, segment 0: seg_h/w/ho/wo, seg_xoff = pixel offset of x, seg_yoff = of dy);
 *          one source; segments = the pyramid levels a shared filter ran on (3x3 / stride 1 / pad 1: one launch over the patches of
 *          all levels; other filters: one launch per segment)
 *   dy     fp16 NHWC gradient of the conv output (before the activation), row stride dy_ld
 *   dw     fp32, the packed weight layout [groups][w_rows][k_pad]; ACCUMULATED into (fp32 atomics): zero it first
 *   db     optional fp32 [groups*cout]: db[g*cout + co] += sum_p dy[p][g*cout + co] (the bias / BN-shift gradient,
 *          formed by one extra MFMA against a ones operand in the workgroups that already hold the dY tiles)
 * dgrad (below) is llv_conv2d_nhwc_fwd on re-packed weights (flipped taps, in/out transposed; stride 2 as a small
 * stride-1 window + PixelShuffle). */
int llv_conv2d_nhwc_wgrad(const llv_conv_desc* d, const void* dy, int dy_ld, float* dw, float* db, void* stream);

/* The weight gradients of SEVERAL convolutions as one launch per kernel kind. A single-image training step has ~80 weight-gradient
 * launches and most of them cover a fraction of the chip; side by side they share one launch ramp and need fewer pixel splits (each
 * split ends in fp32 atomics over its dW tile).
 *   llv_conv2d_nhwc_wgrad_plan   host only, no launch: the job(s) of one convolution (arguments as llv_conv2d_nhwc_wgrad) -- one job
 *                                for a 3x3 / stride-1 / pad-1 filter over all its segments (kind 2: all nine taps in a workgroup), one
 *                                per segment otherwise (kind 0 / 1: one tap per workgroup, 64 x 64 / 128 x 128 tiles). `share` in (0, 1]:
 *                                the fraction of the launch's work this convolution is (sizes its pixel splits; 1 = a launch of its own).
 *                                jobs: room for LLV_MAX_SEG entries; n_jobs: how many were written; n_wg of each is set, wg_begin is 0.
 *   llv_conv2d_nhwc_wgrad_jobs   one launch over a DEVICE table of jobs of ONE kind. The caller lays the jobs' workgroups side by side:
 *                                wg_begin (a multiple of 8: the kernels place splits on XCDs by workgroup index) and
 *                                wg2job[w] = index of the job that owns workgroup w, for w < total_wgs. */
#define LLV_WGRAD_JOB_ARGS 368
typedef struct llv_wgrad_job {
    int32_t kind, wg_begin, n_wg, reserved;
    unsigned char args[LLV_WGRAD_JOB_ARGS]; /* opaque: the kernel's argument block */
} llv_wgrad_job;
int llv_conv2d_nhwc_wgrad_plan(const llv_conv_desc* d, const void* dy, int dy_ld, float* dw, float* db, float share, llv_wgrad_job* jobs,
                               int* n_jobs);
int llv_conv2d_nhwc_wgrad_jobs(const llv_wgrad_job* jobs, const uint16_t* wg2job, int kind, int total_wgs, void* stream);

/* Once per optimizer step: module parameters (fp32, torch's (Cout, Cin, kh, kw)) -> the fp16 layouts the conv kernels read,
 * eval-mode BatchNorm folded in (s = gamma * rsqrt(var + eps); gamma = var = NULL: no BN). Replaces ops.pack_conv's host
 * packing while training (the reference keeps cuDNN's layout = the parameter itself, resnet.py:260-299).
 *   mode 0 forward  dst[row0+o][(r*kw+q)*cpad + i]; mode 1 data gradient of a stride-1 conv (flipped taps, in/out
 *   transposed: dst[i][..*cpad + row0+o]); mode 2 data gradient of a stride-2 conv as a KHxKW stride-1 conv over dY that
 *   emits the four output phases (rows 4i + 2py + px, PixelShuffle epilogue). cpad = channel pitch of dst's K axis,
 *   rows = destination rows written (modes 1, 2), padp = padding of the KHxKW window (backward.dgrad_geometry). */
int llv_pack_conv_weights(const float* w, const float* gamma, const float* var, float eps, void* dst, int cout, int cin,
                          int kh, int kw, int cpad, int k_pad, int row0, int mode, int pad, int KH, int KW, int padp, int rows,
                          void* stream);
/* out[o] = beta + ((b ? b[o] : 0) - mean) * gamma * rsqrt(var + eps)   (gamma == NULL: out = b or 0) */
int llv_fold_bias(const float* b, const float* gamma, const float* beta, const float* mean, const float* var, float eps,
                  float* out, int n, void* stream);
/* packed gradient block (dw: first row of the part, db: its first channel) -> parameter gradients, any output may be NULL:
 * gw = dw*s (Cout,Cin,kh,kw); ggamma = sum(dw*w)/sigma - mean/sigma*db; gbeta = db; gbias = db*s (norm_eval training). */
int llv_unpack_wgrad(const float* dw, const float* db, const float* w, const float* gamma, const float* mean, const float* var,
                     float eps, float* gw, float* ggamma, float* gbeta, float* gbias, int cout, int cin, int kh, int kw,
                     int cpad, int k_pad, void* stream);

/* The three bookkeeping launches above over tables of jobs that live in DEVICE memory (one launch per table, blockIdx.y = job): the
 * fields of a job are the arguments of the single-job entry point of the same name. blocks_per_job: workgroups of 256 threads each
 * job is given (it loops; size it for the largest job). Pointers inside the jobs are device pointers the caller keeps valid. */
typedef struct llv_pack_job {
    const float* w; const float* gamma; const float* var; float eps; void* dst;
    int cout, cin, kh, kw, cpad, k_pad, row0, mode, pad, KH, KW, padp, rows;
} llv_pack_job;
typedef struct llv_fold_job {
    const float* b; const float* gamma; const float* beta; const float* mean; const float* var; float eps; float* out; int n;
} llv_fold_job;
typedef struct llv_unpack_job {
    const float* dw; const float* db; const float* w; const float* gamma; const float* mean; const float* var; float eps;
    float* gw; float* ggamma; float* gbeta; float* gbias;
    int cout, cin, kh, kw, cpad, k_pad;
} llv_unpack_job;
int llv_pack_conv_weights_batched(const llv_pack_job* jobs, int n_jobs, int blocks_per_job, void* stream);
/* jobs of mode 1 / 2 only (data-gradient filters), moved as 32 x 32 x taps tiles through LDS; the destination must have been zero
 * when first packed (positions no tap maps to are not written). Same result as llv_pack_conv_weights_batched on the same jobs. */
int llv_pack_dgrad_weights_batched(const llv_pack_job* jobs, int n_jobs, int blocks_per_job, void* stream);
int llv_fold_bias_batched(const llv_fold_job* jobs, int n_jobs, int blocks_per_job, void* stream);
int llv_unpack_wgrad_batched(const llv_unpack_job* jobs, int n_jobs, int blocks_per_job, void* stream);

/* dx (+)= data gradient of the convolution described by `fwd` (its x/w/y pointers are not read; geometry, groups, batch and
 * segments are). w_dgrad: filter packed by llv_pack_conv_weights mode 1 (stride 1) or 2 (stride 2) with wd_rows x wd_k_pad,
 * window kh x kw and padding `pad` (backward.dgrad_geometry); zero_bias: fp32 zeros, one per output row; dy / dx fp16 NHWC with
 * row strides dy_ld / dx_ld; accumulate != 0 adds to dx in place. Stride 1 and 2, one source; stride 2 one segment per call.
 * Two optional epilogue inputs ride in `fwd` (stride 1 only): flags & LLV_CONV_MASK_POS -- dx also takes the ReLU mask of the forward
 * INPUT fwd->x[0] (x_ld == dx_ld), so that the launch whose output gradient dx is needs no llv_relu_bwd pass; fwd->res != NULL
 * (accumulate must be 0) -- dx = gradient + res instead of an accumulation: a residual block's first conv, res = the gradient of the
 * block output, whose identity path ends at the same tensor.
 * Workspace as for llv_conv2d_nhwc_fwd (NULL: the launch must not need split-K). */
int llv_conv2d_nhwc_dgrad(const llv_conv_desc* fwd, const void* w_dgrad, const float* zero_bias, int wd_rows, int wd_k_pad,
                          int kh, int kw, int pad, const void* dy, int dy_ld, void* dx, int dx_ld, int accumulate,
                          void* workspace, size_t workspace_bytes, void* stream);

/* dx = (dy [+ dy2]) * (y > 0); dy2 and y may be NULL (then: plain sum / plain copy). n fp16 elements, multiple of 8.
 * torch autograd of F.relu(x [+ identity]) (resnet.py:292-296). */
int llv_relu_bwd(const void* dy, const void* dy2, const void* y, void* dx, int64_t n, void* stream);
/* The same for the block output y = relu(v + identity) in one pass: dy *= (y > 0) in place, then gres += dy (the identity's gradient). */
int llv_relu_res_bwd(void* dy, const void* y, void* gres, int64_t n, void* stream);
/* Backward of a LeakyReLU epilogue (Darknet / YOLOV3Neck ConvModules), with the residual of a ResBlock (y = res + act(v), darknet.py:50-57)
 * when res != NULL: gres += dy (when gres != NULL), then dy *= (y - res > 0 ? 1 : slope) in place. `rows` rows of c halfs, own pitches. */
int llv_act_bwd(void* dy, const void* y, const void* res, void* gres, int64_t rows, int c, int dy_ld, int y_ld, int res_ld, int gres_ld,
                float slope, void* stream);
/* Backward of llv_upsample2x_nhwc: gsrc[n][y][x][c] += the four fine pixels of gdst (n, 2h, 2w, gdst_ld). */
int llv_upsample2x_nhwc_bwd(const void* gdst, void* gsrc, int n, int h, int w, int c, int gdst_ld, int gsrc_ld, void* stream);

/* out[c] += sum_p dy[p][c]  (fp32 atomics; dy fp16 [m][ld]): the bias / BN-shift gradient of a conv */
int llv_bias_grad(const void* dy, int64_t m, int c, int ld, float* out, void* stream);

/* gradient of FPN's top-down path (fpn.py:186-195): dcoarse (n,h,w,c) (+)= sum of dfine (n,H,W,c) over the fine pixels
 * whose F.interpolate(mode='nearest', size=(H,W)) source is that coarse pixel. fp16 NHWC, c multiple of 8. */
int llv_upsample_add_bwd(const void* dfine, void* dcoarse, int n, int H, int W, int h, int w, int c, int accumulate, void* stream);

/* torch.optim.SGD step on fp32 tensors: d = g*grad_scale + weight_decay*w; buf = first_step ? d : momentum*buf + d;
 * w -= lr*buf  (schedule_1x.py: lr 0.01, momentum 0.9, weight_decay 1e-4; grad_scale = 1/loss_scale of fp16 training) */
int llv_sgd_momentum(float* w, const float* g, float* mom, int64_t n, float lr, float momentum, float weight_decay,
                     float grad_scale, int first_step, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* LLV_H_ */
