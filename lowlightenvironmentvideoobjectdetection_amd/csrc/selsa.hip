// selsa.hip -- the kernels around the matrix products of the SELSA RoI head.
//   SelsaAggregator.forward   mmtracking/mmtrack/models/aggregators/selsa_aggregator.py:29-73
//   SelsaBBoxHead.forward     mmtracking/mmtrack/models/roi_heads/bbox_heads/selsa_bbox_head.py:60-84
//   BBoxHead.get_bboxes       mmdetection/mmdet/models/roi_heads/bbox_heads/bbox_head.py:270-371 (softmax over classes)
// The products themselves run on the convolution engine (llv_conv2d_nhwc_fwd): a fully connected layer is a 1x1 (the first one a
// 7x7 "valid") convolution over the RoI batch; the proposal attention of one aggregator is two GROUPED 1x1 convolutions, one group
// per attention block: scores = x_embed . ref_embed^T (the reference embeddings packed as that convolution's weights, [block][ref row]
// [d]) and x_new = softmax(scores / sqrt(d)) . ref_new (the reference values packed transposed, [block][d][ref row]). What is here:
//   llv_selsa_pack_kv         writes rows of (ref rows, blocks*d) fp16 into either packed-weight layout at a row offset (the static
//                             reference frames of a video once, the key frame's own proposals every frame)
//   llv_selsa_softmax         the scaled softmax over the reference rows of every (proposal, block), fp32 scores -> fp16 weights;
//                             rows past the valid count (padding of the fixed-shape step) get weight 0
//   llv_softmax_rows_masked   class softmax of the head's logits; proposals past the valid count score 0 everywhere (they then fail
//                             score_thr in the NMS)
//   llv_dets_to_rois          (x1, y1, x2, y2, score) rows of the RPN's NMS output -> RoIAlign's (batch index, x1, y1, x2, y2) rows
// The count words are device memory: a captured step never reads them on the host.

#include <hip/hip_runtime.h>
#include <math.h>
#include "llv.h"

namespace {

__global__ void dets_to_rois_kernel(const float* __restrict__ dets, const int* __restrict__ count, float* __restrict__ rois, float* __restrict__ boxes4,
                                    int max_rows, float batch_index) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= max_rows) return;
    const bool ok = i < *count;
    float* r = rois + (long)i * 5;
    r[0] = ok ? batch_index : 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float v = ok ? dets[(long)i * 5 + e] : 0.f;
        r[1 + e] = v;
        if (boxes4) boxes4[(long)i * 4 + e] = v;
    }
}

// mode 0: dst[g][row0 + r][k] = src[r][g*d + k]            (dst: [nb][w_rows][d])
// mode 1: dst[g][k][row0 + r] = src[r][g*d + k]            (dst: [nb][d][k_pad])
__global__ void selsa_pack_kv_kernel(const _Float16* __restrict__ src, _Float16* __restrict__ dst, int rows, int row0, int nb, int d,
                                     int src_ld, long g_stride, int row_stride, int mode) {
    const long total = (long)rows * nb * d;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int k = (int)(i % d);
        const long t = i / d;
        const int g = (int)(t % nb);
        const int r = (int)(t / nb);
        const _Float16 v = src[(long)r * src_ld + g * d + k];
        if (mode == 0) dst[g * g_stride + (long)(row0 + r) * row_stride + k] = v;
        else dst[g * g_stride + (long)k * row_stride + row0 + r] = v;
    }
}

// one wavefront per (proposal, block): softmax over the valid reference rows of scores[n][g*m_ld .. + m_total)
__global__ void selsa_softmax_kernel(const float* __restrict__ s, _Float16* __restrict__ p, long n_rows, int nb, int m_ld, int m_static,
                                     int m_live_max, const int* __restrict__ live_count, float scale) {
    const int lane = threadIdx.x & 63;
    const long wave = (blockIdx.x * (long)blockDim.x + threadIdx.x) >> 6;
    const long nwaves = ((long)gridDim.x * blockDim.x) >> 6;
    int live = live_count ? *live_count : m_live_max;
    live = live < 0 ? 0 : (live > m_live_max ? m_live_max : live);
    const int m_valid = m_static + live;
    for (long w = wave; w < n_rows * nb; w += nwaves) {
        const float* row = s + w * m_ld;
        _Float16* out = p + w * m_ld;
        float mx = -INFINITY;
        for (int j = lane; j < m_valid; j += 64) mx = fmaxf(mx, row[j] * scale);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
        float den = 0.f;
        for (int j = lane; j < m_valid; j += 64) den += expf(row[j] * scale - mx);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) den += __shfl_xor(den, o);
        const float inv = m_valid > 0 ? 1.f / den : 0.f;
        for (int j = lane; j < m_ld; j += 64) out[j] = (_Float16)(j < m_valid ? expf(row[j] * scale - mx) * inv : 0.f);
    }
}

__global__ void softmax_rows_masked_kernel(const float* __restrict__ x, float* __restrict__ y, int n, int c, int x_ld, int y_ld,
                                           const int* __restrict__ count) {
    const int lane = threadIdx.x & 63;
    const int r = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (r >= n) return;
    const bool live = !count || r < *count;
    const float* row = x + (long)r * x_ld;
    float mx = -INFINITY;
    for (int j = lane; j < c; j += 64) mx = fmaxf(mx, row[j]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    float den = 0.f;
    for (int j = lane; j < c; j += 64) den += expf(row[j] - mx);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) den += __shfl_xor(den, o);
    for (int j = lane; j < c; j += 64) y[(long)r * y_ld + j] = live ? expf(row[j] - mx) / den : 0.f;
}

inline int grid1(long n, int per) {
    long g = (n + per - 1) / per;
    if (g < 1) g = 1;
    if (g > 256L * 16) g = 256L * 16;
    return (int)g;
}

}  // namespace

extern "C" int llv_dets_to_rois(const float* dets, const int32_t* count, float* rois, float* boxes4, int max_rows, int batch_index, void* stream) {
    if (!dets || !count || !rois || max_rows < 1) return LLV_E_BADARG;
    hipLaunchKernelGGL(dets_to_rois_kernel, dim3((max_rows + 255) / 256), dim3(256), 0, (hipStream_t)stream, dets, count, rois, boxes4,
                       max_rows, (float)batch_index);
    return hipGetLastError() == hipSuccess ? LLV_OK : LLV_E_LAUNCH;
}

extern "C" int llv_selsa_pack_kv(const void* src, void* dst, int rows, int row0, int blocks, int d, int src_ld, int dst_rows, int dst_kpad,
                                 int transposed, void* stream) {
    if (rows < 0 || row0 < 0 || blocks < 1 || d < 1 || src_ld < blocks * d) return LLV_E_BADARG;
    if (rows == 0) return LLV_OK;
    if (!src || !dst) return LLV_E_BADARG;
    // plain: dst [blocks][dst_rows][dst_kpad] with rows = reference rows, k = d;  transposed: rows = d, k = reference rows
    if (!transposed && (row0 + rows > dst_rows || d > dst_kpad)) return LLV_E_BADARG;
    if (transposed && (row0 + rows > dst_kpad || d > dst_rows)) return LLV_E_BADARG;
    const long total = (long)rows * blocks * d;
    hipLaunchKernelGGL(selsa_pack_kv_kernel, dim3(grid1(total, 256)), dim3(256), 0, (hipStream_t)stream, (const _Float16*)src, (_Float16*)dst,
                       rows, row0, blocks, d, src_ld, (long)dst_rows * dst_kpad, dst_kpad, transposed ? 1 : 0);
    return hipGetLastError() == hipSuccess ? LLV_OK : LLV_E_LAUNCH;
}

extern "C" int llv_selsa_softmax(const float* scores, void* weights, int64_t n_rows, int blocks, int m_ld, int m_static, int m_live_max,
                                 const int32_t* live_count, float scale, void* stream) {
    if (n_rows < 0 || blocks < 1 || m_ld < 1 || m_static < 0 || m_live_max < 0 || m_static + m_live_max > m_ld) return LLV_E_BADARG;
    if (n_rows == 0) return LLV_OK;
    if (!scores || !weights) return LLV_E_BADARG;
    hipLaunchKernelGGL(selsa_softmax_kernel, dim3(grid1(n_rows * blocks, 4)), dim3(256), 0, (hipStream_t)stream, scores, (_Float16*)weights,
                       (long)n_rows, blocks, m_ld, m_static, m_live_max, live_count, scale);
    return hipGetLastError() == hipSuccess ? LLV_OK : LLV_E_LAUNCH;
}

extern "C" int llv_softmax_rows_masked(const float* x, float* y, int n, int c, int x_ld, int y_ld, const int32_t* count, void* stream) {
    if (n < 0 || c < 1 || x_ld < c || y_ld < c) return LLV_E_BADARG;
    if (n == 0) return LLV_OK;
    if (!x || !y) return LLV_E_BADARG;
    hipLaunchKernelGGL(softmax_rows_masked_kernel, dim3((n + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, y, n, c, x_ld, y_ld, count);
    return hipGetLastError() == hipSuccess ? LLV_OK : LLV_E_LAUNCH;
}
