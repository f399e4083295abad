// troi.hip -- the element-wise / selection kernels of TemporalRoIAlign (the RoI extractor of the authors' own model,
// mmtracking/configs/vid/llvod/llvod_fastdvdnet_darkfarm.py:49-59; module
// mmtracking/mmtrack/models/roi_heads/roi_extractors/temporal_roi_align.py). The two dense products of the module -- the cosine
// similarity maps (RoI points x reference pixels) and the 3x3 embed convolution over the 7x7 RoI maps -- run on the convolution
// engine (llv_conv2d_nhwc_fwd: a 1x1 convolution whose "weights" are the normalised RoI points; a 3x3 convolution over k*(img_n+1)
// small images); what is here is the rest:
//   llv_l2norm_rows_f16      x / ||x||_2 over the channels of every row (temporal_roi_align.py:123-126), fp16 out
//   llv_msra_top2_gather     per (RoI point, reference image): the two most similar reference pixels, softmax of their two cosines,
//                            weighted sum of the two reference feature vectors (:141-168, num_most_similar_points = 2)
//   llv_troi_attention_fuse  temporal_attentional_feature_aggregation after the embed conv (:76-96): per RoI point and attention
//                            block the dot products with the key RoI's embedding, / sqrt(C / blocks), softmax over the images,
//                            weighted sum of the (un-embedded) features
// All NHWC / row-major fp16 with 16-byte lane accesses; one wavefront per row where a reduction over channels is needed.

#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <math.h>
#include "llv.h"

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

namespace {

inline int waves_grid(long rows) {
    long blocks = (rows + 3) / 4;
    if (blocks < 1) blocks = 1;
    if (blocks > 256L * 16) blocks = 256L * 16;
    return (int)blocks;
}

// one wave per row: y[row][0..c) = x[row][0..c) / ||x[row]||  (fp32 or fp16 in, fp16 out; c % 8 == 0)
template <typename T>
__global__ __launch_bounds__(256) void l2norm_rows_kernel(const T* __restrict__ x, _Float16* __restrict__ y, long rows, int c, long x_ld,
                                                         long y_ld, float* __restrict__ inv_norm) {
    const int lane = threadIdx.x & 63;
    const long wave = (blockIdx.x * (long)blockDim.x + threadIdx.x) >> 6, nwaves = ((long)gridDim.x * blockDim.x) >> 6;
    for (long r = wave; r < rows; r += nwaves) {
        float ss = 0.f;
        for (int ch = lane * 8; ch < c; ch += 512) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float v = (float)x[r * x_ld + ch + e]; ss += v * v; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
        const float inv = 1.f / sqrtf(ss);                   // the reference divides by the norm as it is (a zero row gives nan there too)
        if (lane == 0 && inv_norm) inv_norm[r] = inv;
        for (int ch = lane * 8; ch < c; ch += 512) {
            f16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (_Float16)((float)x[r * x_ld + ch + e] * inv);
            *reinterpret_cast<f16x8*>(y + r * y_ld + ch) = o;
        }
    }
}

// sim [img_n * hw][sim_ld] fp32: sim[(i*hw + p)*sim_ld + r] = cos(roi point r, pixel p of reference image i)
// ref [img_n * hw][ref_ld] fp16 (the reference feature maps, NOT normalised); out [img_n][rows][c] fp16.
// One lane owns one RoI point r over all pixels of image i (the 64 lanes of a wave read 64 consecutive floats of a sim row), keeps
// its two best (value desc, pixel index asc on ties); then the wave walks its 64 points and gathers / blends the two feature rows.
__global__ __launch_bounds__(256) void msra_top2_gather_kernel(const float* __restrict__ sim, const _Float16* __restrict__ ref,
                                                              _Float16* __restrict__ out, int* __restrict__ idx_out,
                                                              float* __restrict__ w_out, int img_n, int hw, int rows, int sim_ld,
                                                              int c, int ref_ld) {
    const int lane = threadIdx.x & 63;
    const long wave = (blockIdx.x * (long)blockDim.x + threadIdx.x) >> 6, nwaves = ((long)gridDim.x * blockDim.x) >> 6;
    const int rgroups = (rows + 63) / 64;
    for (long t = wave; t < (long)img_n * rgroups; t += nwaves) {
        const int i = (int)(t / rgroups), r0 = (int)(t % rgroups) * 64, r = r0 + lane;
        float v0 = -INFINITY, v1 = -INFINITY;
        int p0 = 0, p1 = 0;
        if (r < rows) {
            const float* s = sim + (long)i * hw * sim_ld + r;
            for (int p = 0; p < hw; ++p) {
                const float v = s[(long)p * sim_ld];
                if (v > v0) { v1 = v0; p1 = p0; v0 = v; p0 = p; }
                else if (v > v1) { v1 = v; p1 = p; }
            }
        }
        // softmax over the two values (temporal_roi_align.py:149)
        const float e1 = expf(v1 - v0), w0 = 1.f / (1.f + e1), w1 = e1 / (1.f + e1);
        if (r < rows && idx_out) {
            idx_out[((long)i * rows + r) * 2] = p0; idx_out[((long)i * rows + r) * 2 + 1] = p1;
            w_out[((long)i * rows + r) * 2] = w0; w_out[((long)i * rows + r) * 2 + 1] = w1;
        }
        const int nr = rows - r0 < 64 ? rows - r0 : 64;
        for (int j = 0; j < nr; ++j) {
            const int q0 = __shfl(p0, j), q1 = __shfl(p1, j);
            const float a0 = __shfl(w0, j), a1 = __shfl(w1, j);
            const _Float16* f0 = ref + ((long)i * hw + q0) * ref_ld;
            const _Float16* f1 = ref + ((long)i * hw + q1) * ref_ld;
            _Float16* o = out + ((long)i * rows + r0 + j) * c;
            for (int ch = lane * 8; ch < c; ch += 512) {
                const f16x8 x0 = *reinterpret_cast<const f16x8*>(f0 + ch), x1 = *reinterpret_cast<const f16x8*>(f1 + ch);
                f16x8 y;
#pragma unroll
                for (int e = 0; e < 8; ++e) y[e] = (_Float16)((float)x0[e] * a0 + (float)x1[e] * a1);
                *reinterpret_cast<f16x8*>(o + ch) = y;
            }
        }
    }
}

// x, emb [n_img][rows][c] fp16 (image 0 = the key frame's RoI features; rows = roi * 49 points); out [rows][c] fp32.
// One wave per row; lane l holds channels 8l..8l+7 (+512 per pass); block of a channel = ch / (c / blocks). The per-block dot
// products are segmented wave reductions (a block spans cb/8 consecutive lanes, a power of two <= 64; c <= 512 per pass).
__global__ __launch_bounds__(256) void troi_attention_fuse_kernel(const _Float16* __restrict__ x, const _Float16* __restrict__ emb,
                                                                 float* __restrict__ out, int n_img, long rows, int c, int blocks) {
    const int lane = threadIdx.x & 63;
    const long wave = (blockIdx.x * (long)blockDim.x + threadIdx.x) >> 6, nwaves = ((long)gridDim.x * blockDim.x) >> 6;
    const int cb = c / blocks;                 // channels per block (multiple of 8)
    const int lpb = cb / 8;                    // lanes per block
    const float scale = 1.f / sqrtf((float)cb);
    for (long r = wave; r < rows; r += nwaves) {
        for (int c0 = 0; c0 < c; c0 += 512) {
            const int ch = c0 + lane * 8;
            const bool ok = ch < c;
            f16x8 t8 = {};
            if (ok) t8 = *reinterpret_cast<const f16x8*>(emb + r * c + ch);
            // pass 1: logits per image (kept in registers: n_img <= 16), max
            float lg[16];
            float mx = -INFINITY;
            for (int n = 0; n < n_img; ++n) {
                float d = 0.f;
                if (ok) {
                    const f16x8 e8 = *reinterpret_cast<const f16x8*>(emb + ((long)n * rows + r) * c + ch);
#pragma unroll
                    for (int e = 0; e < 8; ++e) d += (float)e8[e] * (float)t8[e];
                }
                for (int o = 1; o < lpb; o <<= 1) d += __shfl_xor(d, o);      // lanes of one block are lpb-aligned
                lg[n] = d * scale;
                mx = fmaxf(mx, lg[n]);
            }
            float den = 0.f;
            for (int n = 0; n < n_img; ++n) den += expf(lg[n] - mx);
            float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (int n = 0; n < n_img; ++n) {
                const float wgt = expf(lg[n] - mx) / den;
                if (ok) {
                    const f16x8 v = *reinterpret_cast<const f16x8*>(x + ((long)n * rows + r) * c + ch);
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc[e] += wgt * (float)v[e];
                }
            }
            if (ok) {
                float* o = out + r * c + ch;
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = acc[e];
            }
        }
    }
}

}  // namespace

extern "C" int llv_l2norm_rows_f16(const void* x, int x_is_f32, void* y, int64_t rows, int c, int64_t x_ld, int64_t y_ld, float* inv_norm,
                                   void* stream) {
    if (!x || !y || rows < 0 || c < 8 || (c & 7) || x_ld < c || y_ld < c || (y_ld & 7) || (!x_is_f32 && (x_ld & 7))) return LLV_E_BADARG;
    if (rows == 0) return LLV_OK;
    if (x_is_f32)
        hipLaunchKernelGGL(l2norm_rows_kernel<float>, dim3(waves_grid(rows)), dim3(256), 0, (hipStream_t)stream, (const float*)x, (_Float16*)y,
                           (long)rows, c, (long)x_ld, (long)y_ld, inv_norm);
    else
        hipLaunchKernelGGL(l2norm_rows_kernel<_Float16>, dim3(waves_grid(rows)), dim3(256), 0, (hipStream_t)stream, (const _Float16*)x,
                           (_Float16*)y, (long)rows, c, (long)x_ld, (long)y_ld, inv_norm);
    return hipGetLastError() == hipSuccess ? LLV_OK : LLV_E_LAUNCH;
}

extern "C" int llv_msra_top2_gather(const float* sim, const void* ref, void* out, int32_t* idx_out, float* w_out, int img_n, int hw,
                                    int rows, int sim_ld, int c, int ref_ld, void* stream) {
    if (!sim || !ref || !out || img_n < 1 || hw < 2 || rows < 1 || sim_ld < rows || c < 8 || (c & 7) || (ref_ld & 7) || ref_ld < c)
        return LLV_E_BADARG;
    if ((idx_out == nullptr) != (w_out == nullptr)) return LLV_E_BADARG;
    hipLaunchKernelGGL(msra_top2_gather_kernel, dim3(waves_grid((long)img_n * ((rows + 63) / 64))), dim3(256), 0, (hipStream_t)stream, sim,
                       (const _Float16*)ref, (_Float16*)out, idx_out, w_out, img_n, hw, rows, sim_ld, c, ref_ld);
    return hipGetLastError() == hipSuccess ? LLV_OK : LLV_E_LAUNCH;
}

extern "C" int llv_troi_attention_fuse(const void* x, const void* emb, float* out, int n_img, int64_t rows, int c, int blocks, void* stream) {
    if (!x || !emb || !out || n_img < 1 || n_img > 16 || rows < 0 || c < 8 || (c & 7) || blocks < 1 || c % blocks) return LLV_E_BADARG;
    const int cb = c / blocks;
    // a block must be a power-of-two number of lanes (8 channels each) that does not straddle a 512-channel pass
    if ((cb & 7) || ((cb / 8) & (cb / 8 - 1)) || cb > 512 || (512 % cb)) return LLV_E_UNSUPPORTED;
    if (rows == 0) return LLV_OK;
    hipLaunchKernelGGL(troi_attention_fuse_kernel, dim3(waves_grid(rows)), dim3(256), 0, (hipStream_t)stream, (const _Float16*)x,
                       (const _Float16*)emb, out, n_img, (long)rows, c, blocks);
    return hipGetLastError() == hipSuccess ? LLV_OK : LLV_E_LAUNCH;
}
