// elementwise.hip -- HBM-bound helpers around the conv path: max pooling, layout packing at the
// model boundary, EmbedAggregator tail, flow warp. All NHWC, 16-byte (8 x fp16) per lane accesses.

#include <hip/hip_runtime.h>
#include <cstdlib>
#include <hip/hip_fp16.h>
#include "llv.h"

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

namespace {

constexpr int kBlock = 256;
inline int grid_for(long work) {
    long g = (work + kBlock - 1) / kBlock;
    if (g < 1) g = 1;
    if (g > 256L * 16) g = 256L * 16;  // grid-stride beyond 16 blocks per CU
    return (int)g;
}

// nn.MaxPool2d(kernel_size=3, stride=2, padding=1) (resnet.py:570); padding never wins the max.
__global__ void maxpool3x3s2_kernel(const _Float16* __restrict__ x, _Float16* __restrict__ y, int batch,
                                    int h, int w, int c8, int ho, int wo) {
    const long total = (long)batch * ho * wo * c8;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < (unsigned)total; i += gridDim.x * blockDim.x) {   // total < 2^31 (host check): 32-bit divisions
        const int cv = (int)(i % c8);
        unsigned p = i / c8;
        const int ox = (int)(p % wo); p /= wo;
        const int oy = (int)(p % ho);
        const int n = (int)(p / ho);
        f16x8 m;
#pragma unroll
        for (int e = 0; e < 8; ++e) m[e] = (_Float16)(-65504.f);
        bool any = false;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            const int iy = oy * 2 - 1 + dy;
            if ((unsigned)iy >= (unsigned)h) continue;
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int ix = ox * 2 - 1 + dx;
                if ((unsigned)ix >= (unsigned)w) continue;
                const f16x8 v = *reinterpret_cast<const f16x8*>(x + (((long)n * h + iy) * w + ix) * (c8 * 8) + cv * 8);
#pragma unroll
                for (int e = 0; e < 8; ++e) m[e] = v[e] > m[e] ? v[e] : m[e];
                any = true;
            }
        }
        (void)any;
        *reinterpret_cast<f16x8*>(y + (long)i * 8) = m;
    }
}

// NCHW fp32 -> NHWC fp16 (c_pad channels, zeros beyond c); one lane per pixel.
__global__ void pack_kernel(const float* __restrict__ x, _Float16* __restrict__ y, int batch, int c, long hw,
                            int c_pad, const float* __restrict__ scale, const float* __restrict__ shift) {
    const long total = (long)batch * hw;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < (unsigned)total; i += gridDim.x * blockDim.x) {   // total < 2^31 (host check): 32-bit divisions
        const unsigned n = i / (unsigned)hw, p = i - n * (unsigned)hw;
        for (int cb = 0; cb < c_pad; cb += 8) {
            f16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int ch = cb + e;
                float v = 0.f;
                if (ch < c) {
                    v = x[(n * c + ch) * hw + p];
                    if (scale) v = v * scale[ch] + shift[ch];
                }
                o[e] = (_Float16)v;
            }
            *reinterpret_cast<f16x8*>(y + (long)i * c_pad + cb) = o;
        }
    }
}

template <typename T>
__global__ void unpack_kernel(const T* __restrict__ x, float* __restrict__ y, int batch, int c, long hw, int ld) {
    const long total = (long)batch * c * hw;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < (unsigned)total; i += gridDim.x * blockDim.x) {   // total < 2^31 (host check): 32-bit divisions
        const unsigned p = i % (unsigned)hw;
        const unsigned t = i / (unsigned)hw;
        const int ch = (int)(t % c);
        const unsigned n = t / c;
        y[i] = (float)x[(n * hw + p) * ld + ch];
    }
}

// EmbedAggregator tail (embed_aggregator.py:70-81). One wave per pixel: lanes stride the channels.
//   xe = x_embed / ||x_embed||, re_n = ref_embed_n / ||ref_embed_n||, w_n = softmax_n(sum_c re_n*xe),
//   out = sum_n w_n * ref_n.
// key_slot >= 0 selects the ring form: re/ref hold n_ref slots of a ring buffer, the key frame is slot key_slot
// (its embedding is read from re, xe is ignored) and is excluded from the references, as LLVOD does
// (mmtracking/mmtrack/models/vid/llvod.py:201-206).
__global__ void embed_aggregate_kernel(const _Float16* __restrict__ xe, const _Float16* __restrict__ re,
                                       const _Float16* __restrict__ ref, _Float16* __restrict__ out, int n_ref,
                                       long hw, int c, int embed_ld, int ref_ld, int out_ld, int key_slot) {
    if (key_slot >= 0) xe = re + (long)key_slot * hw * embed_ld;
    const int lane = threadIdx.x & 63;
    const long wave = (blockIdx.x * (long)blockDim.x + threadIdx.x) >> 6;
    const long nwaves = ((long)gridDim.x * blockDim.x) >> 6;
    extern __shared__ float sw[];  // [waves per block][n_ref]
    float* myw = sw + (threadIdx.x >> 6) * n_ref;
    for (long p = wave; p < hw; p += nwaves) {
        // |x|^2
        float xx = 0.f;
        for (int ch = lane * 8; ch < c; ch += 512) {
            const f16x8 v = *reinterpret_cast<const f16x8*>(xe + p * embed_ld + ch);
#pragma unroll
            for (int e = 0; e < 8; ++e) xx += (float)v[e] * (float)v[e];
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) xx += __shfl_xor(xx, o);
        const float xn = sqrtf(xx);
        float mx = -INFINITY;
        for (int n = 0; n < n_ref; ++n) {
            if (n == key_slot) { if (lane == 0) myw[n] = -INFINITY; continue; }
            float rr = 0.f, rx = 0.f;
            for (int ch = lane * 8; ch < c; ch += 512) {
                const f16x8 v = *reinterpret_cast<const f16x8*>(re + ((long)n * hw + p) * embed_ld + ch);
                const f16x8 u = *reinterpret_cast<const f16x8*>(xe + p * embed_ld + ch);
#pragma unroll
                for (int e = 0; e < 8; ++e) { rr += (float)v[e] * (float)v[e]; rx += (float)v[e] * (float)u[e]; }
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { rr += __shfl_xor(rr, o); rx += __shfl_xor(rx, o); }
            const float cs = rx / (sqrtf(rr) * xn);
            if (lane == 0) myw[n] = cs;
            mx = fmaxf(mx, cs);
        }
        __builtin_amdgcn_wave_barrier();
        float den = 0.f;
        for (int n = 0; n < n_ref; ++n)
            if (n != key_slot) den += expf(myw[n] - mx);
        for (int ch = lane * 8; ch < c; ch += 512) {
            float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (int n = 0; n < n_ref; ++n) {
                if (n == key_slot) continue;
                const float wgt = expf(myw[n] - mx) / den;
                const f16x8 v = *reinterpret_cast<const f16x8*>(ref + ((long)n * hw + p) * ref_ld + ch);
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] += wgt * (float)v[e];
            }
            f16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (_Float16)acc[e];
            *reinterpret_cast<f16x8*>(out + p * out_ld + ch) = o;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// flow_warp_feats (mmtracking/mmtrack/core/motion/flow.py:4-41), reproduced operation by operation:
//  1. flow -> bilinear resize (align_corners=False, scale_factor = W_x/W_f) to (h,w), times scale;
//  2. grid = (x + u)/W*2-1, (y + v)/H*2-1   (divides by W,H -- not W-1 -- as the reference does);
//  3. grid_sample(bilinear, padding_mode='border', align_corners=True).
__device__ __forceinline__ float bilerp_flow(const float* f, int hf, int wf, float sy, float sx) {
    // PyTorch upsample_bilinear2d, align_corners=False: src = max(0, (dst+0.5)*scale - 0.5)
    sy = sy < 0.f ? 0.f : sy;
    sx = sx < 0.f ? 0.f : sx;
    int y0 = (int)sy, x0 = (int)sx;
    y0 = y0 < hf - 1 ? y0 : hf - 1;
    x0 = x0 < wf - 1 ? x0 : wf - 1;
    const int y1 = y0 < hf - 1 ? y0 + 1 : y0, x1 = x0 < wf - 1 ? x0 + 1 : x0;
    const float ly = sy - (float)y0, lx = sx - (float)x0;
    const float hy = 1.f - ly, hx = 1.f - lx;
    return hy * (hx * f[y0 * wf + x0] + lx * f[y0 * wf + x1]) + ly * (hx * f[y1 * wf + x0] + lx * f[y1 * wf + x1]);
}

__global__ void flow_warp_kernel(const _Float16* __restrict__ x, const float* __restrict__ flow,
                                 _Float16* __restrict__ y, int nb, int h, int w, int c8, int ld, int hf, int wf,
                                 float rscale_h, float rscale_w, float mul) {
    const long total = (long)nb * h * w * c8;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < (unsigned)total; i += gridDim.x * blockDim.x) {   // total < 2^31 (host check): 32-bit divisions
        const int cv = (int)(i % c8);
        unsigned p = i / c8;
        const int ox = (int)(p % w); p /= w;
        const int oy = (int)(p % h);
        const int n = (int)(p / h);
        const float sy = ((float)oy + 0.5f) * rscale_h - 0.5f;
        const float sx = ((float)ox + 0.5f) * rscale_w - 0.5f;
        const float* fu = flow + ((long)n * 2 + 0) * hf * wf;
        const float* fv = flow + ((long)n * 2 + 1) * hf * wf;
        const float u = bilerp_flow(fu, hf, wf, sy, sx) * mul;
        const float v = bilerp_flow(fv, hf, wf, sy, sx) * mul;
        float gx = ((float)ox + u) / (float)w * 2.f - 1.f;
        float gy = ((float)oy + v) / (float)h * 2.f - 1.f;
        // grid_sample align_corners=True: pix = (g+1)/2*(size-1); border: clamp to [0,size-1]
        float fx = (gx + 1.f) * 0.5f * (float)(w - 1);
        float fy = (gy + 1.f) * 0.5f * (float)(h - 1);
        fx = fminf(fmaxf(fx, 0.f), (float)(w - 1));
        fy = fminf(fmaxf(fy, 0.f), (float)(h - 1));
        const int x0 = (int)floorf(fx), y0 = (int)floorf(fy);
        const int x1 = x0 + 1, y1 = y0 + 1;
        const float lx = fx - (float)x0, ly = fy - (float)y0;
        const float w00 = (1.f - lx) * (1.f - ly), w01 = lx * (1.f - ly), w10 = (1.f - lx) * ly, w11 = lx * ly;
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        const _Float16* xb = x + (long)n * h * w * ld + cv * 8;
        auto tap = [&](int yy, int xx, float wt) {
            if ((unsigned)yy < (unsigned)h && (unsigned)xx < (unsigned)w) {
                const f16x8 t = *reinterpret_cast<const f16x8*>(xb + ((long)yy * w + xx) * ld);
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] += wt * (float)t[e];
            }
        };
        tap(y0, x0, w00); tap(y0, x1, w01); tap(y1, x0, w10); tap(y1, x1, w11);
        f16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (_Float16)acc[e];
        *reinterpret_cast<f16x8*>(y + (((long)n * h + oy) * w + ox) * ld + cv * 8) = o;
    }
}

// Backward of the EmbedAggregator tail (ring form: the key frame is slot key_slot, the references are the other slots). One wave per
// pixel. With e_n = embed slot n, k = e_key, c_n = <e_n, k> / (|e_n| |k|), w = softmax_n(c), out = sum_n w_n f_n and g = d loss / d out:
//   d f_n = w_n g;   d w_n = <g, f_n>;   d c_n = w_n (d w_n - sum_m w_m d w_m);
//   d e_n = d c_n (k / |k| - c_n e_n / |e_n|) / |e_n|;   d k = sum_n d c_n (e_n / |e_n| - c_n k / |k|) / |k|.
// All three gradients are ADDED to fp16 buffers (the training graph's gradient twins accumulate over consumers).
__global__ void embed_aggregate_bwd_kernel(const _Float16* __restrict__ emb, const _Float16* __restrict__ feat,
                                           const _Float16* __restrict__ gout, _Float16* __restrict__ d_emb,
                                           _Float16* __restrict__ d_feat, int n_slots, int key_slot, long hw, int c, int embed_ld,
                                           int feat_ld, int gout_ld) {
    const int lane = threadIdx.x & 63;
    const long wave = (blockIdx.x * (long)blockDim.x + threadIdx.x) >> 6;
    const long nwaves = ((long)gridDim.x * blockDim.x) >> 6;
    extern __shared__ float sw[];                            // [waves per block][3 * n_slots]: cosine, |e_n|, d c_n
    float* cs = sw + (threadIdx.x >> 6) * 3 * n_slots;
    float* rn = cs + n_slots;
    float* dc = rn + n_slots;
    for (long p = wave; p < hw; p += nwaves) {
        const _Float16* kp = emb + ((long)key_slot * hw + p) * embed_ld;
        float kk = 0.f;
        for (int ch = lane * 8; ch < c; ch += 512) {
            const f16x8 v = *reinterpret_cast<const f16x8*>(kp + ch);
#pragma unroll
            for (int e = 0; e < 8; ++e) kk += (float)v[e] * (float)v[e];
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) kk += __shfl_xor(kk, o);
        const float kn = sqrtf(kk);
        float mx = -INFINITY;
        for (int n = 0; n < n_slots; ++n) {
            if (n == key_slot) continue;
            float rr = 0.f, rx = 0.f, gf = 0.f;
            for (int ch = lane * 8; ch < c; ch += 512) {
                const f16x8 v = *reinterpret_cast<const f16x8*>(emb + ((long)n * hw + p) * embed_ld + ch);
                const f16x8 u = *reinterpret_cast<const f16x8*>(kp + ch);
                const f16x8 f = *reinterpret_cast<const f16x8*>(feat + ((long)n * hw + p) * feat_ld + ch);
                const f16x8 g = *reinterpret_cast<const f16x8*>(gout + p * gout_ld + ch);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    rr += (float)v[e] * (float)v[e];
                    rx += (float)v[e] * (float)u[e];
                    gf += (float)g[e] * (float)f[e];
                }
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { rr += __shfl_xor(rr, o); rx += __shfl_xor(rx, o); gf += __shfl_xor(gf, o); }
            const float r = sqrtf(rr), cv = rx / (r * kn);
            if (lane == 0) { cs[n] = cv; rn[n] = r; dc[n] = gf; }         // dc holds d w_n until the softmax below
            mx = fmaxf(mx, cv);
        }
        __builtin_amdgcn_wave_barrier();
        float den = 0.f, s = 0.f;
        for (int n = 0; n < n_slots; ++n)
            if (n != key_slot) den += expf(cs[n] - mx);
        for (int n = 0; n < n_slots; ++n)
            if (n != key_slot) s += expf(cs[n] - mx) / den * dc[n];
        __builtin_amdgcn_wave_barrier();
        for (int ch = lane * 8; ch < c; ch += 512) {
            const f16x8 u = *reinterpret_cast<const f16x8*>(kp + ch);
            const f16x8 g = *reinterpret_cast<const f16x8*>(gout + p * gout_ld + ch);
            float dk[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (int n = 0; n < n_slots; ++n) {
                if (n == key_slot) continue;
                const float wn = expf(cs[n] - mx) / den;
                const float dcn = wn * (dc[n] - s);
                const float ir = 1.f / rn[n], ik = 1.f / kn;
                const f16x8 v = *reinterpret_cast<const f16x8*>(emb + ((long)n * hw + p) * embed_ld + ch);
                _Float16* de = d_emb + ((long)n * hw + p) * embed_ld + ch;
                _Float16* df = d_feat + ((long)n * hw + p) * feat_ld + ch;
                f16x8 oe = *reinterpret_cast<const f16x8*>(de), of = *reinterpret_cast<const f16x8*>(df);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float vn = (float)v[e] * ir, un = (float)u[e] * ik;
                    oe[e] = (_Float16)((float)oe[e] + dcn * (un - cs[n] * vn) * ir);
                    of[e] = (_Float16)((float)of[e] + wn * (float)g[e]);
                    dk[e] += dcn * (vn - cs[n] * un) * ik;
                }
                *reinterpret_cast<f16x8*>(de) = oe;
                *reinterpret_cast<f16x8*>(df) = of;
            }
            _Float16* dkp = d_emb + ((long)key_slot * hw + p) * embed_ld + ch;
            f16x8 ok = *reinterpret_cast<const f16x8*>(dkp);
#pragma unroll
            for (int e = 0; e < 8; ++e) ok[e] = (_Float16)((float)ok[e] + dk[e]);
            *reinterpret_cast<f16x8*>(dkp) = ok;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// Backward of flow_warp_feats with respect to the warped features: every output pixel's gradient is scattered to the (up to) four
// input pixels it was interpolated from, with the forward kernel's weights (same coordinate arithmetic, operation by operation).
// One thread per (pixel, channel), channels fastest: a wave-instruction's atomics cover 64 consecutive floats of one input pixel.
// dx is fp32 [n][h][w][c] (zeroed by the caller); llv_add_f32_to_f16 folds it into the fp16 gradient buffer afterwards.
__global__ void flow_warp_bwd_kernel(const _Float16* __restrict__ dy, const float* __restrict__ flow, float* __restrict__ dx,
                                     int nb, int h, int w, int c, int ld, int hf, int wf, float rscale_h, float rscale_w, float mul) {
    const long total = (long)nb * h * w * c;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < (unsigned)total; i += gridDim.x * blockDim.x) {
        const int ch = (int)(i % c);
        unsigned p = i / c;
        const int ox = (int)(p % w); p /= w;
        const int oy = (int)(p % h);
        const int n = (int)(p / h);
        const float g = (float)dy[(((long)n * h + oy) * w + ox) * ld + ch];
        if (g == 0.f) continue;
        const float sy = ((float)oy + 0.5f) * rscale_h - 0.5f;
        const float sx = ((float)ox + 0.5f) * rscale_w - 0.5f;
        const float* fu = flow + ((long)n * 2 + 0) * hf * wf;
        const float* fv = flow + ((long)n * 2 + 1) * hf * wf;
        const float u = bilerp_flow(fu, hf, wf, sy, sx) * mul;
        const float v = bilerp_flow(fv, hf, wf, sy, sx) * mul;
        const float gx = ((float)ox + u) / (float)w * 2.f - 1.f;
        const float gy = ((float)oy + v) / (float)h * 2.f - 1.f;
        float fx = (gx + 1.f) * 0.5f * (float)(w - 1);
        float fy = (gy + 1.f) * 0.5f * (float)(h - 1);
        fx = fminf(fmaxf(fx, 0.f), (float)(w - 1));
        fy = fminf(fmaxf(fy, 0.f), (float)(h - 1));
        const int x0 = (int)floorf(fx), y0 = (int)floorf(fy);
        const int x1 = x0 + 1, y1 = y0 + 1;
        const float lx = fx - (float)x0, ly = fy - (float)y0;
        float* xb = dx + (long)n * h * w * c + ch;
        auto tap = [&](int yy, int xx, float wt) {
            if ((unsigned)yy < (unsigned)h && (unsigned)xx < (unsigned)w && wt != 0.f) atomicAdd(xb + ((long)yy * w + xx) * c, wt * g);
        };
        tap(y0, x0, (1.f - lx) * (1.f - ly)); tap(y0, x1, lx * (1.f - ly)); tap(y1, x0, (1.f - lx) * ly); tap(y1, x1, lx * ly);
    }
}

// d(loss)/d(flow) of flow_warp_feats (mmtracking/mmtrack/core/motion/flow.py:4-41): the grid gradient of grid_sample (bilinear, border
// padding, align_corners=True) chained through grid = (x + u) / W * 2 - 1 and the bilinear resize of the flow field (x scale factor).
// One wavefront per output pixel (n, oy, ox): the lanes split the channels, d out_c / d fx = (1 - ly)(X01 - X00) + ly (X11 - X10) (and the
// same in y) are summed over the channels against dY, times d fx / d u = (w - 1) / w * mul -- zero where the sample was clamped to the
// border, as ATen's clip_coordinates_set_grad has it -- and scattered with the resize's four weights into dflow (fp32 atomics).
__global__ void flow_warp_bwd_flow_kernel(const _Float16* __restrict__ dy, const _Float16* __restrict__ x, const float* __restrict__ flow,
                                          float* __restrict__ dflow, int nb, int h, int w, int c, int dy_ld, int x_ld, int hf, int wf,
                                          float rscale, float mul) {
    const int lane = threadIdx.x & 63;
    const long wave = (blockIdx.x * (long)blockDim.x + threadIdx.x) >> 6;
    const long nwaves = ((long)gridDim.x * blockDim.x) >> 6;
    const long total = (long)nb * h * w;
    for (long p = wave; p < total; p += nwaves) {
        const int ox = (int)(p % w);
        const long t = p / w;
        const int oy = (int)(t % h);
        const int n = (int)(t / h);
        float sy = ((float)oy + 0.5f) * rscale - 0.5f, sx = ((float)ox + 0.5f) * rscale - 0.5f;
        const float* fu = flow + ((long)n * 2 + 0) * hf * wf;
        const float* fv = flow + ((long)n * 2 + 1) * hf * wf;
        const float u = bilerp_flow(fu, hf, wf, sy, sx) * mul;
        const float v = bilerp_flow(fv, hf, wf, sy, sx) * mul;
        const float rx = (((float)ox + u) / (float)w * 2.f - 1.f + 1.f) * 0.5f * (float)(w - 1);
        const float ry = (((float)oy + v) / (float)h * 2.f - 1.f + 1.f) * 0.5f * (float)(h - 1);
        const bool inx = rx > 0.f && rx < (float)(w - 1), iny = ry > 0.f && ry < (float)(h - 1);
        const float fx = fminf(fmaxf(rx, 0.f), (float)(w - 1)), fy = fminf(fmaxf(ry, 0.f), (float)(h - 1));
        const int x0 = (int)floorf(fx), y0 = (int)floorf(fy);
        const int x1 = x0 + 1, y1 = y0 + 1;
        const float lx = fx - (float)x0, ly = fy - (float)y0;
        const bool x1ok = x1 < w, y1ok = y1 < h;
        const _Float16* xb = x + (long)n * h * w * x_ld;
        const _Float16* g = dy + (((long)n * h + oy) * w + ox) * dy_ld;
        float gx = 0.f, gy = 0.f;
        for (int ch = lane; ch < c; ch += 64) {
            const float d = (float)g[ch];
            const float v00 = (float)xb[((long)y0 * w + x0) * x_ld + ch];
            const float v01 = x1ok ? (float)xb[((long)y0 * w + x1) * x_ld + ch] : 0.f;
            const float v10 = y1ok ? (float)xb[((long)y1 * w + x0) * x_ld + ch] : 0.f;
            const float v11 = (x1ok && y1ok) ? (float)xb[((long)y1 * w + x1) * x_ld + ch] : 0.f;
            gx += d * ((1.f - ly) * (v01 - v00) + ly * (v11 - v10));
            gy += d * ((1.f - lx) * (v10 - v00) + lx * (v11 - v01));
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            gx += __shfl_xor(gx, o);
            gy += __shfl_xor(gy, o);
        }
        if (lane == 0) {
            const float du = inx ? gx * (float)(w - 1) / (float)w * mul : 0.f;
            const float dv = iny ? gy * (float)(h - 1) / (float)h * mul : 0.f;
            // the transpose of bilerp_flow at (sy, sx)
            sy = sy < 0.f ? 0.f : sy;
            sx = sx < 0.f ? 0.f : sx;
            int fy0 = (int)sy, fx0 = (int)sx;
            fy0 = fy0 < hf - 1 ? fy0 : hf - 1;
            fx0 = fx0 < wf - 1 ? fx0 : wf - 1;
            const int fy1 = fy0 < hf - 1 ? fy0 + 1 : fy0, fx1 = fx0 < wf - 1 ? fx0 + 1 : fx0;
            const float wy = sy - (float)fy0, wx = sx - (float)fx0;
            float* du_p = dflow + ((long)n * 2 + 0) * hf * wf;
            float* dv_p = dflow + ((long)n * 2 + 1) * hf * wf;
            const float w00 = (1.f - wy) * (1.f - wx), w01 = (1.f - wy) * wx, w10 = wy * (1.f - wx), w11 = wy * wx;
            if (du != 0.f) {
                atomicAdd(du_p + fy0 * wf + fx0, w00 * du); atomicAdd(du_p + fy0 * wf + fx1, w01 * du);
                atomicAdd(du_p + fy1 * wf + fx0, w10 * du); atomicAdd(du_p + fy1 * wf + fx1, w11 * du);
            }
            if (dv != 0.f) {
                atomicAdd(dv_p + fy0 * wf + fx0, w00 * dv); atomicAdd(dv_p + fy0 * wf + fx1, w01 * dv);
                atomicAdd(dv_p + fy1 * wf + fx0, w10 * dv); atomicAdd(dv_p + fy1 * wf + fx1, w11 * dv);
            }
        }
    }
}

// Backward of llv_bilinear_resize_f32: dx (nc, h, w) += the transpose of the resize applied to dy (nc, ho, wo) * mul (fp32 atomics)
__global__ void bilinear_resize_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int nc, int h, int w, int ho, int wo,
                                           float rscale, float mul) {
    const long total = (long)nc * ho * wo;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < (unsigned)total; i += gridDim.x * blockDim.x) {
        const int ox = (int)(i % wo);
        const unsigned t = i / wo;
        const int oy = (int)(t % ho);
        const unsigned p = t / ho;
        const float g = dy[i] * mul;
        if (g == 0.f) continue;
        float sy = ((float)oy + 0.5f) * rscale - 0.5f, sx = ((float)ox + 0.5f) * rscale - 0.5f;
        sy = sy < 0.f ? 0.f : sy;
        sx = sx < 0.f ? 0.f : sx;
        int y0 = (int)sy, x0 = (int)sx;
        y0 = y0 < h - 1 ? y0 : h - 1;
        x0 = x0 < w - 1 ? x0 : w - 1;
        const int y1 = y0 < h - 1 ? y0 + 1 : y0, x1 = x0 < w - 1 ? x0 + 1 : x0;
        const float ly = sy - (float)y0, lx = sx - (float)x0;
        float* d = dx + (long)p * h * w;
        atomicAdd(d + y0 * w + x0, (1.f - ly) * (1.f - lx) * g); atomicAdd(d + y0 * w + x1, (1.f - ly) * lx * g);
        atomicAdd(d + y1 * w + x0, ly * (1.f - lx) * g); atomicAdd(d + y1 * w + x1, ly * lx * g);
    }
}

__global__ void add_f32_to_f16_kernel(const float* __restrict__ src, _Float16* __restrict__ dst, long rows, int c, int ld) {
    const long total = rows * c;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / c;
        const int ch = (int)(i - r * c);
        _Float16* d = dst + r * ld + ch;
        *d = (_Float16)((float)*d + src[i]);
    }
}

// bilinear sample of plane f (h x w) at PyTorch's align_corners=False source coordinate of output (oy, ox)
__device__ __forceinline__ float bilerp_plane(const float* f, int h, int w, float sy, float sx) {
    sy = sy < 0.f ? 0.f : sy;
    sx = sx < 0.f ? 0.f : sx;
    int y0 = (int)sy, x0 = (int)sx;
    y0 = y0 < h - 1 ? y0 : h - 1;
    x0 = x0 < w - 1 ? x0 : w - 1;
    const int y1 = y0 < h - 1 ? y0 + 1 : y0, x1 = x0 < w - 1 ? x0 + 1 : x0;
    const float ly = sy - (float)y0, lx = sx - (float)x0;
    const float hy = 1.f - ly, hx = 1.f - lx;
    return hy * (hx * f[y0 * w + x0] + lx * f[y0 * w + x1]) + ly * (hx * f[y1 * w + x0] + lx * f[y1 * w + x1]);
}

__global__ void bilinear_resize_kernel(const float* __restrict__ x, float* __restrict__ y, int nc, int h, int w, int ho,
                                       int wo, float rscale, float mul) {
    const long total = (long)nc * ho * wo;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < (unsigned)total; i += gridDim.x * blockDim.x) {   // total < 2^31 (host check): 32-bit divisions
        const int ox = (int)(i % wo);
        const unsigned t = i / wo;
        const int oy = (int)(t % ho);
        const unsigned p = t / ho;
        const float sy = ((float)oy + 0.5f) * rscale - 0.5f, sx = ((float)ox + 0.5f) * rscale - 0.5f;
        y[i] = bilerp_plane(x + (long)p * h * w, h, w, sy, sx) * mul;
    }
}

// FlowNetSimple.prepare_imgs: affine per channel, zero outside the un-padded image, bilinear resize, NHWC8 fp16
__global__ void flow_prepare_kernel(const float* __restrict__ x, _Float16* __restrict__ y, int n, int h, int w, int ho,
                                    int wo, int img_h, int img_w, const float* __restrict__ scale,
                                    const float* __restrict__ shift, float rscale) {
    const long total = (long)n * ho * wo;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < (unsigned)total; i += gridDim.x * blockDim.x) {   // total < 2^31 (host check): 32-bit divisions
        const int ox = (int)(i % wo);
        const unsigned t = i / wo;
        const int oy = (int)(t % ho);
        const unsigned b = t / ho;
        float sy = ((float)oy + 0.5f) * rscale - 0.5f, sx = ((float)ox + 0.5f) * rscale - 0.5f;
        sy = sy < 0.f ? 0.f : sy;
        sx = sx < 0.f ? 0.f : sx;
        int y0 = (int)sy, x0 = (int)sx;
        y0 = y0 < h - 1 ? y0 : h - 1;
        x0 = x0 < w - 1 ? x0 : w - 1;
        const int y1 = y0 < h - 1 ? y0 + 1 : y0, x1 = x0 < w - 1 ? x0 + 1 : x0;
        const float ly = sy - (float)y0, lx = sx - (float)x0, hy = 1.f - ly, hx = 1.f - lx;
        f16x8 o;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            float v = 0.f;
            if (c < 6) {
                const float* f = x + (b * 6 + c) * (long)h * w;
                const float sc = scale[c], sh = shift[c];
                auto px = [&](int yy, int xx) { return (yy < img_h && xx < img_w) ? f[yy * w + xx] * sc + sh : 0.f; };
                v = hy * (hx * px(y0, x0) + lx * px(y0, x1)) + ly * (hx * px(y1, x0) + lx * px(y1, x1));
            }
            o[c] = (_Float16)v;
        }
        *reinterpret_cast<f16x8*>(y + (long)i * 8) = o;
    }
}

}  // namespace

extern "C" int llv_flow_prepare(const float* x, void* y, int n, int h, int w, int ho, int wo, int img_h, int img_w,
                                const float* scale, const float* shift, float resize_scale, void* stream) {
    if (!x || !y || !scale || !shift || n < 1 || h < 1 || w < 1 || ho < 1 || wo < 1 || !(resize_scale > 0.f)) return LLV_E_BADARG;
    if ((int)floor((double)h * (double)resize_scale) != ho || (int)floor((double)w * (double)resize_scale) != wo) return LLV_E_BADARG;
    const long total = (long)n * ho * wo;
    if (total >= (1L << 31)) return LLV_E_UNSUPPORTED;   // the kernel indexes with 32-bit arithmetic
    hipLaunchKernelGGL(flow_prepare_kernel, dim3(grid_for(total)), dim3(kBlock), 0, (hipStream_t)stream, x, (_Float16*)y, n, h, w,
                       ho, wo, img_h, img_w, scale, shift, 1.f / resize_scale);
    return hipGetLastError() == hipSuccess ? LLV_OK : LLV_E_LAUNCH;
}

// dst[n][2y+dy][2x+dx][c] = src[n][y][x][c] (F.interpolate(scale_factor=2), nearest; yolo_neck.py:123), 16 B per lane
__global__ void upsample2x_kernel(const _Float16* src, _Float16* dst, int n, int h, int w, int c8, int src_ld, int dst_ld) {
    const long total = (long)n * (2 * h) * (2 * w) * c8;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < (unsigned)total; i += gridDim.x * blockDim.x) {   // total < 2^31 (host check): 32-bit divisions
        const int cv = (int)(i % c8);
        unsigned t = i / c8;
        const int x = (int)(t % (2 * w)); t /= 2 * w;
        const int y = (int)(t % (2 * h));
        const int im = (int)(t / (2 * h));
        const uint4 v = *reinterpret_cast<const uint4*>(src + (((long)im * h + (y >> 1)) * w + (x >> 1)) * src_ld + cv * 8);
        *reinterpret_cast<uint4*>(dst + (((long)im * 2 * h + y) * (2 * w) + x) * dst_ld + cv * 8) = v;
    }
}

// one lane per anchor: sigmoid(xy) -> centre, exp(wh) * anchor size -> box; objectness = sigmoid(attr 4)
// (yolo_head.py:262-279, yolo_bbox_coder.py:61-89). The anchor of cell (cy, cx) is centred at (cx + 0.5, cy + 0.5) * stride.
struct YoloArgs { const float* pred; float* boxes; float* conf; int n, h, w, A, na, ld, stride; float aw[8], ah[8]; };
__global__ void yolo_decode_kernel(const YoloArgs k) {
    const long per = (long)k.h * k.w * k.A, total = per * k.n;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < (unsigned)total; i += gridDim.x * blockDim.x) {   // total < 2^31 (host check): 32-bit divisions
        const int a = (int)(i % k.A);
        unsigned t = i / k.A;
        const int cx = (int)(t % k.w); t /= k.w;
        const int cy = (int)(t % k.h);
        const int im = (int)(t / k.h);
        const float* p = k.pred + (((long)im * k.h + cy) * k.w + cx) * k.ld + a * k.na;
        const float sx = 1.f / (1.f + expf(-p[0])), sy = 1.f / (1.f + expf(-p[1]));
        const float st = (float)k.stride;
        // anchor corners as the generator makes them (base anchor + shift), then centre / size as the coder recomputes them
        const float bx0 = 0.5f * st - 0.5f * k.aw[a] + (float)(cx * k.stride), bx1 = 0.5f * st + 0.5f * k.aw[a] + (float)(cx * k.stride);
        const float by0 = 0.5f * st - 0.5f * k.ah[a] + (float)(cy * k.stride), by1 = 0.5f * st + 0.5f * k.ah[a] + (float)(cy * k.stride);
        const float xc = (bx0 + bx1) * 0.5f, yc = (by0 + by1) * 0.5f, aw = bx1 - bx0, ah = by1 - by0;
        const float px = (sx - 0.5f) * st + xc, py = (sy - 0.5f) * st + yc;
        const float pw = expf(p[2]) * aw, ph = expf(p[3]) * ah;
        reinterpret_cast<float4*>(k.boxes)[i] = make_float4(px - pw / 2, py - ph / 2, px + pw / 2, py + ph / 2);
        k.conf[i] = 1.f / (1.f + expf(-p[4]));
    }
}

// ---- test-time image pipeline fused: Brighten (one gain per clip, from frame 0) -> uint8 truncation -> BGR->RGB ->
// (x - mean) * (1/std) -> zero pad -> NHWC8 fp16 (transforms.py:376-432 of mmtrack, Normalize / Pad of mmdet)
__global__ void sum_u8_kernel(const unsigned char* x, long n, unsigned long long* out) {
    unsigned long long s = 0;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) s += x[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if ((threadIdx.x & 63) == 0 && s) atomicAdd(out, s);
}

struct PrepArgs { const unsigned char* x; const unsigned long long* sum; _Float16* y; int t, h, w, hp, wp, to_rgb; double m_hw3_255; float mean[3], stdinv[3]; };
__global__ void preprocess_kernel(const PrepArgs k) {
    const long total = (long)k.t * k.hp * k.wp;
    const unsigned long long s = *k.sum;
    const double amp = k.m_hw3_255 / (double)(s ? s : 1ull);          // m*h*w*3 / sum(x/255)
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < (unsigned)total; i += gridDim.x * blockDim.x) {   // total < 2^31 (host check): 32-bit divisions
        const int px = (int)(i % k.wp);
        unsigned r = i / k.wp;
        const int py = (int)(r % k.hp);
        const int f = (int)(r / k.hp);
        _Float16 o[8] = {(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
        if (py < k.h && px < k.w) {
            const unsigned char* p = k.x + (((long)f * k.h + py) * k.w + px) * 3;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                double v = (double)p[k.to_rgb ? 2 - c : c] * amp;        // np.clip(img * amp, 0, 255).astype(uint8): truncation
                v = v < 0.0 ? 0.0 : (v > 255.0 ? 255.0 : v);
                const float q = (float)(int)v;
                o[c] = (_Float16)((q - k.mean[c]) * k.stdinv[c]);
            }
        }
        *reinterpret_cast<uint4*>(k.y + (long)i * 8) = *reinterpret_cast<const uint4*>(o);
    }
}

extern "C" int llv_brighten_normalize_pad_u8(const void* frames, int t, int h, int w, int hp, int wp, float m, const float* mean3,
                                             const float* std3, int to_rgb, void* out, void* workspace, void* stream) {
    if (!frames || !out || !workspace || !mean3 || !std3 || t < 1 || h < 1 || w < 1 || hp < h || wp < w || !(m > 0.f)) return LLV_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(workspace, 0, 8, st) != hipSuccess) return LLV_E_LAUNCH;
    const long n0 = (long)h * w * 3;
    hipLaunchKernelGGL(sum_u8_kernel, dim3(grid_for(n0)), dim3(kBlock), 0, st, (const unsigned char*)frames, n0, (unsigned long long*)workspace);
    PrepArgs k;
    k.x = (const unsigned char*)frames; k.sum = (const unsigned long long*)workspace; k.y = (_Float16*)out;
    k.t = t; k.h = h; k.w = w; k.hp = hp; k.wp = wp; k.to_rgb = to_rgb;
    k.m_hw3_255 = (double)m * (double)h * (double)w * 3.0 * 255.0;
    for (int c = 0; c < 3; ++c) { k.mean[c] = mean3[c]; k.stdinv[c] = (float)(1.0 / (double)std3[c]); }
    if ((long)t * hp * wp >= (1L << 31)) return LLV_E_UNSUPPORTED;
    hipLaunchKernelGGL(preprocess_kernel, dim3(grid_for((long)t * hp * wp)), dim3(kBlock), 0, st, k);
    return hipGetLastError() == hipSuccess ? LLV_OK : LLV_E_LAUNCH;
}

extern "C" int llv_upsample2x_nhwc(const void* src, void* dst, int n, int h, int w, int c, int src_ld, int dst_ld, void* stream) {
    if (!src || !dst || n < 1 || h < 1 || w < 1 || c < 8 || (c & 7) || (src_ld & 7) || (dst_ld & 7) || src_ld < c || dst_ld < c) return LLV_E_BADARG;
    const long total = (long)n * 4 * h * w * (c / 8);
    if (total >= (1L << 31)) return LLV_E_UNSUPPORTED;   // the kernel indexes with 32-bit arithmetic
    hipLaunchKernelGGL(upsample2x_kernel, dim3(grid_for(total)), dim3(kBlock), 0, (hipStream_t)stream, (const _Float16*)src, (_Float16*)dst,
                       n, h, w, c / 8, src_ld, dst_ld);
    return hipGetLastError() == hipSuccess ? LLV_OK : LLV_E_LAUNCH;
}

extern "C" int llv_yolo_decode(const float* pred, int n, int h, int w, int num_anchors, int num_attrib, int ld, int stride,
                               const float* anchors_wh, float* boxes, float* conf, void* stream) {
    if (!pred || !boxes || !conf || !anchors_wh || n < 1 || h < 1 || w < 1 || num_anchors < 1 || num_anchors > 8 || num_attrib < 5 ||
        ld < num_anchors * num_attrib || stride < 1)
        return LLV_E_BADARG;
    YoloArgs k;
    k.pred = pred; k.boxes = boxes; k.conf = conf; k.n = n; k.h = h; k.w = w; k.A = num_anchors; k.na = num_attrib; k.ld = ld; k.stride = stride;
    for (int a = 0; a < 8; ++a) { k.aw[a] = a < num_anchors ? anchors_wh[2 * a] : 0.f; k.ah[a] = a < num_anchors ? anchors_wh[2 * a + 1] : 0.f; }
    const long total = (long)n * h * w * num_anchors;
    if (total >= (1L << 31)) return LLV_E_UNSUPPORTED;   // the kernel indexes with 32-bit arithmetic
    hipLaunchKernelGGL(yolo_decode_kernel, dim3(grid_for(total)), dim3(kBlock), 0, (hipStream_t)stream, k);
    return hipGetLastError() == hipSuccess ? LLV_OK : LLV_E_LAUNCH;
}

extern "C" int llv_bilinear_resize_f32(const float* x, float* y, int nc, int h, int w, int ho, int wo, float scale_factor,
                                       float mul, void* stream) {
    if (!x || !y || nc < 1 || h < 1 || w < 1 || ho < 1 || wo < 1 || !(scale_factor > 0.f)) return LLV_E_BADARG;
    if ((int)floor((double)h * (double)scale_factor) != ho || (int)floor((double)w * (double)scale_factor) != wo) return LLV_E_BADARG;
    const long total = (long)nc * ho * wo;
    if (total >= (1L << 31)) return LLV_E_UNSUPPORTED;   // the kernel indexes with 32-bit arithmetic
    hipLaunchKernelGGL(bilinear_resize_kernel, dim3(grid_for(total)), dim3(kBlock), 0, (hipStream_t)stream, x, y, nc, h, w, ho, wo,
                       1.f / scale_factor, mul);
    return hipGetLastError() == hipSuccess ? LLV_OK : LLV_E_LAUNCH;
}

extern "C" int llv_maxpool3x3s2_nhwc(const void* x, void* y, int batch, int h, int w, int c, void* stream) {
    if (!x || !y || batch < 1 || h < 1 || w < 1 || c < 8 || (c & 7)) return LLV_E_BADARG;
    const int ho = (h + 2 - 3) / 2 + 1, wo = (w + 2 - 3) / 2 + 1;
    const long total = (long)batch * ho * wo * (c / 8);
    if (total >= (1L << 31)) return LLV_E_UNSUPPORTED;   // the kernel indexes with 32-bit arithmetic
    hipLaunchKernelGGL(maxpool3x3s2_kernel, dim3(grid_for(total)), dim3(kBlock), 0, (hipStream_t)stream,
                       (const _Float16*)x, (_Float16*)y, batch, h, w, c / 8, ho, wo);
    return hipGetLastError() == hipSuccess ? LLV_OK : LLV_E_LAUNCH;
}

// Backward of nn.MaxPool2d(3, 2, 1): dx[n, iy, ix, c] += sum over the (at most four) windows that contain (iy, ix) of dy[window] where
// (iy, ix) is the window's arg-max. Gather form (one thread per 8 input channels, no atomics). ATen's rule for ties (post-ReLU maps
// are full of equal zeros): the FIRST maximum in (dy, dx) scan order takes the gradient (`val > maxval` while scanning).
__global__ void maxpool3x3s2_bwd_kernel(const _Float16* __restrict__ x, const _Float16* __restrict__ dy, _Float16* __restrict__ dx,
                                        int batch, int h, int w, int c8, int ho, int wo, int x_ld, int dy_ld, int dx_ld) {
    const long total = (long)batch * h * w * c8;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < (unsigned)total; i += gridDim.x * blockDim.x) {
        const int cv = (int)(i % c8);
        unsigned p = i / c8;
        const int ix = (int)(p % w); p /= w;
        const int iy = (int)(p % h);
        const int n = (int)(p / h);
        const f16x8 me = *reinterpret_cast<const f16x8*>(x + (((long)n * h + iy) * w + ix) * x_ld + cv * 8);
        float acc[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = 0.f;
        // windows (oy, ox) with oy*2-1 <= iy <= oy*2+1
        const int oy0 = iy / 2, oy1 = (iy + 1) / 2, ox0 = ix / 2, ox1 = (ix + 1) / 2;
        for (int oy = oy0; oy <= oy1; ++oy) {
            if (oy >= ho) continue;
            for (int ox = ox0; ox <= ox1; ++ox) {
                if (ox >= wo) continue;
                // is (iy, ix) the first maximum of this window?  earlier positions must be strictly smaller, later ones not larger
                bool win[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) win[e] = true;
                const int my = iy - (oy * 2 - 1), mx = ix - (ox * 2 - 1);
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    const int yy = oy * 2 - 1 + ky;
                    if ((unsigned)yy >= (unsigned)h) continue;
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        const int xx = ox * 2 - 1 + kx;
                        if ((unsigned)xx >= (unsigned)w || (ky == my && kx == mx)) continue;
                        const f16x8 v = *reinterpret_cast<const f16x8*>(x + (((long)n * h + yy) * w + xx) * x_ld + cv * 8);
                        const bool earlier = ky < my || (ky == my && kx < mx);
#pragma unroll
                        for (int e = 0; e < 8; ++e) win[e] = win[e] && (earlier ? (v[e] < me[e]) : !(v[e] > me[e]));
                    }
                }
                const f16x8 g = *reinterpret_cast<const f16x8*>(dy + (((long)n * ho + oy) * wo + ox) * dy_ld + cv * 8);
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] += win[e] ? (float)g[e] : 0.f;
            }
        }
        _Float16* d = dx + (((long)n * h + iy) * w + ix) * dx_ld + cv * 8;
        f16x8 o = *reinterpret_cast<const f16x8*>(d);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (_Float16)((float)o[e] + acc[e]);
        *reinterpret_cast<f16x8*>(d) = o;
    }
}

extern "C" int llv_maxpool3x3s2_nhwc_bwd(const void* x, const void* dy, void* dx, int batch, int h, int w, int c, int x_ld, int dy_ld, int dx_ld,
                                         void* stream) {
    if (!x || !dy || !dx || batch < 1 || h < 1 || w < 1 || c < 8 || (c & 7) || x_ld < c || dy_ld < c || dx_ld < c || ((x_ld | dy_ld | dx_ld) & 7))
        return LLV_E_BADARG;
    const int ho = (h + 2 - 3) / 2 + 1, wo = (w + 2 - 3) / 2 + 1;
    const long total = (long)batch * h * w * (c / 8);
    if (total >= (1L << 31)) return LLV_E_UNSUPPORTED;
    hipLaunchKernelGGL(maxpool3x3s2_bwd_kernel, dim3(grid_for(total)), dim3(kBlock), 0, (hipStream_t)stream, (const _Float16*)x, (const _Float16*)dy,
                       (_Float16*)dx, batch, h, w, c / 8, ho, wo, x_ld, dy_ld, dx_ld);
    return hipGetLastError() == hipSuccess ? LLV_OK : LLV_E_LAUNCH;
}

// The ConvLSTM update of DarkResNet's DenoisingBlock (mmtracking/mmtrack/models/backbones/dark_resnet.py:226-246) for one frame:
// gates (rows, 4P) fp16 = conv2([x_t | h_{t-1}]) split along channels into i | f | o | g, ALL FOUR through a sigmoid (the reference
// applies sigmoid to g as well, :240); c <- f * c + i * g (fp32 state, in place), h = o * tanh(c). Two outputs: h (fp16, the next
// frame's recurrent input) and act = relu(h * scale + shift) (fp16: norm2 + ReLU of the block, :267-269, eval-mode BN as an affine).
__global__ void convlstm_gate_kernel(const _Float16* __restrict__ gates, float* __restrict__ c, const float* __restrict__ scale,
                                     const float* __restrict__ shift, _Float16* __restrict__ h, _Float16* __restrict__ act, long rows, int p8,
                                     int gates_ld, int h_ld, int act_ld) {
    const long total = rows * p8;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / p8;
        const int cv = (int)(i - r * p8) * 8;
        const int P = p8 * 8;
        const _Float16* gr = gates + r * gates_ld + cv;
        const f16x8 gi = *reinterpret_cast<const f16x8*>(gr), gf = *reinterpret_cast<const f16x8*>(gr + P);
        const f16x8 go = *reinterpret_cast<const f16x8*>(gr + 2 * P), gg = *reinterpret_cast<const f16x8*>(gr + 3 * P);
        float* cr = c + r * P + cv;
        f16x8 ho, ao;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float si = 1.f / (1.f + expf(-(float)gi[e])), sf = 1.f / (1.f + expf(-(float)gf[e]));
            const float so = 1.f / (1.f + expf(-(float)go[e])), sg = 1.f / (1.f + expf(-(float)gg[e]));
            const float cn = sf * cr[e] + si * sg;
            cr[e] = cn;
            const float hv = so * tanhf(cn);
            ho[e] = (_Float16)hv;
            const float a = hv * scale[cv + e] + shift[cv + e];
            ao[e] = (_Float16)(a > 0.f ? a : 0.f);
        }
        *reinterpret_cast<f16x8*>(h + r * h_ld + cv) = ho;
        *reinterpret_cast<f16x8*>(act + r * act_ld + cv) = ao;
    }
}

extern "C" int llv_convlstm_gate(const void* gates, float* c, const float* scale, const float* shift, void* h, void* act, int64_t rows, int p,
                                 int gates_ld, int h_ld, int act_ld, void* stream) {
    if (!gates || !c || !scale || !shift || !h || !act || rows < 1 || p < 8 || (p & 7) || gates_ld < 4 * p || h_ld < p || act_ld < p ||
        ((gates_ld | h_ld | act_ld) & 7))
        return LLV_E_BADARG;
    const long total = rows * (p / 8);
    hipLaunchKernelGGL(convlstm_gate_kernel, dim3(grid_for(total)), dim3(kBlock), 0, (hipStream_t)stream, (const _Float16*)gates, c, scale, shift,
                       (_Float16*)h, (_Float16*)act, (long)rows, p / 8, gates_ld, h_ld, act_ld);
    return hipGetLastError() == hipSuccess ? LLV_OK : LLV_E_LAUNCH;
}

// Backward of convlstm_gate_kernel. Per element, with the forward's si, sf, so, sg (sigmoid of the four gate blocks -- the reference puts
// the sigmoid on g as well), c_prev, cn = sf c_prev + si sg, hv = so tanh(cn), a = hv scale + shift:
//   d(hv) = dh (the recurrent consumer's gradient, the next frame's conv2) + (a > 0 ? dact * scale : 0)
//   d(cn) = dc (from the next frame: sf_next * its d(cn)) + d(hv) * so * (1 - tanh(cn)^2)
//   dgates = (d(cn) sg si(1-si), d(cn) c_prev sf(1-sf), d(hv) tanh(cn) so(1-so), d(cn) si sg(1-sg)); dc_prev = d(cn) * sf
//   dscale[ch] += (a > 0) dact hv, dshift[ch] += (a > 0) dact (the affine of norm2: d(gamma) = dscale / sqrt(var + eps), d(beta) = dshift)
// dh / dc_in may be null (the last frame). dc_prev may alias dc_in. dgates fp16 (the conv's gradient operand), the state gradients fp32.
__global__ void convlstm_gate_bwd_kernel(const _Float16* __restrict__ gates, const float* __restrict__ c_prev, const float* __restrict__ scale,
                                         const float* __restrict__ shift, const _Float16* __restrict__ dh, const _Float16* __restrict__ dact,
                                         const float* dc_in, _Float16* __restrict__ dgates, float* dc_prev, float* __restrict__ dscale,
                                         float* __restrict__ dshift, long rows, int p8, int gates_ld, int h_ld, int act_ld) {
    const long total = rows * p8;
    const int P = p8 * 8;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / p8;
        const int cv = (int)(i - r * p8) * 8;
        const _Float16* gr = gates + r * gates_ld + cv;
        const f16x8 gi = *reinterpret_cast<const f16x8*>(gr), gf = *reinterpret_cast<const f16x8*>(gr + P);
        const f16x8 go = *reinterpret_cast<const f16x8*>(gr + 2 * P), gg = *reinterpret_cast<const f16x8*>(gr + 3 * P);
        const f16x8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
        const f16x8 dhv = dh ? *reinterpret_cast<const f16x8*>(dh + r * h_ld + cv) : zero;
        const f16x8 dav = *reinterpret_cast<const f16x8*>(dact + r * act_ld + cv);
        f16x8 di, df, dO, dG;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float si = 1.f / (1.f + expf(-(float)gi[e])), sf = 1.f / (1.f + expf(-(float)gf[e]));
            const float so = 1.f / (1.f + expf(-(float)go[e])), sg = 1.f / (1.f + expf(-(float)gg[e]));
            const float cp = c_prev[r * P + cv + e];
            const float cn = sf * cp + si * sg;
            const float th = tanhf(cn);
            const float hv = so * th;
            const float a = hv * scale[cv + e] + shift[cv + e];
            const float da = a > 0.f ? (float)dav[e] : 0.f;
            if (dscale && da != 0.f) { atomicAdd(dscale + cv + e, da * hv); atomicAdd(dshift + cv + e, da); }
            const float dhv_ = (float)dhv[e] + da * scale[cv + e];
            const float dcn = (dc_in ? dc_in[r * P + cv + e] : 0.f) + dhv_ * so * (1.f - th * th);
            di[e] = (_Float16)(dcn * sg * si * (1.f - si));
            df[e] = (_Float16)(dcn * cp * sf * (1.f - sf));
            dO[e] = (_Float16)(dhv_ * th * so * (1.f - so));
            dG[e] = (_Float16)(dcn * si * sg * (1.f - sg));
            dc_prev[r * P + cv + e] = dcn * sf;
        }
        _Float16* dr = dgates + r * gates_ld + cv;
        *reinterpret_cast<f16x8*>(dr) = di;
        *reinterpret_cast<f16x8*>(dr + P) = df;
        *reinterpret_cast<f16x8*>(dr + 2 * P) = dO;
        *reinterpret_cast<f16x8*>(dr + 3 * P) = dG;
    }
}

extern "C" int llv_convlstm_gate_bwd(const void* gates, const float* c_prev, const float* scale, const float* shift, const void* dh, const void* dact,
                                     const float* dc_in, void* dgates, float* dc_prev, float* dscale, float* dshift, int64_t rows, int p,
                                     int gates_ld, int h_ld, int act_ld, void* stream) {
    if (!gates || !c_prev || !scale || !shift || !dact || !dgates || !dc_prev || rows < 1 || p < 8 || (p & 7) || gates_ld < 4 * p || h_ld < p ||
        act_ld < p || ((gates_ld | h_ld | act_ld) & 7) || ((dscale == nullptr) != (dshift == nullptr)))
        return LLV_E_BADARG;
    const long total = rows * (p / 8);
    hipLaunchKernelGGL(convlstm_gate_bwd_kernel, dim3(grid_for(total)), dim3(kBlock), 0, (hipStream_t)stream, (const _Float16*)gates, c_prev, scale,
                       shift, (const _Float16*)dh, (const _Float16*)dact, dc_in, (_Float16*)dgates, dc_prev, dscale, dshift, (long)rows, p / 8,
                       gates_ld, h_ld, act_ld);
    return hipGetLastError() == hipSuccess ? LLV_OK : LLV_E_LAUNCH;
}

// y[n][oy][ox][:] = x[n][2 oy][2 ox][:]: the pixels a 1x1 / stride-2 convolution reads, as a dense map (the downsample conv of a ResNet
// stage's first block then shares its launch with conv3: engine.build_resnet). 16 bytes per lane.
__global__ void subsample2_kernel(const _Float16* __restrict__ x, _Float16* __restrict__ y, int n, int h, int w, int c8, int ho, int wo, int x_ld,
                                  int y_ld) {
    const long total = (long)n * ho * wo * c8;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < (unsigned)total; i += gridDim.x * blockDim.x) {
        const int cv = (int)(i % c8);
        unsigned p = i / c8;
        const int ox = (int)(p % wo); p /= wo;
        const int oy = (int)(p % ho);
        const int b = (int)(p / ho);
        *reinterpret_cast<f16x8*>(y + (((long)b * ho + oy) * wo + ox) * y_ld + cv * 8) =
            *reinterpret_cast<const f16x8*>(x + (((long)b * h + 2 * oy) * w + 2 * ox) * x_ld + cv * 8);
    }
}

extern "C" int llv_subsample2_nhwc(const void* x, void* y, int n, int h, int w, int c, int x_ld, int y_ld, void* stream) {
    if (!x || !y || n < 1 || h < 1 || w < 1 || c < 8 || (c & 7) || x_ld < c || y_ld < c || ((x_ld | y_ld) & 7)) return LLV_E_BADARG;
    const int ho = (h - 1) / 2 + 1, wo = (w - 1) / 2 + 1;
    const long total = (long)n * ho * wo * (c / 8);
    if (total >= (1L << 31)) return LLV_E_UNSUPPORTED;
    hipLaunchKernelGGL(subsample2_kernel, dim3(grid_for(total)), dim3(kBlock), 0, (hipStream_t)stream, (const _Float16*)x, (_Float16*)y, n, h, w, c / 8,
                       ho, wo, x_ld, y_ld);
    return hipGetLastError() == hipSuccess ? LLV_OK : LLV_E_LAUNCH;
}

extern "C" int llv_pack_nchw_f32_to_nhwc_f16(const float* x, void* y, int batch, int c, int h, int w, int c_pad,
                                             const float* scale, const float* shift, void* stream) {
    if (!x || !y || batch < 1 || c < 1 || h < 1 || w < 1 || c_pad < c || (c_pad & 7)) return LLV_E_BADARG;
    if ((scale == nullptr) != (shift == nullptr)) return LLV_E_BADARG;
    const long total = (long)batch * h * w;
    if (total >= (1L << 31)) return LLV_E_UNSUPPORTED;   // the kernel indexes with 32-bit arithmetic
    hipLaunchKernelGGL(pack_kernel, dim3(grid_for(total)), dim3(kBlock), 0, (hipStream_t)stream, x, (_Float16*)y,
                       batch, c, (long)h * w, c_pad, scale, shift);
    return hipGetLastError() == hipSuccess ? LLV_OK : LLV_E_LAUNCH;
}

extern "C" int llv_unpack_nhwc_f16_to_nchw_f32(const void* x, float* y, int batch, int c, int h, int w, int ld,
                                               void* stream) {
    if (!x || !y || batch < 1 || c < 1 || h < 1 || w < 1 || ld < c) return LLV_E_BADARG;
    const long total = (long)batch * c * h * w;
    if (total >= (1L << 31)) return LLV_E_UNSUPPORTED;   // the kernel indexes with 32-bit arithmetic
    hipLaunchKernelGGL((unpack_kernel<_Float16>), dim3(grid_for(total)), dim3(kBlock), 0, (hipStream_t)stream,
                       (const _Float16*)x, y, batch, c, (long)h * w, ld);
    return hipGetLastError() == hipSuccess ? LLV_OK : LLV_E_LAUNCH;
}

extern "C" int llv_unpack_nhwc_f32_to_nchw_f32(const float* x, float* y, int batch, int c, int h, int w, int ld,
                                               void* stream) {
    if (!x || !y || batch < 1 || c < 1 || h < 1 || w < 1 || ld < c) return LLV_E_BADARG;
    const long total = (long)batch * c * h * w;
    if (total >= (1L << 31)) return LLV_E_UNSUPPORTED;   // the kernel indexes with 32-bit arithmetic
    hipLaunchKernelGGL((unpack_kernel<float>), dim3(grid_for(total)), dim3(kBlock), 0, (hipStream_t)stream, x, y,
                       batch, c, (long)h * w, ld);
    return hipGetLastError() == hipSuccess ? LLV_OK : LLV_E_LAUNCH;
}

extern "C" int llv_embed_aggregate(const void* x_embed, const void* ref_embed, const void* ref, void* out,
                                   int n_ref, int64_t hw, int c, int embed_ld, int ref_ld, int out_ld,
                                   void* stream) {
    if (!x_embed || !ref_embed || !ref || !out || n_ref < 1 || n_ref > 256 || hw < 1 || c < 8 || (c & 7))
        return LLV_E_BADARG;
    if ((embed_ld & 7) || (ref_ld & 7) || (out_ld & 7) || embed_ld < c || ref_ld < c || out_ld < c) return LLV_E_BADARG;
    long waves = hw;
    long blocks = (waves + 3) / 4;
    if (blocks > 256L * 8) blocks = 256L * 8;
    hipLaunchKernelGGL(embed_aggregate_kernel, dim3((int)blocks), dim3(256), 4 * n_ref * sizeof(float),
                       (hipStream_t)stream, (const _Float16*)x_embed, (const _Float16*)ref_embed,
                       (const _Float16*)ref, (_Float16*)out, n_ref, (long)hw, c, embed_ld, ref_ld, out_ld, -1);
    return hipGetLastError() == hipSuccess ? LLV_OK : LLV_E_LAUNCH;
}

extern "C" int llv_embed_aggregate_ring(const void* embed, const void* feat, void* out, int n_slots, int key_slot,
                                        int64_t hw, int c, int embed_ld, int feat_ld, int out_ld, void* stream) {
    if (!embed || !feat || !out || n_slots < 2 || n_slots > 256 || key_slot < 0 || key_slot >= n_slots || hw < 1 || c < 8 || (c & 7))
        return LLV_E_BADARG;
    if ((embed_ld & 7) || (feat_ld & 7) || (out_ld & 7) || embed_ld < c || feat_ld < c || out_ld < c) return LLV_E_BADARG;
    long blocks = ((long)hw + 3) / 4;
    if (blocks > 256L * 8) blocks = 256L * 8;
    hipLaunchKernelGGL(embed_aggregate_kernel, dim3((int)blocks), dim3(256), 4 * n_slots * sizeof(float), (hipStream_t)stream,
                       (const _Float16*)embed, (const _Float16*)embed, (const _Float16*)feat, (_Float16*)out, n_slots, (long)hw,
                       c, embed_ld, feat_ld, out_ld, key_slot);
    return hipGetLastError() == hipSuccess ? LLV_OK : LLV_E_LAUNCH;
}

extern "C" int llv_embed_aggregate_bwd(const void* embed, const void* feat, const void* grad_out, void* d_embed, void* d_feat,
                                       int n_slots, int key_slot, int64_t hw, int c, int embed_ld, int feat_ld, int gout_ld,
                                       void* stream) {
    if (!embed || !feat || !grad_out || !d_embed || !d_feat || n_slots < 2 || n_slots > 256 || key_slot < 0 || key_slot >= n_slots ||
        hw < 1 || c < 8 || (c & 7))
        return LLV_E_BADARG;
    if ((embed_ld & 7) || (feat_ld & 7) || (gout_ld & 7) || embed_ld < c || feat_ld < c || gout_ld < c) return LLV_E_BADARG;
    long blocks = ((long)hw + 3) / 4;
    if (blocks > 256L * 8) blocks = 256L * 8;
    hipLaunchKernelGGL(embed_aggregate_bwd_kernel, dim3((int)blocks), dim3(256), 4 * 3 * n_slots * sizeof(float), (hipStream_t)stream,
                       (const _Float16*)embed, (const _Float16*)feat, (const _Float16*)grad_out, (_Float16*)d_embed, (_Float16*)d_feat,
                       n_slots, key_slot, (long)hw, c, embed_ld, feat_ld, gout_ld);
    return hipGetLastError() == hipSuccess ? LLV_OK : LLV_E_LAUNCH;
}

extern "C" int llv_flow_warp_bwd(const void* dy, const float* flow, float* dx, int n, int h, int w, int c, int ld, int hf, int wf,
                                 void* stream) {
    if (!dy || !flow || !dx || n < 1 || h < 1 || w < 1 || c < 1 || ld < c || hf < 1 || wf < 1) return LLV_E_BADARG;
    const float s = (float)w / (float)wf;
    if ((int)floor((double)hf * (double)s) != h || (int)floor((double)wf * (double)s) != w) return LLV_E_BADARG;
    const float rs = 1.f / s;
    const long total = (long)n * h * w * c;
    if (total >= (1L << 31)) return LLV_E_UNSUPPORTED;
    hipLaunchKernelGGL(flow_warp_bwd_kernel, dim3(grid_for(total)), dim3(kBlock), 0, (hipStream_t)stream, (const _Float16*)dy, flow, dx,
                       n, h, w, c, ld, hf, wf, rs, rs, s);
    return hipGetLastError() == hipSuccess ? LLV_OK : LLV_E_LAUNCH;
}

extern "C" int llv_add_f32_to_f16(const float* src, void* dst, int64_t rows, int c, int ld, void* stream) {
    if (rows < 0 || c < 1 || ld < c) return LLV_E_BADARG;
    if (rows == 0) return LLV_OK;
    if (!src || !dst) return LLV_E_BADARG;
    hipLaunchKernelGGL(add_f32_to_f16_kernel, dim3(grid_for(rows * c)), dim3(kBlock), 0, (hipStream_t)stream, src, (_Float16*)dst,
                       (long)rows, c, ld);
    return hipGetLastError() == hipSuccess ? LLV_OK : LLV_E_LAUNCH;
}

extern "C" int llv_flow_warp_bwd_flow(const void* dy, const void* x, const float* flow, float* dflow, int n, int h, int w, int c, int dy_ld,
                                      int x_ld, int hf, int wf, void* stream) {
    if (!dy || !x || !flow || !dflow || n < 1 || h < 1 || w < 1 || c < 1 || dy_ld < c || x_ld < c || hf < 1 || wf < 1) return LLV_E_BADARG;
    const float s = (float)w / (float)wf;
    if ((int)floor((double)hf * (double)s) != h || (int)floor((double)wf * (double)s) != w) return LLV_E_BADARG;
    const long waves = (long)n * h * w;
    long grid = (waves + 3) / 4;                                 // four wavefronts per workgroup
    grid = grid > 256L * 16 ? 256L * 16 : grid;
    hipLaunchKernelGGL(flow_warp_bwd_flow_kernel, dim3((int)grid), dim3(256), 0, (hipStream_t)stream, (const _Float16*)dy, (const _Float16*)x, flow,
                       dflow, n, h, w, c, dy_ld, x_ld, hf, wf, 1.f / s, s);
    return hipGetLastError() == hipSuccess ? LLV_OK : LLV_E_LAUNCH;
}

extern "C" int llv_bilinear_resize_f32_bwd(const float* dy, float* dx, int nc, int h, int w, int ho, int wo, float scale_factor, float mul,
                                           void* stream) {
    if (!dy || !dx || nc < 1 || h < 1 || w < 1 || ho < 1 || wo < 1 || !(scale_factor > 0.f)) return LLV_E_BADARG;
    if ((int)floor((double)h * (double)scale_factor) != ho || (int)floor((double)w * (double)scale_factor) != wo) return LLV_E_BADARG;
    const long total = (long)nc * ho * wo;
    if (total >= (1L << 31)) return LLV_E_UNSUPPORTED;
    hipLaunchKernelGGL(bilinear_resize_bwd_kernel, dim3(grid_for(total)), dim3(kBlock), 0, (hipStream_t)stream, dy, dx, nc, h, w, ho, wo,
                       1.f / scale_factor, mul);
    return hipGetLastError() == hipSuccess ? LLV_OK : LLV_E_LAUNCH;
}

extern "C" int llv_flow_warp(const void* x, const float* flow, void* y, int n, int h, int w, int c, int ld,
                             int hf, int wf, void* stream) {
    if (!x || !flow || !y || n < 1 || h < 1 || w < 1 || c < 8 || (c & 7) || (ld & 7) || ld < c || hf < 1 || wf < 1)
        return LLV_E_BADARG;
    // F.interpolate(scale_factor=s) with s = W_x / W_f: output size floor(in*s); the reference relies on
    // it matching (h,w). recompute_scale_factor is None -> the given scale_factor is used for the
    // coordinate map: src = (dst + 0.5)/s - 0.5 for both axes.
    const float s = (float)w / (float)wf;
    if ((int)floor((double)hf * (double)s) != h || (int)floor((double)wf * (double)s) != w) return LLV_E_BADARG;
    const float rs = 1.f / s;
    const long total = (long)n * h * w * (c / 8);
    if (total >= (1L << 31)) return LLV_E_UNSUPPORTED;   // the kernel indexes with 32-bit arithmetic
    hipLaunchKernelGGL(flow_warp_kernel, dim3(grid_for(total)), dim3(kBlock), 0, (hipStream_t)stream,
                       (const _Float16*)x, flow, (_Float16*)y, n, h, w, c / 8, ld, hf, wf, rs, rs, s);
    return hipGetLastError() == hipSuccess ? LLV_OK : LLV_E_LAUNCH;
}

// A copy done by a KERNEL (16 bytes per lane, byte tail by the first lanes): inside a captured step it is a kernel node like
// every other launch of the plan, where hipMemcpyAsync would insert a memcpy node (the step graph is
// kept homogeneous). dst may be host memory that is mapped for the device (torch pinned memory).
namespace {
__global__ void copy_kernel(const unsigned char* __restrict__ src, unsigned char* __restrict__ dst, size_t n16, size_t bytes) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n16; i += stride)
        reinterpret_cast<uint4*>(dst)[i] = reinterpret_cast<const uint4*>(src)[i];
    for (size_t i = n16 * 16 + blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < bytes; i += stride) dst[i] = src[i];
}

// (dets [batch][max_num][5] f32, labels [batch][max_num] i64, count [batch] i32) -> rows [batch][max_num][6] f32 (x1, y1, x2, y2,
// score, label) + counts [batch] i32, e.g. straight into pinned host memory: the D2H step of bbox2result
// (mmdetection/mmdet/core/bbox/transforms.py:99-116) as one kernel node
__global__ void pack_results_kernel(const float* __restrict__ dets, const long* __restrict__ labels, const int* __restrict__ count,
                                    int total_rows, int batch, float* __restrict__ rows, int* __restrict__ count_out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < total_rows) {
        const float* d = dets + (size_t)i * 5;
        float* o = rows + (size_t)i * 6;
        o[0] = d[0]; o[1] = d[1]; o[2] = d[2]; o[3] = d[3]; o[4] = d[4];
        o[5] = (float)labels[i];
    }
    if (i < batch) count_out[i] = count[i];
}
}  // namespace

extern "C" int llv_copy_d2d(void* dst, const void* src, size_t bytes, void* stream) {
    if (!dst || !src) return LLV_E_BADARG;
    if (bytes == 0) return LLV_OK;
    // LLV_GRAPH_MEMNODES=1 (diagnosis only): round 1's behaviour, a memcpy node inside the captured step
    static const bool memnodes = getenv("LLV_GRAPH_MEMNODES") != nullptr && atoi(getenv("LLV_GRAPH_MEMNODES")) != 0;
    if (memnodes) return hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream) == hipSuccess ? LLV_OK : LLV_E_LAUNCH;
    const bool aligned = ((((uintptr_t)dst) | ((uintptr_t)src)) & 15) == 0;
    const size_t n16 = aligned ? bytes / 16 : 0;
    hipLaunchKernelGGL(copy_kernel, dim3(grid_for((long)(n16 ? n16 : bytes))), dim3(kBlock), 0, (hipStream_t)stream,
                       (const unsigned char*)src, (unsigned char*)dst, n16, bytes);
    return hipGetLastError() == hipSuccess ? LLV_OK : LLV_E_LAUNCH;
}

extern "C" int llv_pack_results(const float* dets, const int64_t* labels, const int32_t* count, int batch, int max_num, float* rows,
                                int32_t* count_out, void* stream) {
    if (!dets || !labels || !count || !rows || !count_out || batch < 1 || max_num < 1) return LLV_E_BADARG;
    const int total = batch * max_num;
    hipLaunchKernelGGL(pack_results_kernel, dim3((total + kBlock - 1) / kBlock), dim3(kBlock), 0, (hipStream_t)stream, dets,
                       (const long*)labels, count, total, batch, rows, count_out);
    return hipGetLastError() == hipSuccess ? LLV_OK : LLV_E_LAUNCH;
}

extern "C" int llv_abi_version(void) { return LLV_ABI_VERSION; }

extern "C" const char* llv_strerror(int code) {
    switch (code) {
        case LLV_OK: return "ok";
        case LLV_E_BADARG: return "bad argument";
        case LLV_E_LAUNCH: return "kernel launch failed";
        case LLV_E_UNSUPPORTED: return "unsupported configuration";
        case LLV_E_WORKSPACE: return "workspace too small";
        case LLV_E_EXTENT: return "buffer extent exceeded";
        default: return "unknown error";
    }
}
