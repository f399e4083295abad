// dcn.hip -- modulated deformable convolution (DCNv2), column step, for the authors' temporal fusion
// (mmtracking/mmtrack/models/aggregators/denoising_aggregator.py:69-111,
// ModulatedDCNPack.forward -> mmcv.ops.modulated_deform_conv2d; mmcv-full 1.3.0 is un-vendored, its
// modulated_deformable_im2col / dmcn_im2col_bilinear are restated in oracle/dcn.py).
//
//   out[n, co, y, x] = bias[co] + sum_{k=(i,j), c} W[co, c, i, j] * m[n, g(c), k, y, x] * bilinear(x[n, c], py, px)
//   py = y*stride - pad + i*dilation + dy[n, g(c), k, y, x],  px likewise with dx
//
// The product runs it as  col = llv_modulated_deform_im2col(x, offsets/mask)  followed by the ordinary implicit-GEMM
// kernel as a 1x1 convolution over col (K = kh*kw*C, the same tap-major / channel-fastest packing a 3x3 weight has).
// The column tensor is (n, ho, wo, kh*kw*C) fp16: channel (k*C + c). One thread = 8 channels of one tap of one pixel:
// three fp32 reads (dy, dx, mask logit), four 16-byte corner reads, one 16-byte write; consecutive lanes write
// consecutive bytes. HBM-bound: per output pixel kh*kw*C*2 B written + 3*dg*kh*kw*4 B read (the input corners hit L2).
//
// `om` is the raw conv_offset output in fp32 NHWC (n, ho, wo, om_ld >= 3*dg*K): the reference splits it with
// torch.chunk(out, 3, dim=1) into o1, o2, mask and uses offset = cat(o1, o2) = channels [0, 2*dg*K); mmcv reads
// offset channel (g*2K + 2k) as dy and (g*2K + 2k + 1) as dx, mask channel (g*K + k) (sigmoid applied here).
#include <hip/hip_runtime.h>
#include "llv.h"

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

struct MdcnItem {          // one (image, pixel, tap, 8-channel vector): where to sample, what to scale with, where to write
    const _Float16* xb;
    _Float16* dst;
    float py, px, m;
};

// row = img*ho + oy (blockIdx.y), t = index inside the row: ((ox*K + k)*c8 + cv). 32-bit arithmetic only: 64-bit division by
// run-time values costs hundreds of instructions per item and made the first version of this kernel ALU-bound.
__device__ __forceinline__ MdcnItem mdcn_locate(unsigned row, unsigned t, const _Float16* x, const float* om, _Float16* col, int h,
                                                int w, unsigned c8, int x_ld, int om_ld, int col_ld, unsigned kw, unsigned K,
                                                int stride, int pad, int dilation, int dg, unsigned cg8, unsigned ho, int wo) {
    const unsigned cv = t % c8;
    const unsigned p = t / c8;
    const unsigned ox = p / K, k = p - ox * K;
    const unsigned img = row / ho, oy = row - img * ho;
    const unsigned g = cv / cg8;
    const float* o = om + ((long)row * wo + ox) * om_ld;
    const float2 d = *reinterpret_cast<const float2*>(o + (g * K + k) * 2);     // (dy, dx): om_ld is even, the pair 8-byte aligned
    const float ml = o[2 * dg * K + g * K + k];
    const unsigned ki = k / kw, kj = k - ki * kw;
    MdcnItem it;
    it.m = 1.f / (1.f + __expf(-ml));
    it.py = (float)((int)oy * stride - pad + (int)ki * dilation) + d.x;
    it.px = (float)((int)ox * stride - pad + (int)kj * dilation) + d.y;
    it.xb = x + (long)img * h * w * x_ld + cv * 8;
    it.dst = col + ((long)row * wo + ox) * col_ld + (k * c8 + cv) * 8;
    return it;
}

__device__ __forceinline__ void mdcn_sample(const MdcnItem& it, int h, int w, int x_ld) {
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const float py = it.py, px = it.px;
    if (py > -1.f && px > -1.f && py < (float)h && px < (float)w) {
        const int y0 = (int)floorf(py), x0 = (int)floorf(px);
        const int y1 = y0 + 1, x1 = x0 + 1;
        const float ly = py - (float)y0, lx = px - (float)x0;
        const float hy = 1.f - ly, hx = 1.f - lx;
        // the four corners are requested together (a corner outside the image reads pixel (0,0) with weight zero)
        const bool v00 = y0 >= 0 && x0 >= 0, v01 = y0 >= 0 && x1 <= w - 1, v10 = y1 <= h - 1 && x0 >= 0, v11 = y1 <= h - 1 && x1 <= w - 1;
        const f16x8 c00 = *reinterpret_cast<const f16x8*>(it.xb + (v00 ? ((long)y0 * w + x0) * x_ld : 0));
        const f16x8 c01 = *reinterpret_cast<const f16x8*>(it.xb + (v01 ? ((long)y0 * w + x1) * x_ld : 0));
        const f16x8 c10 = *reinterpret_cast<const f16x8*>(it.xb + (v10 ? ((long)y1 * w + x0) * x_ld : 0));
        const f16x8 c11 = *reinterpret_cast<const f16x8*>(it.xb + (v11 ? ((long)y1 * w + x1) * x_ld : 0));
        const float w00 = v00 ? hy * hx : 0.f, w01 = v01 ? hy * lx : 0.f, w10 = v10 ? ly * hx : 0.f, w11 = v11 ? ly * lx : 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e)
            acc[e] = w00 * (float)c00[e] + w01 * (float)c01[e] + w10 * (float)c10[e] + w11 * (float)c11[e];
    }
    f16x8 r;
#pragma unroll
    for (int e = 0; e < 8; ++e) r[e] = (_Float16)(acc[e] * it.m);
    __builtin_nontemporal_store(r, reinterpret_cast<f16x8*>(it.dst));
}

__global__ __launch_bounds__(256) void mdcn_im2col_kernel(const _Float16* __restrict__ x, const float* __restrict__ om,
                                                         _Float16* __restrict__ col, int h, int w, int c8, int x_ld,
                                                         int om_ld, int col_ld, int kh, int kw, int stride, int pad,
                                                         int dilation, int dg, int cg8, int ho, int wo) {
    const unsigned K = kh * kw;
    const unsigned total = (unsigned)wo * K * c8;              // items of one output row
    const unsigned step = gridDim.x * blockDim.x;
    const unsigned row = blockIdx.y;
    unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
    // two items per iteration: both offset reads, then both sets of corner reads, are in flight together
    for (; t + step < total; t += 2 * step) {
        const MdcnItem a = mdcn_locate(row, t, x, om, col, h, w, c8, x_ld, om_ld, col_ld, kw, K, stride, pad, dilation, dg, cg8, ho, wo);
        const MdcnItem b = mdcn_locate(row, t + step, x, om, col, h, w, c8, x_ld, om_ld, col_ld, kw, K, stride, pad, dilation, dg, cg8, ho, wo);
        mdcn_sample(a, h, w, x_ld);
        mdcn_sample(b, h, w, x_ld);
    }
    if (t < total)
        mdcn_sample(mdcn_locate(row, t, x, om, col, h, w, c8, x_ld, om_ld, col_ld, kw, K, stride, pad, dilation, dg, cg8, ho, wo), h, w, x_ld);
}


// ---- backward of the column step (mmcv modulated_deformable_col2im_gpu_kernel + modulated_deformable_col2im_coord_gpu_kernel in ONE
// pass): one thread = the forward's item (pixel, tap k, 8 channels). Reads the column gradient g[8], re-derives the sample position and
// the four corners, then
//   dx[corner][c]          += g[c] * m * w_corner                                   (32 fp32 atomics at most)
//   d(dy), d(dx) of the tap += sum_c g[c] * m * d(bilinear)/d(py, px)                (dmcn_get_coordinate_weight: -hx v00 - lx v01 + hx v10 + lx v11,
//                                                                                     -hy v00 + hy v01 - ly v10 + ly v11; a corner outside the image is 0)
//   d(mask logit)           += sum_c g[c] * bilinear(x)[c] * m (1 - m)               (mmcv returns d/d(mask); the sigmoid is the caller's in the reference
//                                                                                     and is folded in here, as the forward folds it)
// A sample outside (-1, h) x (-1, w) contributes nothing, as in the forward. dx fp32 (n, h, w, c), dom fp32 (n, ho, wo, om_ld): both accumulated.
__global__ __launch_bounds__(256) void mdcn_col2im_kernel(const _Float16* __restrict__ dcol, const _Float16* __restrict__ x,
                                                         const float* __restrict__ om, float* __restrict__ dx, float* __restrict__ dom, int h, int w,
                                                         int c8, int x_ld, int om_ld, int col_ld, int kh, int kw, int stride, int pad, int dilation,
                                                         int dg, int cg8, int ho, int wo) {
    const unsigned K = kh * kw;
    const unsigned total = (unsigned)wo * K * c8;
    const unsigned row = blockIdx.y;
    const int c = c8 * 8;
    for (unsigned t = blockIdx.x * blockDim.x + threadIdx.x; t < total; t += gridDim.x * blockDim.x) {
        const unsigned cv = t % (unsigned)c8;
        const unsigned p = t / (unsigned)c8;
        const unsigned ox = p / K, k = p - ox * K;
        const unsigned img = row / (unsigned)ho, oy = row - img * ho;
        const unsigned g = cv / (unsigned)cg8;
        const long pix = (long)row * wo + ox;
        const float* o = om + pix * om_ld;
        const float2 d = *reinterpret_cast<const float2*>(o + (g * K + k) * 2);
        const float ml = o[2 * dg * K + g * K + k];
        const unsigned ki = k / (unsigned)kw, kj = k - ki * kw;
        const float m = 1.f / (1.f + __expf(-ml));
        const float py = (float)((int)oy * stride - pad + (int)ki * dilation) + d.x;
        const float px = (float)((int)ox * stride - pad + (int)kj * dilation) + d.y;
        if (!(py > -1.f && px > -1.f && py < (float)h && px < (float)w)) continue;
        const f16x8 gv = *reinterpret_cast<const f16x8*>(dcol + pix * col_ld + (k * c8 + cv) * 8);
        const int y0 = (int)floorf(py), x0 = (int)floorf(px);
        const int y1 = y0 + 1, x1 = x0 + 1;
        const float ly = py - (float)y0, lx = px - (float)x0, hy = 1.f - ly, hx = 1.f - lx;
        const bool v00 = y0 >= 0 && x0 >= 0, v01 = y0 >= 0 && x1 <= w - 1, v10 = y1 <= h - 1 && x0 >= 0, v11 = y1 <= h - 1 && x1 <= w - 1;
        const _Float16* xb = x + (long)img * h * w * x_ld + cv * 8;
        const f16x8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
        const f16x8 c00 = v00 ? *reinterpret_cast<const f16x8*>(xb + ((long)y0 * w + x0) * x_ld) : zero;
        const f16x8 c01 = v01 ? *reinterpret_cast<const f16x8*>(xb + ((long)y0 * w + x1) * x_ld) : zero;
        const f16x8 c10 = v10 ? *reinterpret_cast<const f16x8*>(xb + ((long)y1 * w + x0) * x_ld) : zero;
        const f16x8 c11 = v11 ? *reinterpret_cast<const f16x8*>(xb + ((long)y1 * w + x1) * x_ld) : zero;
        float* db = dx + (long)img * h * w * c + cv * 8;
        float gy = 0.f, gx = 0.f, gm = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float ge = (float)gv[e];
            const float a = (float)c00[e], b = (float)c01[e], cc = (float)c10[e], dd = (float)c11[e];
            gy += ge * (-hx * a - lx * b + hx * cc + lx * dd);
            gx += ge * (-hy * a + hy * b - ly * cc + ly * dd);
            gm += ge * (hy * hx * a + hy * lx * b + ly * hx * cc + ly * lx * dd);
            const float gmv = ge * m;
            if (gmv != 0.f) {
                if (v00) atomicAdd(db + ((long)y0 * w + x0) * c + e, gmv * hy * hx);
                if (v01) atomicAdd(db + ((long)y0 * w + x1) * c + e, gmv * hy * lx);
                if (v10) atomicAdd(db + ((long)y1 * w + x0) * c + e, gmv * ly * hx);
                if (v11) atomicAdd(db + ((long)y1 * w + x1) * c + e, gmv * ly * lx);
            }
        }
        float* go = dom + pix * om_ld;
        atomicAdd(go + (g * K + k) * 2, gy * m);
        atomicAdd(go + (g * K + k) * 2 + 1, gx * m);
        atomicAdd(go + 2 * dg * K + g * K + k, gm * m * (1.f - m));
    }
}


// ---- element-wise steps of TemporalAttentionFusion.forward (denoising_aggregator.py:135-152), NHWC fp16, 8 channels per lane.
// The reference loops over the reference frame i and works on the n frames j at once; here the n*n (i, j) pairs are one
// batch, pair (i, j) at image index i*n + j.

// y[i*n + j] = cat(x[j], x[i]) over channels (x_cat = cat([x, x_ref], dim=1), :141-143). blockIdx.y = pair.
__global__ __launch_bounds__(256) void taf_pair_concat_kernel(const f16x8* __restrict__ x, f16x8* __restrict__ y, unsigned n,
                                                             unsigned hw, unsigned m8) {
    const unsigned pair = blockIdx.y, i = pair / n, j = pair - i * n;
    const unsigned total = hw * 2 * m8;
    const f16x8* xj = x + (long)j * hw * m8;
    const f16x8* xi = x + (long)i * hw * m8;
    f16x8* yp = y + (long)pair * total;
    for (unsigned t = blockIdx.x * blockDim.x + threadIdx.x; t < total; t += gridDim.x * blockDim.x) {
        const unsigned pix = t / (2 * m8), cv = t - pix * 2 * m8;
        yp[t] = cv < m8 ? xj[pix * m8 + cv] : xi[pix * m8 + cv - m8];
    }
}

// y[i*n + j] = xd[i*n + j] * x[i]   (x_dcn * x_ref, :146). blockIdx.y = pair.
__global__ __launch_bounds__(256) void taf_mul_ref_kernel(const f16x8* __restrict__ xd, const f16x8* __restrict__ x,
                                                         f16x8* __restrict__ y, unsigned n, unsigned per) {
    const unsigned pair = blockIdx.y;
    const f16x8* a = xd + (long)pair * per;
    const f16x8* b = x + (long)(pair / n) * per;
    f16x8* yp = y + (long)pair * per;
    for (unsigned t = blockIdx.x * blockDim.x + threadIdx.x; t < per; t += gridDim.x * blockDim.x) {
        const f16x8 u = a[t], v = b[t];
        f16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (_Float16)((float)u[e] * (float)v[e]);
        yp[t] = o;
    }
}

// out[i] = sum_j softmax_j(xcor[i*n + j]) * x[j]   (softmax over dim 0 and the weighted sum, :147-148), fp32 arithmetic.
// blockIdx.y = reference frame i.
__global__ __launch_bounds__(256) void taf_softmax_fuse_kernel(const f16x8* __restrict__ xcor, const f16x8* __restrict__ x,
                                                              f16x8* __restrict__ out, unsigned n, unsigned per) {
    const unsigned i = blockIdx.y;
    const f16x8* xc = xcor + (long)i * n * per;
    for (unsigned r = blockIdx.x * blockDim.x + threadIdx.x; r < per; r += gridDim.x * blockDim.x) {
        float mx[8], den[8], acc[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { mx[e] = -INFINITY; den[e] = 0.f; acc[e] = 0.f; }
        for (unsigned j = 0; j < n; ++j) {
            const f16x8 v = xc[(long)j * per + r];
#pragma unroll
            for (int e = 0; e < 8; ++e) mx[e] = fmaxf(mx[e], (float)v[e]);
        }
        for (unsigned j = 0; j < n; ++j) {
            const f16x8 v = xc[(long)j * per + r];
            const f16x8 xv = x[(long)j * per + r];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float wgt = __expf((float)v[e] - mx[e]);
                den[e] += wgt;
                acc[e] += wgt * (float)xv[e];
            }
        }
        f16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (_Float16)(acc[e] / den[e]);
        out[(long)i * per + r] = o;
    }
}

// y = a + b over rows of c halfs with row strides (the two inner skips of DenoisingAggregator.forward, :193,197), fp32 add
__global__ __launch_bounds__(256) void add_f16_kernel(const _Float16* __restrict__ a, const _Float16* __restrict__ b,
                                                     _Float16* __restrict__ y, long rows, int c8, int a_ld, int b_ld, int y_ld) {
    const unsigned total = (unsigned)(rows * c8);
    for (unsigned t = blockIdx.x * blockDim.x + threadIdx.x; t < total; t += gridDim.x * blockDim.x) {
        const unsigned rr = t / (unsigned)c8;
        const long r = rr;
        const int cv = (int)(t - rr * (unsigned)c8) * 8;
        const f16x8 u = *reinterpret_cast<const f16x8*>(a + r * a_ld + cv), v = *reinterpret_cast<const f16x8*>(b + r * b_ld + cv);
        f16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (_Float16)((float)u[e] + (float)v[e]);
        *reinterpret_cast<f16x8*>(y + r * y_ld + cv) = o;
    }
}

inline int grid_of(long total) {
    long b = (total + 255) / 256;
    return (int)(b > 256 * 32 ? 256 * 32 : (b < 1 ? 1 : b));
}

}  // namespace

extern "C" int llv_modulated_deform_im2col(const void* x, const float* om, void* col, int n, int h, int w, int c, int x_ld,
                                           int om_ld, int col_ld, int kh, int kw, int stride, int pad, int dilation,
                                           int deform_groups, int ho, int wo, void* stream) {
    if (!x || !om || !col || n <= 0 || h <= 0 || w <= 0 || c <= 0 || kh <= 0 || kw <= 0 || stride <= 0 || dilation <= 0 ||
        deform_groups <= 0 || ho <= 0 || wo <= 0)
        return LLV_E_BADARG;
    if ((c & 7) || (x_ld & 7) || (col_ld & 7) || x_ld < c || col_ld < kh * kw * c || om_ld < 3 * deform_groups * kh * kw || (om_ld & 1))
        return LLV_E_BADARG;
    if (c % deform_groups || ((c / deform_groups) & 7)) return LLV_E_UNSUPPORTED;   // a lane's 8 channels share one offset
    if (ho != (h + 2 * pad - (dilation * (kh - 1) + 1)) / stride + 1 || wo != (w + 2 * pad - (dilation * (kw - 1) + 1)) / stride + 1)
        return LLV_E_BADARG;
    if ((long)wo * kh * kw * (c / 8) >= (1L << 30)) return LLV_E_UNSUPPORTED;
    const long row_items = (long)wo * kh * kw * (c / 8);
    long bx = (row_items + 511) / 512;                        // two items per thread
    bx = bx < 1 ? 1 : bx;
    // grid.y is limited to 65535: fold rows into y only (n*ho of a 736-row, 9-pair clip is 6624)
    if ((long)n * ho > 65535) return LLV_E_UNSUPPORTED;
    hipLaunchKernelGGL(mdcn_im2col_kernel, dim3((unsigned)bx, (unsigned)(n * ho)), dim3(256), 0, (hipStream_t)stream, (const _Float16*)x, om,
                       (_Float16*)col, h, w, c / 8, x_ld, om_ld, col_ld, kh, kw, stride, pad, dilation, deform_groups,
                       c / deform_groups / 8, ho, wo);
    return hipGetLastError() == hipSuccess ? LLV_OK : LLV_E_LAUNCH;
}

extern "C" int llv_modulated_deform_col2im(const void* dcol, const void* x, const float* om, float* dx, float* dom, int n, int h, int w, int c,
                                           int x_ld, int om_ld, int col_ld, int kh, int kw, int stride, int pad, int dilation, int deform_groups,
                                           int ho, int wo, void* stream) {
    if (!dcol || !x || !om || !dx || !dom || n <= 0 || h <= 0 || w <= 0 || c <= 0 || kh <= 0 || kw <= 0 || stride <= 0 || dilation <= 0 ||
        deform_groups <= 0 || ho <= 0 || wo <= 0)
        return LLV_E_BADARG;
    if ((c & 7) || (x_ld & 7) || (col_ld & 7) || x_ld < c || col_ld < kh * kw * c || om_ld < 3 * deform_groups * kh * kw || (om_ld & 1))
        return LLV_E_BADARG;
    if (c % deform_groups || ((c / deform_groups) & 7)) return LLV_E_UNSUPPORTED;
    if (ho != (h + 2 * pad - (dilation * (kh - 1) + 1)) / stride + 1 || wo != (w + 2 * pad - (dilation * (kw - 1) + 1)) / stride + 1)
        return LLV_E_BADARG;
    if ((long)wo * kh * kw * (c / 8) >= (1L << 30) || (long)n * ho > 65535) return LLV_E_UNSUPPORTED;
    const long row_items = (long)wo * kh * kw * (c / 8);
    long bx = (row_items + 255) / 256;
    bx = bx < 1 ? 1 : bx;
    hipLaunchKernelGGL(mdcn_col2im_kernel, dim3((unsigned)bx, (unsigned)(n * ho)), dim3(256), 0, (hipStream_t)stream, (const _Float16*)dcol,
                       (const _Float16*)x, om, dx, dom, h, w, c / 8, x_ld, om_ld, col_ld, kh, kw, stride, pad, dilation, deform_groups,
                       c / deform_groups / 8, ho, wo);
    return hipGetLastError() == hipSuccess ? LLV_OK : LLV_E_LAUNCH;
}

static bool taf_ok(int n, int h, int w, int m) {
    return n > 0 && h > 0 && w > 0 && m > 0 && !(m & 7) && (long)n * n <= 65535 && (long)h * w * (m / 4) < (1L << 31);
}

extern "C" int llv_taf_pair_concat(const void* x, void* y, int n, int h, int w, int m, void* stream) {
    if (!x || !y || !taf_ok(n, h, w, m)) return LLV_E_BADARG;
    const long per = (long)h * w * 2 * (m / 8);
    hipLaunchKernelGGL(taf_pair_concat_kernel, dim3(grid_of(per), n * n), dim3(256), 0, (hipStream_t)stream, (const f16x8*)x, (f16x8*)y,
                       (unsigned)n, (unsigned)(h * w), (unsigned)(m / 8));
    return hipGetLastError() == hipSuccess ? LLV_OK : LLV_E_LAUNCH;
}

extern "C" int llv_taf_mul_ref(const void* xd, const void* x, void* y, int n, int h, int w, int m, void* stream) {
    if (!xd || !x || !y || !taf_ok(n, h, w, m)) return LLV_E_BADARG;
    const long per = (long)h * w * (m / 8);
    hipLaunchKernelGGL(taf_mul_ref_kernel, dim3(grid_of(per), n * n), dim3(256), 0, (hipStream_t)stream, (const f16x8*)xd, (const f16x8*)x,
                       (f16x8*)y, (unsigned)n, (unsigned)per);
    return hipGetLastError() == hipSuccess ? LLV_OK : LLV_E_LAUNCH;
}

extern "C" int llv_taf_softmax_fuse(const void* xcor, const void* x, void* out, int n, int h, int w, int m, void* stream) {
    if (!xcor || !x || !out || !taf_ok(n, h, w, m)) return LLV_E_BADARG;
    const long per = (long)h * w * (m / 8);
    hipLaunchKernelGGL(taf_softmax_fuse_kernel, dim3(grid_of(per), n), dim3(256), 0, (hipStream_t)stream, (const f16x8*)xcor,
                       (const f16x8*)x, (f16x8*)out, (unsigned)n, (unsigned)per);
    return hipGetLastError() == hipSuccess ? LLV_OK : LLV_E_LAUNCH;
}

extern "C" int llv_add_f16(const void* a, const void* b, void* y, int64_t rows, int c, int a_ld, int b_ld, int y_ld, void* stream) {
    if (!a || !b || !y || rows <= 0 || c <= 0 || ((c | a_ld | b_ld | y_ld) & 7) || a_ld < c || b_ld < c || y_ld < c) return LLV_E_BADARG;
    if (rows * (c / 8) >= (1L << 31)) return LLV_E_UNSUPPORTED;
    hipLaunchKernelGGL(add_f16_kernel, dim3(grid_of(rows * (c / 8))), dim3(256), 0, (hipStream_t)stream, (const _Float16*)a,
                       (const _Float16*)b, (_Float16*)y, (long)rows, c / 8, a_ld, b_ld, y_ld);
    return hipGetLastError() == hipSuccess ? LLV_OK : LLV_E_LAUNCH;
}

// ------------------------------------------------------------------------------------------------
// RoIAlign (first piece of SURVEY.md section 8f item 4, the authors' two-stage SELSA path:
// mmtracking/configs/vid/llvod/*.py `roi_layer=dict(type='RoIAlign', output_size=7, sampling_ratio=2)` ->
// mmdet SingleRoIExtractor -> mmcv.ops.roi_align (mmcv-full 1.3.0, un-vendored), pool_mode 'avg', aligned=True).
//   feat (n, h, w, ld) NHWC fp16; rois [k][5] fp32 = (batch index, x1, y1, x2, y2) in image coordinates;
//   out (k, ph, pw, c) NHWC fp16 (or fp32). One lane = 8 channels of one output bin: grid x grid bilinear samples averaged.
// mmcv's rules: start = coord * spatial_scale - 0.5 (aligned), bin = roi size / pooled size (no minimum size when aligned),
// sample (iy + 0.5) * bin / grid inside the bin; a sample outside [-1, H] x [-1, W] contributes zero, coordinates are clamped
// to >= 0, and the upper neighbour is clamped to the last row / column.
// ------------------------------------------------------------------------------------------------
namespace {

template <typename OutT>
__global__ __launch_bounds__(256) void roi_align_kernel(const _Float16* __restrict__ feat, const float* __restrict__ rois,
                                                       OutT* __restrict__ out, int n, int h, int w, int c8, int ld, int ph, int pw,
                                                       float scale, int grid, int aligned, unsigned total) {
    for (unsigned t = blockIdx.x * blockDim.x + threadIdx.x; t < total; t += gridDim.x * blockDim.x) {
        const unsigned cv = t % (unsigned)c8;
        unsigned p = t / (unsigned)c8;
        const unsigned ox = p % (unsigned)pw; p /= (unsigned)pw;
        const unsigned oy = p % (unsigned)ph;
        const unsigned r = p / (unsigned)ph;
        const float* roi = rois + (long)r * 5;
        const int b = (int)roi[0];
        const float off = aligned ? 0.5f : 0.f;
        const float x1 = roi[1] * scale - off, y1 = roi[2] * scale - off, x2 = roi[3] * scale - off, y2 = roi[4] * scale - off;
        float rw = x2 - x1, rh = y2 - y1;
        if (!aligned) { rw = fmaxf(rw, 1.f); rh = fmaxf(rh, 1.f); }
        const float bw = rw / (float)pw, bh = rh / (float)ph;
        const int gh = grid > 0 ? grid : (int)ceilf(rh / (float)ph), gw = grid > 0 ? grid : (int)ceilf(rw / (float)pw);
        const float cnt = fmaxf((float)(gh * gw), 1.f);
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (b >= 0 && b < n) {
            const _Float16* fb = feat + (long)b * h * w * ld + cv * 8;
            for (int iy = 0; iy < gh; ++iy) {
                float y = y1 + (float)oy * bh + ((float)iy + 0.5f) * bh / (float)gh;
                for (int ix = 0; ix < gw; ++ix) {
                    float x = x1 + (float)ox * bw + ((float)ix + 0.5f) * bw / (float)gw;
                    if (y < -1.f || y > (float)h || x < -1.f || x > (float)w) continue;
                    float yy = y <= 0.f ? 0.f : y, xx = x <= 0.f ? 0.f : x;
                    int yl = (int)yy, xl = (int)xx, yh, xh;
                    if (yl >= h - 1) { yh = yl = h - 1; yy = (float)yl; } else yh = yl + 1;
                    if (xl >= w - 1) { xh = xl = w - 1; xx = (float)xl; } else xh = xl + 1;
                    const float ly = yy - (float)yl, lx = xx - (float)xl, hy = 1.f - ly, hx = 1.f - lx;
                    const f16x8 v1 = *reinterpret_cast<const f16x8*>(fb + ((long)yl * w + xl) * ld);
                    const f16x8 v2 = *reinterpret_cast<const f16x8*>(fb + ((long)yl * w + xh) * ld);
                    const f16x8 v3 = *reinterpret_cast<const f16x8*>(fb + ((long)yh * w + xl) * ld);
                    const f16x8 v4 = *reinterpret_cast<const f16x8*>(fb + ((long)yh * w + xh) * ld);
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        acc[e] += hy * hx * (float)v1[e] + hy * lx * (float)v2[e] + ly * hx * (float)v3[e] + ly * lx * (float)v4[e];
                }
            }
        }
        OutT* o = out + (((long)r * ph + oy) * pw + ox) * (c8 * 8) + cv * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (OutT)(acc[e] / cnt);
    }
}


// Backward of RoIAlign with respect to the feature map (mmcv roi_align_backward_cuda_kernel: every bilinear sample's gradient
// g / count goes to its four neighbours with the forward weights). One thread per (output bin, channel), channels fastest, so the
// atomics of a wave-instruction hit 64 consecutive floats of one map pixel. dfeat fp32 [n][h][w][c], accumulated (zero it first).
__global__ __launch_bounds__(256) void roi_align_bwd_kernel(const float* __restrict__ gout, const float* __restrict__ rois,
                                                           float* __restrict__ dfeat, int n, int h, int w, int c, int ph, int pw, float scale,
                                                           int grid, int aligned, unsigned total) {
    for (unsigned t = blockIdx.x * blockDim.x + threadIdx.x; t < total; t += gridDim.x * blockDim.x) {
        const unsigned ch = t % (unsigned)c;
        unsigned p = t / (unsigned)c;
        const unsigned ox = p % (unsigned)pw; p /= (unsigned)pw;
        const unsigned oy = p % (unsigned)ph;
        const unsigned r = p / (unsigned)ph;
        const float* roi = rois + (long)r * 5;
        const int b = (int)roi[0];
        if (b < 0 || b >= n) continue;
        const float off = aligned ? 0.5f : 0.f;
        const float x1 = roi[1] * scale - off, y1 = roi[2] * scale - off, x2 = roi[3] * scale - off, y2 = roi[4] * scale - off;
        float rw = x2 - x1, rh = y2 - y1;
        if (!aligned) { rw = fmaxf(rw, 1.f); rh = fmaxf(rh, 1.f); }
        const float bw = rw / (float)pw, bh = rh / (float)ph;
        const int gh = grid > 0 ? grid : (int)ceilf(rh / (float)ph), gw = grid > 0 ? grid : (int)ceilf(rw / (float)pw);
        const float g = gout[t] / fmaxf((float)(gh * gw), 1.f);
        if (g == 0.f) continue;
        float* fb = dfeat + (long)b * h * w * c + ch;
        for (int iy = 0; iy < gh; ++iy) {
            const float y = y1 + (float)oy * bh + ((float)iy + 0.5f) * bh / (float)gh;
            for (int ix = 0; ix < gw; ++ix) {
                const float x = x1 + (float)ox * bw + ((float)ix + 0.5f) * bw / (float)gw;
                if (y < -1.f || y > (float)h || x < -1.f || x > (float)w) continue;
                float yy = y <= 0.f ? 0.f : y, xx = x <= 0.f ? 0.f : x;
                int yl = (int)yy, xl = (int)xx, yh, xh;
                if (yl >= h - 1) { yh = yl = h - 1; yy = (float)yl; } else yh = yl + 1;
                if (xl >= w - 1) { xh = xl = w - 1; xx = (float)xl; } else xh = xl + 1;
                const float ly = yy - (float)yl, lx = xx - (float)xl, hy = 1.f - ly, hx = 1.f - lx;
                atomicAdd(fb + ((long)yl * w + xl) * c, g * hy * hx);
                atomicAdd(fb + ((long)yl * w + xh) * c, g * hy * lx);
                atomicAdd(fb + ((long)yh * w + xl) * c, g * ly * hx);
                atomicAdd(fb + ((long)yh * w + xh) * c, g * ly * lx);
            }
        }
    }
}

}  // namespace

extern "C" int llv_roi_align_bwd(const float* grad_out, const float* rois, float* dfeat, int n, int h, int w, int c, int k, int ph, int pw,
                                 float spatial_scale, int sampling_ratio, int aligned, void* stream) {
    if (n <= 0 || h <= 0 || w <= 0 || c <= 0 || k < 0 || ph <= 0 || pw <= 0 || !(spatial_scale > 0.f) || sampling_ratio < 0) return LLV_E_BADARG;
    if (k == 0) return LLV_OK;
    if (!grad_out || !rois || !dfeat) return LLV_E_BADARG;
    const long total = (long)k * ph * pw * c;
    if (total >= (1L << 31)) return LLV_E_UNSUPPORTED;
    hipLaunchKernelGGL(roi_align_bwd_kernel, dim3(grid_of(total)), dim3(256), 0, (hipStream_t)stream, grad_out, rois, dfeat, n, h, w, c, ph, pw,
                       spatial_scale, sampling_ratio, aligned, (unsigned)total);
    return hipGetLastError() == hipSuccess ? LLV_OK : LLV_E_LAUNCH;
}

extern "C" int llv_roi_align(const void* feat, const float* rois, void* out, int n, int h, int w, int c, int ld, int k, int ph, int pw,
                             float spatial_scale, int sampling_ratio, int aligned, int out_f32, void* stream) {
    if (!feat || !out || n <= 0 || h <= 0 || w <= 0 || c <= 0 || (c & 7) || (ld & 7) || ld < c || k < 0 || ph <= 0 || pw <= 0 ||
        !(spatial_scale > 0.f) || sampling_ratio < 0)
        return LLV_E_BADARG;
    if (k == 0) return LLV_OK;
    if (!rois) return LLV_E_BADARG;
    const long total = (long)k * ph * pw * (c / 8);
    if (total >= (1L << 31)) return LLV_E_UNSUPPORTED;
    if (out_f32)
        hipLaunchKernelGGL(roi_align_kernel<float>, dim3(grid_of(total)), dim3(256), 0, (hipStream_t)stream, (const _Float16*)feat, rois,
                           (float*)out, n, h, w, c / 8, ld, ph, pw, spatial_scale, sampling_ratio, aligned, (unsigned)total);
    else
        hipLaunchKernelGGL(roi_align_kernel<_Float16>, dim3(grid_of(total)), dim3(256), 0, (hipStream_t)stream, (const _Float16*)feat,
                           rois, (_Float16*)out, n, h, w, c / 8, ld, ph, pw, spatial_scale, sampling_ratio, aligned, (unsigned)total);
    return hipGetLastError() == hipSuccess ? LLV_OK : LLV_E_LAUNCH;
}
