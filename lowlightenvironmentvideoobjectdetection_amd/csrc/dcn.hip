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

__global__ __launch_bounds__(256) void mdcn_im2col_kernel(const _Float16* __restrict__ x, const float* __restrict__ om,
                                                         _Float16* __restrict__ col, int n, int h, int w, int c8, int x_ld,
                                                         int om_ld, int col_ld, int kh, int kw, int stride, int pad,
                                                         int dilation, int dg, int cg8, int ho, int wo) {
    const int K = kh * kw;
    const long total = (long)n * ho * wo * K * c8;
    for (long t = blockIdx.x * (long)blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
        const int cv = (int)(t % c8);
        long p = t / c8;
        const int k = (int)(p % K); p /= K;
        const int ox = (int)(p % wo); p /= wo;
        const int oy = (int)(p % ho);
        const int img = (int)(p / ho);
        const int g = cv / cg8;
        const float* o = om + (((long)img * ho + oy) * wo + ox) * om_ld;
        const float dy = o[(g * K + k) * 2], dx = o[(g * K + k) * 2 + 1];
        const float ml = o[2 * dg * K + g * K + k];
        const float m = 1.f / (1.f + __expf(-ml));
        const int ki = k / kw, kj = k - ki * kw;
        const float py = (float)(oy * stride - pad + ki * dilation) + dy;
        const float px = (float)(ox * stride - pad + kj * dilation) + dx;
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (py > -1.f && px > -1.f && py < (float)h && px < (float)w) {
            const int y0 = (int)floorf(py), x0 = (int)floorf(px);
            const int y1 = y0 + 1, x1 = x0 + 1;
            const float ly = py - (float)y0, lx = px - (float)x0;
            const float hy = 1.f - ly, hx = 1.f - lx;
            const _Float16* xb = x + (long)img * h * w * x_ld + cv * 8;
            auto corner = [&](int yy, int xx, float wt) {
                if (yy >= 0 && yy <= h - 1 && xx >= 0 && xx <= w - 1) {
                    const f16x8 v = *reinterpret_cast<const f16x8*>(xb + ((long)yy * w + xx) * x_ld);
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc[e] += wt * (float)v[e];
                }
            };
            corner(y0, x0, hy * hx); corner(y0, x1, hy * lx); corner(y1, x0, ly * hx); corner(y1, x1, ly * lx);
        }
        f16x8 r;
#pragma unroll
        for (int e = 0; e < 8; ++e) r[e] = (_Float16)(acc[e] * m);
        *reinterpret_cast<f16x8*>(col + (((long)img * ho + oy) * wo + ox) * col_ld + (long)k * c8 * 8 + cv * 8) = r;
    }
}


// ---- element-wise steps of TemporalAttentionFusion.forward (denoising_aggregator.py:135-152), NHWC fp16, 8 channels per lane.
// The reference loops over the reference frame i and works on the n frames j at once; here the n*n (i, j) pairs are one
// batch, pair (i, j) at image index i*n + j.

// y[i*n + j] = cat(x[j], x[i]) over channels (x_cat = cat([x, x_ref], dim=1), :141-143)
__global__ __launch_bounds__(256) void taf_pair_concat_kernel(const f16x8* __restrict__ x, f16x8* __restrict__ y, int n, long hw,
                                                             int m8) {
    const long total = (long)n * n * hw * 2 * m8;
    for (long t = blockIdx.x * (long)blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
        const int cv = (int)(t % (2 * m8));
        long p = t / (2 * m8);
        const long pix = p % hw;
        const int pair = (int)(p / hw);
        const int i = pair / n, j = pair - i * n;
        const int src = cv < m8 ? j : i;
        y[t] = x[((long)src * hw + pix) * m8 + (cv < m8 ? cv : cv - m8)];
    }
}

// y[i*n + j] = xd[i*n + j] * x[i]   (x_dcn * x_ref, :146)
__global__ __launch_bounds__(256) void taf_mul_ref_kernel(const f16x8* __restrict__ xd, const f16x8* __restrict__ x,
                                                         f16x8* __restrict__ y, int n, long hw, int m8) {
    const long per = hw * m8, total = (long)n * n * per;
    for (long t = blockIdx.x * (long)blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
        const int pair = (int)(t / per);
        const long r = t - (long)pair * per;
        const f16x8 a = xd[t], b = x[(long)(pair / n) * per + r];
        f16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (_Float16)((float)a[e] * (float)b[e]);
        y[t] = o;
    }
}

// out[i] = sum_j softmax_j(xcor[i*n + j]) * x[j]   (softmax over dim 0 and the weighted sum, :147-148), fp32 arithmetic
__global__ __launch_bounds__(256) void taf_softmax_fuse_kernel(const f16x8* __restrict__ xcor, const f16x8* __restrict__ x,
                                                              f16x8* __restrict__ out, int n, long hw, int m8) {
    const long per = hw * m8, total = (long)n * per;
    for (long t = blockIdx.x * (long)blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
        const int i = (int)(t / per);
        const long r = t - (long)i * per;
        float mx[8], den[8], acc[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { mx[e] = -INFINITY; den[e] = 0.f; acc[e] = 0.f; }
        for (int j = 0; j < n; ++j) {
            const f16x8 v = xcor[((long)i * n + j) * per + r];
#pragma unroll
            for (int e = 0; e < 8; ++e) mx[e] = fmaxf(mx[e], (float)v[e]);
        }
        for (int j = 0; j < n; ++j) {
            const f16x8 v = xcor[((long)i * n + j) * per + r];
            const f16x8 xv = x[(long)j * per + r];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float w = __expf((float)v[e] - mx[e]);
                den[e] += w;
                acc[e] += w * (float)xv[e];
            }
        }
        f16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (_Float16)(acc[e] / den[e]);
        out[t] = o;
    }
}

// y = a + b over rows of c halfs with row strides (the two inner skips of DenoisingAggregator.forward, :193,197), fp32 add
__global__ __launch_bounds__(256) void add_f16_kernel(const _Float16* __restrict__ a, const _Float16* __restrict__ b,
                                                     _Float16* __restrict__ y, long rows, int c8, int a_ld, int b_ld, int y_ld) {
    const long total = rows * c8;
    for (long t = blockIdx.x * (long)blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
        const long r = t / c8;
        const int cv = (int)(t - r * c8) * 8;
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
    if ((c & 7) || (x_ld & 7) || (col_ld & 7) || x_ld < c || col_ld < kh * kw * c || om_ld < 3 * deform_groups * kh * kw)
        return LLV_E_BADARG;
    if (c % deform_groups || ((c / deform_groups) & 7)) return LLV_E_UNSUPPORTED;   // a lane's 8 channels share one offset
    if (ho != (h + 2 * pad - (dilation * (kh - 1) + 1)) / stride + 1 || wo != (w + 2 * pad - (dilation * (kw - 1) + 1)) / stride + 1)
        return LLV_E_BADARG;
    const long total = (long)n * ho * wo * kh * kw * (c / 8);
    long blocks = (total + 255) / 256;
    blocks = blocks > 256 * 32 ? 256 * 32 : blocks;
    hipLaunchKernelGGL(mdcn_im2col_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, (const _Float16*)x, om,
                       (_Float16*)col, n, h, w, c / 8, x_ld, om_ld, col_ld, kh, kw, stride, pad, dilation, deform_groups,
                       c / deform_groups / 8, ho, wo);
    return hipGetLastError() == hipSuccess ? LLV_OK : LLV_E_LAUNCH;
}

extern "C" int llv_taf_pair_concat(const void* x, void* y, int n, int h, int w, int m, void* stream) {
    if (!x || !y || n <= 0 || h <= 0 || w <= 0 || m <= 0 || (m & 7)) return LLV_E_BADARG;
    const long total = (long)n * n * h * w * 2 * (m / 8);
    hipLaunchKernelGGL(taf_pair_concat_kernel, dim3(grid_of(total)), dim3(256), 0, (hipStream_t)stream, (const f16x8*)x, (f16x8*)y, n,
                       (long)h * w, m / 8);
    return hipGetLastError() == hipSuccess ? LLV_OK : LLV_E_LAUNCH;
}

extern "C" int llv_taf_mul_ref(const void* xd, const void* x, void* y, int n, int h, int w, int m, void* stream) {
    if (!xd || !x || !y || n <= 0 || h <= 0 || w <= 0 || m <= 0 || (m & 7)) return LLV_E_BADARG;
    const long total = (long)n * n * h * w * (m / 8);
    hipLaunchKernelGGL(taf_mul_ref_kernel, dim3(grid_of(total)), dim3(256), 0, (hipStream_t)stream, (const f16x8*)xd, (const f16x8*)x,
                       (f16x8*)y, n, (long)h * w, m / 8);
    return hipGetLastError() == hipSuccess ? LLV_OK : LLV_E_LAUNCH;
}

extern "C" int llv_taf_softmax_fuse(const void* xcor, const void* x, void* out, int n, int h, int w, int m, void* stream) {
    if (!xcor || !x || !out || n <= 0 || h <= 0 || w <= 0 || m <= 0 || (m & 7)) return LLV_E_BADARG;
    const long total = (long)n * h * w * (m / 8);
    hipLaunchKernelGGL(taf_softmax_fuse_kernel, dim3(grid_of(total)), dim3(256), 0, (hipStream_t)stream, (const f16x8*)xcor,
                       (const f16x8*)x, (f16x8*)out, n, (long)h * w, m / 8);
    return hipGetLastError() == hipSuccess ? LLV_OK : LLV_E_LAUNCH;
}

extern "C" int llv_add_f16(const void* a, const void* b, void* y, int64_t rows, int c, int a_ld, int b_ld, int y_ld, void* stream) {
    if (!a || !b || !y || rows <= 0 || c <= 0 || ((c | a_ld | b_ld | y_ld) & 7) || a_ld < c || b_ld < c || y_ld < c) return LLV_E_BADARG;
    hipLaunchKernelGGL(add_f16_kernel, dim3(grid_of(rows * (c / 8))), dim3(256), 0, (hipStream_t)stream, (const _Float16*)a,
                       (const _Float16*)b, (_Float16*)y, (long)rows, c / 8, a_ld, b_ld, y_ld);
    return hipGetLastError() == hipSuccess ? LLV_OK : LLV_E_LAUNCH;
}
