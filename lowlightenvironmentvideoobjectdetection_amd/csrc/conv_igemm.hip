// conv_igemm.hip -- NHWC fp16 implicit-GEMM convolution on gfx950 MFMA (v_mfma_f32_16x16x32_f16).
//
// Orientation: the WEIGHT tile is the MFMA A operand (rows = output channels) and the PIXEL tile is
// the B operand (cols = output pixels), so a lane's 4 accumulator registers are 4 consecutive output
// channels of one pixel: the NHWC store is an 8-byte packed write and bias/residual are 4-wide.
//
// Per k-step (BK = 32 halfs = one 64-byte slab of K per row):
//   global -> registers (16 B per lane per row, predicated zero for padding / tails)
//   registers -> LDS   (double buffered, 16-B chunks XOR-swizzled so ds_read_b128 is conflict free)
//   LDS -> fragments -> MFMA, fp32 accumulate.
// K is ordered (r, q, c) with c fastest, so a 16-B chunk never straddles a filter tap when the
// per-group channel count is a multiple of 8.
//
// Replaces torch.nn.Conv2d (+BN/ReLU/add/upsample/PixelShuffle) in the reference; see include/llv.h.

#include <hip/hip_runtime.h>
#include <type_traits>
#include <cstdlib>
#include <hip/hip_fp16.h>
#include "llv.h"

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

struct ConvKArgs {
    const _Float16* x0;
    const _Float16* x1;
    const _Float16* x2;
    const _Float16* w;
    const float* bias;
    const _Float16* res;
    const _Float16* mask;      // LLV_CONV_MASK_POS
    int mask_ld;
    void* y;
    int n_src, src_cvec;
    int x_ld, y_ld, res_ld;
    int cin, cout, w_rows, k_pad;
    int kh, kw, stride, pad;
    int dil;            // dilation (>= 1): tap (r, q) reads input (y*stride - pad + r*dil, x*stride - pad + q*dil)
    int n_seg;
    int seg_h[LLV_MAX_SEG], seg_w[LLV_MAX_SEG], seg_ho[LLV_MAX_SEG], seg_wo[LLV_MAX_SEG];
    int seg_m0[LLV_MAX_SEG + 1];
    int seg_xoff[LLV_MAX_SEG], seg_yoff[LLV_MAX_SEG];  // pixel offsets (fit int32, checked by host)
    int flags;
    int res_h, res_w;
    int tiles_n, nblk;
    int m_total;
    long w_gstride;  // elements between groups in w
    int src_per_group;  // groups == n_src: group g reads source tensor g
    float act_slope;    // LLV_CONV_LEAKY: y = y > 0 ? y : act_slope * y
    int ps_h, ps_w;     // PIXSHUF: stored output extent (<= 2Ho, 2Wo): crop_like of a transposed conv
    long y_gstride;     // != 0: group g writes its own plane y + g*y_gstride (channel index restarts at 0)
    int stage_out;      // epilogue goes through an LDS tile so that global stores are full 16-byte vectors
    int k_splits;       // > 1: blockIdx.z owns k-steps [z*nk_per, ...) and writes fp32 partials to ws
    int nk_per;
    float* ws;          // [k_splits][m_total][groups*cout] fp32 partial sums
    unsigned x_bytes, w_bytes;   // buffer-resource bounds of one source tensor / one group's weights (0: not usable)
    const long* off_dev;         // convpair2 only: *off_dev (fp16 elements) is added to x0..x2 (off_x) / res (off_res) at kernel start
    int off_x, off_res;
    long off_max;                // the largest offset the caller declared (llv_conv_desc.offset_max): the word is clamped to [0, off_max]
    unsigned* clsmax;            // LLV_CONV_CLSMAX: [rows][cout / cls_c] running maximum logit per anchor (ordered-uint encoding)
    unsigned char* segw;         //                  [rows][tiles_n] segment-stored flags
    int cls_c;
    float store_thr;
};

// order-preserving map float -> unsigned (a < b  <=>  enc(a) < enc(b)); 0 is below every float
__device__ __forceinline__ unsigned f32_ordered(float f) {
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// source tensor s (0..2) of a multi-source launch, without a pointer table (see conv_igemm_glds_kernel)
__device__ __forceinline__ const _Float16* src_sel(const ConvKArgs& a, int s) {
    const long d1 = reinterpret_cast<const char*>(a.x1) - reinterpret_cast<const char*>(a.x0);
    const long d2 = reinterpret_cast<const char*>(a.x2) - reinterpret_cast<const char*>(a.x0);
    const long off = (d1 & -(long)(s == 1)) | (d2 & -(long)(s == 2));
    return reinterpret_cast<const _Float16*>(reinterpret_cast<const char*>(a.x0) + off);
}

// Shared epilogue: v[4] = conv sum (no bias yet) of channels gco..gco+3 of output row m.
struct EpiCtx {
    long opix, rpix;
    int ox, oy, n, Ho, Wo, yo;
};

// residual read through a nearest-neighbour upsample: F.interpolate(mode='nearest', size=(Ho,Wo)) is
// src = min(floor(dst * in/out), in-1) (fpn.py:186-195)
__device__ __forceinline__ void epi_res_ups(const ConvKArgs& a, EpiCtx& c) {
    const float sy = (float)a.res_h / (float)c.Ho, sx = (float)a.res_w / (float)c.Wo;
    int ry = (int)floorf((float)c.oy * sy); ry = ry < a.res_h - 1 ? ry : a.res_h - 1;
    int rx = (int)floorf((float)c.ox * sx); rx = rx < a.res_w - 1 ? rx : a.res_w - 1;
    c.rpix = ((long)c.n * a.res_h + ry) * a.res_w + rx;
}

__device__ __forceinline__ void epi_locate(const ConvKArgs& a, int m, EpiCtx& c) {
    int Ho = a.seg_ho[0], Wo = a.seg_wo[0], m0 = 0, yo = a.seg_yoff[0];
    if (a.n_seg > 1) {       // uniform branch: the common single-segment launch skips the search
#pragma unroll
        for (int s = 1; s < LLV_MAX_SEG; ++s) {
            if (s < a.n_seg && m >= a.seg_m0[s]) {
                Ho = a.seg_ho[s]; Wo = a.seg_wo[s]; m0 = a.seg_m0[s]; yo = a.seg_yoff[s];
            }
        }
    }
    const int ml = m - m0;
    c.opix = (long)yo + ml;  // output pixel index (dense batch x Ho x Wo per segment)
    c.rpix = c.opix;
    c.ox = 0; c.oy = 0; c.n = 0; c.Ho = Ho; c.Wo = Wo; c.yo = yo;
    if (a.flags & (LLV_CONV_PIXSHUF | LLV_CONV_RES_UPS)) {
        c.ox = ml % Wo;
        const int t = ml / Wo;
        c.oy = t % Ho;
        c.n = t / Ho;
    }
    if (a.flags & LLV_CONV_RES_UPS) epi_res_ups(a, c);
}

__device__ __forceinline__ float act_fn(const ConvKArgs& a, float v) {
    if (a.flags & LLV_CONV_RELU) return fmaxf(v, 0.f);
    if (a.flags & LLV_CONV_LEAKY) return v > 0.f ? v : a.act_slope * v;
    return v;
}

// bias / residual / ReLU on 4 consecutive channels. With PIXSHUF the residual and ReLU are applied at the
// shuffled output location instead (epi_store_direct / the staged store), so only the bias is added here.
__device__ __forceinline__ void epi_compute_b(const ConvKArgs& a, const EpiCtx& c, int gco, float (&v)[4], const float4 b4) {
    const int flags = a.flags;
    v[0] += b4.x; v[1] += b4.y; v[2] += b4.z; v[3] += b4.w;
    if (flags & LLV_CONV_PIXSHUF) return;
    if (flags & LLV_CONV_ACT_FIRST) {        // Darknet ResBlock: x + act(bn(conv(.))) (darknet.py:47-53)
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = act_fn(a, v[e]);
        const f16x4 r4 = *reinterpret_cast<const f16x4*>(a.res + c.rpix * a.res_ld + gco);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += (float)r4[e];
        return;
    }
    if (flags & (LLV_CONV_RES_ADD | LLV_CONV_RES_RSUB)) {
        const f16x4 r4 = *reinterpret_cast<const f16x4*>(a.res + c.rpix * a.res_ld + gco);
        if (flags & LLV_CONV_RES_RSUB) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = (float)r4[e] - v[e];
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += (float)r4[e];
        }
    }
    if (flags & (LLV_CONV_RELU | LLV_CONV_LEAKY)) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = act_fn(a, v[e]);
    }
    if (flags & LLV_CONV_MASK_POS) {          // a data gradient takes the ReLU mask of the activation it is the gradient of
        const f16x4 m4 = *reinterpret_cast<const f16x4*>(a.mask + c.opix * a.mask_ld + gco);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (float)m4[e] > 0.f ? v[e] : 0.f;
    }
}

__device__ __forceinline__ void epi_compute(const ConvKArgs& a, const EpiCtx& c, int gco, float (&v)[4]) {
    epi_compute_b(a, c, gco, v, *reinterpret_cast<const float4*>(a.bias + gco));
}

__device__ __forceinline__ void epi_store(const ConvKArgs& a, const EpiCtx& c, int gco, float v0, float v1, float v2, float v3) {
    const int flags = a.flags;
    float v[4] = {v0, v1, v2, v3};
    epi_compute(a, c, gco, v);
    if (a.y_gstride) {   // planar groups (plain fp16 store only; host checks the flags)
        const int g = gco / a.cout, co = gco - g * a.cout;
        f16x4 o4 = {(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
        *reinterpret_cast<f16x4*>(reinterpret_cast<_Float16*>(a.y) + g * a.y_gstride + c.opix * a.y_ld + co) = o4;
        return;
    }
    if (flags & LLV_CONV_PIXSHUF) {
        // conv channel ch = 4*cc + 2*dy + dx -> out pixel (2oy+dy, 2ox+dx), channel cc
        const int cc = gco >> 2;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int dy = e >> 1, dx = e & 1;
            if (2 * c.oy + dy >= a.ps_h || 2 * c.ox + dx >= a.ps_w) continue;
            const long p = (long)c.yo + ((long)(c.n * a.ps_h + 2 * c.oy + dy)) * a.ps_w + (2 * c.ox + dx);
            float o = v[e];
            if (flags & LLV_CONV_RES_ADD) o += (float)a.res[p * a.res_ld + cc];
            o = act_fn(a, o);
            reinterpret_cast<_Float16*>(a.y)[p * a.y_ld + cc] = (_Float16)o;
        }
        return;
    }
    if (flags & LLV_CONV_OUT_F32) {
        *reinterpret_cast<float4*>(reinterpret_cast<float*>(a.y) + c.opix * a.y_ld + gco) = make_float4(v[0], v[1], v[2], v[3]);
    } else {
        f16x4 o4 = {(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
        *reinterpret_cast<f16x4*>(reinterpret_cast<_Float16*>(a.y) + c.opix * a.y_ld + gco) = o4;
    }
}

// LDS reads the compiler must not guard with vmcnt waits (see the implicit-GEMM main loop): the read is inline assembly, the wave's
// LDS counter is waited for by hand (lds_wait_all_but<N>: at most the N youngest reads still outstanding) and lds_use ties a register
// to a point after that wait, so its first use cannot be scheduled in front of it.
__device__ __forceinline__ unsigned lds_addr(const void* p) {
    return (unsigned)(unsigned long)((__attribute__((address_space(3))) const unsigned char*)p);
}
__device__ __forceinline__ f16x8 lds_read128(unsigned addr) {
    f16x8 v;
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr));
    return v;
}
template <int N>
__device__ __forceinline__ void lds_wait_all_but() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N)); }
__device__ __forceinline__ void lds_use(f16x8& v) { asm volatile("" : "+v"(v)); }

// Phase 2 of the LDS-staged epilogue: the [BM pixels][BN couts] fp16 tile (row stride BN+8 halfs) goes to
// global memory 16 bytes per lane, consecutive lanes = consecutive channels of a pixel. row_to_m maps a tile
// row to the output row index m (or -1 when the row is outside the image / past the end).
template <int BN, int BM, typename LocFn>
__device__ __forceinline__ void staged_store_loc(const ConvKArgs& a, const _Float16* tile, int g, int tile_n, LocFn loc) {
    constexpr int TS = BN + 8;
    const int tid = threadIdx.x;
    if (!(a.flags & LLV_CONV_PIXSHUF)) {
        constexpr int CPR = BN / 8;  // 16-B chunks per tile row; a lane keeps its chunk, rows advance by 256/CPR
        static_assert(256 % CPR == 0, "lane -> chunk mapping is fixed across iterations");
        const int cc = tid % CPR;
        const int co = tile_n * BN + cc * 8;
        if (co < a.cout) {
            const long cofs = a.y_gstride ? g * a.y_gstride + co : (long)g * a.cout + co;
            const bool full = co + 8 <= a.cout;
#pragma unroll 4
            for (int prow = tid / CPR; prow < BM; prow += 256 / CPR) {
                EpiCtx c;
                if (!loc(prow, c)) continue;
                _Float16* dst = reinterpret_cast<_Float16*>(a.y) + c.opix * a.y_ld + cofs;
                const uint4 val = *reinterpret_cast<const uint4*>(tile + prow * TS + cc * 8);
                if (full) *reinterpret_cast<uint4*>(dst) = val;
                else *reinterpret_cast<uint2*>(dst) = make_uint2(val.x, val.y);   // cout is a multiple of 4
            }
        }
    } else if constexpr (BN >= 32) {
        // PixelShuffle(2): out pixel (2oy+dy, 2ox+dx), channel cc <- conv channel 4*cc + 2*dy + dx.
        // One lane = 8 consecutive output channels of one of the four output pixels of a conv pixel.
        constexpr int OC8 = BN / 32;  // groups of 8 output channels per tile row
        for (int q = tid; q < BM * 4 * OC8; q += 256) {
            const int o8 = q % OC8;
            const int e = (q / OC8) & 3;
            const int prow = q / (OC8 * 4);
            const int co = tile_n * BN + o8 * 32;          // first conv channel of this group
            EpiCtx c;
            if (co >= a.cout || !loc(prow, c)) continue;
            const int dy = e >> 1, dx = e & 1;
            if (2 * c.oy + dy >= a.ps_h || 2 * c.ox + dx >= a.ps_w) continue;
            const long p = (long)c.yo + ((long)(c.n * a.ps_h + 2 * c.oy + dy)) * a.ps_w + (2 * c.ox + dx);
            const int occ = (g * a.cout + co) >> 2;         // first output channel
            const _Float16* src = tile + prow * TS + o8 * 32 + e;
            float v[8];
#pragma unroll
            for (int t = 0; t < 8; ++t) v[t] = (float)src[t * 4];
            if (a.flags & LLV_CONV_RES_ADD) {
                const f16x8 r8 = *reinterpret_cast<const f16x8*>(a.res + p * a.res_ld + occ);
#pragma unroll
                for (int t = 0; t < 8; ++t) v[t] += (float)r8[t];
            }
            if (a.flags & (LLV_CONV_RELU | LLV_CONV_LEAKY)) {
#pragma unroll
                for (int t = 0; t < 8; ++t) v[t] = act_fn(a, v[t]);
            }
            f16x8 o8v;
#pragma unroll
            for (int t = 0; t < 8; ++t) o8v[t] = (_Float16)v[t];
            *reinterpret_cast<f16x8*>(reinterpret_cast<_Float16*>(a.y) + p * a.y_ld + occ) = o8v;
        }
    }
}

template <int BN, int BM, typename RowFn>
__device__ __forceinline__ void staged_store(const ConvKArgs& a, const _Float16* tile, int g, int tile_n, RowFn row_to_m) {
    staged_store_loc<BN, BM>(a, tile, g, tile_n, [&](int prow, EpiCtx& c) {
        const int m = row_to_m(prow);
        if (m < 0) return false;
        epi_locate(a, m, c);
        return true;
    });
}

// 4 x 4 transpose across the four 16-lane rows of a wave: lane (r, row f) holding v[e] ends up holding the old
// v[f] of row e. Two rounds of gfx950's lane-swap instructions (v_permlane16_swap / v_permlane32_swap), no LDS.
__device__ __forceinline__ void row_transpose4(float (&v)[4]) {
    unsigned u0 = __float_as_uint(v[0]), u1 = __float_as_uint(v[1]), u2 = __float_as_uint(v[2]), u3 = __float_as_uint(v[3]);
    auto s01 = __builtin_amdgcn_permlane16_swap(u0, u1, false, false);   // odd rows of [0] <-> even rows of [1]
    auto s23 = __builtin_amdgcn_permlane16_swap(u2, u3, false, false);
    u0 = s01[0]; u1 = s01[1]; u2 = s23[0]; u3 = s23[1];
    auto t02 = __builtin_amdgcn_permlane32_swap(u0, u2, false, false);   // rows 2,3 of [0] <-> rows 0,1 of [2]
    auto t13 = __builtin_amdgcn_permlane32_swap(u1, u3, false, false);
    v[0] = __uint_as_float(t02[0]); v[2] = __uint_as_float(t02[1]);
    v[1] = __uint_as_float(t13[0]); v[3] = __uint_as_float(t13[1]);
}

// PixelShuffle(2) store of a tile staged in SHUFFLED order: row prow*4 + e (e = 2*dy + dx) holds the BN/4 output
// channels of output pixel (2oy+dy, 2ox+dx) contiguously (row stride BN/4 + 8 halfs), bias already added. One lane =
// 8 channels: 16-B LDS read, 16-B skip-connection read, activation, 16-B store.
template <int BN, int BM, typename LocFn>
__device__ __forceinline__ void staged_store_shuffled(const ConvKArgs& a, const _Float16* tile, int g, int tile_n, LocFn loc) {
    constexpr int OC = BN / 4, TS2 = OC + 8, CPR = OC / 8;
    static_assert(OC % 8 == 0 && 256 % CPR == 0, "BN >= 32");
    const int tid = threadIdx.x;
    const int cc = tid % CPR;
    const int co = tile_n * BN + cc * 32;                     // first conv channel behind this lane's 8 outputs
    if (co >= a.cout) return;
    const int occ = (g * a.cout + co) >> 2;
    for (int row = tid / CPR; row < BM * 4; row += 256 / CPR) {
        const int prow = row >> 2, e = row & 3;
        EpiCtx c;
        if (!loc(prow, c)) continue;
        const int dy = e >> 1, dx = e & 1;
        if (2 * c.oy + dy >= a.ps_h || 2 * c.ox + dx >= a.ps_w) continue;
        const long p = (long)c.yo + ((long)(c.n * a.ps_h + 2 * c.oy + dy)) * a.ps_w + (2 * c.ox + dx);
        const f16x8 t8 = *reinterpret_cast<const f16x8*>(tile + row * TS2 + cc * 8);
        float v[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) v[t] = (float)t8[t];
        if (a.flags & LLV_CONV_RES_ADD) {
            const f16x8 r8 = *reinterpret_cast<const f16x8*>(a.res + p * a.res_ld + occ);
#pragma unroll
            for (int t = 0; t < 8; ++t) v[t] += (float)r8[t];
        }
        if (a.flags & (LLV_CONV_RELU | LLV_CONV_LEAKY)) {
#pragma unroll
            for (int t = 0; t < 8; ++t) v[t] = act_fn(a, v[t]);
        }
        f16x8 o8v;
#pragma unroll
        for (int t = 0; t < 8; ++t) o8v[t] = (_Float16)v[t];
        *reinterpret_cast<f16x8*>(reinterpret_cast<_Float16*>(a.y) + p * a.y_ld + occ) = o8v;
    }
}

// split-K second pass: sum the fp32 partial slabs, then the normal epilogue. One lane per (row, 4 channels).
__global__ __launch_bounds__(256) void splitk_epilogue_kernel(const ConvKArgs a, int ctot) {
    const int c4 = ctot >> 2;
    const long total = (long)a.m_total * c4;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int m = (int)(i / c4);
        const int gco = (int)(i - (long)m * c4) * 4;
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int z = 0; z < a.k_splits; ++z) {
            const float4 p = *reinterpret_cast<const float4*>(a.ws + ((long)z * a.m_total + m) * ctot + gco);
            s.x += p.x; s.y += p.y; s.z += p.z; s.w += p.w;
        }
        EpiCtx c;
        epi_locate(a, m, c);
        epi_store(a, c, gco, s.x, s.y, s.z, s.w);
    }
}

// byte offset of 16-B chunk `ch` (0..3) of row `row` inside a [rows][32 halfs] LDS tile.
// Fragment reads take row = 16*t + (lane&15), ch = lane>>4; ds_read_b128 is served in four
// 16-lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31}, ... so each group sees all 16 rows once
// with chunk c for rows 0-3,12-15 and chunk c^1 for rows 4-11: XOR with (-(row>>2))&3 = {0,3,2,1}
// sends the 16 accesses of a group to 16 distinct 16-B slots of the 256-B bank row.
__device__ __forceinline__ int lds_off(int row, int ch) {
    return row * 64 + (((ch ^ ((0 - (row >> 2)) & 3)) & 3) << 4);
}

template <int BN, int BM, int WN, int WM>
__global__ __launch_bounds__(256) void conv_igemm_kernel(const ConvKArgs a) {
    constexpr int WAVES_N = BN / WN;
    constexpr int WAVES_M = BM / WM;
    static_assert(WAVES_N * WAVES_M == 4, "4 waves per workgroup");
    constexpr int TN = WN / 16, TM = WM / 16;
    constexpr int WVEC = (BN * 4 + 255) / 256;
    constexpr int XVEC = (BM * 4 + 255) / 256;
    constexpr int STAGE = (BN + BM) * 64;  // bytes per LDS stage

    constexpr int TS = BN + 8;                    // epilogue tile row stride in halfs (+16 B: spreads rows over banks)
    constexpr int EPI = BM * TS * 2;              // bytes of the [BM pixels][BN couts] fp16 output tile
    constexpr int SMEM = 2 * STAGE > EPI ? 2 * STAGE : EPI;
    __shared__ __attribute__((aligned(16))) unsigned char smem[SMEM];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wn = wave % WAVES_N;
    const int wm = wave / WAVES_N;

    // XCD-aware, bijective block remap: blocks sharing bid%8 share an XCD (L2); give each XCD a
    // contiguous run of tiles so the n-tiles of one pixel tile and neighbouring halos hit one L2.
    int bid = blockIdx.x;
    {
        const int nblk = a.nblk;
        const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tile_n = bid % a.tiles_n;
    const int tile_m = bid / a.tiles_n;
    const int g = blockIdx.y;

    const int ch = tid & 3;   // which 16-B chunk of the 64-B k-slab this lane moves
    const int r0 = tid >> 2;  // row 0..63 (+64*i)

    const _Float16* wbase = a.w + (long)g * a.w_gstride;
    const int cvec = a.cin >> 3;  // 8-channel vectors per tap

    // ---- per-lane pixel rows (fixed for the whole K loop)
    int px_iy0[XVEC], px_ix0[XVEC], px_base[XVEC], px_hw[XVEC];
#pragma unroll
    for (int i = 0; i < XVEC; ++i) {
        const int row = r0 + 64 * i;
        const int m = tile_m * BM + row;
        px_iy0[i] = 0; px_ix0[i] = 0; px_base[i] = 0; px_hw[i] = 0;  // hw = 0 -> every tap out of range
        if (row < BM && m < a.m_total) {
            int H = a.seg_h[0], W = a.seg_w[0], Ho = a.seg_ho[0], Wo = a.seg_wo[0], m0 = 0, xo = a.seg_xoff[0];
#pragma unroll
            for (int s = 1; s < LLV_MAX_SEG; ++s) {
                if (s < a.n_seg && m >= a.seg_m0[s]) {
                    H = a.seg_h[s]; W = a.seg_w[s]; Ho = a.seg_ho[s]; Wo = a.seg_wo[s];
                    m0 = a.seg_m0[s]; xo = a.seg_xoff[s];
                }
            }
            const int ml = m - m0;
            const int ox = ml % Wo;
            const int t = ml / Wo;
            const int oy = t % Ho;
            const int n = t / Ho;
            px_iy0[i] = oy * a.stride - a.pad;
            px_ix0[i] = ox * a.stride - a.pad;
            px_base[i] = xo + n * H * W;
            px_hw[i] = (H << 16) | W;
        }
    }

    // ---- K walk state: (r, q, c8) of this lane's chunk, starting at this block's first k-step
    const int ks_begin = blockIdx.z * a.nk_per;
    int nk_all = a.k_pad >> 5;
    const int ks_end = (ks_begin + a.nk_per < nk_all) ? ks_begin + a.nk_per : nk_all;
    int kc8, kr, kq;
    {
        const int kvec = ks_begin * 4 + ch;
        const int tap = kvec / cvec;
        kc8 = kvec - tap * cvec;
        kr = tap / a.kw;
        kq = tap - kr * a.kw;
    }

    uint4 wreg[WVEC], xreg[XVEC];

    auto gload = [&](int ks) {
#pragma unroll
        for (int i = 0; i < WVEC; ++i) {
            const int row = r0 + 64 * i;
            const int grow = tile_n * BN + row;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (row < BN && grow < a.w_rows)
                v = *reinterpret_cast<const uint4*>(wbase + (long)grow * a.k_pad + ks * 32 + ch * 8);
            wreg[i] = v;
        }
        // the source tensor of this lane's chunk, selected with integer masks on the distance from x0: a ternary chain over
        // (x0, x1, x2) with a per-lane index is compiled into a 24-byte table in scratch memory (private segment 32 B per lane,
        // a scratch_load per k-step), which is how every launch of this kernel came to need a scratch allocation
        const _Float16* xb;
        int cofs;
        if (a.src_per_group) {
            xb = src_sel(a, g);
            cofs = kc8 * 8;
        } else if (a.n_src > 1) {
            const int s = kc8 / a.src_cvec;
            xb = src_sel(a, s);
            cofs = (kc8 - s * a.src_cvec) * 8;
        } else {
            xb = a.x0;
            cofs = g * a.cin + kc8 * 8;
        }
        const bool tap_ok = kr < a.kh;
#pragma unroll
        for (int i = 0; i < XVEC; ++i) {
            const int H = px_hw[i] >> 16, W = px_hw[i] & 0xffff;
            const int y = px_iy0[i] + kr * a.dil, x = px_ix0[i] + kq * a.dil;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (tap_ok && (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W)
                v = *reinterpret_cast<const uint4*>(xb + (long)(px_base[i] + y * W + x) * a.x_ld + cofs);
            xreg[i] = v;
        }
        kc8 += 4;
        while (kc8 >= cvec) { kc8 -= cvec; if (++kq == a.kw) { kq = 0; ++kr; } }
    };

    auto lstore = [&](int buf) {
        unsigned char* sW = smem + buf * STAGE;
        unsigned char* sX = sW + BN * 64;
#pragma unroll
        for (int i = 0; i < WVEC; ++i) {
            const int row = r0 + 64 * i;
            if (row < BN) *reinterpret_cast<uint4*>(sW + lds_off(row, ch)) = wreg[i];
        }
#pragma unroll
        for (int i = 0; i < XVEC; ++i) {
            const int row = r0 + 64 * i;
            if (row < BM) *reinterpret_cast<uint4*>(sX + lds_off(row, ch)) = xreg[i];
        }
    };

    f32x4 acc[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nk = ks_end - ks_begin;
    gload(ks_begin);
    lstore(0);
    __syncthreads();

    const int frow = lane & 15, fch = lane >> 4;
    for (int ks = 0; ks < nk; ++ks) {
        const int cur = ks & 1;
        if (ks + 1 < nk) gload(ks_begin + ks + 1);

        const unsigned char* sW = smem + cur * STAGE;
        const unsigned char* sX = sW + BN * 64;
        f16x8 af[TN], bf[TM];
#pragma unroll
        for (int i = 0; i < TN; ++i)
            af[i] = *reinterpret_cast<const f16x8*>(sW + lds_off(wn * WN + i * 16 + frow, fch));
#pragma unroll
        for (int j = 0; j < TM; ++j)
            bf[j] = *reinterpret_cast<const f16x8*>(sX + lds_off(wm * WM + j * 16 + frow, fch));
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
            for (int j = 0; j < TM; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[i], bf[j], acc[i][j], 0, 0, 0);

        if (ks + 1 < nk) lstore(cur ^ 1);
        __syncthreads();
    }

    // ---- epilogue: lane holds channels co..co+3 (rows of D) of pixel column lane&15
    if (a.k_splits > 1) {
        const int ctot = (int)gridDim.y * a.cout;
#pragma unroll
        for (int j = 0; j < TM; ++j) {
            const int m = tile_m * BM + wm * WM + j * 16 + frow;
            if (m >= a.m_total) continue;
#pragma unroll
            for (int i = 0; i < TN; ++i) {
                const int co = tile_n * BN + wn * WN + i * 16 + fch * 4;
                if (co >= a.cout) continue;
                *reinterpret_cast<float4*>(a.ws + ((long)blockIdx.z * a.m_total + m) * ctot + g * a.cout + co) =
                    make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
            }
        }
        return;
    }
    if (a.stage_out) {
        // ---- phase 1: final fp16 values -> LDS tile [pixel][cout]  (the K loop ended with a barrier: LDS is free)
        _Float16* tile = reinterpret_cast<_Float16*>(smem);
#pragma unroll
        for (int j = 0; j < TM; ++j) {
            const int prow = wm * WM + j * 16 + frow;
            const int m = tile_m * BM + prow;
            EpiCtx c;
            const bool mok = m < a.m_total;
            if (mok) epi_locate(a, m, c);
#pragma unroll
            for (int i = 0; i < TN; ++i) {
                const int col = wn * WN + i * 16 + fch * 4;
                const int co = tile_n * BN + col;
                float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
                if (mok && co < a.cout) epi_compute(a, c, g * a.cout + co, v);
                const f16x4 o4 = {(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
                *reinterpret_cast<f16x4*>(tile + prow * TS + col) = o4;
            }
        }
        __syncthreads();
        staged_store<BN, BM>(a, tile, g, tile_n, [&](int prow) { const int m = tile_m * BM + prow; return m < a.m_total ? m : -1; });
        return;
    }
#pragma unroll
    for (int j = 0; j < TM; ++j) {
        const int m = tile_m * BM + wm * WM + j * 16 + frow;
        if (m >= a.m_total) continue;
        EpiCtx c;
        epi_locate(a, m, c);
#pragma unroll
        for (int i = 0; i < TN; ++i) {
            const int co = tile_n * BN + wn * WN + i * 16 + fch * 4;
            if (co >= a.cout) continue;
            epi_store(a, c, g * a.cout + co, acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
        }
    }
}


// ------------------------------------------------------------------------------------------------
// conv_igemm_glds_kernel<BN,BM,WN,WM>: the implicit GEMM every plan uses (BN x BM x 32 tile), with the operand tiles written
// straight into LDS by global_load_lds_dwordx4 (no VGPR staging, no ds_write pass). The LDS image of one
// wave-instruction is lane-linear (16 rows x 4 chunks of 16 B), so the bank swizzle is applied to the SOURCE
// address: lane (row = l>>2, physical chunk = l&3) fetches logical chunk (l&3) ^ f(row). Padding taps and
// tail rows read a 16-byte zero page. Next k-step's loads are in flight while this one's MFMAs run.
// ------------------------------------------------------------------------------------------------
__device__ __attribute__((aligned(16))) const unsigned int g_zero_page[4] = {0u, 0u, 0u, 0u};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), (short)0, (int)bytes, 0x00020000);
}
// one wave-instruction of LDS DMA: lane l's 16 bytes at base + voff + soff land at lds + 16*l; an offset at or
// past the resource's size reads as zero (how padding taps, tile edges and tail rows are produced)
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t r, unsigned char* lds, int voff, int soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds, 16, voff, soff, 0, 0);
}

#ifndef GLDS_STAGES
#define GLDS_STAGES 2
#endif
// LDS image of a BK-wide operand tile: row-major rows of BK halfs, 16-byte chunks XOR-swizzled so that both the
// lane-linear DMA writes and the MFMA-fragment ds_read_b128 (16 rows x one chunk per 16-lane group) are conflict-free.
template <int BK>
__device__ __forceinline__ int glds_swz(int row) {
    if constexpr (BK == 32) return (0 - (row >> 2)) & 3;
    else return (row >> 1) & 7;
}
template <int BK>
__device__ __forceinline__ int glds_off(int row, int ch) { return row * (BK * 2) + ((ch ^ glds_swz<BK>(row)) << 4); }

// UT ("uniform tap"): Cin is a multiple of BK and there is one segment, so a k-step lies inside one filter tap and one
// source: the tap / channel part of every address is a scalar offset of a buffer load, each lane keeps ONE constant
// byte offset per pixel piece plus a bit mask of the taps that fall inside the image, and the per-step address work
// drops from ~18 vector instructions per piece to 3 (0 for 1x1 convolutions). The vector ALU shares its issue port
// with the matrix core, so this is what the main loop's speed hangs on.
template <int BN, int BM, int WN, int WM, int BK, bool UT>
__global__ __launch_bounds__(256, (BK == 64 ? 2 : (BN * BM >= 128 * 128 ? 3 : 4))) void conv_igemm_glds_kernel(const ConvKArgs a) {
    constexpr int WAVES_N = BN / WN;
    constexpr int WAVES_M = BM / WM;
    static_assert(WAVES_N * WAVES_M == 4, "4 waves per workgroup");
    constexpr int TN = WN / 16, TM = WM / 16;
    constexpr int RB = BK * 2;                               // bytes per LDS row
    constexpr int CPR = BK / 8;                              // 16-byte chunks per row
    constexpr int RPP = 1024 / RB;                           // rows per wave-instruction ("piece" = 1 KiB)
    constexpr int STAGE = (BN + BM) * RB;
    constexpr int TS = BN + 8;
    constexpr int EPI = BM * TS * 2;
    constexpr bool ASM_READS = BN == 64 && BM == 64;         // the small-map tile
    constexpr int NST = ASM_READS ? 4 : GLDS_STAGES;         // LDS ring depth: NST-1 k-steps of loads in flight
    constexpr int SMEM = NST * STAGE > EPI ? NST * STAGE : EPI;
    constexpr int NWI = BN / RPP, NPI = BM / RPP;            // pieces per k-step: weights, pixels
    constexpr int WPW = (NWI + 3) / 4, PPW = (NPI + 3) / 4;  // per wave (wave w takes pieces w, w+4, ...)
    __shared__ __attribute__((aligned(16))) unsigned char smem[SMEM];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = wave % WAVES_N, wm = wave / WAVES_N;
    int bid = blockIdx.x;
    {
        const int nblk = a.nblk;
        const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tile_n = bid % a.tiles_n, tile_m = bid / a.tiles_n;
    const int g = blockIdx.y;
    const _Float16* wbase = a.w + (long)g * a.w_gstride;
    const int cvec = a.cin >> 3;

    // this lane's slot inside a piece: row lane/CPR, physical chunk lane%CPR; the swizzle is applied to the SOURCE
    // chunk (pieces start at multiples of RPP rows, and glds_swz of row q*RPP + r depends on q only through q&1 for BK=64)
    const int lrow = lane / CPR;
    const _Float16* zero = reinterpret_cast<const _Float16*>(g_zero_page);

    int px_iy0[PPW], px_ix0[PPW], px_base[PPW], px_hw[PPW];
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        const int row = (wave + 4 * i) * RPP + lrow;
        const int m = tile_m * BM + row;
        px_iy0[i] = 0; px_ix0[i] = 0; px_base[i] = 0; px_hw[i] = 0;
        if (wave + 4 * i < NPI && m < a.m_total) {
            int H = a.seg_h[0], W = a.seg_w[0], Ho = a.seg_ho[0], Wo = a.seg_wo[0], m0 = 0, xo = a.seg_xoff[0];
#pragma unroll
            for (int s = 1; s < LLV_MAX_SEG; ++s) {
                if (s < a.n_seg && m >= a.seg_m0[s]) {
                    H = a.seg_h[s]; W = a.seg_w[s]; Ho = a.seg_ho[s]; Wo = a.seg_wo[s];
                    m0 = a.seg_m0[s]; xo = a.seg_xoff[s];
                }
            }
            const int ml = m - m0;
            const int ox = ml % Wo;
            const int t = ml / Wo;
            const int oy = t % Ho;
            const int n = t / Ho;
            px_iy0[i] = oy * a.stride - a.pad;
            px_ix0[i] = ox * a.stride - a.pad;
            px_base[i] = xo + n * H * W;
            px_hw[i] = (H << 16) | W;
        }
    }
    // every piece a wave owns has the same (piece & 1) when pieces are dealt w, w+4, ...: one logical chunk per lane
    const int ch = (lane % CPR) ^ glds_swz<BK>(wave * RPP + lrow);
    const _Float16* wrow[WPW];
#pragma unroll
    for (int i = 0; i < WPW; ++i) {
        const int grow = tile_n * BN + (wave + 4 * i) * RPP + lrow;
        wrow[i] = (wave + 4 * i < NWI && grow < a.w_rows) ? wbase + (long)grow * a.k_pad + ch * 8 : nullptr;
    }

    const int ks_begin = blockIdx.z * a.nk_per;
    const int nk_all = (a.k_pad + BK - 1) / BK;
    const int ks_end = (ks_begin + a.nk_per < nk_all) ? ks_begin + a.nk_per : nk_all;
    int kc8, kr, kq;
    {
        const int kvec = ks_begin * CPR + ch;
        const int tap = kvec / cvec;
        kc8 = kvec - tap * cvec;
        kr = tap / a.kw;
        kq = tap - kr * a.kw;
    }

    // ---- UT mode state
    int pv[PPW];                                             // byte offset of (pixel at tap (0,0), this lane's chunk) + pad shift
    unsigned pmask[PPW];                                     // bit t: tap t of this pixel is inside the image
    int wv[WPW];
    const int padb = UT ? (a.pad * a.seg_w[0] + a.pad) * a.x_ld * 2 : 0;   // resources start padb bytes early: offsets stay >= 0
    const int oob = (int)a.x_bytes + padb;
    __amdgpu_buffer_rsrc_t rW, rX0, rX1, rX2;
    int u_kc = 0, u_tap = 0, u_kr = 0, u_kq = 0, u_toff = 0;
    if constexpr (UT) {
        rW = make_rsrc(wbase, a.w_bytes);
        const _Float16* b0 = a.src_per_group ? (g == 0 ? a.x0 : (g == 1 ? a.x1 : a.x2)) : a.x0;
        rX0 = make_rsrc(reinterpret_cast<const unsigned char*>(b0) - padb, (unsigned)oob);
        rX1 = make_rsrc(reinterpret_cast<const unsigned char*>(a.n_src > 1 ? a.x1 : a.x0) - padb, (unsigned)oob);
        rX2 = make_rsrc(reinterpret_cast<const unsigned char*>(a.n_src > 2 ? a.x2 : a.x0) - padb, (unsigned)oob);
        const int chan0 = (a.n_src == 1 && !a.src_per_group) ? g * a.cin : 0;
        const int ntap = a.kh * a.kw;
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            const int H = px_hw[i] >> 16, W = px_hw[i] & 0xffff;
            pv[i] = ((px_base[i] + px_iy0[i] * W + px_ix0[i]) * a.x_ld + chan0 + ch * 8) * 2 + padb;
            unsigned mk = 0;
            for (int t = 0; t < ntap; ++t) {
                const int r = t / a.kw, q = t - r * a.kw;
                if ((unsigned)(px_iy0[i] + r * a.dil) < (unsigned)H && (unsigned)(px_ix0[i] + q * a.dil) < (unsigned)W) mk |= 1u << t;
            }
            pmask[i] = mk;                                   // H = W = 0 for rows past the end: no bit set
        }
#pragma unroll
        for (int i = 0; i < WPW; ++i) {
            const int grow = tile_n * BN + (wave + 4 * i) * RPP + lrow;
            wv[i] = grow < a.w_rows ? (grow * a.k_pad + ch * 8) * 2 : (int)a.w_bytes;
        }
        const int k0 = ks_begin * BK;
        u_tap = k0 / a.cin;
        u_kc = k0 - u_tap * a.cin;
        u_kr = u_tap / a.kw;
        u_kq = u_tap - u_kr * a.kw;
        u_toff = (u_kr * a.seg_w[0] + u_kq) * a.dil * a.x_ld * 2;
    }

    auto issue = [&](int ks, int buf) {
        unsigned char* sW = smem + buf * STAGE;
        unsigned char* sX = sW + BN * RB;
        if constexpr (UT) {
#pragma unroll
            for (int i = 0; i < WPW; ++i)
                if (wave + 4 * i < NWI) dma16(rW, sW + (wave + 4 * i) * 1024, wv[i], ks * (BK * 2));
            int kc = u_kc, sidx = 0;
            if (a.n_src > 1 && !a.src_per_group) {
                sidx = kc / (a.src_cvec * 8);
                kc -= sidx * a.src_cvec * 8;
            }
            const int soff = u_toff + kc * 2;
#pragma unroll
            for (int i = 0; i < PPW; ++i) {
                if (wave + 4 * i < NPI) {
                    const int voff = ((pmask[i] >> u_tap) & 1u) ? pv[i] : oob;
                    unsigned char* dst = sX + (wave + 4 * i) * 1024;
                    if (sidx == 0) dma16(rX0, dst, voff, soff);
                    else if (sidx == 1) dma16(rX1, dst, voff, soff);
                    else dma16(rX2, dst, voff, soff);
                }
            }
            u_kc += BK;
            if (u_kc >= a.cin) {
                u_kc = 0; ++u_tap;
                if (++u_kq == a.kw) { u_kq = 0; ++u_kr; }
                u_toff = (u_kr * a.seg_w[0] + u_kq) * a.dil * a.x_ld * 2;
            }
            return;
        }
        const bool k_in = ks * BK + ch * 8 < a.k_pad;          // BK = 64 with k_pad = 32 (mod 64): the last half-step is zero
#pragma unroll
        for (int i = 0; i < WPW; ++i) {
            if (wave + 4 * i < NWI) {       // wave-uniform
                const _Float16* src = (wrow[i] && k_in) ? wrow[i] + ks * BK : zero;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src),
                                                 (__attribute__((address_space(3))) void*)(sW + (wave + 4 * i) * 1024), 16, 0, 0);
            }
        }
        // the source tensor of this lane's chunk, selected with integer masks on the distance from x0: a ternary chain over
        // (x0, x1, x2) with a per-lane index is compiled into a 24-byte table in scratch memory (private segment 32 B per lane,
        // a scratch_load per k-step), which is how every launch of this kernel came to need a scratch allocation
        const _Float16* xb;
        int cofs;
        if (a.src_per_group) {
            xb = src_sel(a, g);
            cofs = kc8 * 8;
        } else if (a.n_src > 1) {
            const int s = kc8 / a.src_cvec;
            xb = src_sel(a, s);
            cofs = (kc8 - s * a.src_cvec) * 8;
        } else {
            xb = a.x0;
            cofs = g * a.cin + kc8 * 8;
        }
        const bool tap_ok = kr < a.kh;
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            if (wave + 4 * i < NPI) {
                const int H = px_hw[i] >> 16, W = px_hw[i] & 0xffff;
                const int y = px_iy0[i] + kr * a.dil, x = px_ix0[i] + kq * a.dil;
                const _Float16* src = zero;
                if (tap_ok && (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W)
                    src = xb + (long)(px_base[i] + y * W + x) * a.x_ld + cofs;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src),
                                                 (__attribute__((address_space(3))) void*)(sX + (wave + 4 * i) * 1024), 16, 0, 0);
            }
        }
        kc8 += CPR;
        while (kc8 >= cvec) { kc8 -= cvec; if (++kq == a.kw) { kq = 0; ++kr; } }
    };

    // staged, unsplit launches carry the bias in the accumulator (the epilogue then adds nothing)
    const bool bias_in_acc = a.stage_out && a.k_splits <= 1;
    f32x4 acc[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; ++i) {
        const int co = tile_n * BN + wn * WN + i * 16 + (lane >> 4) * 4;
        const float4 b4 = (bias_in_acc && co < a.cout) ? *reinterpret_cast<const float4*>(a.bias + g * a.cout + co) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int j = 0; j < TM; ++j) acc[i][j] = (f32x4){b4.x, b4.y, b4.z, b4.w};
    }

    // vmcnt counts this wave's loads in issue order, so "all but the youngest stage's pieces" = stage ks has landed
    constexpr int PIECES_FULL = WPW + PPW;
    const bool short_wave = (NWI & 3) != 0 && wave >= (NWI & 3);   // waves that own one weight piece fewer
    const int nk = ks_end - ks_begin;
    int cur = 0, nxt = 0;
#pragma unroll
    for (int s = 0; s < NST - 1; ++s)
        if (s < nk) { issue(ks_begin + s, nxt); nxt = nxt + 1 == NST ? 0 : nxt + 1; }
    const int frow = lane & 15, fch = lane >> 4;
    int foff_w[BK / 32][TN], foff_x[BK / 32][TM];             // byte offsets of this lane's fragment rows inside a stage (ASM_READS)
#pragma unroll
    for (int kk = 0; kk < BK / 32; ++kk) {
#pragma unroll
        for (int i = 0; i < TN; ++i) foff_w[kk][i] = glds_off<BK>(wn * WN + i * 16 + frow, kk * 4 + fch);
#pragma unroll
        for (int j = 0; j < TM; ++j) foff_x[kk][j] = glds_off<BK>(wm * WM + j * 16 + frow, kk * 4 + fch);
    }
    for (int ks = 0; ks < nk; ++ks) {
        // stages ks .. min(ks+NST-2, nk-1) are outstanding; wait until only those after ks remain
        const int ahead = nk - 1 - ks < NST - 2 ? nk - 1 - ks : NST - 2;
        if (ahead == 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else if (NST == 3 || ahead == 1) {
            if (short_wave) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES_FULL - 1) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES_FULL) : "memory");
        } else {
            if (short_wave) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (PIECES_FULL - 1)) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PIECES_FULL) : "memory");
        }
        // stage ks visible to all; everyone is done reading stage ks-1 (__syncthreads() would drain the deeper pipeline: vmcnt(0))
        if constexpr (ASM_READS) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        else __syncthreads();
        if (ks + NST - 1 < nk) { issue(ks_begin + ks + NST - 1, nxt); nxt = nxt + 1 == NST ? 0 : nxt + 1; }
        if constexpr (ASM_READS) {
            // Small tiles (single-image maps: one workgroup per CU or fewer, nothing else to hide a load behind) keep three k-steps
            // of DMA loads in flight. That needs operand reads the compiler does not guard: through plain LDS loads it cannot tell
            // the reads from the LDS destination of the DMA loads just issued and puts `s_waitcnt vmcnt(0)` in front of the first
            // fragment read, so every prefetch was waited for before the k-step it was issued under was computed (round 4, found in
            // the ISA). Inline-assembly reads, the LDS counter waited for by hand, lds_use ties the registers behind the wait.
            const unsigned sW = lds_addr(smem) + cur * STAGE;
            const unsigned sX = sW + BN * RB;
            cur = cur + 1 == NST ? 0 : cur + 1;
            constexpr int KK = BK / 32;
            f16x8 af[KK][TN], bf[KK][TM];
#pragma unroll
            for (int kk = 0; kk < KK; ++kk) {
#pragma unroll
                for (int i = 0; i < TN; ++i) af[kk][i] = lds_read128(sW + foff_w[kk][i]);
#pragma unroll
                for (int j = 0; j < TM; ++j) bf[kk][j] = lds_read128(sX + foff_x[kk][j]);
            }
#pragma unroll
            for (int kk = 0; kk < KK; ++kk) {
                if (kk + 1 < KK) lds_wait_all_but<(KK - 1) * (TN + TM)>(); else lds_wait_all_but<0>();
#pragma unroll
                for (int i = 0; i < TN; ++i) lds_use(af[kk][i]);
#pragma unroll
                for (int j = 0; j < TM; ++j) lds_use(bf[kk][j]);
#pragma unroll
                for (int i = 0; i < TN; ++i)
#pragma unroll
                    for (int j = 0; j < TM; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[kk][i], bf[kk][j], acc[i][j], 0, 0, 0);
            }
            continue;
        }
        const unsigned char* sW = smem + cur * STAGE;
        const unsigned char* sX = sW + BN * RB;
        cur = cur + 1 == NST ? 0 : cur + 1;
#pragma unroll
        for (int kk = 0; kk < BK / 32; ++kk) {
            f16x8 af[TN], bf[TM];
#pragma unroll
            for (int i = 0; i < TN; ++i)
                af[i] = *reinterpret_cast<const f16x8*>(sW + glds_off<BK>(wn * WN + i * 16 + frow, kk * 4 + fch));
#pragma unroll
            for (int j = 0; j < TM; ++j)
                bf[j] = *reinterpret_cast<const f16x8*>(sX + glds_off<BK>(wm * WM + j * 16 + frow, kk * 4 + fch));
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
                for (int j = 0; j < TM; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
    }
    __syncthreads();                                         // LDS is reused by the staged epilogue

    if (a.k_splits > 1) {
        const int ctot = (int)gridDim.y * a.cout;
#pragma unroll
        for (int j = 0; j < TM; ++j) {
            const int m = tile_m * BM + wm * WM + j * 16 + frow;
            if (m >= a.m_total) continue;
#pragma unroll
            for (int i = 0; i < TN; ++i) {
                const int co = tile_n * BN + wn * WN + i * 16 + fch * 4;
                if (co >= a.cout) continue;
                *reinterpret_cast<float4*>(a.ws + ((long)blockIdx.z * a.m_total + m) * ctot + g * a.cout + co) =
                    make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
            }
        }
        return;
    }
    if (a.stage_out) {
        _Float16* tile = reinterpret_cast<_Float16*>(smem);
        if (!(a.flags & ~LLV_CONV_RELU) && a.y_gstride == 0 && a.n_seg == 1) {
            // the common epilogue (bias [+ ReLU] -> fp16, one segment): no pixel lookup, ReLU on the rounded halves
            const bool relu = (a.flags & LLV_CONV_RELU) != 0;
            const f16x4 z4 = {(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
#pragma unroll
            for (int j = 0; j < TM; ++j) {
                const int prow = wm * WM + j * 16 + frow;
#pragma unroll
                for (int i = 0; i < TN; ++i) {
                    f16x4 o4 = {(_Float16)acc[i][j][0], (_Float16)acc[i][j][1], (_Float16)acc[i][j][2], (_Float16)acc[i][j][3]};
                    if (relu) o4 = __builtin_elementwise_max(o4, z4);
                    *reinterpret_cast<f16x4*>(tile + prow * TS + wn * WN + i * 16 + fch * 4) = o4;
                }
            }
            __syncthreads();
            constexpr int CPR = BN / 8;
            const int cc = tid % CPR, co = tile_n * BN + cc * 8;
            if (co < a.cout) {
                const bool full = co + 8 <= a.cout;
                _Float16* yb = reinterpret_cast<_Float16*>(a.y) + ((long)a.seg_yoff[0] + (long)tile_m * BM) * a.y_ld + (long)g * a.cout + co;
                const int rows = a.m_total - tile_m * BM < BM ? a.m_total - tile_m * BM : BM;
#pragma unroll 4
                for (int prow = tid / CPR; prow < rows; prow += 256 / CPR) {
                    const uint4 val = *reinterpret_cast<const uint4*>(tile + prow * TS + cc * 8);
                    _Float16* dst = yb + (long)prow * a.y_ld;
                    if (full) *reinterpret_cast<uint4*>(dst) = val;
                    else *reinterpret_cast<uint2*>(dst) = make_uint2(val.x, val.y);
                }
            }
            return;
        }
        if ((a.flags & ~(LLV_CONV_RELU | LLV_CONV_MASK_POS)) == LLV_CONV_RES_ADD && a.y_gstride == 0 && a.n_seg == 1 && (a.cout & 7) == 0 &&
            (!(a.flags & LLV_CONV_MASK_POS) || (a.mask_ld & 7) == 0)) {
            // residual add (+ ReLU), one segment (Bottleneck conv3 + identity, resnet.py:285-296; in-place gradient
            // accumulation): the residual tile comes in as 16 bytes per lane, whole cache lines per wave, is parked
            // in the staging tile, and each lane then adds its own 4 channels in fp32. Gathering it 8 bytes per lane
            // straight from memory touched sixteen 128-byte lines per wave-load for a quarter of their bytes.
            const bool relu = (a.flags & LLV_CONV_RELU) != 0;
            constexpr int CPR = BN / 8;
            const int cc = tid % CPR, co8 = tile_n * BN + cc * 8;
            const int rows = a.m_total - tile_m * BM < BM ? a.m_total - tile_m * BM : BM;
            const bool cok = co8 < a.cout;
            {
                const _Float16* rb = a.res + ((long)a.seg_yoff[0] + (long)tile_m * BM) * a.res_ld + (long)g * a.cout + co8;
#pragma unroll 4
                for (int prow = tid / CPR; prow < BM; prow += 256 / CPR) {
                    uint4 val = make_uint4(0, 0, 0, 0);
                    if (cok && prow < rows) val = *reinterpret_cast<const uint4*>(rb + (long)prow * a.res_ld);
                    *reinterpret_cast<uint4*>(tile + prow * TS + cc * 8) = val;
                }
            }
            __syncthreads();
#pragma unroll
            for (int j = 0; j < TM; ++j) {
                const int prow = wm * WM + j * 16 + frow;
#pragma unroll
                for (int i = 0; i < TN; ++i) {
                    _Float16* tp = tile + prow * TS + wn * WN + i * 16 + fch * 4;
                    const f16x4 r4 = *reinterpret_cast<const f16x4*>(tp);
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v[e] = acc[i][j][e] + (float)r4[e];
                        if (relu) v[e] = fmaxf(v[e], 0.f);
                    }
                    const f16x4 o4 = {(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
                    *reinterpret_cast<f16x4*>(tp) = o4;
                }
            }
            __syncthreads();
            if (cok) {
                _Float16* yb = reinterpret_cast<_Float16*>(a.y) + ((long)a.seg_yoff[0] + (long)tile_m * BM) * a.y_ld + (long)g * a.cout + co8;
                if (a.flags & LLV_CONV_MASK_POS) {       // (accumulating data gradient) the ReLU mask of the activation, 16 bytes per lane
                    const _Float16* mb = a.mask + ((long)a.seg_yoff[0] + (long)tile_m * BM) * a.mask_ld + (long)g * a.cout + co8;
#pragma unroll 4
                    for (int prow = tid / CPR; prow < rows; prow += 256 / CPR) {
                        f16x8 v8 = *reinterpret_cast<const f16x8*>(tile + prow * TS + cc * 8);
                        const f16x8 m8 = *reinterpret_cast<const f16x8*>(mb + (long)prow * a.mask_ld);
#pragma unroll
                        for (int t = 0; t < 8; ++t) v8[t] = m8[t] > (_Float16)0.f ? v8[t] : (_Float16)0.f;
                        *reinterpret_cast<f16x8*>(yb + (long)prow * a.y_ld) = v8;
                    }
                } else
#pragma unroll 4
                for (int prow = tid / CPR; prow < rows; prow += 256 / CPR)
                    *reinterpret_cast<uint4*>(yb + (long)prow * a.y_ld) = *reinterpret_cast<const uint4*>(tile + prow * TS + cc * 8);
            }
            return;
        }
        const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int j = 0; j < TM; ++j) {
            const int prow = wm * WM + j * 16 + frow;
            const int m = tile_m * BM + prow;
            EpiCtx c;
            const bool mok = m < a.m_total;
            if (mok) epi_locate(a, m, c);
#pragma unroll
            for (int i = 0; i < TN; ++i) {
                const int col = wn * WN + i * 16 + fch * 4;
                const int co = tile_n * BN + col;
                float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
                if (mok && co < a.cout) epi_compute_b(a, c, g * a.cout + co, v, zero4);      // bias is in the accumulator
                const f16x4 o4 = {(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
                *reinterpret_cast<f16x4*>(tile + prow * TS + col) = o4;
            }
        }
        __syncthreads();
        staged_store<BN, BM>(a, tile, g, tile_n, [&](int prow) { const int m = tile_m * BM + prow; return m < a.m_total ? m : -1; });
        return;
    }
#pragma unroll
    for (int j = 0; j < TM; ++j) {
        const int m = tile_m * BM + wm * WM + j * 16 + frow;
        if (m >= a.m_total) continue;
        EpiCtx c;
        epi_locate(a, m, c);
#pragma unroll
        for (int i = 0; i < TN; ++i) {
            const int co = tile_n * BN + wn * WN + i * 16 + fch * 4;
            if (co >= a.cout) continue;
            epi_store(a, c, g * a.cout + co, acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// conv3x3_halo_kernel: direct 3x3 / stride 1 / pad 1 convolution for the full-resolution, few-channel layers
// (FastDVDnet.py:22-74). One workgroup = an 8 x 32 pixel output tile x BN output channels. Per 32-channel
// chunk of the input it stages the (8+2) x (32+2) pixel halo tile AND that chunk's weights in LDS once; all
// nine taps are then formed from LDS (MFMA B fragments read at shifted pixel addresses), so every input byte
// crosses L1 once instead of nine times and a chunk is 9 barrier-free MFMA k-steps.
// K order inside a chunk: k = tap*CK + c (CK = min(Cin, 32) channels), weights packed per (chunk, n-tile) as
// [BN rows][wrow halfs] with wrow = Kc_pad + 8 (the +16 B spreads the 16 fragment rows over all LDS slots).
// ------------------------------------------------------------------------------------------------
struct HaloArgs {
    const _Float16* w;   // halo-packed weights
    int H, W;            // image size (= output size)
    int tiles_x, tiles_y;
    int n_chunks, ck, kc_pad, wrow;
    int ntiles_n;
    int nblk;
    long w_gstride;      // halfs between groups
};

constexpr int HALO_TH = 8, HALO_TW = 32, HALO_PX = (HALO_TH + 2) * (HALO_TW + 2);

__device__ __forceinline__ int halo_off(int px, int c8, int ck) {
    // byte offset of 16-B piece c8 of halo pixel px. For 64-B pixels XOR piece bit 1 with pixel bit 2: the 16
    // consecutive pixels of a fragment read then hit 16 distinct LDS slots whatever their alignment.
    if (ck == 32) return px * 64 + ((c8 ^ (((px >> 2) & 1) << 1)) << 4);
    return px * (ck * 2) + (c8 << 4);
}

template <int BN>
__global__ __launch_bounds__(256) void conv3x3_halo_kernel(const ConvKArgs a, const HaloArgs h) {
    constexpr int TN = BN / 16, TM = 4;
    constexpr int HV = (HALO_PX * 4 + 255) / 256;            // halo 16-B vectors per lane (CK = 32)
    constexpr int WV = (BN * 37 + 255) / 256;                // weight vectors per lane (wrow = 296 halfs = 37 vectors)
    extern __shared__ __attribute__((aligned(16))) unsigned char hsm[];
    unsigned char* s_halo = hsm;
    unsigned char* s_w = hsm + HALO_PX * 64;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int bid = blockIdx.x;
    {
        const int nblk = h.nblk;
        const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tx0 = (bid % h.tiles_x) * HALO_TW;
    const int t1 = bid / h.tiles_x;
    const int ty0 = (t1 % h.tiles_y) * HALO_TH;
    const int n = t1 / h.tiles_y;
    const int tile_n = blockIdx.y % h.ntiles_n;
    const int g = blockIdx.y / h.ntiles_n;

    const _Float16* xb = a.src_per_group ? (g == 0 ? a.x0 : (g == 1 ? a.x1 : a.x2)) : a.x0;
    const int xc0 = a.src_per_group ? 0 : g * a.cin;
    const bool multi = a.n_src > 1 && !a.src_per_group;   // sources concatenated along C, a chunk lies in one source
    const int cvec = h.ck >> 3;
    const int nvec_h = HALO_PX * cvec;
    const int wvec_row = h.wrow >> 3;
    const int nvec_w = BN * wvec_row;
    const long img_base = (long)a.seg_xoff[0] + (long)n * h.H * h.W;

    f32x4 acc[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int frow = lane & 15, fch = lane >> 4;
    int pbase[TM];
#pragma unroll
    for (int j = 0; j < TM; ++j) pbase[j] = (2 * wave + (j >> 1)) * (HALO_TW + 2) + (j & 1) * 16 + frow;

    const int nks = h.kc_pad >> 5;
    for (int ci = 0; ci < h.n_chunks; ++ci) {
        // ---- stage halo + weights of this chunk (all loads issued before the first LDS write)
        uint4 hreg[HV], wreg[WV];
#pragma unroll
        for (int i = 0; i < HV; ++i) {
            const int v = tid + i * 256;
            uint4 val = make_uint4(0, 0, 0, 0);
            if (v < nvec_h) {
                const int px = v / cvec, c8 = v - px * cvec;
                const int hy = px / (HALO_TW + 2), hx = px - hy * (HALO_TW + 2);
                const int gy = ty0 + hy - 1, gx = tx0 + hx - 1;
                if ((unsigned)gy < (unsigned)h.H && (unsigned)gx < (unsigned)h.W) {
                    const _Float16* xs = xb;
                    int coff = xc0 + ci * h.ck + c8 * 8;
                    if (multi) {
                        const int cv = (ci * h.ck >> 3) + c8, sidx = cv / a.src_cvec;
                        xs = sidx == 0 ? a.x0 : (sidx == 1 ? a.x1 : a.x2);
                        coff = (cv - sidx * a.src_cvec) * 8;
                    }
                    val = *reinterpret_cast<const uint4*>(xs + (img_base + (long)gy * h.W + gx) * a.x_ld + coff);
                }
            }
            hreg[i] = val;
        }
        const _Float16* wsrc = h.w + (long)g * h.w_gstride + ((long)(ci * h.ntiles_n + tile_n) * BN) * h.wrow;
#pragma unroll
        for (int i = 0; i < WV; ++i) {
            const int v = tid + i * 256;
            uint4 val = make_uint4(0, 0, 0, 0);
            if (v < nvec_w) val = *reinterpret_cast<const uint4*>(wsrc + (long)v * 8);
            wreg[i] = val;
        }
        if (ci > 0) __syncthreads();   // previous chunk's fragment reads are done
#pragma unroll
        for (int i = 0; i < HV; ++i) {
            const int v = tid + i * 256;
            if (v < nvec_h) {
                const int px = v / cvec, c8 = v - px * cvec;
                *reinterpret_cast<uint4*>(s_halo + halo_off(px, c8, h.ck)) = hreg[i];
            }
        }
#pragma unroll
        for (int i = 0; i < WV; ++i) {
            const int v = tid + i * 256;
            if (v < nvec_w) *reinterpret_cast<uint4*>(s_w + v * 16) = wreg[i];
        }
        __syncthreads();
        // ---- nks barrier-free k-steps out of LDS
        for (int ks = 0; ks < nks; ++ks) {
            const int kvec = ks * 4 + fch;
            int tap = kvec / cvec;
            const int c8 = kvec - tap * cvec;
            tap = tap > 8 ? 8 : tap;                           // k beyond 9*CK: weights are zero, keep the read in range
            const int tr = tap / 3, tq = tap - tr * 3;
            const int poff = tr * (HALO_TW + 2) + tq;
            f16x8 af[TN], bf[TM];
#pragma unroll
            for (int i = 0; i < TN; ++i)
                af[i] = *reinterpret_cast<const f16x8*>(s_w + ((i * 16 + frow) * h.wrow + ks * 32 + fch * 8) * 2);
#pragma unroll
            for (int j = 0; j < TM; ++j)
                bf[j] = *reinterpret_cast<const f16x8*>(s_halo + halo_off(pbase[j] + poff, c8, h.ck));
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
                for (int j = 0; j < TM; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
    }
    __syncthreads();

    // ---- epilogue through the LDS tile (rows = ty*32 + tx)
    auto row_to_m = [&](int prow) {
        const int oy = ty0 + (prow >> 5), ox = tx0 + (prow & 31);
        return (oy < h.H && ox < h.W) ? (int)(((long)n * h.H + oy) * h.W + ox) : -1;
    };
    constexpr int TS = BN + 8;
    _Float16* tile = reinterpret_cast<_Float16*>(hsm);
#pragma unroll
    for (int j = 0; j < TM; ++j) {
        const int prow = (2 * wave + (j >> 1)) * 32 + (j & 1) * 16 + frow;
        const int m = row_to_m(prow);
        EpiCtx c;
        if (m >= 0) epi_locate(a, m, c);
#pragma unroll
        for (int i = 0; i < TN; ++i) {
            const int col = i * 16 + fch * 4;
            const int co = tile_n * BN + col;
            float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
            if (a.stage_out) {
                if (m >= 0 && co < a.cout) epi_compute(a, c, g * a.cout + co, v);
                const f16x4 o4 = {(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
                *reinterpret_cast<f16x4*>(tile + prow * TS + col) = o4;
            } else if (m >= 0 && co < a.cout) {
                epi_store(a, c, g * a.cout + co, v[0], v[1], v[2], v[3]);
            }
        }
    }
    if (a.stage_out) {
        __syncthreads();
        staged_store<BN, 256>(a, tile, g, tile_n, row_to_m);
    }
}

// ------------------------------------------------------------------------------------------------
// conv3x3_tap_kernel<BN>: 3x3 / stride 1 / pad 1 convolution, Cin a multiple of 32, any number of segments
// (FastDVDnet inner layers FastDVDnet.py:39-74, RetinaNet head towers retina_head.py:50-81, FPN output convs
// fpn.py:139-150, bottleneck conv2 resnet.py:151-167). The implicit GEMM above fetches every input pixel nine
// times (once per tap), which makes it L1/L2-throughput bound at ~64 flop per fetched byte. Here one workgroup
// owns an 8 x 32 output-pixel tile x BN output channels; per 32-channel chunk the (8+2) x (32+2) halo tile is
// brought into LDS ONCE by global_load_lds, and the k loop runs chunk-major, tap-minor: each (chunk, tap) step
// DMAs only that step's BN x 32 weight slice (8 KiB for BN = 128) and forms the pixel operand from the resident
// halo at a shifted address. ~200 flop per fetched byte, 32 MFMAs per wave per barrier.
//   ring: 2 halo buffers (22 KiB each; chunk c+1 streams in one piece per wave per step while chunk c computes)
//         2 weight buffers (BN x 64 B); everything issued in step s has landed before step s+1 starts (vmcnt(0)).
// Weights are the ordinary [row][k_pad] packing with k = tap*Cin + c (same buffer the implicit GEMM reads).
// ------------------------------------------------------------------------------------------------
// Dispatch order of the tap kernels: the grid is one-dimensional in (spatial tile, n-tile); hardware deals workgroup L to XCD L & 7.
// Every XCD walks ITS contiguous run of spatial tiles with the n-tiles of one spatial tile next to each other (n fastest), so the
// workgroups that read the same halo run at the same time behind the same L2. With n-tiles in blockIdx.y the second n-tile of a
// spatial tile started a whole grid later and re-read the input from HBM (measured per launch, profiles/r03a_traffic.txt: UpBlock
// convs 2.2x / 3.1x, retina_cls 4.8x the algorithmic bytes). The grid is padded to 8 * ceil(nblk / 8) * ntn; the (< 8 * ntn)
// workgroups without a tile return at once.
__device__ __forceinline__ bool tap_dispatch(int lin, int nblk, int ntn, int& bid, int& tile_n) {
    const int q = nblk >> 3, r = nblk & 7, xcd = lin & 7, idx = lin >> 3;
    tile_n = idx % ntn;
    const int sl = idx / ntn;
    if (sl >= (xcd < r ? q + 1 : q)) return false;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + sl;
    return true;
}
static inline unsigned tap_grid_x(int nblk, int ntn) { return 8u * (unsigned)((nblk + 7) / 8) * (unsigned)ntn; }

struct TapArgs {
    int tiles_x[LLV_MAX_SEG], tiles_y[LLV_MAX_SEG], blk0[LLV_MAX_SEG + 1];
    int nblk, n_chunks, ntn;     // ntn = n-tiles (output-channel tiles) per group
    unsigned x_bytes, w_bytes;   // sizes of the source tensors / weight buffer (buffer-resource bounds, < 2 GiB)
};

constexpr int TAP_HPIECES = (HALO_PX * 4 + 63) / 64;        // 1-KiB pieces per halo tile (64-B pixels): 22
constexpr int TAP_HBYTES = TAP_HPIECES * 1024;

// halo image: pixel (hy, hx) of the (8+2) x (32+2) tile at byte (hy*34 + hx)*64, its 16-B channel piece c8 in slot
// c8 ^ (((hx>>2)&1)<<1). Keyed on the COLUMN only, so a tap's row shift is a pure immediate offset, and the 16
// consecutive pixels of an MFMA fragment hit 16 distinct LDS slots at any column alignment.
__device__ __forceinline__ int tap_halo_off(int hy, int hx, int c8) {
    return (hy * (HALO_TW + 2) + hx) * 64 + ((c8 ^ (((hx >> 2) & 1) << 1)) << 4);
}

// R = depth of the weight ring: the slice of step s + R - 1 is requested while step s computes (R = 2: one step ahead).
// MSRC: the input is several tensors concatenated along C (a chunk's source is looked up per request); the common
// single-source instantiation has no such scalar work in its step.
template <int BN, int R = 2, bool MSRC = false>
__global__ __launch_bounds__(256, 2) void conv3x3_tap_kernel(const ConvKArgs a, const TapArgs h) {
    constexpr int TN = BN / 16, TM = 4;
    constexpr int WBYTES = BN * 64;
    constexpr int WPW = BN / 64;                             // weight pieces per wave per step
    constexpr int HPW = (TAP_HPIECES + 3) / 4;               // halo pieces per wave per chunk (<= 6)
    constexpr int TS = BN + 8;
    // R = 0: the grouped main loop -- one barrier per tap ROW (3 taps, 3 * TN * TM MFMAs per wave), weight ring of 3 groups requested
    // two groups ahead, counted s_waitcnt vmcnt(N) and a barrier that leaves the DMA queue alone (__syncthreads() drains vmcnt to 0).
    // Same k order (chunk-major, tap-minor): bit-identical results. Measured 3 % SLOWER than the stepwise loop (80 KiB of LDS: two
    // workgroups per CU instead of three): the kernel is bound by the CU's vector-memory path (one 1-KiB LDS-DMA piece per ~72 cycles,
    // one 1-KiB store per ~114 cycles), which pipelining does not touch. Kept as LLV_TILE_TAP64_G3 for A/B.
    constexpr bool G3 = R == 0;
    constexpr int NWS = G3 ? 9 : R;                          // weight slices resident
    constexpr int RING = 2 * TAP_HBYTES + NWS * WBYTES;
    constexpr int D = G3 ? 1 : R - 1;                        // prefetch distance in steps (stepwise loop)
    static_assert(!G3 || (BN == 64 && HPW == 6), "grouped loop: 64-channel tiles, two halo pieces per wave per group");
    constexpr int EPI0 = 256 * TS * 2;
    constexpr int EPI1 = 256 * 4 * (BN / 4 + 8) * 2;         // PixelShuffle: rows in (pixel, sub-pixel) order
    constexpr int EPI = EPI0 > EPI1 ? EPI0 : EPI1;
    constexpr int SMEM = RING > EPI ? RING : EPI;
    static_assert(2 * SMEM <= 160 * 1024, "two workgroups per CU");
    constexpr int ROWB = (HALO_TW + 2) * 64;                 // bytes per halo row
    static_assert(HPW <= 8, "halo pieces are dealt one per tap step, with one step to spare");
    __shared__ __attribute__((aligned(16))) unsigned char smem[SMEM];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int bid, tile_n;
    if (!tap_dispatch(blockIdx.x, h.nblk, h.ntn, bid, tile_n)) return;
    int seg = 0;
#pragma unroll
    for (int s = 1; s < LLV_MAX_SEG; ++s)
        if (s < a.n_seg && bid >= h.blk0[s]) seg = s;
    const int local = bid - h.blk0[seg];
    const int H = a.seg_h[seg], W = a.seg_w[seg];
    const int tx0 = (local % h.tiles_x[seg]) * HALO_TW;
    const int t1 = local / h.tiles_x[seg];
    const int ty0 = (t1 % h.tiles_y[seg]) * HALO_TH;
    const int n = t1 / h.tiles_y[seg];
    const int g = blockIdx.z;                                // group: input channels [g*Cin, ..), its own weight block
    const int img_base = a.seg_xoff[seg] + n * H * W;        // pixel index of this image's (0, 0)

    const __amdgpu_buffer_rsrc_t rW = make_rsrc(a.w + (long)g * a.w_gstride, h.w_bytes);
    const __amdgpu_buffer_rsrc_t rX0 = make_rsrc(a.x0, h.x_bytes);

    // ---- per-lane DMA offsets, fixed for the whole k loop (the chunk / tap part is a scalar offset)
    int hv[HPW];
#pragma unroll
    for (int i = 0; i < HPW; ++i) {
        // grouped loop: every wave issues HPW pieces (the counted waits need equal counts); a wave without a last piece repeats
        // its previous one (same bytes to the same place)
        const int pc = (G3 && wave + 4 * i >= TAP_HPIECES) ? wave + 4 * (i - 1) : wave + 4 * i;
        const int px = pc * 16 + (lane >> 2);
        hv[i] = (int)h.x_bytes;                              // out of range: zeros
        if (px < HALO_PX) {
            const int hy = px / (HALO_TW + 2), hx = px - hy * (HALO_TW + 2);
            const int gy = ty0 + hy - 1, gx = tx0 + hx - 1;
            const int c8 = (lane & 3) ^ (((hx >> 2) & 1) << 1);
            if ((unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W) hv[i] = ((img_base + gy * W + gx) * a.x_ld + g * a.cin + c8 * 8) * 2;
        }
    }
    const int wch = (lane & 3) ^ ((0 - (lane >> 4)) & 3);
    int wv[WPW];
#pragma unroll
    for (int i = 0; i < WPW; ++i) {
        const int grow = tile_n * BN + (wave + 4 * i) * 16 + (lane >> 2);
        wv[i] = grow < a.w_rows ? (grow * a.k_pad + wch * 8) * 2 : (int)h.w_bytes;
    }

    unsigned char* const sHalo = smem;
    unsigned char* const sWgt = smem + 2 * TAP_HBYTES;

    auto issue_w = [&](int c, int t, int buf) __attribute__((always_inline)) {   // weights of step (chunk c, tap t): k offset t*Cin + 32c
        const int soff = (t * a.cin + c * 32) * 2;
#pragma unroll
        for (int i = 0; i < WPW; ++i) dma16(rW, sWgt + buf * WBYTES + (wave + 4 * i) * 1024, wv[i], soff);
    };
    auto issue_h = [&](int c, int i, int buf) __attribute__((always_inline)) {   // piece wave + 4i of chunk c's halo (i static)
        if (!G3 && wave + 4 * i >= TAP_HPIECES) return;       // wave-uniform
        const int pc = (G3 && wave + 4 * i >= TAP_HPIECES) ? wave + 4 * (i - 1) : wave + 4 * i;
        int cv = c * 4;
        unsigned char* dst = sHalo + buf * TAP_HBYTES + pc * 1024;
        if (MSRC) {
            // the chunk's source tensor: a resource built from a masked pointer sum (scalar), not a choice among three
            // resources (which the compiler turned into a table in scratch memory: 512 / 1040 B of private segment)
            const int sidx = cv / a.src_cvec;
            cv -= sidx * a.src_cvec;
            dma16(make_rsrc(src_sel(a, sidx), h.x_bytes), dst, hv[i], cv * 16);
            return;
        }
        dma16(rX0, dst, hv[i], cv * 16);
    };
    // the first chunk's halo and the first weight slice are on their way while the rest of the set-up runs
#pragma unroll
    for (int i = 0; i < HPW; ++i) issue_h(0, i, 0);
    const int nsteps = h.n_chunks * 9;
    auto issue_wg = [&](int c, int tr) __attribute__((always_inline)) {   // the three slices of tap row tr of chunk c -> ring group tr
#pragma unroll
        for (int tq = 0; tq < 3; ++tq) issue_w(c, tr * 3 + tq, tr * 3 + tq);
    };
    if (G3) {
        issue_wg(0, 0);
        issue_wg(0, 1);
    } else {
#pragma unroll
        for (int s0 = 0; s0 < D; ++s0)
            if (s0 < nsteps) issue_w(s0 / 9, s0 % 9, s0 % (G3 ? 1 : R));
    }

    const int frow = lane & 15, fch = lane >> 4;
    // MFMA fragment read addresses: weights row i*16 + frow -> wa + i*1024; halo pixel (2*wave + (j>>1) + tr,
    // (j&1)*16 + frow + tq) -> hb[j][tq] + tr*ROWB
    const int wa = lds_off(frow, fch);
    int hb[TM][3];
#pragma unroll
    for (int j = 0; j < TM; ++j)
#pragma unroll
        for (int tq = 0; tq < 3; ++tq) hb[j][tq] = tap_halo_off(2 * wave + (j >> 1), (j & 1) * 16 + frow + tq, fch);

    // on the LDS-staged paths the bias rides in the accumulator (one add per output saved in the epilogue)
    float4 bias4[TN];
#pragma unroll
    for (int i = 0; i < TN; ++i) {
        const int co = tile_n * BN + i * 16 + fch * 4;
        bias4[i] = co < a.cout ? *reinterpret_cast<const float4*>(a.bias + g * a.cout + co) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const bool bias_in_acc = a.stage_out != 0;
    f32x4 acc[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j)
            acc[i][j] = bias_in_acc ? (f32x4){bias4[i].x, bias4[i].y, bias4[i].z, bias4[i].w} : (f32x4){0.f, 0.f, 0.f, 0.f};

    // one (chunk, tap) step with static tap T, halo buffer HB and weight buffer WB. A macro, not a lambda: with the
    // accumulators captured by reference the compiler leaves the 512-byte array of the BN = 128 variant in scratch.
#define TAP_STEP(T, HB, c, more)                                                                              \
    {                                                                                                         \
        /* loads retire in order: all but the D-1 youngest weight groups landed = this step's operands are in */ \
        if (D == 1 || sidx + D > nsteps) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                     \
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((D - 1) * WPW) : "memory");                             \
        __syncthreads(); /* this step's operands are in LDS; the buffers of step s-1 are free */              \
        if ((T) < HPW && (more)) issue_h((c) + 1, (T), (HB) ^ 1);                                             \
        if (sidx + D < nsteps) issue_w((c) + ((T) + D) / 9, ((T) + D) % 9, (unsigned)(sidx + D) % R);         \
        const unsigned char* sWc = sWgt + ((unsigned)sidx % R) * WBYTES + wa;                                 \
        f16x8 af[TN], bf[TM];                                                                                 \
        _Pragma("unroll") for (int j = 0; j < TM; ++j)                                                        \
            bf[j] = *reinterpret_cast<const f16x8*>(sHalo + (HB) * TAP_HBYTES + ((T) / 3) * ROWB + hb[j][(T) % 3]); \
        _Pragma("unroll") for (int i = 0; i < TN; ++i)                                                        \
            af[i] = *reinterpret_cast<const f16x8*>(sWc + i * 1024);                                          \
        _Pragma("unroll") for (int i = 0; i < TN; ++i)                                                        \
            _Pragma("unroll") for (int j = 0; j < TM; ++j)                                                    \
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[i], bf[j], acc[i][j], 0, 0, 0);         \
        ++sidx;                                                                                               \
    }
#define TAP_CHUNK(HB, c, more)                                                                                \
    TAP_STEP(0, HB, c, more) TAP_STEP(1, HB, c, more) TAP_STEP(2, HB, c, more)                                \
    TAP_STEP(3, HB, c, more) TAP_STEP(4, HB, c, more) TAP_STEP(5, HB, c, more)                                \
    TAP_STEP(6, HB, c, more) TAP_STEP(7, HB, c, more) TAP_STEP(8, HB, c, more)

    // grouped loop, tap row TR of chunk c in halo buffer HB. Issue order per group: next chunk's halo pieces, then the weights of the
    // group after next; so at the top of a group everything older than [those halo pieces (tap rows 1, 2 only: row 0 needs them),
    // the next group's weights] has to have landed.
#define TAP_GROUP(TR, HB, c, more)                                                                            \
    {                                                                                                         \
        const int gi = (c) * 3 + (TR);                                                                        \
        const int allow = (gi + 1 < ngroups ? 3 * WPW : 0) + (((TR) != 0 && (more)) ? 2 : 0);                 \
        if (allow >= 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");                                      \
        else if (allow >= 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");                                 \
        else if (allow >= 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");                                 \
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                 \
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");                                       \
        if (more) { issue_h((c) + 1, 2 * (TR), (HB) ^ 1); issue_h((c) + 1, 2 * (TR) + 1, (HB) ^ 1); }         \
        if (gi + 2 < ngroups) issue_wg((c) + ((TR) + 2) / 3, ((TR) + 2) % 3);                                 \
        __builtin_amdgcn_sched_barrier(0);                                                                    \
        _Pragma("unroll") for (int tq = 0; tq < 3; ++tq) {                                                    \
            const unsigned char* sWc = sWgt + ((TR) * 3 + tq) * WBYTES + wa;                                  \
            f16x8 af[TN], bf[TM];                                                                             \
            _Pragma("unroll") for (int j = 0; j < TM; ++j)                                                    \
                bf[j] = *reinterpret_cast<const f16x8*>(sHalo + (HB) * TAP_HBYTES + (TR) * ROWB + hb[j][tq]); \
            _Pragma("unroll") for (int i = 0; i < TN; ++i)                                                    \
                af[i] = *reinterpret_cast<const f16x8*>(sWc + i * 1024);                                      \
            _Pragma("unroll") for (int i = 0; i < TN; ++i)                                                    \
                _Pragma("unroll") for (int j = 0; j < TM; ++j)                                                \
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[i], bf[j], acc[i][j], 0, 0, 0);     \
        }                                                                                                     \
    }
#define TAP_GCHUNK(HB, c, more) TAP_GROUP(0, HB, c, more) TAP_GROUP(1, HB, c, more) TAP_GROUP(2, HB, c, more)

    int sidx = 0;
    if (G3) {
        const int ngroups = h.n_chunks * 3;
        for (int c = 0; c < h.n_chunks; c += 2) {
            { const bool more = c + 1 < h.n_chunks; TAP_GCHUNK(0, c, more) }
            if (c + 1 < h.n_chunks) { const bool more = c + 2 < h.n_chunks; TAP_GCHUNK(1, c + 1, more) }
        }
    } else {
        for (int c = 0; c < h.n_chunks; c += 2) {
            { const bool more = c + 1 < h.n_chunks; TAP_CHUNK(0, c, more) }
            if (c + 1 < h.n_chunks) { const bool more = c + 2 < h.n_chunks; TAP_CHUNK(1, c + 1, more) }
        }
    }
#undef TAP_STEP
#undef TAP_CHUNK
#undef TAP_GROUP
#undef TAP_GCHUNK
    __syncthreads();

    // ---- epilogue through the LDS tile (rows = ty*32 + tx); the tile's position is known, no row -> pixel search
    const int ybase = a.seg_yoff[seg] + n * H * W;
    auto loc = [&](int prow, EpiCtx& c) {
        const int oy = ty0 + (prow >> 5), ox = tx0 + (prow & 31);
        if (oy >= H || ox >= W) return false;
        c.opix = ybase + oy * W + ox;
        c.rpix = c.opix;
        c.ox = ox; c.oy = oy; c.n = n; c.Ho = H; c.Wo = W; c.yo = a.seg_yoff[seg];
        if (a.flags & LLV_CONV_RES_UPS) epi_res_ups(a, c);
        return true;
    };
    _Float16* tile = reinterpret_cast<_Float16*>(smem);
    if (a.stage_out && !(a.flags & ~LLV_CONV_RELU) && a.y_gstride == 0) {
        // the common epilogue (bias [+ ReLU] -> fp16): bias is already in the accumulator, ReLU on the rounded halves,
        // no pixel lookup before the staging (rows of pixels outside the image are simply never stored)
        const bool relu = (a.flags & LLV_CONV_RELU) != 0;
        const f16x4 z4 = {(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
#pragma unroll
        for (int j = 0; j < TM; ++j) {
            const int prow = (2 * wave + (j >> 1)) * 32 + (j & 1) * 16 + frow;
#pragma unroll
            for (int i = 0; i < TN; ++i) {
                f16x4 o4 = {(_Float16)acc[i][j][0], (_Float16)acc[i][j][1], (_Float16)acc[i][j][2], (_Float16)acc[i][j][3]};
                if (relu) o4 = __builtin_elementwise_max(o4, z4);
                *reinterpret_cast<f16x4*>(tile + prow * TS + i * 16 + fch * 4) = o4;
            }
        }
        __syncthreads();
        constexpr int CPR = BN / 8;
        const int cc = tid % CPR, co = tile_n * BN + cc * 8;
        if (co < a.cout) {
            const bool full = co + 8 <= a.cout;
            _Float16* ybase_p = reinterpret_cast<_Float16*>(a.y) + (long)g * a.cout + co;
#pragma unroll 4
            for (int prow = tid / CPR; prow < 256; prow += 256 / CPR) {
                const int oy = ty0 + (prow >> 5), ox = tx0 + (prow & 31);
                if (oy >= H || ox >= W) continue;
                const uint4 val = *reinterpret_cast<const uint4*>(tile + prow * TS + cc * 8);
                _Float16* dst = ybase_p + (long)(ybase + oy * W + ox) * a.y_ld;
                if (full) *reinterpret_cast<uint4*>(dst) = val;
                else *reinterpret_cast<uint2*>(dst) = make_uint2(val.x, val.y);
            }
        }
        return;
    }
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (a.stage_out && (a.flags & LLV_CONV_PIXSHUF) && (a.flags & LLV_CONV_PS_EMAJOR)) {
        // PixelShuffle with the conv channels packed sub-pixel-major (channel = e*OC + cc, e = 2*dy + dx, OC = Cout/4
        // output channels): a lane's 4 consecutive channels are 4 consecutive OUTPUT channels of one sub-pixel, so the
        // tile is staged as it is and a row's 16-byte chunks go to the sub-pixel they belong to. A 64-channel workgroup
        // then writes whole output pixels (OC <= 64) instead of one eighth of four of them.
        const int OC = a.cout >> 2;
#pragma unroll
        for (int j = 0; j < TM; ++j) {
            const int prow = (2 * wave + (j >> 1)) * 32 + (j & 1) * 16 + frow;
#pragma unroll
            for (int i = 0; i < TN; ++i) {
                const f16x4 o4 = {(_Float16)acc[i][j][0], (_Float16)acc[i][j][1], (_Float16)acc[i][j][2], (_Float16)acc[i][j][3]};
                *reinterpret_cast<f16x4*>(tile + prow * TS + i * 16 + fch * 4) = o4;
            }
        }
        // destination pixels and the skip values of this lane's CPR rows: requested BEFORE the barrier, so the global reads
        // are in flight while the tile is being staged (one exposed latency per tile instead of one per row)
        constexpr int CPR = BN / 8, RPI = 256 / CPR;
        const int cc = tid % CPR, ch = tile_n * BN + cc * 8;      // first conv channel of this lane's 8
        const bool chok = ch < a.cout;
        const int e = chok ? ch / OC : 0, oc = ch - e * OC;       // sub-pixel, first output channel
        const int dy = e >> 1, dx = e & 1;
        const bool has_res = (a.flags & LLV_CONV_RES_ADD) != 0;
        // BN = 64 (8 rows per lane): pixel indices and skip values are requested BEFORE the barrier; the 128-channel
        // variant (16 rows per lane) would spill and keeps the row-at-a-time loop
        constexpr bool PRE = BN <= 64;
        long pixs[PRE ? CPR : 1];
        f16x8 r8s[PRE ? CPR : 1];
        if (PRE) {
#pragma unroll
            for (int it = 0; it < CPR; ++it) {
                const int prow = tid / CPR + it * RPI;
                const int oy = ty0 + (prow >> 5), ox = tx0 + (prow & 31);
                const bool ok = chok && oy < H && ox < W && 2 * oy + dy < a.ps_h && 2 * ox + dx < a.ps_w;
                pixs[it] = ok ? (long)a.seg_yoff[seg] + ((long)(n * a.ps_h + 2 * oy + dy)) * a.ps_w + (2 * ox + dx) : -1;
                if (ok && has_res) r8s[it] = *reinterpret_cast<const f16x8*>(a.res + pixs[it] * a.res_ld + g * OC + oc);
            }
        }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < CPR; ++it) {
            const int prow = tid / CPR + it * RPI;
            long pix;
            f16x8 r8;
            if (PRE) {
                pix = pixs[it];
                r8 = r8s[it];
            } else {
                const int oy = ty0 + (prow >> 5), ox = tx0 + (prow & 31);
                const bool ok = chok && oy < H && ox < W && 2 * oy + dy < a.ps_h && 2 * ox + dx < a.ps_w;
                pix = ok ? (long)a.seg_yoff[seg] + ((long)(n * a.ps_h + 2 * oy + dy)) * a.ps_w + (2 * ox + dx) : -1;
                if (ok && has_res) r8 = *reinterpret_cast<const f16x8*>(a.res + pix * a.res_ld + g * OC + oc);
            }
            if (pix < 0) continue;
            const f16x8 t8 = *reinterpret_cast<const f16x8*>(tile + prow * TS + cc * 8);
            float v[8];
#pragma unroll
            for (int t = 0; t < 8; ++t) v[t] = (float)t8[t];
            if (has_res) {
#pragma unroll
                for (int t = 0; t < 8; ++t) v[t] += (float)r8[t];
            }
            if (a.flags & (LLV_CONV_RELU | LLV_CONV_LEAKY)) {
#pragma unroll
                for (int t = 0; t < 8; ++t) v[t] = act_fn(a, v[t]);
            }
            f16x8 o8;
#pragma unroll
            for (int t = 0; t < 8; ++t) o8[t] = (_Float16)v[t];
            *reinterpret_cast<f16x8*>(reinterpret_cast<_Float16*>(a.y) + pix * a.y_ld + g * OC + oc) = o8;
        }
        return;
    }
    constexpr int TS2 = BN / 4 + 8;
    constexpr bool SHUF_OK = 256 * 4 * TS2 * 2 <= SMEM;       // the shuffled tile fits (BN = 64)
    if (SHUF_OK && a.stage_out && (a.flags & LLV_CONV_PIXSHUF)) {
        // PixelShuffle: transpose each 4-channel group across the four fragment rows so that a lane holds 4
        // consecutive OUTPUT channels of one sub-pixel, and stage rows in (pixel, sub-pixel) order
#pragma unroll
        for (int j = 0; j < TM; ++j) {
            const int prow = (2 * wave + (j >> 1)) * 32 + (j & 1) * 16 + frow;
#pragma unroll
            for (int i = 0; i < TN; ++i) {
                float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};      // bias already inside
                row_transpose4(v);
                const f16x4 o4 = {(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
                *reinterpret_cast<f16x4*>(tile + (prow * 4 + fch) * TS2 + i * 4) = o4;
            }
        }
        __syncthreads();
        staged_store_shuffled<BN, 256>(a, tile, g, tile_n, loc);
        return;
    }
    if (a.flags & LLV_CONV_CLSMAX) {
        // class convolution of a dense head (host: OUT_F32, one group, no staging, cout % 16 == 0, cls_c % 16 == 0): a lane holds, per
        // pixel row j, TN blocks of 4 channels; the four lanes fch = 0..3 of a pixel hold one 16-channel block together, and a block lies
        // inside one anchor. Per anchor touched by this tile (at most three): maximum over the lane's blocks, over the four lanes (two
        // lane swaps), one atomicMax per pixel. The pixel's 128-channel segment is stored only when it has a logit above store_thr.
        const int A = a.cout / a.cls_c;
        // (the bias is re-read here, as epi_compute does: keeping the pre-loop copy alive across the main loop would cost 32 registers --
        // but ONCE for the tile, all TN loads in flight together: inside the row loop every one of the TM x TN loads was waited for by
        // itself, 64 exposed load latencies per tile; found in the ISA in round 4)
        float4 b4s[TN];
#pragma unroll
        for (int i = 0; i < TN; ++i) {
            const int co0 = tile_n * BN + i * 16;
            b4s[i] = co0 < a.cout ? *reinterpret_cast<const float4*>(a.bias + co0 + fch * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int j = 0; j < TM; ++j) {
            const int prow = (2 * wave + (j >> 1)) * 32 + (j & 1) * 16 + frow;
            EpiCtx ec;
            const bool ok = loc(prow, ec);
            unsigned bits = 0;                                // bit i: this pixel's 16-channel block i has a logit above store_thr
            float run = -INFINITY;
#pragma unroll
            for (int i = 0; i < TN; ++i) {
                const int co0 = tile_n * BN + i * 16;
                const bool live = co0 < a.cout;
                if (live) {
                    const float4 b4 = b4s[i];
                    acc[i][j][0] += b4.x; acc[i][j][1] += b4.y; acc[i][j][2] += b4.z; acc[i][j][3] += b4.w;
                    const float m4 = fmaxf(fmaxf(acc[i][j][0], acc[i][j][1]), fmaxf(acc[i][j][2], acc[i][j][3]));
                    run = fmaxf(run, m4);
                    // the block's four lanes are frow, frow + 16, frow + 32, frow + 48: one ballot instead of two lane swaps
                    const unsigned long long mk = __ballot(m4 > a.store_thr);
                    bits |= ((mk >> frow) & 0x0001000100010001ull) ? (1u << i) : 0u;
                }
                const int an = co0 / a.cls_c;
                const bool last = i == TN - 1 || (co0 + 16) / a.cls_c != an || !(co0 + 16 < a.cout);
                if (live && last) {                           // uniform over the wave: depends on i and the tile only
                    float r = fmaxf(run, __shfl_xor(run, 16));
                    r = fmaxf(r, __shfl_xor(r, 32));
                    if (ok && fch == 0) atomicMax(a.clsmax + (long)ec.opix * A + an, f32_ordered(r));
                    run = -INFINITY;
                }
            }
            if (ok) {
                if (fch == 0) a.segw[(long)ec.opix * h.ntn + tile_n] = (unsigned char)bits;
#pragma unroll
                for (int i = 0; i < TN; ++i) {
                    const int co = tile_n * BN + i * 16 + fch * 4;
                    if ((bits >> i) & 1u)                     // (a block past cout never sets its bit)
                        *reinterpret_cast<float4*>(reinterpret_cast<float*>(a.y) + (long)ec.opix * a.y_ld + co) =
                            make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
                }
            }
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < TM; ++j) {
        const int prow = (2 * wave + (j >> 1)) * 32 + (j & 1) * 16 + frow;
        EpiCtx ec;
        const bool ok = loc(prow, ec);
#pragma unroll
        for (int i = 0; i < TN; ++i) {
            const int col = i * 16 + fch * 4;
            const int co = tile_n * BN + col;
            float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
            if (a.stage_out) {
                if (ok && co < a.cout) epi_compute_b(a, ec, g * a.cout + co, v, zero4);
                const f16x4 o4 = {(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
                *reinterpret_cast<f16x4*>(tile + prow * TS + col) = o4;
            } else if (ok && co < a.cout) {
                epi_store(a, ec, g * a.cout + co, v[0], v[1], v[2], v[3]);
            }
        }
    }
    if (a.stage_out) {
        __syncthreads();
        staged_store_loc<BN, 256>(a, tile, g, tile_n, loc);
    }
}

// inputcv_kernel: FastDVDnet's InputCvBlock in one launch (FastDVDnet.py:22-36):
//   Conv2d(3F, 30F, 3, groups=F) + BN + ReLU  ->  Conv2d(30F, 32, 3) + BN + ReLU        (F = 3 frames)
// One workgroup = 8 x 32 output pixels x 32 channels. For each frame g it loads the (8+4) x (32+4) pixel patch of
// frame g (NHWC8) and the group's weights, computes the FIRST conv for the (8+2) x (32+2) halo tile with MFMA and
// writes it (bias + ReLU, zero outside the image = the second conv's zero padding) into the LDS halo buffer,
// then runs the second conv's 9 k-steps for that 32-channel chunk exactly like conv3x3_halo_kernel.
// The 90-channel full-resolution intermediate never exists in HBM.
// ------------------------------------------------------------------------------------------------
struct PreArgs {
    const _Float16* w;    // [group][32 rows][wrow] halo packing of the grouped conv (K = tap*8 + c, padded to 96)
    const float* bias;    // [groups*32]
    int wrow;             // 104
};
constexpr int PATCH_W = HALO_TW + 4, PATCH_H = HALO_TH + 4, PATCH_PX = PATCH_W * PATCH_H;   // 36 x 12

// CK0 = channels per pixel of the first conv's input (8: one NHWC8 frame per group, InputCvBlock;
//       32: one 32-channel tensor, OutputCvBlock FastDVDnet.py:64-74), BN2 = output tile of the second conv.
template <int CK0, int BN2>
__global__ __launch_bounds__(256) void convpair_kernel(const ConvKArgs a, const HaloArgs h, const PreArgs pre) {
    constexpr int BN = 32;                                   // channels of the intermediate per chunk
    constexpr int TN2 = BN2 / 16, TM = 4;
    constexpr int NMT = (HALO_PX + 15) / 16;                 // 22 M-tiles of 16 halo pixels
    constexpr int PV = (PATCH_PX * (CK0 / 8) + 255) / 256;   // patch vectors per lane
    constexpr int NKS0 = (9 * CK0 + 31) / 32;                // k-steps of the first conv
    constexpr int W0V = (BN * (NKS0 * 32 + 8) / 8 + 255) / 256;
    constexpr int W1V = (BN2 * 37 + 255) / 256;
    extern __shared__ __attribute__((aligned(16))) unsigned char hsm[];
    unsigned char* s_halo = hsm;                             // 340 x 64 B   (also the epilogue tile later)
    unsigned char* s_w = s_halo + HALO_PX * 64;              // BN2 x wrow halfs (second conv, this chunk)
    unsigned char* s_in = s_w + BN2 * h.wrow * 2;            // 432 x CK0*2 B (input patch)
    unsigned char* s_w0 = s_in + PATCH_PX * CK0 * 2;         // 32 x pre.wrow halfs (first conv, this group)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int bid = blockIdx.x;
    {
        const int nblk = h.nblk;
        const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tx0 = (bid % h.tiles_x) * HALO_TW;
    const int t1 = bid / h.tiles_x;
    const int ty0 = (t1 % h.tiles_y) * HALO_TH;
    const int n = t1 / h.tiles_y;
    const long img_base = (long)a.seg_xoff[0] + (long)n * h.H * h.W;
    const int frow = lane & 15, fch = lane >> 4;

    f32x4 acc[TN2][TM];
#pragma unroll
    for (int i = 0; i < TN2; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    int pbase[TM];
#pragma unroll
    for (int j = 0; j < TM; ++j) pbase[j] = (2 * wave + (j >> 1)) * (HALO_TW + 2) + (j & 1) * 16 + frow;

    constexpr int cv0 = CK0 / 8;
    const int nvec_w = BN2 * (h.wrow >> 3);
    const int nvec_w0 = BN * (pre.wrow >> 3);
    for (int g = 0; g < h.n_chunks; ++g) {
        const _Float16* xg = g == 0 ? a.x0 : (g == 1 ? a.x1 : a.x2);
        // ---- loads of this group: input patch, first-conv weights, second-conv chunk weights
        uint4 preg[PV], w0reg[W0V], wreg[W1V];
#pragma unroll
        for (int i = 0; i < PV; ++i) {
            const int v = tid + i * 256;
            uint4 val = make_uint4(0, 0, 0, 0);
            if (v < PATCH_PX * cv0) {
                const int pp = v / cv0, c8 = v - pp * cv0;
                const int py = pp / PATCH_W, px = pp - py * PATCH_W;
                const int gy = ty0 + py - 2, gx = tx0 + px - 2;
                if ((unsigned)gy < (unsigned)h.H && (unsigned)gx < (unsigned)h.W)
                    val = *reinterpret_cast<const uint4*>(xg + (img_base + (long)gy * h.W + gx) * a.x_ld + c8 * 8);
            }
            preg[i] = val;
        }
        const _Float16* w0src = pre.w + (long)g * BN * pre.wrow;
#pragma unroll
        for (int i = 0; i < W0V; ++i) {
            const int v = tid + i * 256;
            w0reg[i] = v < nvec_w0 ? *reinterpret_cast<const uint4*>(w0src + (long)v * 8) : make_uint4(0, 0, 0, 0);
        }
        const _Float16* wsrc = h.w + ((long)g * BN2) * h.wrow;
#pragma unroll
        for (int i = 0; i < W1V; ++i) {
            const int v = tid + i * 256;
            wreg[i] = v < nvec_w ? *reinterpret_cast<const uint4*>(wsrc + (long)v * 8) : make_uint4(0, 0, 0, 0);
        }
        if (g > 0) __syncthreads();                            // previous chunk's fragment reads are done
#pragma unroll
        for (int i = 0; i < PV; ++i) {
            const int v = tid + i * 256;
            if (v < PATCH_PX * cv0) {
                const int pp = v / cv0, c8 = v - pp * cv0;
                *reinterpret_cast<uint4*>(s_in + halo_off(pp, c8, CK0)) = preg[i];
            }
        }
#pragma unroll
        for (int i = 0; i < W0V; ++i) {
            const int v = tid + i * 256;
            if (v < nvec_w0) *reinterpret_cast<uint4*>(s_w0 + v * 16) = w0reg[i];
        }
#pragma unroll
        for (int i = 0; i < W1V; ++i) {
            const int v = tid + i * 256;
            if (v < nvec_w) *reinterpret_cast<uint4*>(s_w + v * 16) = wreg[i];
        }
        __syncthreads();
        // ---- stage A: first conv on the halo tile (22 M-tiles of 16 halo pixels), k = tap*CK0 + c
        for (int mt = wave; mt < NMT; mt += 4) {
            const int hp = mt * 16 + frow;                     // halo pixel of this lane's B column
            const int hpc = hp < HALO_PX ? hp : HALO_PX - 1;
            const int hy = hpc / (HALO_TW + 2), hx = hpc - hy * (HALO_TW + 2);
            f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < NKS0; ++ks) {
                const int kvec = ks * 4 + fch;
                int tap = kvec / cv0;
                const int c8 = kvec - tap * cv0;
                tap = tap > 8 ? 8 : tap;                       // k beyond 9*CK0: zero weights
                const int tr = tap / 3, tq = tap - tr * 3;
                const f16x8 bf = *reinterpret_cast<const f16x8*>(s_in + halo_off((hy + tr) * PATCH_W + hx + tq, c8, CK0));
                const f16x8 a0 = *reinterpret_cast<const f16x8*>(s_w0 + ((frow) * pre.wrow + ks * 32 + fch * 8) * 2);
                const f16x8 a1 = *reinterpret_cast<const f16x8*>(s_w0 + ((16 + frow) * pre.wrow + ks * 32 + fch * 8) * 2);
                c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, bf, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, bf, c1, 0, 0, 0);
            }
            // lane holds channels fch*4..+3 (c0) and 16+fch*4..+3 (c1) of halo pixel hp
            if (hp < HALO_PX) {
                const int gy = ty0 + hy - 1, gx = tx0 + hx - 1;
                const bool inside = (unsigned)gy < (unsigned)h.H && (unsigned)gx < (unsigned)h.W;
                const float4 b0 = *reinterpret_cast<const float4*>(pre.bias + g * BN + fch * 4);
                const float4 b1 = *reinterpret_cast<const float4*>(pre.bias + g * BN + 16 + fch * 4);
                f16x4 o0, o1;
                o0[0] = (_Float16)(inside ? fmaxf(c0[0] + b0.x, 0.f) : 0.f); o0[1] = (_Float16)(inside ? fmaxf(c0[1] + b0.y, 0.f) : 0.f);
                o0[2] = (_Float16)(inside ? fmaxf(c0[2] + b0.z, 0.f) : 0.f); o0[3] = (_Float16)(inside ? fmaxf(c0[3] + b0.w, 0.f) : 0.f);
                o1[0] = (_Float16)(inside ? fmaxf(c1[0] + b1.x, 0.f) : 0.f); o1[1] = (_Float16)(inside ? fmaxf(c1[1] + b1.y, 0.f) : 0.f);
                o1[2] = (_Float16)(inside ? fmaxf(c1[2] + b1.z, 0.f) : 0.f); o1[3] = (_Float16)(inside ? fmaxf(c1[3] + b1.w, 0.f) : 0.f);
                // channel ch lives in 16-B piece ch/8, byte (ch%8)*2 of the halo pixel (same swizzle as the reads)
                *reinterpret_cast<f16x4*>(s_halo + halo_off(hp, fch >> 1, 32) + (fch & 1) * 8) = o0;
                *reinterpret_cast<f16x4*>(s_halo + halo_off(hp, 2 + (fch >> 1), 32) + (fch & 1) * 8) = o1;
            }
        }
        __syncthreads();
        // ---- stage B: second conv, chunk g (9 barrier-free k-steps)
        for (int ks = 0; ks < 9; ++ks) {
            const int tr = ks / 3, tq = ks - tr * 3;
            const int poff = tr * (HALO_TW + 2) + tq;
            f16x8 af[TN2], bf[TM];
#pragma unroll
            for (int i = 0; i < TN2; ++i)
                af[i] = *reinterpret_cast<const f16x8*>(s_w + ((i * 16 + frow) * h.wrow + ks * 32 + fch * 8) * 2);
#pragma unroll
            for (int j = 0; j < TM; ++j)
                bf[j] = *reinterpret_cast<const f16x8*>(s_halo + halo_off(pbase[j] + poff, fch, 32));
#pragma unroll
            for (int i = 0; i < TN2; ++i)
#pragma unroll
                for (int j = 0; j < TM; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
    }
    __syncthreads();
    // ---- epilogue through the LDS tile
    auto row_to_m = [&](int prow) {
        const int oy = ty0 + (prow >> 5), ox = tx0 + (prow & 31);
        return (oy < h.H && ox < h.W) ? (int)(((long)n * h.H + oy) * h.W + ox) : -1;
    };
    constexpr int TS = BN2 + 8;
    _Float16* tile = reinterpret_cast<_Float16*>(hsm);
#pragma unroll
    for (int j = 0; j < TM; ++j) {
        const int prow = (2 * wave + (j >> 1)) * 32 + (j & 1) * 16 + frow;
        const int m = row_to_m(prow);
        EpiCtx c;
        if (m >= 0) epi_locate(a, m, c);
#pragma unroll
        for (int i = 0; i < TN2; ++i) {
            const int col = i * 16 + fch * 4;
            float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
            if (m >= 0 && col < a.cout) epi_compute(a, c, col, v);
            const f16x4 o4 = {(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
            *reinterpret_cast<f16x4*>(tile + prow * TS + col) = o4;
        }
    }
    __syncthreads();
    staged_store<BN2, 256>(a, tile, 0, 0, row_to_m);
}

// ------------------------------------------------------------------------------------------------
// convpair2_kernel<CK0, BN2>: the fused conv pair re-cut for the vector/matrix issue port. Same math and the same
// packed weights as convpair_kernel, but
//   * every global read is an LDS DMA through a buffer resource (flat copies of the weight blocks, a gathered
//     copy of the input patch with out-of-image pixels read as zero); group g+1's patch and first-conv weights
//     stream in while group g's second conv runs, the second conv's weights while the first conv runs;
//   * all LDS fragment addresses are computed once per workgroup (they do not depend on the group) and the taps
//     are immediates, so the two MFMA phases carry almost no address arithmetic;
//   * patch and halo use the column-keyed swizzle of the tap kernel (row shifts are immediates).
// LDS (bytes): patch[2] | w0[2] | halo | w1     InputCvBlock 2x7168 + 2x7168 + 22528 + 19456 = 70656
//                                               OutputCvBlock 27648 + 19456 + 22528 + 10240 = 79872   (2 per CU)
// ------------------------------------------------------------------------------------------------
// Diagnostic build only (tools/cp2_probe.py compiles this file with -DLLV_CP2_STAMPS into its own library): s_memtime stamps of the
// phases of the first 8 tiles of the first 64 workgroups. In the product build the macro is empty and no stamp executes.
#ifdef LLV_CP2_STAMPS
__device__ unsigned long long g_cp2_stamps[64 * 8 * 16];
#define CP2_STAMP(K)                                                                                                   \
    do {                                                                                                               \
        if (tid == 0 && blockIdx.x < 64 && stamp_tile < 8)                                                             \
            g_cp2_stamps[(blockIdx.x * 8 + stamp_tile) * 16 + (K)] = __builtin_amdgcn_s_memtime();                      \
    } while (0)
#else
#define CP2_STAMP(K) do {} while (0)
#endif

static __device__ unsigned int g_conv_guard_trips = 0u;   // clamps of the sliding-window offset word (llv_guard_trips sums it in)

struct Pair2Args {
    unsigned x_bytes, w0_bytes, w1_bytes;   // buffer-resource bounds: one source tensor, ALL first-conv weights, ALL second-conv weights
};

__device__ __forceinline__ int patch_off32(int py, int px, int c8) {   // 64-B pixels, column-keyed swizzle
    return (py * PATCH_W + px) * 64 + ((c8 ^ (((px >> 2) & 1) << 1)) << 4);
}

// LDS layouts of convpair2_kernel<.., PL = true> (round 4; PL = false is the round-3 layout, kept for the same-box A/B LLV_CP2_PLANES=0):
//  * the intermediate (the first conv's output on the halo tile, the second conv's pixel operand) is CHANNEL-CHUNK-MAJOR: four planes of
//    340 x 16 B, plane c8 = channels 8 c8 .. 8 c8 + 7. The first conv's lanes hold 4 channels of 16 CONSECUTIVE halo pixels per fragment
//    row: in the pixel-major image (64-B pixels) those 16 ds_write_b64 land on 4 bank positions (4-way conflict, 16 LDS-array cycles per
//    wave-instruction instead of 4); in a plane they are 16 B apart (2-way). The second conv's ds_read_b128 stay conflict-free: the lane
//    group {fch 0: pixels 0-3, 12-15; fch 1: pixels 4-11} reads two planes a multiple of 256 B apart at complementary pixel sets.
//  * the InputCvBlock's 16-B-pixel patch has a row pitch of 50 pixels instead of 36: with a pitch = 2 (mod 16) the patch index of halo
//    pixel hp under tap (tr, tq) is hp + 2 tr + tq (mod 16) -- consecutive across the wrap from one halo row to the next -- and the two
//    taps a ds_read_b128 lane group mixes (t, t + 1) never put different pixels on one bank (the pair (0,2) | (1,0) did: 2-way).
// PMC round 3: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 29.5 % (OutputCvBlock) and 33.6 % (InputCvBlock) in the round-3 layout.
constexpr int CP2_PLANE = 5632;                              // 340 pixels x 16 B rounded up to a multiple of 256 B; 4 planes = TAP_HBYTES
static_assert(4 * CP2_PLANE <= TAP_HBYTES && CP2_PLANE >= HALO_PX * 16 && CP2_PLANE % 256 == 0, "halo planes");
template <bool PL>
__device__ __forceinline__ int cp2_halo_off(int hy, int hx, int c8) {
    return PL ? c8 * CP2_PLANE + (hy * (HALO_TW + 2) + hx) * 16 : tap_halo_off(hy, hx, c8);
}

template <int CK0, int BN2, bool PL>
__global__ __launch_bounds__(256, 2) void convpair2_kernel(const ConvKArgs a_in, const HaloArgs h, const PreArgs pre, const Pair2Args q) {
    // sliding window over a resident clip (llv_conv_desc.offset_dev): one scalar load, added to the source / residual bases
    ConvKArgs a = a_in;
    if (a_in.off_dev) {
        long xo = *a_in.off_dev;
        // the extents replay (conv_extents.inc) vouches for offsets in [0, offset_max] only: a stale or foreign word is clamped
        // and counted (llv_guard_trips), never followed
        if (xo < 0 || xo > a_in.off_max) {
            if (threadIdx.x == 0 && blockIdx.x == 0) atomicAdd(&g_conv_guard_trips, 1u);
            xo = xo < 0 ? 0 : a_in.off_max;
        }
        if (a_in.off_x) { a.x0 = a_in.x0 + xo; a.x1 = a_in.x1 + xo; a.x2 = a_in.x2 + xo; }
        if (a_in.off_res) a.res = a_in.res + xo;
    }
    constexpr int G = CK0 == 8 ? 3 : 1;                      // groups of the first conv = 32-channel chunks of the second
    constexpr int TN2 = BN2 / 16, TM = 4;
    constexpr int NMT = (HALO_PX + 15) / 16;                 // 22 M-tiles of 16 halo pixels
    constexpr int MTW = (NMT + 3) / 4;                       // per wave (<= 6)
    constexpr int NKS0 = (9 * CK0 + 31) / 32;                // k-steps of the first conv (3 or 9)
    constexpr int WROW0 = CK0 == 8 ? 104 : 296, WROW1 = 296;
    constexpr int PW = (PL && CK0 == 8) ? 50 : PATCH_W;      // patch row pitch in pixels (see above)
    constexpr int PPX = PW * PATCH_H;                        // patch pixels incl. the pitch padding
    constexpr int PATCH_B = PPX * CK0 * 2;                   // 6912 (9600 with the 50-pixel pitch) / 27648
    constexpr int PATCH_PIECES = (PATCH_B + 1023) / 1024;    // 7 (10) / 27
    constexpr int W0_B = 32 * WROW0 * 2, W1_B = BN2 * WROW1 * 2;               // bytes per group in global memory
    // LDS images of the weights: [k-step][row][32 halfs] with lds_off's swizzle (the padded global rows are gathered
    // into it by the DMA), so fragment reads are conflict-free under ds_read_b128's real lane grouping
    constexpr int W0_PIECES = NKS0 * 2, W1_PIECES = 9 * BN2 / 16;              // 6 / 18 and 18 / 9 KiB
    constexpr int NPB = G > 1 ? 2 : 1;                       // patch / w0 buffers
    constexpr int O_PATCH = 0, O_W0 = O_PATCH + NPB * PATCH_PIECES * 1024, O_HALO = O_W0 + NPB * W0_PIECES * 1024;
    constexpr int O_W1 = O_HALO + TAP_HBYTES, SMEM = O_W1 + W1_PIECES * 1024;
    constexpr int ROWB = (HALO_TW + 2) * (PL ? 16 : 64), PROWB = PW * (CK0 * 2);
    constexpr int PPW = (PATCH_PIECES + 3) / 4, W0PW = (W0_PIECES + 3) / 4, W1PW = (W1_PIECES + 3) / 4;
    static_assert(2 * SMEM <= 160 * 1024, "two workgroups per CU");
    static_assert(256 * (BN2 + 8) * 2 <= TAP_HBYTES, "the epilogue tile reuses the halo buffer");
    __shared__ __attribute__((aligned(16))) unsigned char smem[SMEM];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int H = h.H, W = h.W;
    const int frow = lane & 15, fch = lane >> 4;
    // persistent workgroup: tiles v = blockIdx.x, blockIdx.x + gridDim.x, ... (XCD-aware order); everything that does not
    // depend on the tile position (weight gathers, fragment addresses) is computed once
    struct Tile { int tx0, ty0, n, img_base; };
    auto tile_of = [&](int v) {
        const int nblk = h.nblk;
        const int qq = nblk >> 3, r = nblk & 7, xcd = v & 7, idx = v >> 3;
        const int bid = (xcd < r ? xcd * (qq + 1) : r * (qq + 1) + (xcd - r) * qq) + idx;
        Tile t;
        t.tx0 = (bid % h.tiles_x) * HALO_TW;
        const int t1 = bid / h.tiles_x;
        t.ty0 = (t1 % h.tiles_y) * HALO_TH;
        t.n = t1 / h.tiles_y;
        t.img_base = a.seg_xoff[0] + t.n * H * W;
        return t;
    };

    const __amdgpu_buffer_rsrc_t rX0 = make_rsrc(a.x0, q.x_bytes);
    const __amdgpu_buffer_rsrc_t rX1 = make_rsrc(a.x1, q.x_bytes);
    const __amdgpu_buffer_rsrc_t rX2 = make_rsrc(a.x2, q.x_bytes);
    const __amdgpu_buffer_rsrc_t rW0 = make_rsrc(pre.w, q.w0_bytes);
    const __amdgpu_buffer_rsrc_t rW1 = make_rsrc(h.w, q.w1_bytes);

    // ---- patch DMA offsets of a tile (constant over its groups)
    int pvo[PPW];                                            // patch pieces wave, wave+4, ...
    int ppy[PPW], ppx[PPW], pc8[PPW];                        // this lane's patch pixel / channel chunk per piece (tile independent)
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        const int pc = wave + 4 * i;
        int pp, c8;
        if (CK0 == 8) { pp = pc * 64 + lane; c8 = 0; }
        else { pp = pc * 16 + (lane >> 2); c8 = lane & 3; }
        ppy[i] = -1000000; ppx[i] = 0; pc8[i] = 0;
        if (pp < PPX) {
            const int py = pp / PW, px = pp - py * PW;
            if (CK0 == 32) c8 ^= ((px >> 2) & 1) << 1;
            if (px < PATCH_W) { ppy[i] = py - 2; ppx[i] = px - 2; pc8[i] = c8 * 16; }      // pitch padding: reads zero, like outside the image
        }
    }
    auto set_pvo = [&](const Tile& t) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            const int gy = t.ty0 + ppy[i], gx = t.tx0 + ppx[i];
            pvo[i] = ((unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W) ? (t.img_base + gy * W + gx) * a.x_ld * 2 + pc8[i]
                                                                                : (int)q.x_bytes;
        }
    };
    auto issue_patch = [&](int g, int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            if (wave + 4 * i < PATCH_PIECES) {
                unsigned char* dst = smem + O_PATCH + (buf * PATCH_PIECES + wave + 4 * i) * 1024;
                if (g == 0) dma16(rX0, dst, pvo[i], 0);
                else if (g == 1) dma16(rX1, dst, pvo[i], 0);
                else dma16(rX2, dst, pvo[i], 0);
            }
        }
    };
    // weight gathers: image offset o = 1024*piece + 16*lane -> (k-step, row, physical slot) -> padded global row
    int w0v[W0PW], w1v[W1PW];
#pragma unroll
    for (int i = 0; i < W0PW; ++i) {
        const int o = (wave + 4 * i) * 1024 + lane * 16;
        const int ks = o / 2048, row = (o % 2048) / 64, slot = (o % 64) / 16;
        w0v[i] = (row * WROW0 + ks * 32 + ((slot ^ ((0 - (row >> 2)) & 3)) * 8)) * 2;
    }
#pragma unroll
    for (int i = 0; i < W1PW; ++i) {
        const int o = (wave + 4 * i) * 1024 + lane * 16;
        const int t = o / (BN2 * 64), row = (o % (BN2 * 64)) / 64, slot = (o % 64) / 16;
        w1v[i] = (row * WROW1 + t * 32 + ((slot ^ ((0 - (row >> 2)) & 3)) * 8)) * 2;
    }
    auto issue_w0 = [&](int g, int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < W0PW; ++i)
            if (wave + 4 * i < W0_PIECES) dma16(rW0, smem + O_W0 + (buf * W0_PIECES + wave + 4 * i) * 1024, w0v[i], g * W0_B);
    };
    auto issue_w1 = [&](int g) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < W1PW; ++i)
            if (wave + 4 * i < W1_PIECES) dma16(rW1, smem + O_W1 + (wave + 4 * i) * 1024, w1v[i], g * W1_B);
    };

    // ---- fragment addresses (constant over groups)
    // stage A: M-tile it of this wave = halo pixels 16*(wave + 4*it) + frow
    int pa[MTW][CK0 == 8 ? NKS0 : 3];                        // patch read offsets: per k-step (CK0 = 8) / per tap column (CK0 = 32)
    int hw[MTW];                                             // halo write offset of this lane's first 4 channels; -1: no pixel
    int hyx[MTW];                                            // (hy << 8 | hx) of this lane's halo pixel per M-tile, -1: none
#pragma unroll
    for (int it = 0; it < MTW; ++it) {
        const int hp = (wave + 4 * it) * 16 + frow;
        const int hpc = hp < HALO_PX ? hp : HALO_PX - 1;
        const int hy = hpc / (HALO_TW + 2), hx = hpc - hy * (HALO_TW + 2);
        if (CK0 == 8) {
#pragma unroll
            for (int ks = 0; ks < NKS0; ++ks) {
                int tap = ks * 4 + fch;                        // one 8-channel pixel per k chunk
                tap = tap > 8 ? 8 : tap;                       // k >= 72: zero weights, any valid address
                const int tr = tap / 3, tq = tap - 3 * tr;
                pa[it][ks] = ((hy + tr) * PW + hx + tq) * 16;
            }
        } else {
#pragma unroll
            for (int tq = 0; tq < 3; ++tq) pa[it][tq] = patch_off32(hy, hx + tq, fch);
        }
        hw[it] = hp < HALO_PX ? cp2_halo_off<PL>(hy, hx, fch >> 1) + (fch & 1) * 8 : -1;
        hyx[it] = hp < HALO_PX ? (hy << 8) | hx : -1;
    }
    auto inside_of = [&](const Tile& t) __attribute__((always_inline)) {
        unsigned m = 0;
#pragma unroll
        for (int it = 0; it < MTW; ++it) {
            const int gy = t.ty0 + (hyx[it] >> 8) - 1, gx = t.tx0 + (hyx[it] & 255) - 1;
            if (hyx[it] >= 0 && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W) m |= 1u << it;
        }
        return m;
    };
    const int wa = lds_off(frow, fch);                       // weight fragment: row frow (+1024: next n-tile), k chunk fch
    int hb[TM][3];
#pragma unroll
    for (int j = 0; j < TM; ++j)
#pragma unroll
        for (int tq = 0; tq < 3; ++tq) hb[j][tq] = cp2_halo_off<PL>(2 * wave + (j >> 1), (j & 1) * 16 + frow + tq, fch);

    float4 bias4[TN2];
#pragma unroll
    for (int i = 0; i < TN2; ++i) {
        const int co = i * 16 + fch * 4;
        bias4[i] = co < a.cout ? *reinterpret_cast<const float4*>(a.bias + co) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    // bias (+ ReLU) only, as the InputCvBlock's second conv is: the bias rides in the accumulator, ReLU is applied to the rounded
    // halves, and no pixel is located before the staging (the tap kernel's lean epilogue)
    const bool lean = CK0 == 8 && !(a.flags & ~LLV_CONV_RELU) && a.y_gstride == 0;   // compiled out of the OutputCvBlock instantiation (residual subtract)
    // OutputCvBlock: the residual (in1 of `in1 - x`) of the tile's pixels is requested together with the tile's patch and kept in
    // registers until the epilogue, instead of one exposed global-load latency per accumulator row there
    const bool pre_res = CK0 == 32 && (a.flags & (LLV_CONV_RES_ADD | LLV_CONV_RES_RSUB)) != 0 &&
                         !(a.flags & (LLV_CONV_PIXSHUF | LLV_CONV_ACT_FIRST | LLV_CONV_RES_UPS)) && fch * 4 < a.cout;
    f16x4 rres[TM];
    auto load_res = [&](const Tile& t) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < TM; ++j) {
            const int prow = (2 * wave + (j >> 1)) * 32 + (j & 1) * 16 + frow;
            const int oy = t.ty0 + (prow >> 5), ox = t.tx0 + (prow & 31);
            const f16x4 z4 = {(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
            rres[j] = z4;
            if (pre_res && oy < H && ox < W)
                rres[j] = *reinterpret_cast<const f16x4*>(a.res + (long)(a.seg_yoff[0] + t.n * H * W + oy * W + ox) * a.res_ld + fch * 4);
        }
    };
    // first-conv bias of the group that runs next (constant for one group: loaded once)
    float4 nb0 = *reinterpret_cast<const float4*>(pre.bias + fch * 4);
    float4 nb1 = *reinterpret_cast<const float4*>(pre.bias + 16 + fch * 4);
    int v = blockIdx.x;
    Tile cur = tile_of(v);
    set_pvo(cur);
    issue_w1(0);
    issue_patch(0, 0);
    issue_w0(0, 0);
    load_res(cur);
    int stamp_tile = 0;
    (void)stamp_tile;
#pragma unroll 1
    for (;;) {
    CP2_STAMP(0);
    const int tx0 = cur.tx0, ty0 = cur.ty0, n = cur.n;
    const int vn = v + gridDim.x;
    const bool has_next = vn < h.nblk;
    const Tile nxt = has_next ? tile_of(vn) : cur;
    const unsigned inside = inside_of(cur);
    const bool border = ty0 == 0 || tx0 == 0 || ty0 + HALO_TH >= H || tx0 + HALO_TW >= W;   // halo leaves the image
    f32x4 acc[TN2][TM];
#pragma unroll
    for (int i = 0; i < TN2; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j)
            acc[i][j] = lean ? (f32x4){bias4[i].x, bias4[i].y, bias4[i].z, bias4[i].w} : (f32x4){0.f, 0.f, 0.f, 0.f};
    f16x4 rcur[TM];
#pragma unroll 1
    for (int g = 0; g < G; ++g) {
        const int pb = g & (NPB - 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                     // patch/w0 of group g landed; halo and w1 are free (B of g-1 done)
        CP2_STAMP(1 + 4 * g);
        if (g > 0) issue_w1(g);
        const float4 b0 = nb0, b1 = nb1;                     // landed with the group's operands (requested a group ago)
        if (g == G - 1) {
#pragma unroll
            for (int j = 0; j < TM; ++j) rcur[j] = rres[j];  // this tile's residual; rres is re-requested for the next tile below
        }
        // ---- stage A: first conv on the halo tile. The weight fragments are the same for every M-tile: registers.
        const unsigned char* sP = smem + O_PATCH + pb * PATCH_PIECES * 1024;
        const unsigned char* sW0 = smem + O_W0 + pb * W0_PIECES * 1024;
        f16x8 a0[NKS0], a1[NKS0];
#pragma unroll
        for (int ks = 0; ks < NKS0; ++ks) {
            a0[ks] = *reinterpret_cast<const f16x8*>(sW0 + ks * 2048 + wa);
            a1[ks] = *reinterpret_cast<const f16x8*>(sW0 + ks * 2048 + 1024 + wa);
        }
        // One flat stream of (M-tile, k-step) steps, two MFMAs each, with the pixel fragment of step s + PD requested before the
        // MFMAs of step s (a ring of PD + 1 registers): the compiler's own order was read -> lgkmcnt(0) -> 2 MFMAs, i.e. one exposed
        // LDS latency per 32 MFMA cycles (the first conv ran at a quarter of the MFMA rate). Every wave walks MTW M-tiles; where
        // 4 * it + wave runs past the last one the addresses are clamped and nothing is written (the barrier behind this stage
        // waits for the slowest wave anyway).
        {
            constexpr int SA = MTW * NKS0, PD = 3;
            auto lda = [&](int st) __attribute__((always_inline)) {
                const int it = st / NKS0, ks = st % NKS0;
                if (CK0 == 8) return *reinterpret_cast<const f16x8*>(sP + pa[it][ks]);
                return *reinterpret_cast<const f16x8*>(sP + pa[it][ks % 3] + (ks / 3) * PROWB);
            };
            f16x8 ring[PD + 1];
#pragma unroll
            for (int st = 0; st < PD; ++st) ring[st] = lda(st);
            f32x4 c0 = {b0.x, b0.y, b0.z, b0.w}, c1 = {b1.x, b1.y, b1.z, b1.w};      // bias rides in the accumulator
#pragma unroll
            for (int st = 0; st < SA; ++st) {
                const int it = st / NKS0, ks = st % NKS0;
                if (st + PD < SA) ring[(st + PD) % (PD + 1)] = lda(st + PD);
                c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0[ks], ring[st % (PD + 1)], c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1[ks], ring[st % (PD + 1)], c1, 0, 0, 0);
                if (st + PD < SA) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                if (ks == NKS0 - 1) {
                    // ReLU on the rounded halves (max commutes with the monotone rounding); zero outside the image
                    // (= the second conv's zero padding), which only workgroups on the image border can see
                    f16x4 o0 = {(_Float16)c0[0], (_Float16)c0[1], (_Float16)c0[2], (_Float16)c0[3]};
                    f16x4 o1 = {(_Float16)c1[0], (_Float16)c1[1], (_Float16)c1[2], (_Float16)c1[3]};
                    const f16x4 z4 = {(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
                    o0 = __builtin_elementwise_max(o0, z4);
                    o1 = __builtin_elementwise_max(o1, z4);
                    if (border && !((inside >> it) & 1u)) { o0 = z4; o1 = z4; }
                    if (hw[it] >= 0) {
                        *reinterpret_cast<f16x4*>(smem + O_HALO + hw[it]) = o0;          // channels 4*fch .. (16-B slot fch>>1)
                        *reinterpret_cast<f16x4*>(smem + O_HALO + (PL ? hw[it] + 2 * CP2_PLANE : (hw[it] ^ 32))) = o1;   // channels 16 + 4*fch .. (chunk 2 + (fch>>1))
                    }
                    c0 = (f32x4){b0.x, b0.y, b0.z, b0.w};
                    c1 = (f32x4){b1.x, b1.y, b1.z, b1.w};
                }
            }
        }
        CP2_STAMP(2 + 4 * g);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                     // halo complete, w1 of group g landed, patch/w0 buffers of g-1 free
        if (G > 1) {
            const int gn = g + 1 < G ? g + 1 : 0;
            nb0 = *reinterpret_cast<const float4*>(pre.bias + gn * 32 + fch * 4);
            nb1 = *reinterpret_cast<const float4*>(pre.bias + gn * 32 + 16 + fch * 4);
        }
        if (g + 1 < G) {
            issue_patch(g + 1, (g + 1) & (NPB - 1));
            issue_w0(g + 1, (g + 1) & (NPB - 1));
        } else if (has_next) {
            // last group: patch / first-conv buffer 0 is free again, so the NEXT tile's first patch (and first-conv weights)
            // stream in under this group's second conv and the epilogue
            set_pvo(nxt);
            issue_patch(0, 0);
            if (G > 1) issue_w0(0, 0);
            load_res(nxt);
        }
        CP2_STAMP(3 + 4 * g);
        // ---- stage B: second conv, chunk g
        const unsigned char* sH = smem + O_HALO;
        const unsigned char* sW1 = smem + O_W1;
        {
            // tap t + 1's fragments are requested before tap t's MFMAs (two register sets), one read per MFMA
            f16x8 af[2][TN2], bf[2][TM];
#define CP2_LOADB(SET, TP)                                                                                            \
    _Pragma("unroll") for (int j = 0; j < TM; ++j)                                                                    \
        bf[SET][j] = *reinterpret_cast<const f16x8*>(sH + hb[j][(TP) % 3] + ((TP) / 3) * ROWB);                       \
    _Pragma("unroll") for (int i = 0; i < TN2; ++i)                                                                   \
        af[SET][i] = *reinterpret_cast<const f16x8*>(sW1 + (TP) * (BN2 * 64) + i * 1024 + wa);
            CP2_LOADB(0, 0)
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                if (t < 8) { CP2_LOADB((t + 1) & 1, t + 1) }
#pragma unroll
                for (int i = 0; i < TN2; ++i)
#pragma unroll
                    for (int j = 0; j < TM; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[t & 1][i], bf[t & 1][j], acc[i][j], 0, 0, 0);
                if (t < 8) {
                    constexpr int NR = TN2 + TM, NM = TN2 * TM, NI = NR < NM ? NR : NM;
#pragma unroll
                    for (int k = 0; k < NI; ++k) {
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    }
                    if (NR > NM) __builtin_amdgcn_sched_group_barrier(0x100, NR - NM, 0);
                    if (NM > NR) __builtin_amdgcn_sched_group_barrier(0x008, NM - NR, 0);
                }
            }
#undef CP2_LOADB
        }
        CP2_STAMP(4 + 4 * g);
    }
    // Barriers from here to the end of the tile order LDS accesses only: s_waitcnt lgkmcnt(0) + s_barrier. __syncthreads() also emits
    // vmcnt(0), which here would wait for the NEXT tile's patch (requested before the last second conv so that it lands during this
    // epilogue) and for the output stores.
#define CP2_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
    CP2_LDS_BARRIER();       // every wave is done with the patch / weight buffers and the halo
    CP2_STAMP(13);

    // the next tile's second-conv weights can only follow this tile's last second conv; with one group (OutputCvBlock) both
    // weight images simply stay resident in LDS across tiles
    if (has_next && G > 1) issue_w1(0);

    // ---- epilogue through an LDS tile in the halo buffer
    const int ybase = a.seg_yoff[0] + n * H * W;
    auto loc = [&](int prow, EpiCtx& c) {
        const int oy = ty0 + (prow >> 5), ox = tx0 + (prow & 31);
        if (oy >= H || ox >= W) return false;
        c.opix = ybase + oy * W + ox;
        c.rpix = c.opix;
        c.ox = ox; c.oy = oy; c.n = n; c.Ho = H; c.Wo = W; c.yo = a.seg_yoff[0];
        return true;
    };
    constexpr int TS = BN2 + 8;
    _Float16* tile = reinterpret_cast<_Float16*>(smem + O_HALO);
    if (lean) {
        const bool relu = (a.flags & LLV_CONV_RELU) != 0;
        const f16x4 z4 = {(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
#pragma unroll
        for (int j = 0; j < TM; ++j) {
            const int prow = (2 * wave + (j >> 1)) * 32 + (j & 1) * 16 + frow;
#pragma unroll
            for (int i = 0; i < TN2; ++i) {
                f16x4 o4 = {(_Float16)acc[i][j][0], (_Float16)acc[i][j][1], (_Float16)acc[i][j][2], (_Float16)acc[i][j][3]};
                if (relu) o4 = __builtin_elementwise_max(o4, z4);
                *reinterpret_cast<f16x4*>(tile + prow * TS + i * 16 + fch * 4) = o4;
            }
        }
        CP2_LDS_BARRIER();
        CP2_STAMP(14);
        constexpr int CPR = BN2 / 8;
        const int cc = tid % CPR, co = cc * 8;
        if (co < a.cout) {
            const bool full = co + 8 <= a.cout;
            _Float16* yb = reinterpret_cast<_Float16*>(a.y) + co;
#pragma unroll
            for (int prow = tid / CPR; prow < 256; prow += 256 / CPR) {
                const int oy = ty0 + (prow >> 5), ox = tx0 + (prow & 31);
                if (oy >= H || ox >= W) continue;
                const uint4 val = *reinterpret_cast<const uint4*>(tile + prow * TS + cc * 8);
                _Float16* dst = yb + (long)(ybase + oy * W + ox) * a.y_ld;
                if (full) *reinterpret_cast<uint4*>(dst) = val;
                else *reinterpret_cast<uint2*>(dst) = make_uint2(val.x, val.y);
            }
        }
        CP2_STAMP(15);
        ++stamp_tile;
        if (!has_next) break;
        v = vn;
        cur = nxt;
        continue;
    }
#pragma unroll
    for (int j = 0; j < TM; ++j) {
        const int prow = (2 * wave + (j >> 1)) * 32 + (j & 1) * 16 + frow;
        EpiCtx ec;
        const bool ok = loc(prow, ec);
#pragma unroll
        for (int i = 0; i < TN2; ++i) {
            const int col = i * 16 + fch * 4;
            float vv[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
            if (pre_res && i == 0) {                         // the same arithmetic as epi_compute_b, residual from the registers
                if (ok) {
                    vv[0] += bias4[0].x; vv[1] += bias4[0].y; vv[2] += bias4[0].z; vv[3] += bias4[0].w;
#pragma unroll
                    for (int e = 0; e < 4; ++e) vv[e] = (a.flags & LLV_CONV_RES_RSUB) ? (float)rcur[j][e] - vv[e] : vv[e] + (float)rcur[j][e];
                    if (a.flags & (LLV_CONV_RELU | LLV_CONV_LEAKY)) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) vv[e] = act_fn(a, vv[e]);
                    }
                }
            } else if (ok && col < a.cout) epi_compute_b(a, ec, col, vv, bias4[i]);
            const f16x4 o4 = {(_Float16)vv[0], (_Float16)vv[1], (_Float16)vv[2], (_Float16)vv[3]};
            *reinterpret_cast<f16x4*>(tile + prow * TS + col) = o4;
        }
    }
    CP2_LDS_BARRIER();
    CP2_STAMP(14);
    staged_store_loc<BN2, 256>(a, tile, 0, 0, loc);
    CP2_STAMP(15);
    ++stamp_tile;
    if (!has_next) break;
    v = vn;
    cur = nxt;
    }   // tile loop (the next iteration's first barrier orders this tile's staged reads before the next halo writes)
#undef CP2_LDS_BARRIER
}

template <int BN>
int launch_halo(ConvKArgs& k, HaloArgs& h, int groups, int batch, hipStream_t st) {
    h.ntiles_n = (k.cout + BN - 1) / BN;
    h.nblk = h.tiles_x * h.tiles_y * batch;
    const size_t lds_main = (size_t)HALO_PX * 64 + (size_t)BN * h.wrow * 2;
    const size_t lds_epi = (size_t)256 * (BN + 8) * 2;
    const size_t lds = lds_main > lds_epi ? lds_main : lds_epi;
    dim3 grid(h.nblk, h.ntiles_n * groups, 1);
    hipLaunchKernelGGL((conv3x3_halo_kernel<BN>), grid, dim3(256), lds, st, k, h);
    return hipGetLastError() == hipSuccess ? LLV_OK : LLV_E_LAUNCH;
}

// bytes spanned by one source tensor (all sources of a launch share the geometry): the buffer-resource bound
static size_t tap_x_bytes(const llv_conv_desc* d) {
    size_t px = 0;
    for (int s = 0; s < d->n_seg; ++s) {
        const size_t e = (size_t)d->seg_xoff[s] + (size_t)d->batch * d->seg_h[s] * d->seg_w[s];
        px = e > px ? e : px;
    }
    return px * (size_t)d->x_ld * 2;
}

static void fill_tap(const llv_conv_desc* d, TapArgs& h) {
    int blk = 0;
    for (int s = 0; s < LLV_MAX_SEG; ++s) {
        h.tiles_x[s] = h.tiles_y[s] = 0;
        h.blk0[s] = blk;
        if (s < d->n_seg) {
            h.tiles_x[s] = (d->seg_w[s] + HALO_TW - 1) / HALO_TW;
            h.tiles_y[s] = (d->seg_h[s] + HALO_TH - 1) / HALO_TH;
            blk += h.tiles_x[s] * h.tiles_y[s] * d->batch;
        }
    }
    h.blk0[LLV_MAX_SEG] = blk;
    h.nblk = blk;
    h.ntn = 1;
    h.n_chunks = d->cin / 32;
    h.x_bytes = (unsigned)tap_x_bytes(d);
    h.w_bytes = (unsigned)((size_t)d->w_rows * d->k_pad * 2);
}

template <int BN, int R = 2>
int launch_tap(ConvKArgs& k, const llv_conv_desc* d, hipStream_t st) {
    TapArgs h;
    fill_tap(d, h);
    k.k_splits = 1;
    h.ntn = (d->cout + BN - 1) / BN;
    dim3 grid(tap_grid_x(h.nblk, h.ntn), 1, d->groups);
    if (d->n_src > 1) hipLaunchKernelGGL((conv3x3_tap_kernel<BN, R, true>), grid, dim3(256), 0, st, k, h);
    else hipLaunchKernelGGL((conv3x3_tap_kernel<BN, R, false>), grid, dim3(256), 0, st, k, h);
    return hipGetLastError() == hipSuccess ? LLV_OK : LLV_E_LAUNCH;
}

static bool tap_eligible(const llv_conv_desc* d) {
    if (d->kh != 3 || d->kw != 3 || d->stride != 1 || d->pad != 1 || d->dilation > 1 || (d->cin & 31)) return false;
    if (d->n_src > 1 && ((d->src_cvec & 3) || d->groups != 1)) return false;   // a 32-channel chunk must lie in one source
    for (int s = 0; s < d->n_seg; ++s)
        if (d->seg_ho[s] != d->seg_h[s] || d->seg_wo[s] != d->seg_w[s]) return false;
    // 32-bit buffer offsets, with room for the scalar k offset on top of an out-of-range marker
    if (tap_x_bytes(d) >= (1ull << 31) || (size_t)d->w_rows * d->k_pad * 2 >= (1ull << 31)) return false;
    return true;
}

template <int BN, int BM, int WN, int WM>
int launch(ConvKArgs& k, int groups, hipStream_t st) {
    k.tiles_n = (k.cout + BN - 1) / BN;
    const int tiles_m = (k.m_total + BM - 1) / BM;
    k.nblk = k.tiles_n * tiles_m;
    dim3 grid(k.nblk, groups, k.k_splits);
    hipLaunchKernelGGL((conv_igemm_kernel<BN, BM, WN, WM>), grid, dim3(256), 0, st, k);
    if (k.k_splits > 1) {
        const int ctot = groups * k.cout;
        long work = (long)k.m_total * (ctot >> 2);
        long gsz = (work + 255) / 256;
        gsz = gsz > 2048 ? 2048 : (gsz < 1 ? 1 : gsz);
        hipLaunchKernelGGL(splitk_epilogue_kernel, dim3((int)gsz), dim3(256), 0, st, k, ctot);
    }
    return hipGetLastError() == hipSuccess ? LLV_OK : LLV_E_LAUNCH;
}

template <int BN, int BM, int WN, int WM, int BK = 32>
int launch_glds(ConvKArgs& k, int groups, hipStream_t st) {
    const bool ut = k.x_bytes != 0 && (k.cin % BK) == 0;
    k.tiles_n = (k.cout + BN - 1) / BN;
    const int tiles_m = (k.m_total + BM - 1) / BM;
    k.nblk = k.tiles_n * tiles_m;
    dim3 grid(k.nblk, groups, k.k_splits);
    if (ut) hipLaunchKernelGGL((conv_igemm_glds_kernel<BN, BM, WN, WM, BK, true>), grid, dim3(256), 0, st, k);
    else hipLaunchKernelGGL((conv_igemm_glds_kernel<BN, BM, WN, WM, BK, false>), grid, dim3(256), 0, st, k);
    if (k.k_splits > 1) {
        const int ctot = groups * k.cout;
        long work = (long)k.m_total * (ctot >> 2);
        long gsz = (work + 255) / 256;
        gsz = gsz > 2048 ? 2048 : (gsz < 1 ? 1 : gsz);
        hipLaunchKernelGGL(splitk_epilogue_kernel, dim3((int)gsz), dim3(256), 0, st, k, ctot);
    }
    return hipGetLastError() == hipSuccess ? LLV_OK : LLV_E_LAUNCH;
}

void tile_dims(int tile, int& bn, int& bm) {
    switch (tile) {
        case LLV_TILE_128x128: case LLV_TILE_128x128_REGSTAGE: case LLV_TILE_128x128_K64: bn = 128; bm = 128; break;
        case LLV_TILE_TAP128: bn = 128; bm = 256; break;
        case LLV_TILE_TAP64: case LLV_TILE_TAP64_R4: case LLV_TILE_TAP64_G3: bn = 64; bm = 256; break;
        case LLV_TILE_TAP512: bn = 64; bm = 512; break;
        case LLV_TILE_64x256: case LLV_TILE_64x256_K64: bn = 64; bm = 256; break;
        case LLV_TILE_32x256: bn = 32; bm = 256; break;
        case LLV_TILE_16x256: bn = 16; bm = 256; break;
        default: bn = 64; bm = 64; break;
    }
}

// Split K when the launch cannot fill the chip: few tiles and a long K loop (ResNet layer3/4, FPN P5-P7).
int pick_splits(int tile, int cout, long m, int groups, int nk, int flags) {
    if (flags & LLV_CONV_PIXSHUF) return 1;
    int bn, bm;
    tile_dims(tile, bn, bm);
    const long blocks = ((cout + bn - 1) / bn) * ((m + bm - 1) / bm) * groups;
    // at 160-255 workgroups a short reduction (< 48 k-steps) is not worth the extra epilogue launch (single-image layer3 conv1, 232
    // workgroups, K = 1024: 14.7 us unsplit against 24.9 us split in three, in isolation); a long one is (layer3 conv2, K = 2304)
    if (blocks >= 256 || nk < 16 || (blocks >= 160 && nk < 48)) return 1;
    static const long split_target = getenv("LLV_SPLIT_TARGET") && atol(getenv("LLV_SPLIT_TARGET")) > 0 ? atol(getenv("LLV_SPLIT_TARGET")) : 512;
    static const long split_min_nk = getenv("LLV_SPLIT_MIN_NK") && atol(getenv("LLV_SPLIT_MIN_NK")) > 0 ? atol(getenv("LLV_SPLIT_MIN_NK")) : 8;
    long s = (split_target + blocks - 1) / blocks;
    if (s > nk / split_min_nk) s = nk / split_min_nk;
    if (s > 32) s = 32;
    return s < 2 ? 1 : (int)s;
}

int pick_tile(int co, long m, int groups) {
    if (co <= 16) return LLV_TILE_16x256;
    if (co <= 32) return LLV_TILE_32x256;
    if (co <= 64) return m >= 64 * 1024 ? LLV_TILE_64x256 : LLV_TILE_64x64;
    // enough 128x128 tiles to fill 256 CUs a couple of times over? otherwise go small
    const long blocks128 = ((co + 127) / 128) * ((m + 127) / 128) * groups;
    if (blocks128 >= 384) return LLV_TILE_128x128;
    return LLV_TILE_64x64;
}

}  // namespace

#ifdef LLV_CP2_STAMPS
extern "C" int llv_dbg_cp2_stamps(unsigned long long* host, int reset) {
    if (hipDeviceSynchronize() != hipSuccess) return LLV_E_LAUNCH;
    if (host && hipMemcpyFromSymbol(host, HIP_SYMBOL(g_cp2_stamps), sizeof(unsigned long long) * 64 * 8 * 16) != hipSuccess) return LLV_E_LAUNCH;
    if (reset) {
        static unsigned long long zeros[64 * 8 * 16];
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_cp2_stamps), zeros, sizeof(zeros)) != hipSuccess) return LLV_E_LAUNCH;
    }
    return LLV_OK;
}
#endif

// clamps of the sliding-window offset word since the last reset (summed into llv_guard_trips, decode.hip)
unsigned int llv_conv_guard_trips_(int reset) {
    unsigned int v = 0;
    (void)hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_conv_guard_trips), sizeof(v));
    if (reset) {
        const unsigned int z = 0;
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_conv_guard_trips), &z, sizeof(z));
    }
    return v;
}

unsigned int* llv_conv_guard_addr_() {
    void* p = nullptr;
    return hipGetSymbolAddress(&p, HIP_SYMBOL(g_conv_guard_trips)) == hipSuccess ? (unsigned int*)p : nullptr;
}

#include "conv_tap512.inc"

struct LaunchPlan { int tile, splits, nk_per; long m; };

static int tile_bk(int tile) { return (tile == LLV_TILE_128x128_K64 || tile == LLV_TILE_64x256_K64 || tile == LLV_TILE_64x64_K64) ? 64 : 32; }

static int plan_launch(const llv_conv_desc* d, LaunchPlan& lp) {
    long m = 0;
    for (int s = 0; s < d->n_seg; ++s) m += (long)d->batch * d->seg_ho[s] * d->seg_wo[s];
    lp.m = m;
    lp.tile = d->tile ? d->tile : pick_tile(d->cout, m, d->groups);
    if (!d->tile && d->k_splits <= 1 && tap_eligible(d)) {
        // 3x3 / stride 1: the tap-streamed kernel, unless the launch is too small to give every CU a workgroup
        // (then the implicit GEMM with split K fills the chip better)
        long blocks = 0;
        for (int s = 0; s < d->n_seg; ++s)
            blocks += (long)d->batch * ((d->seg_h[s] + HALO_TH - 1) / HALO_TH) * ((d->seg_w[s] + HALO_TW - 1) / HALO_TW);
        blocks *= (long)((d->cout + 63) / 64) * d->groups;
        if (blocks >= 160 || (d->flags & LLV_CONV_PS_EMAJOR)) {      // sub-pixel-major PixelShuffle exists on the tap kernel only
            // PixelShuffle layers take the 128-channel tile: a workgroup then owns 32 output channels of both
            // dx sub-pixels = whole 128-byte lines of the shuffled output (and of the skip tensor it adds)
            lp.tile = ((d->flags & LLV_CONV_PIXSHUF) && !(d->flags & LLV_CONV_PS_EMAJOR) && d->cout % 128 == 0) ? LLV_TILE_TAP128 : LLV_TILE_TAP64;
            // The tap kernel is bound by the CU's vector-memory path (every 1-KiB LDS-DMA piece costs ~72 cycles of
            // it, every 1-KiB store ~114): per (chunk, tile) a 64-channel workgroup requests 22 halo + 36 weight pieces, a 128-channel
            // one 22 + 72 for twice the MFMAs. Where Cout allows, the 128-channel tile is 9-10 % faster in isolation (128 -> 128 at
            // 16x184x320: 276 vs 303 us; 256 -> 256 at 16x92x160: 256 vs 282 us), slower on launches of a few hundred workgroups
            // (layer3/4 conv2) and on narrow layers (its LDS holds one workgroup per CU less). LLV_TAP128_CIN sets the Cin threshold.
            static const int tap128_cin = getenv("LLV_TAP128_CIN") ? atoi(getenv("LLV_TAP128_CIN")) : 128;
            // (a Cout that is not a multiple of 128 takes the tile too while the padding rows cost under 7 %: the 720-channel class
            // conv of the RetinaHead runs 6 tiles of 128 instead of 12 of 64, 208 -> 203 us on four images)
            const int n128 = (d->cout + 127) / 128;
            if (lp.tile == LLV_TILE_TAP64 && !(d->flags & LLV_CONV_PIXSHUF) && d->cin >= tap128_cin && n128 * 128 * 100 <= d->cout * 107 &&
                blocks / ((d->cout + 63) / 64) * n128 >= 1500)
                lp.tile = LLV_TILE_TAP128;
            static const bool grouped = getenv("LLV_TAP64_G3") && atoi(getenv("LLV_TAP64_G3"));      // A/B: the one-barrier-per-tap-row loop
            if (grouped && lp.tile == LLV_TILE_TAP64) lp.tile = LLV_TILE_TAP64_G3;
            // 64-channel layers with the plain epilogue: the 16 x 32-pixel tile (39 + 36 DMA pieces per chunk for 512 pixels instead of
            // 22 + 36 for 256), when the launch still gives every CU several of the twice-as-large workgroups. Same-box A/B per launch:
            // 64 -> 64 at 16x368x640 331 vs 376 us, 32 -> 32 at 16x736x1280 633 vs 671 us, 128 -> 128 285 vs 304 us (the 128-channel tile
            // above takes that one: 276 us). LLV_TAP512=0 turns it off.
            static const int tap512 = getenv("LLV_TAP512") ? atoi(getenv("LLV_TAP512")) : 1;
            static const long tap512_min = getenv("LLV_TAP512_MIN_BLOCKS") ? atol(getenv("LLV_TAP512_MIN_BLOCKS")) : 4096;
            if (tap512 && lp.tile == LLV_TILE_TAP64 && tap512_supported(d) && blocks >= tap512_min) lp.tile = LLV_TILE_TAP512;
            lp.splits = 1;
            lp.nk_per = (d->k_pad + 31) / 32;
            return LLV_OK;
        }
    }
    // a long reduction on a small map (the dilated 3x3 512 -> 512 of a DC5 stage on one frame: K = 4608, 3680 pixels): 128x128 tiles with
    // split K instead of four times as many 64x64 tiles that each stream all of K (45 us against 65 us)
    if (!d->tile && lp.tile == LLV_TILE_64x64 && d->cout >= 128 && d->k_pad >= 2048) {
        const long blocks128 = ((d->cout + 127) / 128) * ((m + 127) / 128) * d->groups;
        if (blocks128 >= 64) lp.tile = LLV_TILE_128x128;   // (not lower: single-image layer3 conv2, 58 blocks, is 26.6 us on 64x64 tiles, 31.3 on these)
    }
    // a long reduction on a map that fills the chip (layer4's conv1, K = 2048, and layer4.0's strided conv2, K = 4608, at 16 frames:
    // 14,720 pixels): 64-wide k-steps, half as many barriers and waits per MFMA -- 75 -> 59 us and 131 -> 110 us (tools/sweep_1x1_tiles.py,
    // tools/sweep_stride2.py); shorter reductions lose with it (K = 256 / 512: 1.2-1.3x slower). LLV_K64_AUTO=0 turns it off.
    static const int k64_auto = getenv("LLV_K64_AUTO") ? atoi(getenv("LLV_K64_AUTO")) : 1;
    if (k64_auto && !d->tile && lp.tile == LLV_TILE_128x128 && d->k_pad >= 2048 && !(d->k_pad & 63) && d->k_splits <= 0) {
        const long blocks128 = ((d->cout + 127) / 128) * ((m + 127) / 128) * d->groups;
        if (blocks128 >= 256) lp.tile = LLV_TILE_128x128_K64;
    }
    // the small-map tile with 64-wide k-steps for reductions of 1024 .. 2304 (single-image layer3 / layer4 conv1, their 3x3s, the data
    // gradients of the conv3s): 5-16 % per launch alone on the device (tools/sweep_single_image.py), +0.3-0.7 % on the per-frame modes --
    // but its 64 KB of LDS cost the training step 1.2 % (235.8 against 238.7 images/s, same box: those launches run beside the weight
    // gradients there). Off unless LLV_K64_SMALL=1; LLV_TILE_64x64_K64 can be forced per launch.
    static const int k64_small = getenv("LLV_K64_SMALL") ? atoi(getenv("LLV_K64_SMALL")) : 0;
    if (k64_small && !d->tile && lp.tile == LLV_TILE_64x64 && d->k_pad >= 1024 && d->k_pad <= 2304 && !(d->k_pad & 63) && d->k_splits <= 0)
        lp.tile = LLV_TILE_64x64_K64;
    const int bk = tile_bk(lp.tile);
    const int nk = (d->k_pad + bk - 1) / bk;
    lp.splits = d->k_splits > 0 ? d->k_splits : pick_splits(lp.tile, d->cout, m, d->groups, nk, d->flags);
    if (lp.splits > nk) lp.splits = nk;
    if (lp.splits < 1) lp.splits = 1;
    lp.nk_per = (nk + lp.splits - 1) / lp.splits;
    lp.splits = (nk + lp.nk_per - 1) / lp.nk_per;
    return LLV_OK;
}

extern "C" int llv_conv2d_plan(const llv_conv_desc* d, int* tile, int* k_splits) {
    if (!d || d->n_seg < 1 || d->n_seg > LLV_MAX_SEG || d->cout <= 0 || d->groups < 1 || d->k_pad < 32) return LLV_E_BADARG;
    LaunchPlan lp;
    if (d->w_halo && !d->tile) { lp.tile = -1; lp.splits = 1; }        // conv3x3_halo_kernel
    else if (plan_launch(d, lp) != LLV_OK) return LLV_E_BADARG;
    if ((d->flags & LLV_CONV_CLSMAX) && !(d->w_halo && !d->tile)) { lp.tile = LLV_TILE_TAP128; lp.splits = 1; }   // as conv_prepare forces it
    if (tile) *tile = lp.tile;
    if (k_splits) *k_splits = lp.splits;
    return LLV_OK;
}

extern "C" size_t llv_conv2d_workspace_bytes(const llv_conv_desc* d) {
    if (!d || d->n_seg < 1 || d->n_seg > LLV_MAX_SEG || d->cout <= 0 || d->groups < 1 || d->k_pad < 32) return 0;
    if (d->w_halo && !d->tile) return 0;
    LaunchPlan lp;
    plan_launch(d, lp);
    if (lp.splits <= 1) return 0;
    return (size_t)lp.splits * (size_t)lp.m * (size_t)d->groups * d->cout * sizeof(float);
}

extern "C" double llv_conv2d_flops(const llv_conv_desc* d) {
    if (!d) return 0.0;
    double m = 0.0;
    for (int s = 0; s < d->n_seg; ++s) m += (double)d->batch * d->seg_ho[s] * d->seg_wo[s];
    return 2.0 * m * (double)d->groups * d->cout * d->kh * d->kw * d->cin;
}

// everything llv_conv2d_nhwc_fwd decides on the host: validated kernel arguments, the kernel variant and its split
struct ConvPrepared {
    ConvKArgs k;
    LaunchPlan lp;
    bool halo;        // conv3x3_halo_kernel (w_halo given, no forced tile)
    HaloArgs h;
    int halo_bn;
};

static int conv_prepare(const llv_conv_desc* d, ConvPrepared& P) {
    ConvKArgs& k = P.k;
    LaunchPlan& lp = P.lp;
    P.halo = false;
    if (!d || !d->x[0] || !d->w || !d->bias || !d->y) return LLV_E_BADARG;
    if (d->n_seg < 1 || d->n_seg > LLV_MAX_SEG) return LLV_E_BADARG;
    if (d->n_src < 1 || d->n_src > LLV_MAX_SRC) return LLV_E_BADARG;
    if (d->cin <= 0 || (d->cin & 7) || d->cout <= 0 || (d->cout & 3)) return LLV_E_BADARG;
    if ((d->w_rows & 15) || d->w_rows < d->cout) return LLV_E_BADARG;
    if ((d->k_pad & 31) || d->k_pad < d->kh * d->kw * d->cin) return LLV_E_BADARG;
    if (d->kh < 1 || d->kw < 1 || d->stride < 1 || d->pad < 0 || d->groups < 1 || d->batch < 1) return LLV_E_BADARG;
    if (d->dilation < 0) return LLV_E_BADARG;
    if (d->dilation > 1 && (d->flags & LLV_CONV_PIXSHUF)) return LLV_E_UNSUPPORTED;
    if (d->offset_dev) return LLV_E_UNSUPPORTED;           // the sliding-window offset exists on the fused CvBlock kernels only
    k.off_dev = nullptr; k.off_x = 0; k.off_res = 0; k.off_max = 0;
    if ((d->x_ld & 7) || (d->y_ld & 3)) return LLV_E_BADARG;
    if (d->n_src > 1 && d->groups == d->n_src) {
        if (d->src_cvec * 8 != d->cin || d->x_ld < d->cin) return LLV_E_BADARG;   // group g reads source g
        for (int s = 0; s < d->n_src; ++s)
            if (!d->x[s]) return LLV_E_BADARG;
    } else if (d->n_src > 1) {
        if (d->groups != 1 || d->src_cvec < 1 || d->src_cvec * 8 * d->n_src != d->cin) return LLV_E_BADARG;
        for (int s = 0; s < d->n_src; ++s)
            if (!d->x[s]) return LLV_E_BADARG;
        if (d->x_ld < d->src_cvec * 8) return LLV_E_BADARG;
    } else if (d->x_ld < d->groups * d->cin) {
        return LLV_E_BADARG;
    }
    const int f = d->flags;
    if ((f & (LLV_CONV_RES_ADD | LLV_CONV_RES_RSUB | LLV_CONV_RES_UPS)) && (!d->res || (d->res_ld & 3))) return LLV_E_BADARG;
    if ((f & LLV_CONV_RES_ADD) && (f & LLV_CONV_RES_RSUB)) return LLV_E_BADARG;
    if (f & LLV_CONV_RES_UPS) {
        if (!(f & LLV_CONV_RES_ADD) || d->n_seg != 1 || d->res_h < 1 || d->res_w < 1 || (f & LLV_CONV_PIXSHUF)) return LLV_E_BADARG;
    }
    if (f & LLV_CONV_PIXSHUF) {
        if (d->n_seg != 1 || (f & (LLV_CONV_OUT_F32 | LLV_CONV_RES_RSUB))) return LLV_E_BADARG;
        if (d->y_ld < d->groups * d->cout / 4) return LLV_E_BADARG;
    } else if (d->y_ld < (d->y_group_stride ? 1 : d->groups) * d->cout) {
        return LLV_E_BADARG;
    }

    k.x0 = (const _Float16*)d->x[0];
    k.x1 = (const _Float16*)(d->n_src > 1 ? d->x[1] : d->x[0]);
    k.x2 = (const _Float16*)(d->n_src > 2 ? d->x[2] : d->x[0]);
    k.w = (const _Float16*)d->w;
    k.bias = d->bias;
    k.res = (const _Float16*)d->res;
    k.mask = (const _Float16*)d->mask; k.mask_ld = d->mask_ld;
    if (f & LLV_CONV_MASK_POS) {
        if (!d->mask || (d->mask_ld & 3) || d->mask_ld < d->groups * d->cout || ((uintptr_t)d->mask & 7) || d->y_group_stride ||
            (f & (LLV_CONV_PIXSHUF | LLV_CONV_OUT_F32 | LLV_CONV_CLSMAX)))
            return LLV_E_BADARG;
    }
    k.y = d->y;
    k.n_src = d->n_src; k.src_cvec = d->n_src > 1 ? d->src_cvec : (d->cin >> 3);
    k.x_ld = d->x_ld; k.y_ld = d->y_ld; k.res_ld = d->res_ld;
    k.cin = d->cin; k.cout = d->cout; k.w_rows = d->w_rows; k.k_pad = d->k_pad;
    k.kh = d->kh; k.kw = d->kw; k.stride = d->stride; k.pad = d->pad;
    k.dil = d->dilation > 1 ? d->dilation : 1;
    k.n_seg = d->n_seg;
    long m = 0;
    for (int s = 0; s < LLV_MAX_SEG; ++s) {
        const int ss = s < d->n_seg ? s : 0;
        const int H = d->seg_h[ss], W = d->seg_w[ss], Ho = d->seg_ho[ss], Wo = d->seg_wo[ss];
        if (s < d->n_seg) {
            if (H < 1 || W < 1 || H > 32767 || W > 65535 || Ho < 1 || Wo < 1) return LLV_E_BADARG;
            // every tap of every output pixel is either in range or predicated off (reads as zero), so the output may
            // extend past the symmetric-padding formula by up to k-1 taps (the stride-2 data gradient needs one
            // extra zero row/column at the bottom/right); it must still touch the padded input with its first tap.
            if (Ho > (H + 2 * d->pad - 1) / d->stride + 1 || Wo > (W + 2 * d->pad - 1) / d->stride + 1) return LLV_E_BADARG;
            const long xend = d->seg_xoff[s] + (long)d->batch * H * W;
            const long yend = d->seg_yoff[s] + (long)d->batch * Ho * Wo * ((f & LLV_CONV_PIXSHUF) ? 4 : 1);  // upper bound (cropped shuffle stores fewer)
            if (d->seg_xoff[s] < 0 || d->seg_yoff[s] < 0 || xend > 0x7fffffffL || yend > 0x7fffffffL) return LLV_E_BADARG;
        }
        k.seg_h[s] = H; k.seg_w[s] = W; k.seg_ho[s] = Ho; k.seg_wo[s] = Wo;
        k.seg_xoff[s] = (int)d->seg_xoff[ss]; k.seg_yoff[s] = (int)d->seg_yoff[ss];
        k.seg_m0[s] = (int)m;
        if (s < d->n_seg) m += (long)d->batch * Ho * Wo;
        if (m > 0x7fffffffL) return LLV_E_BADARG;
    }
    k.seg_m0[LLV_MAX_SEG] = (int)m;
    k.m_total = (int)m;
    k.flags = f; k.res_h = d->res_h; k.res_w = d->res_w;
    k.w_gstride = (long)d->w_rows * d->k_pad;
    k.x_bytes = 0; k.w_bytes = 0;
    if (getenv("LLV_NO_UT") == nullptr && d->n_seg == 1 && d->kh * d->kw <= 32 && (d->cin & 31) == 0 &&
        (d->n_src == 1 || d->groups == d->n_src || (d->src_cvec & 7) == 0)) {
        const size_t xb = tap_x_bytes(d) + (size_t)(d->pad * d->seg_w[0] + d->pad) * d->x_ld * 2, wb = (size_t)d->w_rows * d->k_pad * 2;
        if (xb < (1ull << 30) && wb < (1ull << 31)) { k.x_bytes = (unsigned)tap_x_bytes(d); k.w_bytes = (unsigned)wb; }
    }
    k.src_per_group = (d->n_src > 1 && d->groups == d->n_src) ? 1 : 0;
    k.act_slope = d->act_slope;
    k.ps_h = 2 * d->seg_ho[0]; k.ps_w = 2 * d->seg_wo[0];
    if ((f & LLV_CONV_PIXSHUF) && d->ps_out_h > 0 && d->ps_out_w > 0) {
        if (d->ps_out_h > k.ps_h || d->ps_out_w > k.ps_w) return LLV_E_BADARG;
        k.ps_h = d->ps_out_h; k.ps_w = d->ps_out_w;
    }
    if ((f & LLV_CONV_RELU) && (f & LLV_CONV_LEAKY)) return LLV_E_BADARG;
    if ((f & LLV_CONV_PS_EMAJOR) && (!(f & LLV_CONV_PIXSHUF) || (d->cout & 31) || d->groups != 1)) return LLV_E_BADARG;
    if ((f & LLV_CONV_ACT_FIRST) && (!(f & LLV_CONV_RES_ADD) || (f & (LLV_CONV_RES_RSUB | LLV_CONV_RES_UPS | LLV_CONV_PIXSHUF | LLV_CONV_OUT_F32)) || d->k_splits > 1)) return LLV_E_BADARG;
    k.y_gstride = d->y_group_stride;
    if (d->y_group_stride) {
        if (d->y_group_stride < 0 || (f & (LLV_CONV_OUT_F32 | LLV_CONV_PIXSHUF)) || d->y_ld < d->cout || (d->y_group_stride & 7)) return LLV_E_BADARG;
    }
    // LDS-staged stores need 16-byte aligned channel runs in y (and in res for the shuffled skip add)
    k.stage_out = 0;
    if (!(f & LLV_CONV_OUT_F32) && (d->y_ld & 7) == 0) {
        if (f & LLV_CONV_PIXSHUF) {
            const bool res_ok = !(f & LLV_CONV_RES_ADD) || (d->res_ld & 7) == 0;
            if ((d->cout & 31) == 0 && res_ok) k.stage_out = 1;
        } else if ((d->cout & 7) == 0 || d->groups == 1 || d->y_group_stride) {
            k.stage_out = 1;
        }
    }
    plan_launch(d, lp);
    k.clsmax = nullptr; k.segw = nullptr; k.cls_c = 0; k.store_thr = 0.f;
    if (f & LLV_CONV_CLSMAX) {
        if (!(f & LLV_CONV_OUT_F32) || (f & ~(LLV_CONV_OUT_F32 | LLV_CONV_CLSMAX)) || !d->clsmax || !d->seg_written || d->cls_c < 16 ||
            (d->cls_c & 15) || d->cout % d->cls_c || d->groups != 1 || d->n_src != 1 || d->y_group_stride || d->k_splits > 1)
            return LLV_E_BADARG;
        if (!tap_eligible(d) || (d->tile && d->tile != LLV_TILE_TAP128)) return LLV_E_UNSUPPORTED;
        lp.tile = LLV_TILE_TAP128; lp.splits = 1;          // the segment flags are per 128-channel tile
        k.clsmax = d->clsmax; k.segw = d->seg_written; k.cls_c = d->cls_c; k.store_thr = d->store_thr;
    }
    k.k_splits = lp.splits;
    k.nk_per = lp.nk_per;
    k.ws = (float*)d->workspace;
    if (lp.splits > 1 && !(d->w_halo && !d->tile)) {
        if (f & LLV_CONV_PIXSHUF) return LLV_E_BADARG;
        if (!d->workspace || d->workspace_bytes < llv_conv2d_workspace_bytes(d)) return LLV_E_WORKSPACE;
    }

    if (d->w_halo && !d->tile) {
        // direct (halo-in-LDS) path: 3x3 / stride 1 / pad 1, one segment, Cin in {8,16,24} or a multiple of 32
        if (d->kh != 3 || d->kw != 3 || d->stride != 1 || d->pad != 1 || d->dilation > 1 || d->n_seg != 1) return LLV_E_BADARG;
        if (!(d->cin == 8 || d->cin == 16 || d->cin == 24 || (d->cin & 31) == 0)) return LLV_E_BADARG;
        if (d->n_src > 1 && d->groups != d->n_src && (d->groups != 1 || (d->src_cvec * 8) % (d->cin < 32 ? d->cin : 32) != 0)) return LLV_E_BADARG;
        if (d->halo_bn != 16 && d->halo_bn != 32 && d->halo_bn != 64) return LLV_E_BADARG;
        HaloArgs& h = P.h;
        h.w = (const _Float16*)d->w_halo;
        h.H = d->seg_h[0]; h.W = d->seg_w[0];
        if (d->seg_ho[0] != h.H || d->seg_wo[0] != h.W) return LLV_E_BADARG;
        h.tiles_x = (h.W + HALO_TW - 1) / HALO_TW;
        h.tiles_y = (h.H + HALO_TH - 1) / HALO_TH;
        h.ck = d->cin < 32 ? d->cin : 32;
        h.n_chunks = d->cin / h.ck;
        h.kc_pad = (9 * h.ck + 31) & ~31;
        h.wrow = h.kc_pad + 8;
        const int bn = d->halo_bn;
        const int nt = (d->cout + bn - 1) / bn;
        h.w_gstride = (long)h.n_chunks * nt * bn * h.wrow;
        if ((f & LLV_CONV_PIXSHUF) && bn < 32) k.stage_out = 0;
        k.k_splits = 1;
        P.halo = true;
        P.halo_bn = bn;
        return LLV_OK;
    }
    const int tile = lp.tile;
    if ((f & LLV_CONV_PS_EMAJOR) && !((tile == LLV_TILE_TAP64 || tile == LLV_TILE_TAP64_G3 || tile == LLV_TILE_TAP512) && tap_eligible(d))) return LLV_E_UNSUPPORTED;
    if ((f & LLV_CONV_PIXSHUF) && tile == LLV_TILE_16x256) k.stage_out = 0;
    if ((tile == LLV_TILE_TAP128 || tile == LLV_TILE_TAP64 || tile == LLV_TILE_TAP64_R4 || tile == LLV_TILE_TAP64_G3) && !tap_eligible(d)) return LLV_E_BADARG;
    if (tile == LLV_TILE_TAP512 && !tap512_supported(d)) return LLV_E_UNSUPPORTED;
    if (tile < LLV_TILE_128x128 || tile > LLV_TILE_64x64_K64) return LLV_E_UNSUPPORTED;
    return LLV_OK;
}

static bool extents_enabled() {
#ifdef LLV_DEBUG_EXTENTS
    return true;
#else
    static const bool on = getenv("LLV_CHECK_EXTENTS") != nullptr && atoi(getenv("LLV_CHECK_EXTENTS")) != 0;
    return on;
#endif
}

extern "C" int llv_conv2d_nhwc_fwd(const llv_conv_desc* d, void* stream) {
    ConvPrepared P;
    const int rc = conv_prepare(d, P);
    if (rc != LLV_OK) return rc;
    if (d->extents && extents_enabled()) {
        llv_extents need;
        const int rce = llv_conv2d_extents(d, &need);
        if (rce != LLV_OK) return rce;
        if (llv_extents_fit(&need, d->extents) != LLV_OK) return LLV_E_EXTENT;
    }
    ConvKArgs& k = P.k;
    hipStream_t st = (hipStream_t)stream;
    if (P.halo) {
        switch (P.halo_bn) {
            case 16: return launch_halo<16>(k, P.h, d->groups, d->batch, st);
            case 32: return launch_halo<32>(k, P.h, d->groups, d->batch, st);
            default: return launch_halo<64>(k, P.h, d->groups, d->batch, st);
        }
    }
    const int tile = P.lp.tile;
    switch (tile) {
        case LLV_TILE_128x128: return launch_glds<128, 128, 64, 64>(k, d->groups, st);
        case LLV_TILE_64x256:  return launch_glds<64, 256, 64, 64>(k, d->groups, st);
        case LLV_TILE_32x256:  return launch_glds<32, 256, 32, 64>(k, d->groups, st);
        case LLV_TILE_16x256:  return launch_glds<16, 256, 16, 64>(k, d->groups, st);
        case LLV_TILE_64x64:   return launch_glds<64, 64, 32, 32>(k, d->groups, st);
        case LLV_TILE_64x64_K64: return launch_glds<64, 64, 32, 32, 64>(k, d->groups, st);
        case LLV_TILE_128x128_REGSTAGE: return launch<128, 128, 64, 64>(k, d->groups, st);   // register-staged loop, kept for A/B
        case LLV_TILE_128x128_K64: return launch_glds<128, 128, 64, 64, 64>(k, d->groups, st);
        case LLV_TILE_TAP128: return tap_eligible(d) ? launch_tap<128>(k, d, st) : LLV_E_BADARG;
        case LLV_TILE_TAP64:  return tap_eligible(d) ? launch_tap<64>(k, d, st) : LLV_E_BADARG;      // the tap kernels are the only ones that read LLV_CONV_PS_EMAJOR
        case LLV_TILE_TAP64_R4: return tap_eligible(d) ? launch_tap<64, 4>(k, d, st) : LLV_E_BADARG;
        case LLV_TILE_TAP64_G3: return tap_eligible(d) ? launch_tap<64, 0>(k, d, st) : LLV_E_BADARG;
        case LLV_TILE_TAP512: return tap512_supported(d) ? launch_tap512(k, d, st) : LLV_E_UNSUPPORTED;
        case LLV_TILE_64x256_K64:  return launch_glds<64, 256, 64, 64, 64>(k, d->groups, st);
        default: return LLV_E_UNSUPPORTED;
    }
}


// round-4 LDS layouts of the fused CvBlock kernels (halo planes, 50-pixel patch pitch); LLV_CP2_PLANES=0: the round-3 layouts (A/B)
static bool cp2_planes() { const char* e = getenv("LLV_CP2_PLANES"); return !(e && e[0] == '0'); }

struct PairPrepared {
    ConvKArgs k;
    HaloArgs h;
    PreArgs pre;
    Pair2Args q;
    bool v2;          // convpair2_kernel (DMA staging, persistent workgroups); false: the register-staged convpair_kernel
    size_t lds;
    int grid2;
};

static int convpair_prepare(const llv_inputcv_desc* p, bool output_block, PairPrepared& P) {
    if (!p || !p->w_pre || !p->bias_pre) return LLV_E_BADARG;
    const llv_conv_desc* d = &p->conv;
    if (!d->w_halo || !d->bias || !d->y || d->n_seg != 1 || d->kh != 3 || d->kw != 3 || d->stride != 1 || d->pad != 1) return LLV_E_BADARG;
    if (d->groups != 1 || (d->y_ld & 7) || d->y_ld < d->cout || (d->cout & 7) || d->cout < 8) return LLV_E_BADARG;
    if (d->seg_h[0] < 1 || d->seg_w[0] < 1 || d->seg_ho[0] != d->seg_h[0] || d->seg_wo[0] != d->seg_w[0] || d->batch < 1) return LLV_E_BADARG;
    const long px = (long)d->batch * d->seg_h[0] * d->seg_w[0];
    if (d->seg_xoff[0] < 0 || d->seg_yoff[0] < 0 || d->seg_xoff[0] + px > 0x7fffffffL || d->seg_yoff[0] + px > 0x7fffffffL) return LLV_E_BADARG;
    ConvKArgs& k = P.k;
    k = ConvKArgs{};
    if (!output_block) {
        if (d->n_src != 3 || !d->x[0] || !d->x[1] || !d->x[2] || d->x_ld < 8 || (d->x_ld & 7)) return LLV_E_BADARG;
        if (d->cin != 96 || d->cout != 32 || (d->flags & ~(LLV_CONV_RELU))) return LLV_E_BADARG;
    } else {
        if (d->n_src != 1 || !d->x[0] || d->x_ld < 32 || (d->x_ld & 7)) return LLV_E_BADARG;
        if (d->cin != 32 || d->cout > 16 || (d->flags & ~(LLV_CONV_RELU | LLV_CONV_RES_RSUB | LLV_CONV_RES_ADD))) return LLV_E_BADARG;
        if ((d->flags & (LLV_CONV_RES_RSUB | LLV_CONV_RES_ADD)) && (!d->res || (d->res_ld & 3) || d->res_ld < d->cout)) return LLV_E_BADARG;
    }
    k.x0 = (const _Float16*)d->x[0];
    k.x1 = (const _Float16*)(output_block ? d->x[0] : d->x[1]);
    k.x2 = (const _Float16*)(output_block ? d->x[0] : d->x[2]);
    k.bias = d->bias; k.y = d->y; k.res = (const _Float16*)d->res; k.mask = nullptr; k.mask_ld = 0;
    k.n_src = d->n_src; k.src_cvec = 1; k.x_ld = d->x_ld; k.y_ld = d->y_ld; k.res_ld = d->res_ld;
    k.cin = d->cin; k.cout = d->cout; k.kh = 3; k.kw = 3; k.stride = 1; k.pad = 1; k.dil = 1; k.n_seg = 1;
    for (int s = 0; s < LLV_MAX_SEG; ++s) {
        k.seg_h[s] = d->seg_h[0]; k.seg_w[s] = d->seg_w[0]; k.seg_ho[s] = d->seg_h[0]; k.seg_wo[s] = d->seg_w[0];
        k.seg_xoff[s] = (int)d->seg_xoff[0]; k.seg_yoff[s] = (int)d->seg_yoff[0]; k.seg_m0[s] = 0;
    }
    k.seg_m0[LLV_MAX_SEG] = (int)px; k.m_total = (int)px;
    k.flags = d->flags; k.stage_out = 1; k.k_splits = 1; k.y_gstride = 0; k.src_per_group = 0;
    k.act_slope = 0.f; k.ps_h = 0; k.ps_w = 0;
    k.off_dev = (const long*)d->offset_dev; k.off_x = d->offset_x; k.off_res = d->offset_res; k.off_max = d->offset_max;
    if (d->offset_dev && (d->offset_max < 0 || (!d->offset_x && !d->offset_res))) return LLV_E_BADARG;
    HaloArgs& h = P.h;
    h.w = (const _Float16*)d->w_halo;
    h.H = d->seg_h[0]; h.W = d->seg_w[0];
    h.tiles_x = (h.W + HALO_TW - 1) / HALO_TW; h.tiles_y = (h.H + HALO_TH - 1) / HALO_TH;
    h.ck = 32; h.n_chunks = output_block ? 1 : 3; h.kc_pad = 288; h.wrow = 296; h.ntiles_n = 1;
    h.nblk = h.tiles_x * h.tiles_y * d->batch; h.w_gstride = 0;
    PreArgs& pre = P.pre;
    pre.w = (const _Float16*)p->w_pre; pre.bias = p->bias_pre; pre.wrow = output_block ? 296 : 104;
    const int ck0 = output_block ? 32 : 8, bn2 = output_block ? 16 : 32;
    size_t lds = (size_t)HALO_PX * 64 + (size_t)bn2 * h.wrow * 2 + (size_t)PATCH_PX * ck0 * 2 + (size_t)32 * pre.wrow * 2;
    const size_t epi = (size_t)256 * (bn2 + 8) * 2;
    if (lds < epi) lds = epi;
    P.lds = lds;
    const size_t xb = (size_t)(d->seg_xoff[0] + px) * d->x_ld * 2;
    P.v2 = getenv("LLV_CONVPAIR_V1") == nullptr && xb < (1ull << 31);
    if (d->offset_dev && !P.v2) return LLV_E_UNSUPPORTED;
    Pair2Args& q = P.q;
    q.x_bytes = (unsigned)xb;
    q.w0_bytes = (unsigned)((output_block ? 1 : 3) * 32 * pre.wrow * 2);
    q.w1_bytes = (unsigned)(h.n_chunks * bn2 * h.wrow * 2);
    // persistent workgroups: two per CU (the LDS footprint's limit), each walks tiles v, v + grid, ...
    P.grid2 = h.nblk < 512 ? h.nblk : 512;
    return LLV_OK;
}

static int convpair_launch(const llv_inputcv_desc* p, void* stream, bool output_block) {
    PairPrepared P;
    const int rc = convpair_prepare(p, output_block, P);
    if (rc != LLV_OK) return rc;
    if (p->conv.extents && extents_enabled()) {
        llv_extents need;
        const int rce = llv_convpair_extents(p, output_block ? 1 : 0, &need);
        if (rce != LLV_OK) return rce;
        if (llv_extents_fit(&need, p->conv.extents) != LLV_OK) return LLV_E_EXTENT;
    }
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(convpair_kernel<32, 16>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(convpair_kernel<8, 32>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
        attr_done = true;
    }
    ConvKArgs& k = P.k;
    HaloArgs& h = P.h;
    if (P.v2) {
        static const bool planes = cp2_planes();
        if (planes) {
            if (output_block) hipLaunchKernelGGL((convpair2_kernel<32, 16, true>), dim3(P.grid2), dim3(256), 0, (hipStream_t)stream, k, h, P.pre, P.q);
            else hipLaunchKernelGGL((convpair2_kernel<8, 32, true>), dim3(P.grid2), dim3(256), 0, (hipStream_t)stream, k, h, P.pre, P.q);
        } else {
            if (output_block) hipLaunchKernelGGL((convpair2_kernel<32, 16, false>), dim3(P.grid2), dim3(256), 0, (hipStream_t)stream, k, h, P.pre, P.q);
            else hipLaunchKernelGGL((convpair2_kernel<8, 32, false>), dim3(P.grid2), dim3(256), 0, (hipStream_t)stream, k, h, P.pre, P.q);
        }
        return hipGetLastError() == hipSuccess ? LLV_OK : LLV_E_LAUNCH;
    }
    if (output_block) hipLaunchKernelGGL((convpair_kernel<32, 16>), dim3(h.nblk), dim3(256), P.lds, (hipStream_t)stream, k, h, P.pre);
    else hipLaunchKernelGGL((convpair_kernel<8, 32>), dim3(h.nblk), dim3(256), P.lds, (hipStream_t)stream, k, h, P.pre);
    return hipGetLastError() == hipSuccess ? LLV_OK : LLV_E_LAUNCH;
}

extern "C" int llv_inputcvblock_fwd(const llv_inputcv_desc* p, void* stream) { return convpair_launch(p, stream, false); }
extern "C" int llv_outputcvblock_fwd(const llv_inputcv_desc* p, void* stream) { return convpair_launch(p, stream, true); }

#include "conv_extents.inc"
