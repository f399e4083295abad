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
    void* y;
    int n_src, src_cvec;
    int x_ld, y_ld, res_ld;
    int cin, cout, w_rows, k_pad;
    int kh, kw, stride, pad;
    int n_seg;
    int seg_h[LLV_MAX_SEG], seg_w[LLV_MAX_SEG], seg_ho[LLV_MAX_SEG], seg_wo[LLV_MAX_SEG];
    int seg_m0[LLV_MAX_SEG + 1];
    int seg_xoff[LLV_MAX_SEG], seg_yoff[LLV_MAX_SEG];  // pixel offsets (fit int32, checked by host)
    int flags;
    int res_h, res_w;
    int tiles_n, nblk;
    int m_total;
    long w_gstride;  // elements between groups in w
};

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

    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * STAGE];

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

    // ---- K walk state: (r, q, c8) of this lane's chunk
    int kc8 = ch, kr = 0, kq = 0;
    while (kc8 >= cvec) { kc8 -= cvec; if (++kq == a.kw) { kq = 0; ++kr; } }

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
        const _Float16* xb;
        int cofs;
        if (a.n_src > 1) {
            const int s = kc8 / a.src_cvec;
            xb = s == 0 ? a.x0 : (s == 1 ? a.x1 : a.x2);
            cofs = (kc8 - s * a.src_cvec) * 8;
        } else {
            xb = a.x0;
            cofs = g * a.cin + kc8 * 8;
        }
        const bool tap_ok = kr < a.kh;
#pragma unroll
        for (int i = 0; i < XVEC; ++i) {
            const int H = px_hw[i] >> 16, W = px_hw[i] & 0xffff;
            const int y = px_iy0[i] + kr, x = px_ix0[i] + kq;
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

    const int nk = a.k_pad >> 5;
    gload(0);
    lstore(0);
    __syncthreads();

    const int frow = lane & 15, fch = lane >> 4;
    for (int ks = 0; ks < nk; ++ks) {
        const int cur = ks & 1;
        if (ks + 1 < nk) gload(ks + 1);

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
    const int flags = a.flags;
#pragma unroll
    for (int j = 0; j < TM; ++j) {
        const int m = tile_m * BM + wm * WM + j * 16 + frow;
        if (m >= a.m_total) continue;
        int Ho = a.seg_ho[0], Wo = a.seg_wo[0], m0 = 0, yo = a.seg_yoff[0];
#pragma unroll
        for (int s = 1; s < LLV_MAX_SEG; ++s) {
            if (s < a.n_seg && m >= a.seg_m0[s]) {
                Ho = a.seg_ho[s]; Wo = a.seg_wo[s]; m0 = a.seg_m0[s]; yo = a.seg_yoff[s];
            }
        }
        const int ml = m - m0;
        long opix = (long)yo + ml;  // output pixel index (dense batch x Ho x Wo per segment)
        long rpix = opix;
        int ox = 0, oy = 0, n = 0;
        if (flags & (LLV_CONV_PIXSHUF | LLV_CONV_RES_UPS)) {
            ox = ml % Wo;
            const int t = ml / Wo;
            oy = t % Ho;
            n = t / Ho;
        }
        if (flags & LLV_CONV_RES_UPS) {
            // F.interpolate(mode='nearest', size=(Ho,Wo)): src = min(floor(dst * in/out), in-1)
            const float sy = (float)a.res_h / (float)Ho, sx = (float)a.res_w / (float)Wo;
            int ry = (int)floorf((float)oy * sy); ry = ry < a.res_h - 1 ? ry : a.res_h - 1;
            int rx = (int)floorf((float)ox * sx); rx = rx < a.res_w - 1 ? rx : a.res_w - 1;
            rpix = ((long)n * a.res_h + ry) * a.res_w + rx;
        }
#pragma unroll
        for (int i = 0; i < TN; ++i) {
            const int co = tile_n * BN + wn * WN + i * 16 + fch * 4;
            if (co >= a.cout) continue;
            const int gco = g * a.cout + co;
            const float4 b4 = *reinterpret_cast<const float4*>(a.bias + gco);
            float v[4] = {acc[i][j][0] + b4.x, acc[i][j][1] + b4.y, acc[i][j][2] + b4.z, acc[i][j][3] + b4.w};
            if (flags & LLV_CONV_PIXSHUF) {
                // conv channel c = 4*cc + 2*dy + dx -> out pixel (2oy+dy, 2ox+dx), channel cc
                const int cc = gco >> 2;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int dy = e >> 1, dx = e & 1;
                    const long p = (long)yo + ((long)(n * 2 * Ho + 2 * oy + dy)) * (2 * Wo) + (2 * ox + dx);
                    float o = v[e];
                    if (flags & LLV_CONV_RES_ADD) o += (float)a.res[p * a.res_ld + cc];
                    if (flags & LLV_CONV_RELU) o = fmaxf(o, 0.f);
                    reinterpret_cast<_Float16*>(a.y)[p * a.y_ld + cc] = (_Float16)o;
                }
                continue;
            }
            if (flags & (LLV_CONV_RES_ADD | LLV_CONV_RES_RSUB)) {
                const f16x4 r4 = *reinterpret_cast<const f16x4*>(a.res + rpix * a.res_ld + gco);
                if (flags & LLV_CONV_RES_RSUB) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = (float)r4[e] - v[e];
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += (float)r4[e];
                }
            }
            if (flags & LLV_CONV_RELU) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
            }
            if (flags & LLV_CONV_OUT_F32) {
                *reinterpret_cast<float4*>(reinterpret_cast<float*>(a.y) + opix * a.y_ld + gco) =
                    make_float4(v[0], v[1], v[2], v[3]);
            } else {
                f16x4 o4 = {(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
                *reinterpret_cast<f16x4*>(reinterpret_cast<_Float16*>(a.y) + opix * a.y_ld + gco) = o4;
            }
        }
    }
}

template <int BN, int BM, int WN, int WM>
int launch(ConvKArgs& k, int groups, hipStream_t st) {
    k.tiles_n = (k.cout + BN - 1) / BN;
    const int tiles_m = (k.m_total + BM - 1) / BM;
    k.nblk = k.tiles_n * tiles_m;
    dim3 grid(k.nblk, groups, 1);
    hipLaunchKernelGGL((conv_igemm_kernel<BN, BM, WN, WM>), grid, dim3(256), 0, st, k);
    return hipGetLastError() == hipSuccess ? LLV_OK : LLV_E_LAUNCH;
}

int pick_tile(const ConvKArgs& k, int groups) {
    const int co = k.cout;
    const long m = k.m_total;
    if (co <= 16) return LLV_TILE_16x256;
    if (co <= 32) return LLV_TILE_32x256;
    if (co <= 64) return m >= 64 * 1024 ? LLV_TILE_64x256 : LLV_TILE_64x64;
    // enough 128x128 tiles to fill 256 CUs a couple of times over? otherwise go small
    const long blocks128 = ((co + 127) / 128) * ((m + 127) / 128) * groups;
    if (blocks128 >= 384) return LLV_TILE_128x128;
    return LLV_TILE_64x64;
}

}  // namespace

extern "C" double llv_conv2d_flops(const llv_conv_desc* d) {
    if (!d) return 0.0;
    double m = 0.0;
    for (int s = 0; s < d->n_seg; ++s) m += (double)d->batch * d->seg_ho[s] * d->seg_wo[s];
    return 2.0 * m * (double)d->groups * d->cout * d->kh * d->kw * d->cin;
}

extern "C" int llv_conv2d_nhwc_fwd(const llv_conv_desc* d, void* stream) {
    if (!d || !d->x[0] || !d->w || !d->bias || !d->y) return LLV_E_BADARG;
    if (d->n_seg < 1 || d->n_seg > LLV_MAX_SEG) return LLV_E_BADARG;
    if (d->n_src < 1 || d->n_src > LLV_MAX_SRC) return LLV_E_BADARG;
    if (d->cin <= 0 || (d->cin & 7) || d->cout <= 0 || (d->cout & 3)) return LLV_E_BADARG;
    if ((d->w_rows & 15) || d->w_rows < d->cout) return LLV_E_BADARG;
    if ((d->k_pad & 31) || d->k_pad < d->kh * d->kw * d->cin) return LLV_E_BADARG;
    if (d->kh < 1 || d->kw < 1 || d->stride < 1 || d->pad < 0 || d->groups < 1 || d->batch < 1) return LLV_E_BADARG;
    if ((d->x_ld & 7) || (d->y_ld & 3)) return LLV_E_BADARG;
    if (d->n_src > 1) {
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
    } else if (d->y_ld < d->groups * d->cout) {
        return LLV_E_BADARG;
    }

    ConvKArgs k;
    k.x0 = (const _Float16*)d->x[0];
    k.x1 = (const _Float16*)(d->n_src > 1 ? d->x[1] : d->x[0]);
    k.x2 = (const _Float16*)(d->n_src > 2 ? d->x[2] : d->x[0]);
    k.w = (const _Float16*)d->w;
    k.bias = d->bias;
    k.res = (const _Float16*)d->res;
    k.y = d->y;
    k.n_src = d->n_src; k.src_cvec = d->n_src > 1 ? d->src_cvec : (d->cin >> 3);
    k.x_ld = d->x_ld; k.y_ld = d->y_ld; k.res_ld = d->res_ld;
    k.cin = d->cin; k.cout = d->cout; k.w_rows = d->w_rows; k.k_pad = d->k_pad;
    k.kh = d->kh; k.kw = d->kw; k.stride = d->stride; k.pad = d->pad;
    k.n_seg = d->n_seg;
    long m = 0;
    for (int s = 0; s < LLV_MAX_SEG; ++s) {
        const int ss = s < d->n_seg ? s : 0;
        const int H = d->seg_h[ss], W = d->seg_w[ss], Ho = d->seg_ho[ss], Wo = d->seg_wo[ss];
        if (s < d->n_seg) {
            if (H < 1 || W < 1 || H > 32767 || W > 65535 || Ho < 1 || Wo < 1) return LLV_E_BADARG;
            // geometry must agree with (k, stride, pad): every tap of every output pixel is either in
            // range or predicated off, but the output extent must not exceed the conv formula.
            if (Ho > (H + 2 * d->pad - d->kh) / d->stride + 1 || Wo > (W + 2 * d->pad - d->kw) / d->stride + 1) return LLV_E_BADARG;
            const long xend = d->seg_xoff[s] + (long)d->batch * H * W;
            const long yend = d->seg_yoff[s] + (long)d->batch * Ho * Wo * ((f & LLV_CONV_PIXSHUF) ? 4 : 1);
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

    hipStream_t st = (hipStream_t)stream;
    int tile = d->tile ? d->tile : pick_tile(k, d->groups);
    switch (tile) {
        case LLV_TILE_128x128: return launch<128, 128, 64, 64>(k, d->groups, st);
        case LLV_TILE_64x256:  return launch<64, 256, 64, 64>(k, d->groups, st);
        case LLV_TILE_32x256:  return launch<32, 256, 32, 64>(k, d->groups, st);
        case LLV_TILE_16x256:  return launch<16, 256, 16, 64>(k, d->groups, st);
        case LLV_TILE_64x64:   return launch<64, 64, 32, 32>(k, d->groups, st);
        default: return LLV_E_UNSUPPORTED;
    }
}
