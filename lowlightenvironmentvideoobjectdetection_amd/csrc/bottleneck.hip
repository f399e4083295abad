// bottleneck.hip -- a whole stride-1 ResNet Bottleneck (resnet.py:260-299: conv1 1x1 -> BN -> ReLU -> conv2 3x3 -> BN -> ReLU ->
// conv3 1x1 -> BN -> + identity -> ReLU, BatchNorm folded) in ONE launch for the wide, shallow stages (layer1: 256 <-> 64,
// layer2: 512 <-> 128), where the three separate launches are bound by HBM: they write and re-read the two narrow intermediates
// and read the wide input twice (conv1 and the identity). Here a workgroup owns an 8 x TW pixel tile:
//
//   stage 1  conv1 on the (8+2) x (TW+2) halo: K = C in 32-channel chunks; each chunk's halo pixels and the MID x 32 weight slice
//            arrive in LDS by DMA (double buffered); result (+bias, ReLU, zero outside the image = conv2's padding) -> T1 in LDS
//   stage 2  conv2 exactly as conv3x3_tap_kernel runs it (chunk-major, tap-minor, weight slices streamed by DMA), its pixel operand
//            being T1, which never leaves LDS; result (+bias, ReLU) -> T2 in LDS
//   stage 3  conv3 in 64-channel chunks of the output: K = MID out of T2, weight chunks streamed; + identity (the tile's own pixels
//            of x, hot in L2: stage 1 fetched them moments ago) + ReLU through the LDS staging tile -> 16-byte global stores
//
// HBM traffic per block: x once + y once (964 MB instead of 1926 MB at 16 x 184 x 320 x 256). MFMA work rises by the halo
// recomputation of conv1 (+8 % of the block). fp16 roundings happen at the same three points as in the separate launches and the
// accumulation order is the same (k ascending; chunk-major / tap-minor for conv2), so the result is bit-identical to them.
// One workgroup (4 waves) per CU: LDS 103 KB (MID = 64, TW = 32) / 109 KB (MID = 128, TW = 16).

#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <cstdlib>
#include "llv.h"

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

struct BnkArgs {
    const _Float16* x;
    _Float16* y;
    const _Float16 *w1, *w2, *w3;
    const float *b1, *b2, *b3;
    int batch, H, W, x_ld, y_ld;
    int tiles_x, tiles_y, nblk;
    unsigned x_bytes, w1_bytes, w2_bytes, w3_bytes;
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), (short)0, (int)bytes, 0x00020000);
}
// one wave-instruction of LDS DMA: lane l's 16 bytes at base + voff + soff land at lds + 16*l; voff >= the resource's size reads zero
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t r, unsigned char* lds, int voff, int soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds, 16, voff, soff, 0, 0);
}
// [rows][32 halfs] weight slice image: 16-byte chunk ch of row `row` (conflict-free ds_read_b128 fragments, as conv_igemm.hip)
__device__ __forceinline__ int lds_off(int row, int ch) { return row * 64 + (((ch ^ ((0 - (row >> 2)) & 3)) & 3) << 4); }

constexpr int TH = 8;

template <int MID, int TW>
struct Geo {
    static constexpr int C = 4 * MID;
    static constexpr int HW_ = TW + 2;                       // halo row length
    static constexpr int HP = (TH + 2) * HW_;                // halo pixels
    static constexpr int TPX = TH * TW;                      // output pixels
    static constexpr int XP = (HP * 64 + 1023) / 1024;       // 1-KiB pieces of one 32-channel chunk of the halo
    static constexpr int NT1 = (HP + 15) / 16;               // stage-1 n-tiles (16 halo pixels each)
    static constexpr int NT1W = (NT1 + 3) / 4;               // per wave
    static constexpr int MT = MID / 16;                      // m-tiles (16 rows) of stages 1 and 2
    static constexpr int NT2W = TPX / 16 / 4;                // stage-2 / 3 n-tiles per wave (4 or 2)
    static constexpr int PXB = MID * 2;                      // bytes per pixel of T1 / T2
    static constexpr int NCH = PXB / 16;                     // 16-byte chunks per pixel
    static constexpr int R_A = (HP * PXB > TPX * 72 * 2 ? HP * PXB : TPX * 72 * 2);          // T1 | staging tile [TPX][64 + 8]
    static constexpr int R_B = (2 * XP * 1024 > TPX * PXB ? 2 * XP * 1024 : TPX * PXB);     // x-chunk ring | T2
    static constexpr int WSLOT = 64 * PXB;                   // one conv3 weight chunk (64 rows x MID); conv1/2 slices (MID x 64 B) fit
    static constexpr int SMEM = R_A + R_B + 2 * WSLOT;
    static constexpr int WP12 = MID / 16;                    // 1-KiB pieces of a conv1 / conv2 weight slice (MID rows x 64 B)
    static constexpr int WP3 = (MID / 32) * 4;               // pieces of a conv3 chunk: [k-step][64 rows][64 B]
    static_assert(MID * 64 <= WSLOT && SMEM <= 160 * 1024, "LDS budget");
    static_assert(TPX / 16 % 4 == 0, "n-tiles split over 4 waves");
};

// x-chunk halo image (64-byte pixels): chunk c8 of pixel (hy, hx) -- the column-keyed swizzle of the tap kernel
template <int HW_>
__device__ __forceinline__ int xh_off(int hy, int hx, int c8) { return (hy * HW_ + hx) * 64 + ((c8 ^ (((hx >> 2) & 1) << 1)) << 4); }
// T1 / T2 images (PXB-byte pixels): chunk c8 of linear pixel p whose swizzle key is `key` (the halo column for T1, the pixel index
// for T2): 16 consecutive keys x one chunk land on 16 distinct 16-byte slots for ds_read_b128's lane groups (128-byte pixels)
template <int PXB>
__device__ __forceinline__ int t_off(int p, int key, int c8) { return p * PXB + ((c8 ^ (key & (PXB / 16 - 1))) << 4); }

template <int MID, int TW>
__global__ __launch_bounds__(256, 1) void bottleneck_kernel(const BnkArgs a) {
    using G = Geo<MID, TW>;
    constexpr int C = G::C, HW_ = G::HW_, HP = G::HP, TPX = G::TPX, XP = G::XP, NT1W = G::NT1W, MT = G::MT, NT2W = G::NT2W, PXB = G::PXB;
    constexpr int XPW = (XP + 3) / 4, W12PW = (G::WP12 + 3) / 4, W3PW = G::WP3 / 4;
    constexpr int ROWB = HW_ * PXB;                          // bytes per T1 halo row
    constexpr int CPT = TW / 16;                             // n-tiles per output row
    __shared__ __attribute__((aligned(16))) unsigned char smem[G::SMEM];
    unsigned char* const sA = smem;                          // T1, later the staging tile
    unsigned char* const sB = smem + G::R_A;                 // x-chunk ring, later T2
    unsigned char* const sW = smem + G::R_A + G::R_B;        // two weight slots

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 15, fch = lane >> 4;
    int bid = blockIdx.x;
    {
        const int nblk = a.nblk;
        const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int H = a.H, W = a.W;
    const int tx0 = (bid % a.tiles_x) * TW;
    const int t1_ = bid / a.tiles_x;
    const int ty0 = (t1_ % a.tiles_y) * TH;
    const int n = t1_ / a.tiles_y;
    const int img_base = n * H * W;

    const __amdgpu_buffer_rsrc_t rX = make_rsrc(a.x, a.x_bytes);
    const __amdgpu_buffer_rsrc_t rW1 = make_rsrc(a.w1, a.w1_bytes);
    const __amdgpu_buffer_rsrc_t rW2 = make_rsrc(a.w2, a.w2_bytes);
    const __amdgpu_buffer_rsrc_t rW3 = make_rsrc(a.w3, a.w3_bytes);

    // ---- DMA offsets
    int hv[XPW];                                             // this lane's halo pixel / chunk slot of x pieces wave, wave+4, ...
#pragma unroll
    for (int i = 0; i < XPW; ++i) {
        const int px = (wave + 4 * i) * 16 + (lane >> 2);
        hv[i] = (int)a.x_bytes;
        if (px < HP) {
            const int hy = px / HW_, hx = px - hy * HW_;
            const int gy = ty0 + hy - 1, gx = tx0 + hx - 1;
            const int c8 = (lane & 3) ^ (((hx >> 2) & 1) << 1);
            if ((unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W) hv[i] = ((img_base + gy * W + gx) * a.x_ld + c8 * 8) * 2;
        }
    }
    const int wch = (lane & 3) ^ ((0 - (lane >> 4)) & 3);
    const int wrow = lane >> 2;                              // row inside a 16-row piece
    auto issue_x = [&](int c, int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < XPW; ++i)
            if (wave + 4 * i < XP) dma16(rX, sB + buf * (XP * 1024) + (wave + 4 * i) * 1024, hv[i], c * 64);
    };
    auto issue_w1 = [&](int ks, int slot) __attribute__((always_inline)) {    // rows x channels [32 ks, 32 ks + 32) of conv1
#pragma unroll
        for (int i = 0; i < W12PW; ++i)
            if (wave + 4 * i < G::WP12)
                dma16(rW1, sW + slot * G::WSLOT + (wave + 4 * i) * 1024, (((wave + 4 * i) * 16 + wrow) * C + wch * 8) * 2, ks * 64);
    };
    auto issue_w2 = [&](int c, int t, int slot) __attribute__((always_inline)) {   // k = t*MID + 32c of conv2
#pragma unroll
        for (int i = 0; i < W12PW; ++i)
            if (wave + 4 * i < G::WP12)
                dma16(rW2, sW + slot * G::WSLOT + (wave + 4 * i) * 1024, (((wave + 4 * i) * 16 + wrow) * (9 * MID) + wch * 8) * 2,
                      (t * MID + c * 32) * 2);
    };
    auto issue_w3 = [&](int nc, int slot) __attribute__((always_inline)) {    // rows [64 nc, 64 nc + 64) of conv3, all of K: [k-step][64 rows][64 B]
#pragma unroll
        for (int i = 0; i < W3PW; ++i) {
            const int p = wave + 4 * i, kk = p >> 2, rp = p & 3;
            dma16(rW3, sW + slot * G::WSLOT + p * 1024, ((rp * 16 + wrow) * MID + wch * 8) * 2, (nc * 64 * MID + kk * 32) * 2);
        }
    };

    issue_x(0, 0);
    issue_w1(0, 0);

    // ---- stage-1 addresses: n-tile it of this wave = linear halo pixels 16*(wave + 4*it) + frow
    const int wa = lds_off(frow, fch);
    int xb[NT1W], t1w[NT1W];
    unsigned inside = 0;
#pragma unroll
    for (int it = 0; it < NT1W; ++it) {
        const int hp = (wave + 4 * it) * 16 + frow;
        const int hpc = hp < HP ? hp : HP - 1;
        const int hy = hpc / HW_, hx = hpc - hy * HW_;
        xb[it] = xh_off<HW_>(hy, hx, fch);
        // this lane's 4 channels (16i + 4 fch ..) of halo pixel hp: chunk 2i + (fch>>1), byte (fch&1)*8; i added later as i*32 after the XOR
        t1w[it] = hp < HP ? hpc * PXB : -1;
        const int gy = ty0 + hy - 1, gx = tx0 + hx - 1;
        if (hp < HP && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W) inside |= 1u << it;
    }
    int t1key[NT1W];
#pragma unroll
    for (int it = 0; it < NT1W; ++it) {
        const int hp = (wave + 4 * it) * 16 + frow;
        const int hpc = hp < HP ? hp : HP - 1;
        t1key[it] = (hpc - (hpc / HW_) * HW_) & (G::NCH - 1);
    }

    // ================= stage 1: conv1 on the halo =================
    {
        f32x4 acc[MT][NT1W];
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const float4 b4 = *reinterpret_cast<const float4*>(a.b1 + i * 16 + fch * 4);
#pragma unroll
            for (int it = 0; it < NT1W; ++it) acc[i][it] = (f32x4){b4.x, b4.y, b4.z, b4.w};
        }
        constexpr int NK1 = C / 32;
#pragma unroll 1
        for (int ks = 0; ks < NK1; ++ks) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();                                 // chunk ks and its weights are in LDS; the other buffers are free
            if (ks + 1 < NK1) { issue_x(ks + 1, (ks + 1) & 1); issue_w1(ks + 1, (ks + 1) & 1); }
            const unsigned char* sX = sB + (ks & 1) * (XP * 1024);
            const unsigned char* sWc = sW + (ks & 1) * G::WSLOT + wa;
            f16x8 af[MT], bf[NT1W];
#pragma unroll
            for (int it = 0; it < NT1W; ++it) bf[it] = *reinterpret_cast<const f16x8*>(sX + xb[it]);
#pragma unroll
            for (int i = 0; i < MT; ++i) af[i] = *reinterpret_cast<const f16x8*>(sWc + i * 1024);
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int it = 0; it < NT1W; ++it)
                    if (wave + 4 * it < G::NT1) acc[i][it] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[i], bf[it], acc[i][it], 0, 0, 0);
        }
        __syncthreads();                                     // every wave is done with the x ring and the conv1 weight slots
        issue_w2(0, 0, 0);                                   // conv2's first slice streams in while T1 is written
        const f16x4 z4 = {(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
#pragma unroll
        for (int it = 0; it < NT1W; ++it) {
            if (t1w[it] < 0) continue;
            const bool in = (inside >> it) & 1u;
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                f16x4 o4 = {(_Float16)acc[i][it][0], (_Float16)acc[i][it][1], (_Float16)acc[i][it][2], (_Float16)acc[i][it][3]};
                o4 = __builtin_elementwise_max(o4, z4);
                if (!in) o4 = z4;                            // conv2 pads with zeros, not with conv1(0)
                const int c8 = 2 * i + (fch >> 1);
                *reinterpret_cast<f16x4*>(sA + t1w[it] + ((c8 ^ t1key[it]) << 4) + (fch & 1) * 8) = o4;
            }
        }
    }

    // ================= stage 2: conv2 out of T1 =================
    // pixel operand: output pixel (row 2*wave*(2/CPT).. , col) -> halo pixel (row + tr, col + tq); n-tile j of this wave
    int hb[NT2W][3];                                         // byte offset of halo pixel (row_j, col_j + tq) in T1 (chunk 0, key applied per chunk)
    int hkey[NT2W][3];
#pragma unroll
    for (int j = 0; j < NT2W; ++j) {
        const int nt = wave * NT2W + j;                      // n-tile index inside the output tile: row nt / CPT, column tile nt % CPT
        const int row = nt / CPT, col0 = (nt % CPT) * 16 + frow;
#pragma unroll
        for (int tq = 0; tq < 3; ++tq) {
            hb[j][tq] = (row * HW_ + col0 + tq) * PXB;
            hkey[j][tq] = (col0 + tq) & (G::NCH - 1);
        }
    }
    f32x4 acc2[MT][NT2W];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const float4 b4 = *reinterpret_cast<const float4*>(a.b2 + i * 16 + fch * 4);
#pragma unroll
        for (int j = 0; j < NT2W; ++j) acc2[i][j] = (f32x4){b4.x, b4.y, b4.z, b4.w};
    }
    {
        constexpr int NCK = MID / 32, NS2 = NCK * 9;
#pragma unroll
        for (int c = 0; c < NCK; ++c) {
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int s = c * 9 + t;
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();                             // slice s landed (and, at s = 0, T1 is complete); slot (s+1)&1 is free
                if (s + 1 < NS2) issue_w2((s + 1) / 9, (s + 1) % 9, (s + 1) & 1);
                const unsigned char* sWc = sW + (s & 1) * G::WSLOT + wa;
                f16x8 af[MT], bf[NT2W];
#pragma unroll
                for (int j = 0; j < NT2W; ++j)
                    bf[j] = *reinterpret_cast<const f16x8*>(sA + (t / 3) * ROWB + hb[j][t % 3] + (((4 * c + fch) ^ hkey[j][t % 3]) << 4));
#pragma unroll
                for (int i = 0; i < MT; ++i) af[i] = *reinterpret_cast<const f16x8*>(sWc + i * 1024);
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NT2W; ++j) acc2[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[i], bf[j], acc2[i][j], 0, 0, 0);
            }
        }
    }
    __syncthreads();                                         // T1 and both weight slots are free
    issue_w3(0, 0);
    {   // T2 = relu(conv2) as fp16, [pixel][MID] with the chunk swizzle keyed on the pixel index
        const f16x4 z4 = {(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
#pragma unroll
        for (int j = 0; j < NT2W; ++j) {
            const int px = (wave * NT2W + j) * 16 + frow;
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                f16x4 o4 = {(_Float16)acc2[i][j][0], (_Float16)acc2[i][j][1], (_Float16)acc2[i][j][2], (_Float16)acc2[i][j][3]};
                o4 = __builtin_elementwise_max(o4, z4);
                *reinterpret_cast<f16x4*>(sB + t_off<PXB>(px, px, 2 * i + (fch >> 1)) + (fch & 1) * 8) = o4;
            }
        }
    }

    // ================= stage 3: conv3 in 64-channel chunks + identity + ReLU =================
    constexpr int TS = 64 + 8;                               // staging tile row stride (halfs)
    _Float16* const tile = reinterpret_cast<_Float16*>(sA);
    constexpr int RPI = 256 / 8;                             // pixels covered per pass of the row-wise phase (8 lanes = 128 B per pixel)
    constexpr int NIT = TPX / RPI;                           // passes (8 or 4)
    const int cc = tid & 7;                                  // this lane's 16-byte chunk of a 64-channel row
    long gpix[NIT];                                          // global pixel index of this lane's rows (-1: outside the image)
#pragma unroll
    for (int itp = 0; itp < NIT; ++itp) {
        const int prow = (tid >> 3) + itp * RPI;
        const int oy = ty0 + prow / TW, ox = tx0 + prow % TW;
        gpix[itp] = (oy < H && ox < W) ? (long)img_base + (long)oy * W + ox : -1;
    }
    int tb[NT2W];
#pragma unroll
    for (int j = 0; j < NT2W; ++j) tb[j] = (wave * NT2W + j) * 16 + frow;       // pixel of this lane's B column
    constexpr int NC3 = C / 64;
#pragma unroll 1
    for (int nc = 0; nc < NC3; ++nc) {
        // identity of this chunk: requested before the barrier, consumed after the MFMAs
        uint4 idv[NIT];
#pragma unroll
        for (int itp = 0; itp < NIT; ++itp) {
            idv[itp] = make_uint4(0, 0, 0, 0);
            if (gpix[itp] >= 0) idv[itp] = *reinterpret_cast<const uint4*>(a.x + gpix[itp] * a.x_ld + nc * 64 + cc * 8);
        }
        f32x4 acc3[4][NT2W];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float4 b4 = *reinterpret_cast<const float4*>(a.b3 + nc * 64 + i * 16 + fch * 4);
#pragma unroll
            for (int j = 0; j < NT2W; ++j) acc3[i][j] = (f32x4){b4.x, b4.y, b4.z, b4.w};
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                     // weight chunk nc landed; T2 complete; the staging tile of chunk nc-1 is drained
        if (nc + 1 < NC3) issue_w3(nc + 1, (nc + 1) & 1);
        const unsigned char* sWc = sW + (nc & 1) * G::WSLOT + wa;
#pragma unroll
        for (int kk = 0; kk < MID / 32; ++kk) {
            f16x8 af[4], bf[NT2W];
#pragma unroll
            for (int j = 0; j < NT2W; ++j) bf[j] = *reinterpret_cast<const f16x8*>(sB + t_off<PXB>(tb[j], tb[j], 4 * kk + fch));
#pragma unroll
            for (int i = 0; i < 4; ++i) af[i] = *reinterpret_cast<const f16x8*>(sWc + kk * 4096 + i * 1024);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < NT2W; ++j) acc3[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[i], bf[j], acc3[i][j], 0, 0, 0);
        }
        // identity rows -> staging tile (16 B per lane), then every lane adds its own 4 channels in fp32, ReLU, fp16
#pragma unroll
        for (int itp = 0; itp < NIT; ++itp)
            *reinterpret_cast<uint4*>(tile + ((tid >> 3) + itp * RPI) * TS + cc * 8) = idv[itp];
        __syncthreads();
#pragma unroll
        for (int j = 0; j < NT2W; ++j) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                _Float16* tp = tile + tb[j] * TS + i * 16 + fch * 4;
                const f16x4 r4 = *reinterpret_cast<const f16x4*>(tp);
                f16x4 o4;
#pragma unroll
                for (int e = 0; e < 4; ++e) o4[e] = (_Float16)fmaxf(acc3[i][j][e] + (float)r4[e], 0.f);
                *reinterpret_cast<f16x4*>(tp) = o4;
            }
        }
        __syncthreads();
#pragma unroll
        for (int itp = 0; itp < NIT; ++itp)
            if (gpix[itp] >= 0)
                *reinterpret_cast<uint4*>(a.y + gpix[itp] * a.y_ld + nc * 64 + cc * 8) =
                    *reinterpret_cast<const uint4*>(tile + ((tid >> 3) + itp * RPI) * TS + cc * 8);
    }
}

}  // namespace

extern "C" int llv_bottleneck_fwd(const llv_bottleneck_desc* d, void* stream) {
    if (!d || !d->x || !d->y || !d->w1 || !d->w2 || !d->w3 || !d->b1 || !d->b2 || !d->b3) return LLV_E_BADARG;
    if (d->mid != 64 && d->mid != 128) return LLV_E_UNSUPPORTED;
    const int C = 4 * d->mid;
    if (d->batch < 1 || d->h < 1 || d->w < 1 || d->x_ld < C || d->y_ld < C || (d->x_ld & 7) || (d->y_ld & 7)) return LLV_E_BADARG;
    const size_t xb = (size_t)d->batch * d->h * d->w * d->x_ld * 2;
    if (xb >= (1ull << 31)) return LLV_E_UNSUPPORTED;        // 32-bit DMA offsets
    BnkArgs a;
    a.x = (const _Float16*)d->x; a.y = (_Float16*)d->y;
    a.w1 = (const _Float16*)d->w1; a.w2 = (const _Float16*)d->w2; a.w3 = (const _Float16*)d->w3;
    a.b1 = d->b1; a.b2 = d->b2; a.b3 = d->b3;
    a.batch = d->batch; a.H = d->h; a.W = d->w; a.x_ld = d->x_ld; a.y_ld = d->y_ld;
    const int tw = d->mid == 64 ? 32 : 16;
    a.tiles_x = (d->w + tw - 1) / tw; a.tiles_y = (d->h + TH - 1) / TH;
    a.nblk = a.tiles_x * a.tiles_y * d->batch;
    a.x_bytes = (unsigned)xb;
    a.w1_bytes = (unsigned)((size_t)d->mid * C * 2);
    a.w2_bytes = (unsigned)((size_t)d->mid * 9 * d->mid * 2);
    a.w3_bytes = (unsigned)((size_t)C * d->mid * 2);
    if (d->mid == 64) hipLaunchKernelGGL((bottleneck_kernel<64, 32>), dim3(a.nblk), dim3(256), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL((bottleneck_kernel<128, 16>), dim3(a.nblk), dim3(256), 0, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? LLV_OK : LLV_E_LAUNCH;
}
