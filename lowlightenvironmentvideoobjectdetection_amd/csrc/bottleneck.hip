// bottleneck.hip -- a whole stride-1 ResNet Bottleneck (resnet.py:260-299: conv1 1x1 -> BN -> ReLU -> conv2 3x3 -> BN -> ReLU ->
// conv3 1x1 -> BN -> + identity -> ReLU, BatchNorm folded) in ONE launch for the wide, shallow stages (layer1: 256 <-> 64,
// layer2: 512 <-> 128), where the three separate launches are bound by memory traffic: they write and re-read the two narrow
// intermediates and read the wide input twice (conv1 and the identity). Here a workgroup owns an 8 x TW pixel tile:
//
//   stage 1  conv1 on the (8+2) x (TW+2) halo: K = C in 32-channel units; each unit's halo pixels and the MID x 32 weight slice
//            arrive in LDS by DMA through a ring that spans the whole LDS allocation (NB1 - 1 units in flight); the tile's own pixels
//            of every unit are lifted into registers on the way (the identity of stage 3: x is read from memory exactly once);
//            result (+bias, ReLU, zero outside the image = conv2's padding) -> T1 in LDS
//   stage 2  conv2 exactly as conv3x3_tap_kernel runs it (chunk-major, tap-minor), its pixel operand being T1, which never leaves
//            LDS; the weight slices stream through a ring whose first slots are requested while stage 1 still runs (into unit
//            slots stage 1 has retired); result (+bias, ReLU) -> T2 in LDS
//   stage 3  conv3 in 64-channel chunks of the output: K = MID out of T2, weight chunks requested during stage 2; + identity (from
//            the registers) + ReLU through the LDS staging tile -> 16-byte stores of whole 128-byte rows
//
// Every loop is fully unrolled and every wave issues the same number of vector-memory instructions at every point, so each wait is a
// counted s_waitcnt vmcnt(N) on exactly the DMA the next step consumes and the barriers do not drain the queue (lds_barrier()).
// fp16 roundings happen at the same three points as in the separate launches and the accumulation order is the same, so the result
// is bit-identical to them (tests/test_gpu_bottleneck.py). HBM traffic per block: x once + y once.
//
// Layouts: (MID, TW) = (64, 16): 76 KB LDS, two workgroups per CU -- while one streams its halo in, the other computes or stores, which
// is what keeps both directions of the CU's memory path busy; (64, 32): 148 KB, one per CU (LLV_BNK_SMALL=0; slower: its read, compute
// and write phases run back to back); (128, 16): 157 KB, one per CU (the weights, 544 KB per tile, rule out smaller tiles).
// Measured on MI355X, 16 x 184 x 320 x 256: 292 us against 135 + 91 + 226 = 452 us for the three launches (in-kernel stamps per
// workgroup: stage 1 13.7k, stage 2 7.8k, stage 3 13.3k cycles); 16 x 92 x 160 x 512: 281 us against 300 us.

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
    unsigned x_bytes, y_bytes, w1_bytes, w2_bytes, w3_bytes;
    unsigned long long* stamps;      // diagnostic build only: [nblk][4] s_memrealtime at start / after stage 1 / after stage 2 / end
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

// s_waitcnt vmcnt(n) with n known only after the fully unrolled loops are constant-folded (the switch folds to one case)
#define LLV_WV(n) case n: asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory"); break;
__device__ __forceinline__ void wait_vm(int n) {
    switch (n < 0 ? 0 : n) {
        LLV_WV(0) LLV_WV(1) LLV_WV(2) LLV_WV(3) LLV_WV(4) LLV_WV(5) LLV_WV(6) LLV_WV(7) LLV_WV(8) LLV_WV(9) LLV_WV(10) LLV_WV(11)
        LLV_WV(12) LLV_WV(13) LLV_WV(14) LLV_WV(15) LLV_WV(16) LLV_WV(17) LLV_WV(18) LLV_WV(19) LLV_WV(20) LLV_WV(21) LLV_WV(22)
        LLV_WV(23) LLV_WV(24) LLV_WV(25) LLV_WV(26) LLV_WV(27) LLV_WV(28) LLV_WV(29) LLV_WV(30) LLV_WV(31) LLV_WV(32) LLV_WV(33)
        LLV_WV(34) LLV_WV(35) LLV_WV(36) LLV_WV(37) LLV_WV(38) LLV_WV(39) LLV_WV(40) LLV_WV(41) LLV_WV(42) LLV_WV(43) LLV_WV(44)
        LLV_WV(45) LLV_WV(46) LLV_WV(47) LLV_WV(48) LLV_WV(49) LLV_WV(50) LLV_WV(51) LLV_WV(52) LLV_WV(53) LLV_WV(54) LLV_WV(55)
        LLV_WV(56) LLV_WV(57) LLV_WV(58) LLV_WV(59) LLV_WV(60) LLV_WV(61) LLV_WV(62)
        default: asm volatile("s_waitcnt vmcnt(63)" ::: "memory"); break;
    }
}
#undef LLV_WV
// this wave's LDS writes and reads are complete, then the workgroup barrier. Unlike __syncthreads() it leaves the LDS-DMA queue
// (vmcnt) alone: the prefetched slices keep streaming in across it.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

constexpr int cmax(int a, int b) { return a > b ? a : b; }

template <int MID, int TW>
struct Geo {
    static constexpr int C = 4 * MID;
    static constexpr int HW_ = TW + 2;                       // halo row length
    static constexpr int HP = (TH + 2) * HW_;                // halo pixels
    static constexpr int TPX = TH * TW;                      // output pixels
    static constexpr int XP = (HP * 64 + 1023) / 1024;       // 1-KiB pieces of one 32-channel chunk of the halo
    static constexpr int XPW = (XP + 3) / 4;                 // per wave (the pad pieces load zeros: every wave issues the same count)
    static constexpr int XB = XPW * 4 * 1024;                // bytes of one x-chunk buffer
    static constexpr int NT1 = (HP + 15) / 16;               // stage-1 n-tiles (16 halo pixels each)
    static constexpr int NT1W = (NT1 + 3) / 4;               // per wave
    static constexpr int MT = MID / 16;                      // m-tiles (16 rows) of stages 1 and 2
    static constexpr int NT2W = TPX / 16 / 4;                // stage-2 / 3 n-tiles per wave (4 or 2)
    static constexpr int PXB = MID * 2;                      // bytes per pixel of T1 / T2
    static constexpr int NCH = PXB / 16;                     // 16-byte chunks per pixel
    static constexpr int S12 = MID * 64;                     // a conv1 / conv2 weight slice: MID rows x 32 channels
    static constexpr int W12PW = MID / 64;                   // its 1-KiB pieces per wave
    static constexpr int S3 = 64 * PXB;                      // a conv3 weight chunk: 64 rows x MID, as [k-step][64 rows][64 B]
    static constexpr int W3PW = MID / 32;                    // its pieces per wave
    static constexpr int NK1 = C / 32, NS2 = 9 * MID / 32, NG2 = NS2 / 3, NC3 = C / 64;
    static constexpr bool BIG64 = MID == 64 && TW == 32;     // one workgroup per CU; (64, 16) runs two per CU, (128, 16) one
    static constexpr int RG2 = BIG64 ? 6 : 3;                // stage-2 ring: groups of 3 conv2 slices (BIG64: all of conv2)
    static constexpr int W3R = MID == 64 ? 4 : 6;            // stage-3 ring: conv3 chunks (MID = 64: all of conv3)
    static constexpr int W3EARLY = BIG64 ? 4 : 2;            // chunks requested at the start of stage 2
    static constexpr int WGS = (MID == 64 && TW == 16) ? 2 : 1;   // workgroups per CU
    static constexpr int T1B = HP * PXB, T2B = TPX * PXB, STGB = TPX * 72 * 2;
    // region A: T1 (stages 1-2), then the staging tile (MID = 64) or T2 + staging tile (MID = 128)
    static constexpr int R_A = MID == 64 ? cmax(T1B, STGB) : cmax(T1B, T2B + STGB);
    static constexpr int T2_OFF = MID == 64 ? R_A : 0;       // MID = 64: over the dead conv2 weights
    static constexpr int STG_OFF = MID == 64 ? 0 : T2B;
    // region XW, stage 2: conv2 ring, early conv3 chunks behind it; stage 3: conv3 ring (and T2 for MID = 64)
    static constexpr int XW = R_A;
    static constexpr int W3PRE = XW + 3 * RG2 * S12;
    static constexpr int R_XW = 3 * RG2 * S12 + W3EARLY * S3;
    // stage 1 owns all of it (T1 is written only after the last x chunk is consumed): a ring of NB1 units from offset 0, unit =
    // one 32-channel chunk of the halo + conv1's weight slice for it; NB1 - 1 units are in flight while one is consumed
    static constexpr int UNIT = XB + S12;
    static constexpr int NB1 = (R_A + R_XW) / UNIT;
    static constexpr int BIAS_OFF = R_A + R_XW;              // b1 | b2 | b3 as fp32 (read through LDS: a global load would sit in the DMA queue)
    static constexpr int SMEM = BIAS_OFF + 6 * MID * 4;
    static constexpr int x_off(int k) { return (k % NB1) * UNIT; }
    static constexpr int w1_off(int k) { return (k % NB1) * UNIT + XB; }
    // last stage-1 unit that uses byte `o` of LDS (-1: none)
    static constexpr int last_unit_at(int o) {
        int u = -1;
        for (int k = 0; k < NK1; ++k) if (o < NB1 * UNIT && k % NB1 == o / UNIT) u = k;
        return u;
    }
    // first stage-1 step after whose barrier the ring slot of conv2 slice s (< 3 RG2) is no longer read, under ring rotation rot
    static constexpr int w2_off_rot(int s, int rot) { return XW + ((s + rot) % (3 * RG2)) * S12; }
    static constexpr int w2_avail_rot(int s, int rot) {
        const int lo = w2_off_rot(s, rot), hi = lo + S12 - 1;
        return cmax(cmax(last_unit_at(lo), last_unit_at(hi)), last_unit_at((lo + hi) / 2)) + 1;
    }
    static constexpr int best_rot() {                        // the first slices into the unit slots that retire first
        int best = 0, cost = 1 << 30;
        for (int r = 0; r < 3 * RG2; ++r) {
            int c = 0;
            for (int s2 = 0; s2 < 3 * RG2; ++s2) c += w2_avail_rot(s2, r) * (3 * RG2 - s2);
            if (c < cost) { cost = c; best = r; }
        }
        return best;
    }
    static constexpr int W2ROT = best_rot();
    static constexpr int w2_off(int s) { return w2_off_rot(s, W2ROT); }
    static constexpr int w2_avail(int s) { return w2_avail_rot(s, W2ROT); }
    static constexpr int w3_off(int nc) {
        return BIG64 ? W3PRE + nc * S3
             : MID == 64 ? (nc < 2 ? W3PRE + nc * S3 : XW + T2B + (nc - 2) * S3)       // behind T2, over the dead conv2 ring
                         : ((nc % 6) < 2 ? W3PRE + (nc % 6) * S3 : XW + ((nc % 6) - 2) * S3);
    }
    static_assert(SMEM * WGS <= 160 * 1024, "LDS budget");
    static_assert(BIG64 || MID != 64 || T2B + 2 * S3 <= 3 * RG2 * S12, "T2 and two conv3 chunks over the conv2 ring");
    static_assert(TPX / 16 % 4 == 0 && MID % 64 == 0, "tiles split over 4 waves");
    static_assert(MID == 64 || (W3R - 2) * S3 <= 3 * RG2 * S12, "conv3 ring over the conv2 ring");
    static_assert(T2_OFF != R_A || T2B <= 3 * RG2 * S12, "T2 over the conv2 ring");
};

// x-chunk halo image (64-byte pixels): chunk c8 of pixel (hy, hx) -- the column-keyed swizzle of the tap kernel
template <int HW_>
__device__ __forceinline__ int xh_off(int hy, int hx, int c8) { return (hy * HW_ + hx) * 64 + ((c8 ^ (((hx >> 2) & 1) << 1)) << 4); }
// T1 / T2 images (PXB-byte pixels): chunk c8 of linear pixel p whose swizzle key is `key` (the halo column for T1, the pixel index
// for T2): 16 consecutive keys x one chunk land on 16 distinct 16-byte slots for ds_read_b128's lane groups (128-byte pixels)
template <int PXB>
__device__ __forceinline__ int t_off(int p, int key, int c8) { return p * PXB + ((c8 ^ (key & (PXB / 16 - 1))) << 4); }

template <int MID, int TW, bool PROF>
__global__ __launch_bounds__(256, (Geo<MID, TW>::WGS)) void bottleneck_kernel(const BnkArgs a) {
    using G = Geo<MID, TW>;
    constexpr int C = G::C, HW_ = G::HW_, HP = G::HP, TPX = G::TPX, NT1W = G::NT1W, MT = G::MT, NT2W = G::NT2W, PXB = G::PXB;
    constexpr int XPW = G::XPW, W12PW = G::W12PW, W3PW = G::W3PW, NK1 = G::NK1, NS2 = G::NS2, NG2 = G::NG2, NC3 = G::NC3;
    constexpr int ROWB = HW_ * PXB;                          // bytes per T1 halo row
    constexpr int CPT = TW / 16;                             // n-tiles per output row
    __shared__ __attribute__((aligned(16))) unsigned char smem[G::SMEM];

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
    const __amdgpu_buffer_rsrc_t rY = make_rsrc(a.y, a.y_bytes);
    const __amdgpu_buffer_rsrc_t rW1 = make_rsrc(a.w1, a.w1_bytes);
    const __amdgpu_buffer_rsrc_t rW2 = make_rsrc(a.w2, a.w2_bytes);
    const __amdgpu_buffer_rsrc_t rW3 = make_rsrc(a.w3, a.w3_bytes);

    // Every wave issues the same number of vector-memory instructions at every point (masked lanes and pad pieces go through the
    // buffer range check instead of a branch), so `vm` -- that count, a compile-time constant once the loops are unrolled -- turns
    // "wait until the DMA recorded at mark m has landed" into s_waitcnt vmcnt(vm - m).
    auto stamp = [&](int i) __attribute__((always_inline)) {
        if (PROF && tid == 0) a.stamps[(size_t)blockIdx.x * 4 + i] = __builtin_readcyclecounter();
    };
    stamp(0);
    int vm = 0;

    {
        float* sb = reinterpret_cast<float*>(smem + G::BIAS_OFF);
        for (int i = tid; i < MID; i += 256) { sb[i] = a.b1[i]; sb[MID + i] = a.b2[i]; }
        for (int i = tid; i < C; i += 256) sb[2 * MID + i] = a.b3[i];
    }
    __builtin_amdgcn_sched_barrier(0);
    const float* const sb1 = reinterpret_cast<const float*>(smem + G::BIAS_OFF);
    const float* const sb2 = sb1 + MID;
    const float* const sb3 = sb1 + 2 * MID;

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
    auto issue_unit1 = [&](int k) __attribute__((always_inline)) {     // channels [32k, 32k + 32): the halo's pixels and conv1's rows
#pragma unroll
        for (int i = 0; i < W12PW; ++i)
            dma16(rW1, smem + G::w1_off(k) + (wave + 4 * i) * 1024, (((wave + 4 * i) * 16 + wrow) * C + wch * 8) * 2, k * 64);
#pragma unroll
        for (int i = 0; i < XPW; ++i) dma16(rX, smem + G::x_off(k) + (wave + 4 * i) * 1024, hv[i], k * 64);
        vm += W12PW + XPW;
    };
    auto issue_w2 = [&](int s) __attribute__((always_inline)) {        // slice s = 9 c + t: k = t*MID + 32c of conv2
        const int c = s / 9, t = s % 9;
#pragma unroll
        for (int i = 0; i < W12PW; ++i)
            dma16(rW2, smem + G::w2_off(s) + (wave + 4 * i) * 1024, (((wave + 4 * i) * 16 + wrow) * (9 * MID) + wch * 8) * 2,
                  (t * MID + c * 32) * 2);
        vm += W12PW;
    };
    auto issue_w3 = [&](int nc) __attribute__((always_inline)) {       // rows [64 nc, 64 nc + 64) of conv3, all of K
#pragma unroll
        for (int i = 0; i < W3PW; ++i) {
            const int p = wave + 4 * i, kk = p >> 2, rp = p & 3;
            dma16(rW3, smem + G::w3_off(nc) + p * 1024, ((rp * 16 + wrow) * MID + wch * 8) * 2, (nc * 64 * MID + kk * 32) * 2);
        }
        vm += W3PW;
    };

    int m1[NK1];
#pragma unroll
    for (int k = 0; k < G::NB1 - 1 && k < NK1; ++k) { issue_unit1(k); m1[k] = vm; }

    // ---- stage-1 addresses: n-tile it of this wave = linear halo pixels 16*(wave + 4*it) + frow
    const int wa = lds_off(frow, fch);
    int xb[NT1W], t1w[NT1W], t1key[NT1W];
    unsigned inside = 0;
#pragma unroll
    for (int it = 0; it < NT1W; ++it) {
        const int hp = (wave + 4 * it) * 16 + frow;
        const int hpc = hp < HP ? hp : HP - 1;
        const int hy = hpc / HW_, hx = hpc - hy * HW_;
        xb[it] = xh_off<HW_>(hy, hx, fch);
        t1w[it] = hp < HP ? hpc * PXB : -1;
        t1key[it] = hx & (G::NCH - 1);
        const int gy = ty0 + hy - 1, gx = tx0 + hx - 1;
        if (hp < HP && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W) inside |= 1u << it;
    }

    // identity stash: 16-byte piece (tid & 3) of pixels (tid >> 2) + 64 i of every 32-channel chunk, lifted out of the x buffers
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    constexpr int IDN = TPX / 64;
    u32x4 idk[NK1][IDN];
    int idsrc[IDN];
#pragma unroll
    for (int i = 0; i < IDN; ++i) {
        const int p = (tid >> 2) + 64 * i;
        idsrc[i] = xh_off<HW_>(p / TW + 1, p % TW + 1, tid & 3);
    }

    // ================= stage 1: conv1 on the halo =================
    const bool last_tile = wave + 4 * (NT1W - 1) < G::NT1;
    int m2[NS2], m3[NC3];
    {
        f32x4 acc[MT][NT1W];
#pragma unroll
        for (int ks = 0; ks < NK1; ++ks) {
            wait_vm(vm - m1[ks]);
            lds_barrier();                                   // unit ks is in LDS; the buffers of unit ks - 1 are free
            if (ks + G::NB1 - 1 < NK1) { issue_unit1(ks + G::NB1 - 1); m1[ks + G::NB1 - 1] = vm; }
#pragma unroll
            for (int s = 0; s < 3 * G::RG2; ++s)             // conv2 slices whose ring slot lies in buffers stage 1 has retired
                if (G::w2_avail(s) == ks) { issue_w2(s); m2[s] = vm; }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < IDN; ++i)                    // the tile's own pixels of this chunk: the identity of stage 3
                idk[ks][i] = *reinterpret_cast<const u32x4*>(smem + G::x_off(ks) + idsrc[i]);
            if (ks == 0) {                                   // the biases are in LDS behind the first barrier
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    const f32x4 b4 = *reinterpret_cast<const f32x4*>(sb1 + i * 16 + fch * 4);
#pragma unroll
                    for (int it = 0; it < NT1W; ++it) acc[i][it] = b4;
                }
            }
            const unsigned char* sX = smem + G::x_off(ks);
            const unsigned char* sWc = smem + G::w1_off(ks) + wa;
            f16x8 af[MT], bf[NT1W];
#pragma unroll
            for (int it = 0; it < NT1W; ++it) bf[it] = *reinterpret_cast<const f16x8*>(sX + xb[it]);
#pragma unroll
            for (int i = 0; i < MT; ++i) af[i] = *reinterpret_cast<const f16x8*>(sWc + i * 1024);
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int it = 0; it < NT1W - 1; ++it) acc[i][it] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[i], bf[it], acc[i][it], 0, 0, 0);
            if (last_tile) {                                 // the ragged last n-tile: only the first NT1 % 4 waves have one
#pragma unroll
                for (int i = 0; i < MT; ++i)
                    acc[i][NT1W - 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[i], bf[NT1W - 1], acc[i][NT1W - 1], 0, 0, 0);
            }
        }
        lds_barrier();                                       // every wave is done with the stage-1 rings
        stamp(1);
        // conv2's first RG2 groups and the early conv3 chunks stream in while T1 is written
#pragma unroll
        for (int s = 0; s < 3 * G::RG2; ++s)
            if (G::w2_avail(s) >= NK1) { issue_w2(s); m2[s] = vm; }
#pragma unroll
        for (int nc = 0; nc < G::W3EARLY; ++nc) { issue_w3(nc); m3[nc] = vm; }
        __builtin_amdgcn_sched_barrier(0);
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
                *reinterpret_cast<f16x4*>(smem + t1w[it] + ((c8 ^ t1key[it]) << 4) + (fch & 1) * 8) = o4;
            }
        }
    }

    // ================= stage 2: conv2 out of T1 =================
    int hb[NT2W][3], hkey[NT2W][3];                          // halo pixel (row_j, col_j + tq) of this lane's column in n-tile j
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
        const f32x4 b4 = *reinterpret_cast<const f32x4*>(sb2 + i * 16 + fch * 4);
#pragma unroll
        for (int j = 0; j < NT2W; ++j) acc2[i][j] = b4;
    }
#pragma unroll
    for (int g = 0; g < NG2; ++g) {
        wait_vm(vm - cmax(cmax(m2[3 * g], m2[3 * g + 1]), m2[3 * g + 2]));
        lds_barrier();                                       // group g landed (at g = 0: T1 is complete); the slots of group g - 1 are free
        if (g >= 1 && g - 1 + G::RG2 < NG2) {
#pragma unroll
            for (int s = 3 * (g - 1 + G::RG2); s < 3 * (g + G::RG2); ++s) { issue_w2(s); m2[s] = vm; }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 3 * g; s < 3 * g + 3; ++s) {
            const int c = s / 9, t = s % 9;
            const unsigned char* sWc = smem + G::w2_off(s) + wa;
            f16x8 af[MT], bf[NT2W];
#pragma unroll
            for (int j = 0; j < NT2W; ++j)
                bf[j] = *reinterpret_cast<const f16x8*>(smem + (t / 3) * ROWB + hb[j][t % 3] + (((4 * c + fch) ^ hkey[j][t % 3]) << 4));
#pragma unroll
            for (int i = 0; i < MT; ++i) af[i] = *reinterpret_cast<const f16x8*>(sWc + i * 1024);
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT2W; ++j) acc2[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[i], bf[j], acc2[i][j], 0, 0, 0);
        }
    }
    lds_barrier();                                           // T1 and the conv2 ring are free
    stamp(2);
#pragma unroll
    for (int nc = G::W3EARLY; nc < G::W3R && nc < NC3; ++nc) { issue_w3(nc); m3[nc] = vm; }

    // ================= stage 3: conv3 in 64-channel chunks + identity + ReLU =================
    constexpr int TS = 64 + 8;                               // staging tile row stride (halfs)
    _Float16* const tile = reinterpret_cast<_Float16*>(smem + G::STG_OFF);
    unsigned char* const sT2 = smem + G::T2_OFF;
    constexpr int RPI = 256 / 8;                             // pixels covered per pass of the row-wise phase (8 lanes = 128 B per pixel)
    constexpr int NIT = TPX / RPI;                           // passes (8 or 4)
    const int cc = tid & 7;                                  // this lane's 16-byte chunk of a 64-channel row
    int gy_off[NIT];                                         // byte offsets of this lane's row pieces in y (out of the image: masked)
#pragma unroll
    for (int itp = 0; itp < NIT; ++itp) {
        const int prow = (tid >> 3) + itp * RPI;
        const int oy = ty0 + prow / TW, ox = tx0 + prow % TW;
        gy_off[itp] = (oy < H && ox < W) ? ((img_base + oy * W + ox) * a.y_ld + cc * 8) * 2 : (int)a.y_bytes;
    }
    {   // T2 = relu(conv2) as fp16, [pixel][MID] with the chunk swizzle keyed on the pixel index
        const f16x4 z4 = {(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
#pragma unroll
        for (int j = 0; j < NT2W; ++j) {
            const int px = (wave * NT2W + j) * 16 + frow;
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                f16x4 o4 = {(_Float16)acc2[i][j][0], (_Float16)acc2[i][j][1], (_Float16)acc2[i][j][2], (_Float16)acc2[i][j][3]};
                o4 = __builtin_elementwise_max(o4, z4);
                *reinterpret_cast<f16x4*>(sT2 + t_off<PXB>(px, px, 2 * i + (fch >> 1)) + (fch & 1) * 8) = o4;
            }
        }
    }
    int tb[NT2W];
#pragma unroll
    for (int j = 0; j < NT2W; ++j) tb[j] = (wave * NT2W + j) * 16 + frow;       // pixel of this lane's B column
#pragma unroll
    for (int nc = 0; nc < NC3; ++nc) {
        f32x4 acc3[4][NT2W];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const f32x4 b4 = *reinterpret_cast<const f32x4*>(sb3 + nc * 64 + i * 16 + fch * 4);
#pragma unroll
            for (int j = 0; j < NT2W; ++j) acc3[i][j] = b4;
        }
        wait_vm(vm - m3[nc]);
        lds_barrier();                                       // weight chunk nc landed; T2 complete; the staging tile of chunk nc-1 is drained
        __builtin_amdgcn_sched_barrier(0);
        const unsigned char* sWc = smem + G::w3_off(nc) + wa;
#pragma unroll
        for (int kk = 0; kk < MID / 32; ++kk) {
            f16x8 af[4], bf[NT2W];
#pragma unroll
            for (int j = 0; j < NT2W; ++j) bf[j] = *reinterpret_cast<const f16x8*>(sT2 + t_off<PXB>(tb[j], tb[j], 4 * kk + fch));
#pragma unroll
            for (int i = 0; i < 4; ++i) af[i] = *reinterpret_cast<const f16x8*>(sWc + kk * 4096 + i * 1024);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < NT2W; ++j) acc3[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[i], bf[j], acc3[i][j], 0, 0, 0);
        }
        // identity rows -> staging tile (16 B per lane), then every lane adds its own 4 channels in fp32, ReLU, fp16
#pragma unroll
        for (int hh = 0; hh < 2; ++hh)
#pragma unroll
            for (int i = 0; i < IDN; ++i)
                *reinterpret_cast<u32x4*>(tile + ((tid >> 2) + 64 * i) * TS + hh * 32 + (tid & 3) * 8) = idk[2 * nc + hh][i];
        lds_barrier();                                       // every wave is past the MFMAs of chunk nc: its weight slot is free
        if (nc + G::W3R < NC3) { issue_w3(nc + G::W3R); m3[nc + G::W3R] = vm; }
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
        lds_barrier();
#pragma unroll
        for (int itp = 0; itp < NIT; ++itp)
            __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<const u32x4*>(tile + ((tid >> 3) + itp * RPI) * TS + cc * 8), rY,
                                                   gy_off[itp], nc * 128, 0);
        vm += NIT;
    }
    stamp(3);
}

}  // namespace

static int bottleneck_launch(const llv_bottleneck_desc* d, unsigned long long* stamps, void* stream) {
    if (!d || !d->x || !d->y || !d->w1 || !d->w2 || !d->w3 || !d->b1 || !d->b2 || !d->b3) return LLV_E_BADARG;
    if (d->mid != 64 && d->mid != 128) return LLV_E_UNSUPPORTED;
    const int C = 4 * d->mid;
    if (d->batch < 1 || d->h < 1 || d->w < 1 || d->x_ld < C || d->y_ld < C || (d->x_ld & 7) || (d->y_ld & 7)) return LLV_E_BADARG;
    const size_t xb = (size_t)d->batch * d->h * d->w * d->x_ld * 2;
    const size_t yb = (size_t)d->batch * d->h * d->w * d->y_ld * 2;
    if (xb >= (1ull << 31) || yb >= (1ull << 31)) return LLV_E_UNSUPPORTED;        // 32-bit buffer offsets
    BnkArgs a;
    a.x = (const _Float16*)d->x; a.y = (_Float16*)d->y;
    a.w1 = (const _Float16*)d->w1; a.w2 = (const _Float16*)d->w2; a.w3 = (const _Float16*)d->w3;
    a.b1 = d->b1; a.b2 = d->b2; a.b3 = d->b3;
    a.batch = d->batch; a.H = d->h; a.W = d->w; a.x_ld = d->x_ld; a.y_ld = d->y_ld;
    static const int small64 = [] { const char* e = getenv("LLV_BNK_SMALL"); return e ? atoi(e) : 1; }();
    const int tw = (d->mid == 64 && !small64) ? 32 : 16;
    a.tiles_x = (d->w + tw - 1) / tw; a.tiles_y = (d->h + TH - 1) / TH;
    a.nblk = a.tiles_x * a.tiles_y * d->batch;
    a.x_bytes = (unsigned)xb;
    a.y_bytes = (unsigned)yb;
    a.w1_bytes = (unsigned)((size_t)d->mid * C * 2);
    a.w2_bytes = (unsigned)((size_t)d->mid * 9 * d->mid * 2);
    a.w3_bytes = (unsigned)((size_t)C * d->mid * 2);
    a.stamps = stamps;
    const dim3 grid(a.nblk), block(256);
    hipStream_t st = (hipStream_t)stream;
    if (stamps) {
        if (d->mid == 128) hipLaunchKernelGGL((bottleneck_kernel<128, 16, true>), grid, block, 0, st, a);
        else if (tw == 16) hipLaunchKernelGGL((bottleneck_kernel<64, 16, true>), grid, block, 0, st, a);
        else hipLaunchKernelGGL((bottleneck_kernel<64, 32, true>), grid, block, 0, st, a);
    } else {
        if (d->mid == 128) hipLaunchKernelGGL((bottleneck_kernel<128, 16, false>), grid, block, 0, st, a);
        else if (tw == 16) hipLaunchKernelGGL((bottleneck_kernel<64, 16, false>), grid, block, 0, st, a);
        else hipLaunchKernelGGL((bottleneck_kernel<64, 32, false>), grid, block, 0, st, a);
    }
    return hipGetLastError() == hipSuccess ? LLV_OK : LLV_E_LAUNCH;
}

extern "C" int llv_bottleneck_fwd(const llv_bottleneck_desc* d, void* stream) { return bottleneck_launch(d, nullptr, stream); }

// Diagnostic, not part of include/llv.h: the same launch with per-workgroup clock stamps ([blocks][4] device uint64: start, end of
// stage 1, end of stage 2, end; s_memtime ticks) -- tools/bench_bottleneck.py turns them into a stage breakdown.
extern "C" int llv_dbg_bottleneck_stamps(const llv_bottleneck_desc* d, unsigned long long* stamps, void* stream) {
    return stamps ? bottleneck_launch(d, stamps, stream) : LLV_E_BADARG;
}
