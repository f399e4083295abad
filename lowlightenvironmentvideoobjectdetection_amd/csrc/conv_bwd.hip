// Backward-pass kernels of the convolution engine (gfx950 only).
//
//   llv_conv2d_nhwc_wgrad   dW[g][co][tap][ci] += sum_p dY[p][g*Cout + co] * X[p @ tap][g*Cin + ci]
//   llv_relu_bwd            dx = (dy [+ dy2]) * (y > 0)
//   llv_bias_grad           db[c] += sum_p dy[p][c]
//   llv_upsample_add_bwd    gradient of the FPN top-down "coarse -> nearest upsample -> add" (fpn.py:186-195)
//   llv_sgd_momentum        SGD with momentum and weight decay on fp32 master weights (torch.optim.SGD update rule)
//
// The data gradient needs no kernel of its own: dX of a stride-1 convolution is the forward kernel on the
// spatially flipped, in/out-transposed weights, and dX of a stride-2 convolution is a 2x2 forward convolution over
// dY that produces the four output phases, stored through the PixelShuffle epilogue (ops.pack_dgrad).
//
// Reference: torch autograd of F.conv2d / F.relu / F.interpolate as used by mmdet's ResNet / FPN / RetinaHead
// (resnet.py:260-299, fpn.py:170-221, retina_head.py:93-114) and the optimizer of schedule_1x.py (SGD, momentum 0.9,
// weight decay 1e-4).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <type_traits>
#include "llv.h"

// Knock-out experiments of the weight-gradient kernels (LLV_WGRAD_KNOCK: bit 0 plain stores instead of atomics, bit 1 no stores, bit 2 no
// DMA, bit 3 no MFMA -- wrong results, timing only; tools/wgrad_bench.py) exist only in a build with -DLLV_KNOCKOUTS
// (`make EXTRA=-DLLV_KNOCKOUTS`): the product library ignores the variable.
#ifdef LLV_KNOCKOUTS
#define LLV_KNOCK(a) ((a).knock)
#else
#define LLV_KNOCK(a) 0
#endif

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __fp16 h4 __attribute__((__vector_size__(4 * sizeof(__fp16))));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), (short)0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t r, unsigned char* lds, int voff, int soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds, 16, voff, soff, 0, 0);
}

// Transposing LDS read as inline assembly. Through the builtin the compiler cannot tell the read from the LDS destination of the
// DMA loads and puts `s_waitcnt vmcnt(0)` in front of the first fragment read after every prefetch -- the stages requested for later
// k-steps were waited for at once and a workgroup never had more than one stage in flight (found in round 4 in the ISA of both
// weight-gradient kernels). The counter waits for these reads are therefore written by hand: lds_wait<N>() leaves at most N of the
// wave's LDS reads outstanding and ties the registers it guards, so their first use cannot move in front of it.
__device__ __forceinline__ unsigned lds_addr(const void* p) {
    return (unsigned)(unsigned long)((__attribute__((address_space(3))) const unsigned char*)p);
}
__device__ __forceinline__ h4 tr_read_asm(unsigned addr) {
    h4 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(addr));
    return v;
}
struct Frag2 { h4 lo, hi; };
__device__ __forceinline__ Frag2 frag_read(unsigned lo_addr, unsigned hi_addr) {
    Frag2 f;
    f.lo = tr_read_asm(lo_addr);
    f.hi = tr_read_asm(hi_addr);
    return f;
}
template <int N>
__device__ __forceinline__ void lds_wait(Frag2& f) {
    asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(f.lo), "+v"(f.hi) : "n"(N));
}
template <int N>
__device__ __forceinline__ void lds_wait4(Frag2& a, Frag2& b, Frag2& c, Frag2& d) {
    asm volatile("s_waitcnt lgkmcnt(%8)" : "+v"(a.lo), "+v"(a.hi), "+v"(b.lo), "+v"(b.hi), "+v"(c.lo), "+v"(c.hi), "+v"(d.lo), "+v"(d.hi) : "n"(N));
}
__device__ __forceinline__ f16x8 frag_join(const Frag2& f) {
    f16x8 r;
    __builtin_memcpy(&r, &f.lo, 8);
    __builtin_memcpy(reinterpret_cast<unsigned char*>(&r) + 8, &f.hi, 8);
    return r;
}

// ------------------------------------------------------------------------------------------------
// conv_wgrad_kernel<BCO, BCI>: one workgroup accumulates a BCO x BCI block of dW for ONE filter tap and one group
// over a contiguous range of output pixels (blockIdx.z = pixel split), then adds it to dW with fp32 atomics.
// GEMM view: D[co][ci] = sum_k A[co][k] * B[k][ci] with k = output pixel. Both operands live in memory pixel-major
// (NHWC), i.e. k-major, while the MFMA wants 8 consecutive k per lane: the tiles are copied to LDS as they are
// ([32 pixels][channels], by LDS DMA) and read back through gfx950's transposing ds_read_b64_tr_b16, two reads
// per 16x32 operand fragment. 16-byte chunks are XOR-swizzled by the pixel row so that the 32 lanes of a half-wave
// (8 pixel rows x 4 column groups) hit 32 distinct 8-byte slots.
// ------------------------------------------------------------------------------------------------
struct WgArgs {
    const _Float16* x;
    const _Float16* dy;
    float* dw;
    float* db;                  // optional: db[g*cout + co] += sum_p dy (done by the tap-0 / first-cin-tile workgroups)
    int x_ld, dy_ld;
    int cin, cout, k_pad, w_rows;
    int kh, kw, stride, pad, groups;
    int dil;                   // dilation of the forward conv (>= 1): tap (r, q) reads input (y*stride - pad + r*dil, ...)
    int H, W, Ho, Wo;
    int xoff, yoff;             // pixel offsets of image 0 in x / dy
    int m_total, m_per_split;
    unsigned x_bytes, dy_bytes;
    int tiles_ci, tiles_co, tiles, splits;      // tiles = tiles_co * tiles_ci * taps * groups
    int xq, chunk;              // splits < 8: xq = XCDs per split, chunk = tiles per XCD; splits >= 8: xq = 0
    int knock;                  // experiment knob (LLV_WGRAD_KNOCK): 1 = plain stores instead of atomics (wrong sums, timing only), 2 = no write
};

template <int BC>
__device__ __forceinline__ int wg_swz(int row) {
    if constexpr (BC == 128) return ((row & 3) << 1) | (((row >> 3) & 1) << 3);
    else return (((row >> 1) & 1) << 1) | (((row >> 3) & 1) << 2);
}

template <int BCO, int BCI>
__device__ __forceinline__ void conv_wgrad_body(const WgArgs& a, const int bid) {
    constexpr int TMO = BCO / 32, TNI = BCI / 32;            // MFMA tiles per wave (2 x 2 waves)
    constexpr int RBA = BCO * 2, RBB = BCI * 2;              // LDS row bytes
    constexpr int STAGE = 32 * (RBA + RBB);
    constexpr int PA = BCO / 16, PB = BCI / 16;              // 1-KiB pieces per k-step
    constexpr int PAW = (PA + 3) / 4, PBW = (PB + 3) / 4;
    constexpr int RPA = 1024 / RBA, RPB = 1024 / RBB;        // pixel rows per piece
    // three k-steps in LDS: the loop body is 16 (128 x 128) or 4 (64 x 64) MFMAs per wave, far shorter than a DMA round trip, and a
    // single-image launch has ~8-30 k-steps per workgroup -- with two buffers every step exposed most of that latency (round 3:
    // counted waits, step ks + 2 requested while ks computes)
    constexpr int NST = 3;
    __shared__ __attribute__((aligned(16))) unsigned char smem[NST * STAGE];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave & 1, wc = wave >> 1;
    // One-dimensional grid, XCD-aware (workgroups go round-robin over the 8 XCDs in dispatch order, each XCD has its own L2): every
    // workgroup of a pixel split reads the same slab of dY and X, so a split's tiles stay on ONE XCD (splits >= 8: XCD x takes the
    // splits x, x + 8, ..) or on 8 / splits of them (xq > 0). With the split as the slowest grid dimension every XCD pulled every
    // slab through the fabric: 8x the compulsory traffic on launches that have nothing else to wait for.
    const int xcd = bid & 7, idx = bid >> 3;
    int split, tl;
    if (a.xq == 0) { split = xcd + 8 * (idx / a.tiles); tl = idx % a.tiles; }
    else { split = xcd / a.xq; tl = (xcd % a.xq) * a.chunk + idx; }
    if (split >= a.splits || tl >= a.tiles) return;
    const int tile_ci = tl % a.tiles_ci, t2 = tl / a.tiles_ci;
    const int tile_co = t2 % a.tiles_co, ty = t2 / a.tiles_co;
    const int ntap = a.kh * a.kw;
    const int tap = ty % ntap, g = ty / ntap;
    const int tr = tap / a.kw, tq = tap - tr * a.kw;
    const int m0 = split * a.m_per_split;
    const int m1 = m0 + a.m_per_split < a.m_total ? m0 + a.m_per_split : a.m_total;
    const int nrows = m1 - m0;
    if (nrows <= 0) return;
    const int nk = (nrows + 31) >> 5;
    const int co0 = tile_co * BCO, ci0 = tile_ci * BCI;

    const __amdgpu_buffer_rsrc_t rY = make_rsrc(a.dy, a.dy_bytes);
    const __amdgpu_buffer_rsrc_t rX = make_rsrc(a.x, a.x_bytes);

    // ---- DMA slots. dY: row m0 + 32*ks + row, a pure scalar step per k-step. X: per-lane pixel walk.
    int yv[PAW], yrow[PAW];
#pragma unroll
    for (int i = 0; i < PAW; ++i) {
        const int pc = wave + 4 * i;
        const int row = pc * RPA + lane / (BCO / 8);
        const int ch = (lane % (BCO / 8)) ^ wg_swz<BCO>(row);
        const int co = co0 + ch * 8;
        yrow[i] = row;
        yv[i] = (pc < PA && co < a.cout) ? ((a.yoff + m0 + row) * a.dy_ld + g * a.cout + co) * 2 : (int)a.dy_bytes;
    }
    int xn[PBW], xoy[PBW], xox[PBW], xrow[PBW], xc[PBW];
#pragma unroll
    for (int i = 0; i < PBW; ++i) {
        const int pc = wave + 4 * i;
        const int row = pc * RPB + lane / (BCI / 8);
        const int ch = (lane % (BCI / 8)) ^ wg_swz<BCI>(row);
        const int ci = ci0 + ch * 8;
        xrow[i] = row;
        xc[i] = (pc < PB && ci < a.cin) ? (g * a.cin + ci) * 2 : -1;
        const int m = m0 + row;
        const int ox = m % a.Wo, t = m / a.Wo;
        xox[i] = ox; xoy[i] = t % a.Ho; xn[i] = t / a.Ho;
    }
    auto issue = [&](int ks, int buf) __attribute__((always_inline)) {
        unsigned char* sA = smem + buf * STAGE;
        unsigned char* sB = sA + 32 * RBA;
        const int left = nrows - ks * 32;                    // rows of this k-step that exist
#pragma unroll
        for (int i = 0; i < PAW; ++i) {
            if (wave + 4 * i < PA) {
                const int v = yrow[i] < left ? yv[i] : (int)a.dy_bytes;
                dma16(rY, sA + (wave + 4 * i) * 1024, v, ks * 32 * a.dy_ld * 2);
            }
        }
#pragma unroll
        for (int i = 0; i < PBW; ++i) {
            if (wave + 4 * i < PB) {
                const int iy = xoy[i] * a.stride - a.pad + tr * a.dil, ix = xox[i] * a.stride - a.pad + tq * a.dil;
                int v = (int)a.x_bytes;
                if (xrow[i] < left && xc[i] >= 0 && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W)
                    v = (a.xoff + (xn[i] * a.H + iy) * a.W + ix) * a.x_ld * 2 + xc[i];
                dma16(rX, sB + (wave + 4 * i) * 1024, v, 0);
                xox[i] += 32;
                while (xox[i] >= a.Wo) { xox[i] -= a.Wo; ++xoy[i]; }
                while (xoy[i] >= a.Ho) { xoy[i] -= a.Ho; ++xn[i]; }
            }
        }
    };

    // ---- transposed fragment reads: lane 4q+p of 16-lane group gq supplies row 8*gq + q, columns c0 + 4p ..
    const int gq = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
    const int frow = gq * 8 + q;
    int fa[TMO], fb[TNI];
#pragma unroll
    for (int i = 0; i < TMO; ++i) {
        const int col = wr * (BCO / 2) + i * 16 + 4 * p;
        fa[i] = frow * RBA + ((((col >> 3) ^ wg_swz<BCO>(frow))) << 4) + ((col >> 2) & 1) * 8;
    }
#pragma unroll
    for (int j = 0; j < TNI; ++j) {
        const int col = wc * (BCI / 2) + j * 16 + 4 * p;
        fb[j] = frow * RBB + ((((col >> 3) ^ wg_swz<BCI>(frow))) << 4) + ((col >> 2) & 1) * 8;
    }

    f32x4 acc[TMO][TNI];
#pragma unroll
    for (int i = 0; i < TMO; ++i)
#pragma unroll
        for (int j = 0; j < TNI; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // bias gradient = row sums of the dY^T operand = dY^T x ones: one extra MFMA per m-tile in the workgroups that own it
    const bool do_bias = a.db != nullptr && tap == 0 && tile_ci == 0 && wc == 0;      // wave-uniform
    f32x4 accb[TMO];
#pragma unroll
    for (int i = 0; i < TMO; ++i) accb[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const _Float16 one = (_Float16)1.f;
    const f16x8 ones = {one, one, one, one, one, one, one, one};

    // every wave issues the same number of DMA instructions per step when PA and PB are multiples of 4 (both tile shapes: 8 + 8 or
    // 4 + 4 pieces), so "at most one step's worth outstanding" is a compile-time count
    constexpr int PER_STEP = PAW + PBW;
    static_assert(PA % 4 == 0 && PB % 4 == 0, "equal DMA counts per wave");
    issue(0, 0);
    if (nk > 1) issue(1, 1);
    int cur = 0;
    for (int ks = 0; ks < nk; ++ks) {
        if (ks + 1 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER_STEP) : "memory");      // step ks has landed, ks + 1 may be in flight
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // ... for every wave, and everyone is done reading buffer (ks - 1) % 3
                                                             // (__syncthreads() would drain the DMA queue: it emits vmcnt(0))
        if (ks + 2 < nk) issue(ks + 2, cur >= 1 ? cur - 1 : NST - 1);      // (ks + 2) % 3 == (ks - 1) % 3
        const unsigned sA = lds_addr(smem) + cur * STAGE;
        const unsigned sB = sA + 32 * RBA;
        Frag2 ra[TMO], rb[TNI];
#pragma unroll
        for (int i = 0; i < TMO; ++i) ra[i] = frag_read(sA + fa[i], sA + fa[i] + 4 * RBA);
#pragma unroll
        for (int j = 0; j < TNI; ++j) rb[j] = frag_read(sB + fb[j], sB + fb[j] + 4 * RBB);
        f16x8 af[TMO], bf[TNI];
#pragma unroll
        for (int i = 0; i < TMO; ++i) { lds_wait<0>(ra[i]); af[i] = frag_join(ra[i]); }
#pragma unroll
        for (int j = 0; j < TNI; ++j) { lds_wait<0>(rb[j]); bf[j] = frag_join(rb[j]); }
#pragma unroll
        for (int i = 0; i < TMO; ++i)
#pragma unroll
            for (int j = 0; j < TNI; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[i], bf[j], acc[i][j], 0, 0, 0);
        if (do_bias) {
#pragma unroll
            for (int i = 0; i < TMO; ++i) accb[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[i], ones, accb[i], 0, 0, 0);
        }
        cur = cur + 1 == NST ? 0 : cur + 1;
    }
    if (do_bias && (lane & 15) == 0) {        // every column of D holds the same row sum: take column 0
#pragma unroll
        for (int i = 0; i < TMO; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int co = co0 + wr * (BCO / 2) + i * 16 + 4 * (lane >> 4) + e;
                if (co < a.cout) atomicAdd(a.db + g * a.cout + co, accb[i][e]);
            }
    }

    // D[m = co][n = ci]: lane holds rows 4*(lane>>4) + e, column lane&15
    float* dwg = a.dw + (long)g * a.w_rows * a.k_pad;
#pragma unroll
    for (int i = 0; i < TMO; ++i) {
#pragma unroll
        for (int j = 0; j < TNI; ++j) {
            const int ci = ci0 + wc * (BCI / 2) + j * 16 + (lane & 15);
            if (ci >= a.cin) continue;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int co = co0 + wr * (BCO / 2) + i * 16 + 4 * (lane >> 4) + e;
                if (co < a.cout) {
                    if (LLV_KNOCK(a) == 0) atomicAdd(dwg + (long)co * a.k_pad + tap * a.cin + ci, acc[i][j][e]);
                    else if (LLV_KNOCK(a) == 1) dwg[(long)co * a.k_pad + tap * a.cin + ci] = acc[i][j][e];
                    else if (acc[i][j][e] == 123.456f) dwg[0] = 1.f;
                }
            }
        }
    }
}

// Job tables: the weight gradients of several convolutions as ONE launch (llv_conv2d_nhwc_wgrad_jobs). A single-image training step
// has ~80 of them and most cover a fraction of the chip for 15-30 us each, a third of which is launch ramp and the atomics of pixel
// splits that only exist to fill the chip; side by side they need fewer splits and one ramp. wg2job[] names the job of every
// workgroup; its arguments are read from the table and made wave-uniform (readfirstlane) before the body runs.
template <class A>
__device__ __forceinline__ A load_job_args(const llv_wgrad_job* __restrict__ job) {
    static_assert(sizeof(A) % 4 == 0 && sizeof(A) <= sizeof(job->args), "job arguments fit the table entry");
    A a;
    const int* src = reinterpret_cast<const int*>(job->args);
    int* dst = reinterpret_cast<int*>(&a);
#pragma unroll
    for (int i = 0; i < (int)(sizeof(A) / 4); ++i) dst[i] = __builtin_amdgcn_readfirstlane(src[i]);
    return a;
}

template <int BCO, int BCI>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const WgArgs a) { conv_wgrad_body<BCO, BCI>(a, blockIdx.x); }

template <int BCO, int BCI>
__global__ __launch_bounds__(256) void conv_wgrad_jobs_kernel(const llv_wgrad_job* __restrict__ jobs, const unsigned short* __restrict__ wg2job) {
    const llv_wgrad_job* job = jobs + __builtin_amdgcn_readfirstlane((int)wg2job[blockIdx.x]);
    const int begin = __builtin_amdgcn_readfirstlane(job->wg_begin);
    const WgArgs a = load_job_args<WgArgs>(job);
    conv_wgrad_body<BCO, BCI>(a, (int)blockIdx.x - begin);
}

// tiles, splits -> the one-dimensional XCD-aware grid; returns its size
template <class A>
long set_grid(A& a, long splits) {
    a.splits = (int)splits;
    if (splits >= 8) { a.xq = 0; a.chunk = 0; return 8L * a.tiles * ((splits + 7) / 8); }
    a.xq = 8 / (int)splits; a.chunk = (a.tiles + a.xq - 1) / a.xq;
    return 8L * a.chunk;
}

// ------------------------------------------------------------------------------------------------
// conv_wgrad3x3_kernel: the weight gradient of a 3x3 / stride 1 / pad 1 convolution with all NINE taps in one workgroup.
// conv_wgrad_kernel above gives every tap its own workgroup: a 64 x 64 tile moves 8 KB from L2 to LDS for 16 MFMAs (32 FLOP/B) and
// the launches sit at what the L2s deliver (1.08 GB in 88 us = 12 TB/s on the 92 x 160 head towers, 395 TFLOP/s). Here a workgroup
// owns dW[64 co][9 taps][64 ci]; the pixel loop walks 8 x 8 patches of output pixels, per patch one dY tile (64 px x 64 co, 8 KB) and
// one X tile with its one-pixel halo (10 x 10 px x 64 ci, 12.8 KB) serve 2 x 9 x 16 MFMAs: 72 B per MFMA instead of 512. Wave w holds
// the 16 input channels 16w.. of all nine taps and all 64 output channels (9 x 4 accumulator tiles = 144 registers): the dY fragments
// are read once per 32 pixels and reused by the nine taps, the X fragment of a tap is the same LDS tile at the tap's pixel shift
// (the transposing read takes a row pointer per lane, so a shifted patch costs an instruction offset, not an address computation).
// Both tiles keep memory's pixel-major rows (128 B per pixel); the 32-byte unit of a pixel is XOR-ed with (bit 1 of the column,
// bit 0 of the row), which makes the 8 pixel rows x 32 B of a half-wave's transposing read cover all 64 banks once for every tap.
// ------------------------------------------------------------------------------------------------
struct Wg3Args {
    const _Float16* x;
    const _Float16* dy;
    float* dw;
    float* db;
    int x_ld, dy_ld;
    int cin, cout, k_pad, w_rows, groups, batch;
    // segments (pyramid levels that share the filter): patches of all of them form one list that the pixel splits cut
    int nseg;
    int H[LLV_MAX_SEG], W[LLV_MAX_SEG];                 // = Ho, Wo
    int xoff[LLV_MAX_SEG], yoff[LLV_MAX_SEG];           // pixel offset of image 0
    int py_n[LLV_MAX_SEG], px_n[LLV_MAX_SEG];           // patches per image column / row
    int pstart[LLV_MAX_SEG + 1];                        // first patch of the segment
    int patches, per_split;
    unsigned x_bytes, dy_bytes, dwg_bytes;              // dwg_bytes: cout rows of one group's dW (rows beyond are dropped by the buffer range)
    int tiles_ci, tiles_co, tiles, splits, xq, chunk;   // as WgArgs
    int knock;
};

__device__ __forceinline__ void conv_wgrad3x3_body(const Wg3Args& a, const int bid) {
    constexpr int STAGE = 24 * 1024;          // 8 pieces of dY + 16 of X (13 used: 100 halo pixels), 1 KiB each: six per wave
    constexpr int NST = 3;
    constexpr int HW = 10;                    // halo tile width
    __shared__ __attribute__((aligned(16))) unsigned char smem[NST * STAGE];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int xcd = bid & 7, idx = bid >> 3;
    int split, tl;
    if (a.xq == 0) { split = xcd + 8 * (idx / a.tiles); tl = idx % a.tiles; }
    else { split = xcd / a.xq; tl = (xcd % a.xq) * a.chunk + idx; }
    if (split >= a.splits || tl >= a.tiles) return;
    const int tile_ci = tl % a.tiles_ci, t2 = tl / a.tiles_ci;
    const int tile_co = t2 % a.tiles_co, g = t2 / a.tiles_co;
    const int co0 = tile_co * 64, ci0 = tile_ci * 64;
    const int p0 = split * a.per_split;
    const int p1 = p0 + a.per_split < a.patches ? p0 + a.per_split : a.patches;
    const int np = p1 - p0;
    if (np <= 0) return;

    const __amdgpu_buffer_rsrc_t rY = make_rsrc(a.dy, a.dy_bytes);
    const __amdgpu_buffer_rsrc_t rX = make_rsrc(a.x, a.x_bytes);

    // ---- DMA slots of this wave: dY pieces wave, wave + 4 (patch rows), X pieces wave + 4 i (8 halo pixels each)
    constexpr int NOCH = (int)0x80000000;
    const int slot = lane >> 3, c16 = lane & 7;
    int ych[2], ypy[2];                       // channel byte offset inside a pixel (NOCH: outside), patch row
    const int ypx = slot;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int py = wave + 4 * i;
        const int f = ((ypx >> 1) & 1) | ((py & 1) << 1);
        const int co = co0 + ((c16 ^ (f << 1)) << 3);
        ypy[i] = py;
        ych[i] = co < a.cout ? (g * a.cout + co) * 2 : NOCH;
    }
    int xch[4], xhy[4], xhx[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int hp = (wave + 4 * i) * 8 + slot;
        const int hy = hp / HW, hx = hp - hy * HW;
        const int f = ((hx >> 1) & 1) | ((hy & 1) << 1);
        const int ci = ci0 + ((c16 ^ (f << 1)) << 3);
        xhy[i] = hy - 1; xhx[i] = hx - 1;
        xch[i] = (hp < HW * HW && ci < a.cin) ? (g * a.cin + ci) * 2 : NOCH;
    }
    // ---- patch walk of the issue side: (segment, image, patch row, patch column); the segment's geometry in scalars
    int iseg = 0, in_, ity, itx;
    int sH, sW, sxo, syo, spy, spx;
    // the segment table is read with a run-time index when the walk enters the next level: from LDS (indexing the argument block
    // itself made the compiler spill it to scratch memory)
    __shared__ int segtab[6][LLV_MAX_SEG + 3];
    if (tid == 0) {
#pragma unroll
        for (int k = 0; k < LLV_MAX_SEG; ++k) {
            segtab[0][k] = a.H[k]; segtab[1][k] = a.W[k]; segtab[2][k] = a.xoff[k];
            segtab[3][k] = a.yoff[k]; segtab[4][k] = a.py_n[k]; segtab[5][k] = a.px_n[k];
        }
    }
    __syncthreads();
    auto load_seg = [&]() __attribute__((always_inline)) {
        const int k = iseg < LLV_MAX_SEG ? iseg : LLV_MAX_SEG - 1;          // (one step past the last patch is never issued)
        sH = __builtin_amdgcn_readfirstlane(segtab[0][k]); sW = __builtin_amdgcn_readfirstlane(segtab[1][k]);
        sxo = __builtin_amdgcn_readfirstlane(segtab[2][k]); syo = __builtin_amdgcn_readfirstlane(segtab[3][k]);
        spy = __builtin_amdgcn_readfirstlane(segtab[4][k]); spx = __builtin_amdgcn_readfirstlane(segtab[5][k]);
    };
    {
        int pst = 0;
#pragma unroll
        for (int k = 1; k < LLV_MAX_SEG; ++k)
            if (k < a.nseg && p0 >= a.pstart[k]) { iseg = k; pst = a.pstart[k]; }
        load_seg();
        const int per_img = spy * spx;
        const int rel = p0 - pst;
        in_ = rel / per_img;
        const int r = rel - in_ * per_img;
        ity = r / spx; itx = r - ity * spx;
    }
    auto issue = [&](int buf) __attribute__((always_inline)) {
        unsigned char* sY = smem + buf * STAGE;
        unsigned char* sX = sY + 8 * 1024;
        const int y0 = ity * 8, x0 = itx * 8;
        if (!(LLV_KNOCK(a) & 4)) {
            const int pix0 = (in_ * sH + y0) * sW + x0;
            const int by = (syo + pix0) * a.dy_ld * 2, bx = (sxo + pix0) * a.x_ld * 2;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const bool ok = ych[i] != NOCH && y0 + ypy[i] < sH && x0 + ypx < sW;
                dma16(rY, sY + (wave + 4 * i) * 1024, ok ? by + (ypy[i] * sW + ypx) * a.dy_ld * 2 + ych[i] : (int)a.dy_bytes, 0);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const bool ok = xch[i] != NOCH && (unsigned)(y0 + xhy[i]) < (unsigned)sH && (unsigned)(x0 + xhx[i]) < (unsigned)sW;
                dma16(rX, sX + (wave + 4 * i) * 1024, ok ? bx + (xhy[i] * sW + xhx[i]) * a.x_ld * 2 + xch[i] : (int)a.x_bytes, 0);
            }
        }
        if (++itx == spx) {
            itx = 0;
            if (++ity == spy) {
                ity = 0;
                if (++in_ == a.batch) { in_ = 0; ++iseg; load_seg(); }
            }
        }
    };

    // ---- transposed fragment reads. k = 8 gq + (4 h + q): pixel (row gq, column 4 h + q) of a 4 x 8 half patch
    const int gq = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
    int fa[4][2], fb[3][2][2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int col = 4 * h + q;
        const int f = ((col >> 1) & 1) | ((gq & 1) << 1);
#pragma unroll
        for (int i = 0; i < 4; ++i) fa[i][h] = (gq * 8 + col) * 128 + ((i ^ f) << 5) + p * 8;
    }
#pragma unroll
    for (int s3 = 0; s3 < 3; ++s3)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int hx = 4 * h + q + s3;
            const int fx = (hx >> 1) & 1;
#pragma unroll
            for (int rp = 0; rp < 2; ++rp) {
                const int fy = (gq + rp) & 1;
                fb[s3][h][rp] = 8 * 1024 + (gq * HW + hx) * 128 + ((wave ^ fx ^ (fy << 1)) << 5) + p * 8;
            }
        }

    f32x4 acc[9][4];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[t][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const bool do_bias = a.db != nullptr && tile_ci == 0 && wave == 0;
    f32x4 accb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) accb[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const _Float16 one = (_Float16)1.f;
    const f16x8 ones = {one, one, one, one, one, one, one, one};

    constexpr int PER_STEP = 6;
    issue(0);
    if (np > 1) issue(1);
    int cur = 0;
    // two copies of the pixel loop (with / without the bias row sums) instead of a wave-uniform branch inside it: the branch cut the
    // unrolled body into basic blocks the scheduler cannot move LDS reads across
    auto pixel_loop = [&](auto with_bias) __attribute__((always_inline)) {
    for (int ks = 0; ks < np; ++ks) {
        if (ks + 1 < np) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER_STEP) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (ks + 2 < np) issue(cur >= 1 ? cur - 1 : NST - 1);
        const unsigned st = lds_addr(smem) + cur * STAGE;
        if (LLV_KNOCK(a) & 8) { cur = cur + 1 == NST ? 0 : cur + 1; continue; }
        // the X fragment of tap t + 1 is requested before the MFMAs of tap t are issued: lds_wait<2> = everything but those two reads
        auto bfrag = [&](int t18) __attribute__((always_inline)) {
            const int sub = t18 / 9, t = t18 - sub * 9, r = t / 3, s3 = t - r * 3;
            const unsigned b = st + (4 * sub + r) * (HW * 128);
            return frag_read(b + fb[s3][0][r & 1], b + fb[s3][1][r & 1]);
        };
        Frag2 rb = bfrag(0);
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            Frag2 ra[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) ra[i] = frag_read(st + sub * 4096 + fa[i][0], st + sub * 4096 + fa[i][1]);
            f16x8 af[4];
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                Frag2 rn = rb;
                const bool more = sub * 9 + t + 1 < 18;
                if (more) rn = bfrag(sub * 9 + t + 1);
                if (t == 0) {
                    if (more) lds_wait4<2>(ra[0], ra[1], ra[2], ra[3]); else lds_wait4<0>(ra[0], ra[1], ra[2], ra[3]);
#pragma unroll
                    for (int i = 0; i < 4; ++i) af[i] = frag_join(ra[i]);
                    if constexpr (decltype(with_bias)::value) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) accb[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[i], ones, accb[i], 0, 0, 0);
                    }
                }
                if (more) lds_wait<2>(rb); else lds_wait<0>(rb);
                const f16x8 bf = frag_join(rb);
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[t][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[i], bf, acc[t][i], 0, 0, 0);
                rb = rn;
            }
        }
        cur = cur + 1 == NST ? 0 : cur + 1;
    }
    };
    if (do_bias) pixel_loop(std::true_type{}); else pixel_loop(std::false_type{});
    if (do_bias && (lane & 15) == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int co = co0 + i * 16 + 4 * (lane >> 4) + e;
                if (co < a.cout) atomicAdd(a.db + g * a.cout + co, accb[i][e]);
            }
    }
    // D[m = co][n = ci]: lane holds rows 4 * (lane >> 4) + e, column lane & 15. 576 values per lane: buffer atomics with one lane offset
    // and a scalar offset per value; rows >= cout fall outside the buffer range and are dropped by the hardware (one compare-and-branch
    // plus 64-bit address arithmetic per value made this epilogue 12 us of a 70 us launch)
    if (LLV_KNOCK(a) & 3) return;
    const __amdgpu_buffer_rsrc_t rW = make_rsrc(a.dw + (long)g * a.w_rows * a.k_pad, a.dwg_bytes);
    const int ci = ci0 + wave * 16 + (lane & 15);
    const int lane_off = ci < a.cin ? ((co0 + 4 * (lane >> 4)) * a.k_pad + ci) * 4 : 0x7ffffff0;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int row_off = (i * 16 + e) * a.k_pad * 4;
#pragma unroll
            for (int t = 0; t < 9; ++t)
                __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(acc[t][i][e], rW, lane_off, row_off + t * a.cin * 4, 0);
        }
}

__global__ __launch_bounds__(256, 2) void conv_wgrad3x3_kernel(const Wg3Args a) { conv_wgrad3x3_body(a, blockIdx.x); }

__global__ __launch_bounds__(256, 2) void conv_wgrad3x3_jobs_kernel(const llv_wgrad_job* __restrict__ jobs, const unsigned short* __restrict__ wg2job) {
    const llv_wgrad_job* job = jobs + __builtin_amdgcn_readfirstlane((int)wg2job[blockIdx.x]);
    const int begin = __builtin_amdgcn_readfirstlane(job->wg_begin);
    const Wg3Args a = load_job_args<Wg3Args>(job);
    conv_wgrad3x3_body(a, (int)blockIdx.x - begin);
}

// ------------------------------------------------------------------------------------------------
__global__ void relu_bwd_kernel(const _Float16* dy, const _Float16* dy2, const _Float16* y, _Float16* dx, long nvec) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < nvec; i += (long)gridDim.x * blockDim.x) {
        f16x8 g = reinterpret_cast<const f16x8*>(dy)[i];
        if (dy2) {
            const f16x8 g2 = reinterpret_cast<const f16x8*>(dy2)[i];
#pragma unroll
            for (int e = 0; e < 8; ++e) g[e] = (_Float16)((float)g[e] + (float)g2[e]);
        }
        if (y) {
            const f16x8 o = reinterpret_cast<const f16x8*>(y)[i];
#pragma unroll
            for (int e = 0; e < 8; ++e) g[e] = (float)o[e] > 0.f ? g[e] : (_Float16)0.f;
        }
        reinterpret_cast<f16x8*>(dx)[i] = g;
    }
}

// the block output of a Bottleneck, y = relu(v + identity): dy *= (y > 0) in place and gres += dy, one pass instead of two
__global__ void relu_res_bwd_kernel(_Float16* dy, const _Float16* y, _Float16* gres, long nvec) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < nvec; i += (long)gridDim.x * blockDim.x) {
        f16x8 g = reinterpret_cast<const f16x8*>(dy)[i];
        const f16x8 o = reinterpret_cast<const f16x8*>(y)[i];
        f16x8 r = reinterpret_cast<const f16x8*>(gres)[i];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            g[e] = (float)o[e] > 0.f ? g[e] : (_Float16)0.f;
            r[e] = (_Float16)((float)r[e] + (float)g[e]);
        }
        reinterpret_cast<f16x8*>(dy)[i] = g;
        reinterpret_cast<f16x8*>(gres)[i] = r;
    }
}

// column sums of a [m][c] fp16 matrix (row stride ld) into fp32: one block = 64 rows x 256 channels-of-8... simple form:
// blockIdx.x = channel group of 64, blockIdx.y = row split; lanes = channels, rows strided; fp32 atomics at the end.
__global__ __launch_bounds__(256) void bias_grad_kernel(const _Float16* dy, long m, int c, int ld, float* out, long rows_per) {
    const int ch = blockIdx.x * 64 + (threadIdx.x & 63);
    const int sub = threadIdx.x >> 6;
    const long r0 = blockIdx.y * rows_per, r1 = r0 + rows_per < m ? r0 + rows_per : m;
    float s = 0.f;
    if (ch < c)
        for (long r = r0 + sub; r < r1; r += 4) s += (float)dy[r * ld + ch];
    __shared__ float part[256];
    part[threadIdx.x] = s;
    __syncthreads();
    if (sub == 0 && ch < c) atomicAdd(out + ch, part[threadIdx.x] + part[threadIdx.x + 64] + part[threadIdx.x + 128] + part[threadIdx.x + 192]);
}

// d(coarse) += sum of the fine-grid gradients whose nearest-neighbour source is that coarse pixel
// (F.interpolate(mode='nearest', size=(H,W)): src = min(floor(dst * h/H), h-1))
__global__ void upsample_add_bwd_kernel(const _Float16* dfine, _Float16* dcoarse, int n, int H, int W, int h, int w, int c8, int accumulate) {
    const long total = (long)n * h * w * c8;
    const float sy = (float)h / (float)H, sx = (float)w / (float)W;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int cv = (int)(i % c8);
        long t = i / c8;
        const int cx = (int)(t % w); t /= w;
        const int cy = (int)(t % h);
        const int im = (int)(t / h);
        float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        // candidate fine rows/cols: those y with min(floor(y*sy), h-1) == cy
        int y0 = (int)floorf((float)cy / sy) - 1, y1 = (int)ceilf((float)(cy + 1) / sy) + 1;
        int x0 = (int)floorf((float)cx / sx) - 1, x1 = (int)ceilf((float)(cx + 1) / sx) + 1;
        y0 = y0 < 0 ? 0 : y0; x0 = x0 < 0 ? 0 : x0; y1 = y1 > H - 1 ? H - 1 : y1; x1 = x1 > W - 1 ? W - 1 : x1;
        for (int y = y0; y <= y1; ++y) {
            int ry = (int)floorf((float)y * sy); ry = ry < h - 1 ? ry : h - 1;
            if (ry != cy) continue;
            for (int x = x0; x <= x1; ++x) {
                int rx = (int)floorf((float)x * sx); rx = rx < w - 1 ? rx : w - 1;
                if (rx != cx) continue;
                const f16x8 g = reinterpret_cast<const f16x8*>(dfine)[(((long)im * H + y) * W + x) * c8 + cv];
#pragma unroll
                for (int e = 0; e < 8; ++e) s[e] += (float)g[e];
            }
        }
        f16x8 o;
        if (accumulate) {
            const f16x8 old = reinterpret_cast<const f16x8*>(dcoarse)[i];
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (_Float16)((float)old[e] + s[e]);
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (_Float16)s[e];
        }
        reinterpret_cast<f16x8*>(dcoarse)[i] = o;
    }
}

__global__ void sgd_momentum_kernel(float* w, const float* g, float* mom, long n, float lr, float momentum, float wd, float gscale, int first) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float d = g[i] * gscale + wd * w[i];
        const float b = first ? d : momentum * mom[i] + d;   // torch.optim.SGD: buf = d on the first step
        mom[i] = b;
        w[i] -= lr * b;
    }
}

// ------------------------------------------------------------------------------------------------
// Weight (re)packing on the device, once per optimizer step: module parameters (fp32, OIHW) -> the fp16 layouts the
// conv kernels read, with the eval-mode BatchNorm folded in (s = gamma / sqrt(var + eps), ops.pack_conv's folding).
//   mode 0  forward      dst[row0 + o][(r*kw + q)*cpad + i]                    = w[o][i][r][q] * s[o]
//   mode 1  dgrad, s = 1 dst[i][((kh-1-r)*kw + (kw-1-q))*cpad + row0 + o]      = w[o][i][r][q] * s[o]
//   mode 2  dgrad, s = 2 dst[4i + 2py + px][(dy*KW + dx)*cpad + row0 + o]      = w[o][i][r][q] * s[o]
//           with r = py + pad - 2*(dy - padp), q likewise (backward.dgrad_geometry); taps that do not exist are 0
// One thread per destination element of the part's block; cpad = channel pitch of the destination's K axis.
// ------------------------------------------------------------------------------------------------
struct PackArgs {
    const float* w; const float* gamma; const float* var; float eps;
    _Float16* dst;
    int cout, cin, kh, kw, cpad, k_pad, row0, mode, pad, KH, KW, padp, rows;
};

// (element counts are below 2^31: host check; 32-bit divisions)
__device__ __forceinline__ void pack_one(const PackArgs& a, unsigned idx) {
        if (a.mode == 0) {
            const int i = (int)(idx % a.cpad);
            unsigned t = idx / a.cpad;
            const int q = (int)(t % a.kw); t /= a.kw;
            const int r = (int)(t % a.kh);
            const int o = (int)(t / a.kh);
            float v = 0.f;
            if (i < a.cin) {
                v = a.w[(((long)o * a.cin + i) * a.kh + r) * a.kw + q];
                if (a.gamma) v *= a.gamma[o] * rsqrtf(a.var[o] + a.eps);
            }
            a.dst[(long)(a.row0 + o) * a.k_pad + (r * a.kw + q) * a.cpad + i] = (_Float16)v;
        } else {
            const int o = (int)(idx % a.cout);
            unsigned t = idx / a.cout;
            const int dx = (int)(t % a.KW); t /= a.KW;
            const int dy = (int)(t % a.KH);
            const int row = (int)(t / a.KH);
            int i, r, q;
            if (a.mode == 1) {
                i = row; r = a.kh - 1 - dy; q = a.kw - 1 - dx;
            } else {
                i = row >> 2;
                const int py = (row >> 1) & 1, px = row & 1;
                r = py + a.pad - 2 * (dy - a.padp);
                q = px + a.pad - 2 * (dx - a.padp);
            }
            float v = 0.f;
            if (i < a.cin && r >= 0 && r < a.kh && q >= 0 && q < a.kw) {
                v = a.w[(((long)o * a.cin + i) * a.kh + r) * a.kw + q];
                if (a.gamma) v *= a.gamma[o] * rsqrtf(a.var[o] + a.eps);
            }
            a.dst[(long)row * a.k_pad + (dy * a.KW + dx) * a.cpad + a.row0 + o] = (_Float16)v;
        }
    }

__global__ void pack_weights_kernel(const PackArgs a) {
    const long total = a.mode == 0 ? (long)a.cout * a.kh * a.kw * a.cpad : (long)a.rows * a.KH * a.KW * a.cout;
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) pack_one(a, (unsigned)idx);
}

// folded bias: out[o] = beta[o] + ((b ? b[o] : 0) - mean[o]) * s[o]   (or plain b / 0 without BN)
__global__ void fold_bias_kernel(const float* b, const float* gamma, const float* beta, const float* mean, const float* var, float eps,
                                 float* out, int n) {
    const int o = blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= n) return;
    const float bv = b ? b[o] : 0.f;
    out[o] = gamma ? beta[o] + (bv - mean[o]) * gamma[o] * rsqrtf(var[o] + eps) : bv;
}

// Packed gradient -> module parameter gradients, one workgroup per output channel:
//   gw[o][i][r][q] = dw[row0+o][(r*kw+q)*cpad + i] * s[o]
//   ggamma[o] = (sum_{i,r,q} dw * w) / sigma - mean / sigma * db[o],  gbeta[o] = db[o]       (eval-mode BN, affine trainable)
//   gbias[o]  = db[o] * s[o]
struct UnpackArgs {
    const float* dw; const float* db; const float* w;
    const float* gamma; const float* mean; const float* var; float eps;
    float* gw; float* ggamma; float* gbeta; float* gbias;
    int cin, kh, kw, cpad, k_pad;
};

__device__ __forceinline__ void unpack_one(const UnpackArgs& a, int o) {
    const int n = a.cin * a.kh * a.kw;
    float inv = 1.f, s = 1.f;
    if (a.gamma) { inv = rsqrtf(a.var[o] + a.eps); s = a.gamma[o] * inv; }
    float part = 0.f;
    // one workgroup walks a whole row (up to 4608 values): four independent loads in flight per thread -- with one, a row is a chain of
    // n / 256 exposed memory latencies and the launch waits for its longest rows (245 us per step for 288 MB)
    const int taps = a.kh * a.kw;
    for (int e0 = threadIdx.x; e0 < n; e0 += 1024) {
        float g[4], wv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = e0 + 256 * u;
            g[u] = 0.f; wv[u] = 0.f;
            if (e < n) {
                const int i = e / taps, t = e - i * taps;        // e = (i * kh + r) * kw + q, t = r * kw + q
                g[u] = a.dw[(long)o * a.k_pad + t * a.cpad + i];
                if (a.ggamma) wv[u] = a.w[(long)o * n + e];
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = e0 + 256 * u;
            if (e < n) {
                if (a.gw) a.gw[(long)o * n + e] = g[u] * s;
                part += g[u] * wv[u];
            }
        }
    }
    __shared__ float red[4];
    if (a.ggamma) {                                           // uniform over the block
#pragma unroll
        for (int st = 32; st > 0; st >>= 1) part += __shfl_xor(part, st);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = part;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float d = a.db ? a.db[o] : 0.f;
        if (a.ggamma) a.ggamma[o] = ((red[0] + red[1]) + (red[2] + red[3])) * inv - a.mean[o] * inv * d;
        if (a.gbeta) a.gbeta[o] = d;
        if (a.gbias) a.gbias[o] = d * s;
    }
    __syncthreads();
}

__global__ __launch_bounds__(256) void unpack_wgrad_kernel(const UnpackArgs a) { unpack_one(a, blockIdx.x); }
// Backward of a LeakyReLU epilogue, with the residual of Darknet's ResBlock (y = res + act(v), LLV_CONV_ACT_FIRST) when there is one:
// gres += dy, then dy *= (act(v) > 0 ? 1 : slope) in place, where act(v) = y (- res). Rows of c halfs with their own pitches.
__global__ void act_bwd_kernel(_Float16* __restrict__ dy, const _Float16* __restrict__ y, const _Float16* __restrict__ res,
                               _Float16* __restrict__ gres, long rows, int c8, int dy_ld, int y_ld, int res_ld, int gres_ld, float slope) {
    const long total = rows * c8;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / c8;
        const int cv = (int)(i - r * c8) * 8;
        f16x8 g = *reinterpret_cast<const f16x8*>(dy + r * dy_ld + cv);
        const f16x8 yv = *reinterpret_cast<const f16x8*>(y + r * y_ld + cv);
        f16x8 rv;
#pragma unroll
        for (int e = 0; e < 8; ++e) rv[e] = (_Float16)0.f;
        if (res) rv = *reinterpret_cast<const f16x8*>(res + r * res_ld + cv);
        if (gres) {
            f16x8 gr = *reinterpret_cast<const f16x8*>(gres + r * gres_ld + cv);
#pragma unroll
            for (int e = 0; e < 8; ++e) gr[e] = (_Float16)((float)gr[e] + (float)g[e]);
            *reinterpret_cast<f16x8*>(gres + r * gres_ld + cv) = gr;
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) g[e] = ((float)yv[e] - (float)rv[e]) > 0.f ? g[e] : (_Float16)((float)g[e] * slope);
        *reinterpret_cast<f16x8*>(dy + r * dy_ld + cv) = g;
    }
}

// Backward of llv_upsample2x_nhwc (nearest, x2): gsrc[n][y][x][c] += sum over the four fine pixels of gdst[n][2y+dy][2x+dx][c]
__global__ void upsample2x_bwd_kernel(const _Float16* __restrict__ gdst, _Float16* __restrict__ gsrc, int n, int h, int w, int c8, int gdst_ld,
                                      int gsrc_ld) {
    const long total = (long)n * h * w * c8;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int cv = (int)(i % c8) * 8;
        long p = i / c8;
        const int x = (int)(p % w); p /= w;
        const int y = (int)(p % h);
        const int b = (int)(p / h);
        float acc[8];
        _Float16* d = gsrc + (((long)b * h + y) * w + x) * gsrc_ld + cv;
        const f16x8 cur = *reinterpret_cast<const f16x8*>(d);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = (float)cur[e];
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                const f16x8 v = *reinterpret_cast<const f16x8*>(gdst + (((long)b * 2 * h + 2 * y + dy) * (2 * w) + 2 * x + dx) * gdst_ld + cv);
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] += (float)v[e];
            }
        f16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (_Float16)acc[e];
        *reinterpret_cast<f16x8*>(d) = o;
    }
}

// ---- the same three bookkeeping kernels over a TABLE of jobs in device memory (blockIdx.y = job): a training step re-packs ~120
// parameter tensors and unpacks ~60 gradients, each a few microseconds of work behind a launch of its own; one launch per table
// instead (round 3: 250 launches per step -> 3 + one per all-reduce bucket).
__global__ void pack_weights_batched_kernel(const llv_pack_job* __restrict__ jobs) {
    const llv_pack_job j = jobs[blockIdx.y];
    const PackArgs a{j.w, j.gamma, j.var, j.eps, (_Float16*)j.dst, j.cout, j.cin, j.kh, j.kw, j.cpad, j.k_pad, j.row0, j.mode, j.pad, j.KH, j.KW,
                     j.padp, j.rows};
    const long total = a.mode == 0 ? (long)a.cout * a.kh * a.kw * a.cpad : (long)a.rows * a.KH * a.KW * a.cout;
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) pack_one(a, (unsigned)idx);
}

// The data-gradient filters (modes 1 / 2: input channel i becomes the ROW, output channel o runs along K) are a transpose of the
// parameter tensor: in the element-wise kernel above consecutive threads read addresses cin*kh*kw floats apart (one cache line per
// lane, 2.3 GB of line traffic for the 36 M parameters of the trainable part: ~300 us of every step). Here a workgroup moves a
// 32 (o) x 32 (i) x all-taps tile through LDS: the reads are whole contiguous runs of 32*kh*kw floats per o, the writes 64-byte runs
// of 32 consecutive o per (row, tap). Same values to the same places; positions no source tap maps to keep the zeros they were
// allocated with (the element-wise kernel rewrites those zeros every time).
constexpr int PT = 32, PT_TAPS = 9;
__global__ __launch_bounds__(256) void pack_dgrad_tiled_kernel(const llv_pack_job* __restrict__ jobs) {
    const llv_pack_job j = jobs[blockIdx.y];
    __shared__ _Float16 tile[PT_TAPS][PT][PT + 2];
    const int T = j.kh * j.kw;
    const int tiles_i = (j.cin + PT - 1) / PT, tiles_o = (j.cout + PT - 1) / PT;
    _Float16* const dst = (_Float16*)j.dst;
    for (int tl = blockIdx.x; tl < tiles_i * tiles_o; tl += gridDim.x) {
        const int o0 = (tl / tiles_i) * PT, i0 = (tl % tiles_i) * PT;
        const int ni = j.cin - i0 < PT ? j.cin - i0 : PT, no = j.cout - o0 < PT ? j.cout - o0 : PT;
        for (int t0 = 0; t0 < T; t0 += PT_TAPS) {
            const int nt = T - t0 < PT_TAPS ? T - t0 : PT_TAPS;
            __syncthreads();
            // read: for each o a run of ni * T floats (the taps of ni consecutive input channels); take taps [t0, t0 + nt)
            const int run = ni * T;
            for (int e = threadIdx.x; e < no * run; e += blockDim.x) {
                const int o = e / run, rem = e - o * run;
                const int i = rem / T, t = rem - i * T;
                if (t < t0 || t >= t0 + nt) continue;
                float v = j.w[((long)(o0 + o) * j.cin + i0) * T + rem];
                if (j.gamma) v *= j.gamma[o0 + o] * rsqrtf(j.var[o0 + o] + j.eps);
                tile[t - t0][i][o] = (_Float16)v;
            }
            __syncthreads();
            for (int e = threadIdx.x; e < nt * ni * PT; e += blockDim.x) {
                const int o = e & (PT - 1), ti = e >> 5;
                const int i = ti % ni, t = t0 + ti / ni;
                if (o >= no) continue;
                const int r = t / j.kw, q = t - r * j.kw;
                long at;
                if (j.mode == 1) {
                    const int dy = j.kh - 1 - r, dx = j.kw - 1 - q;
                    at = (long)(i0 + i) * j.k_pad + (dy * j.KW + dx) * j.cpad;
                } else {
                    // r = py + pad - 2 (dy - padp): the phase is the parity of r - pad, the window row follows
                    const int py = (r - j.pad) & 1, px = (q - j.pad) & 1;
                    const int dy = (py + j.pad - r) / 2 + j.padp, dx = (px + j.pad - q) / 2 + j.padp;
                    if ((unsigned)dy >= (unsigned)j.KH || (unsigned)dx >= (unsigned)j.KW) continue;
                    at = (long)(((i0 + i) << 2) + py * 2 + px) * j.k_pad + (dy * j.KW + dx) * j.cpad;
                }
                dst[at + j.row0 + o0 + o] = tile[t - t0][i][o];
            }
        }
    }
}

__global__ void fold_bias_batched_kernel(const llv_fold_job* __restrict__ jobs) {
    const llv_fold_job j = jobs[blockIdx.y];
    for (int o = blockIdx.x * blockDim.x + threadIdx.x; o < j.n; o += gridDim.x * blockDim.x) {
        const float bv = j.b ? j.b[o] : 0.f;
        j.out[o] = j.gamma ? j.beta[o] + (bv - j.mean[o]) * j.gamma[o] * rsqrtf(j.var[o] + j.eps) : bv;
    }
}

__global__ __launch_bounds__(256) void unpack_wgrad_batched_kernel(const llv_unpack_job* __restrict__ jobs) {
    const llv_unpack_job j = jobs[blockIdx.y];
    const UnpackArgs a{j.dw, j.db, j.w, j.gamma, j.mean, j.var, j.eps, j.gw, j.ggamma, j.gbeta, j.gbias, j.cin, j.kh, j.kw, j.cpad, j.k_pad};
    for (int o = blockIdx.x; o < j.cout; o += gridDim.x) unpack_one(a, o);      // workgroup-uniform loop: the barriers inside are safe
}

}  // namespace

static_assert(sizeof(WgArgs) <= LLV_WGRAD_JOB_ARGS && sizeof(Wg3Args) <= LLV_WGRAD_JOB_ARGS, "llv_wgrad_job holds either kernel's arguments");

extern "C" int llv_conv2d_nhwc_wgrad_plan(const llv_conv_desc* d, const void* dy, int dy_ld, float* dw, float* db, float share,
                                          llv_wgrad_job* jobs, int* n_jobs) {
    if (!d || !dy || !dw || !jobs || !n_jobs || !d->x[0] || d->n_seg < 1 || d->n_seg > LLV_MAX_SEG || d->n_src != 1) return LLV_E_BADARG;
    if (d->cin <= 0 || (d->cin & 7) || d->cout <= 0 || d->groups < 1 || d->kh < 1 || d->kw < 1 || d->stride < 1 || d->pad < 0) return LLV_E_BADARG;
    if ((d->x_ld & 7) || (dy_ld & 7) || d->x_ld < d->groups * d->cin || dy_ld < d->groups * d->cout) return LLV_E_BADARG;
    if (d->k_pad < d->kh * d->kw * d->cin || d->w_rows < d->cout || d->batch < 1) return LLV_E_BADARG;
    if (((uintptr_t)d->x[0] | (uintptr_t)dy) & 15) return LLV_E_BADARG;
    if (!(share > 0.f) || share > 1.f) share = 1.f;
    size_t xb = 0, yb = 0;                 // bytes of x / dy the segments reach
    for (int k = 0; k < d->n_seg; ++k) {
        if (d->seg_h[k] < 1 || d->seg_w[k] < 1 || d->seg_ho[k] < 1 || d->seg_wo[k] < 1 || d->seg_xoff[k] < 0 || d->seg_yoff[k] < 0) return LLV_E_BADARG;
        const size_t xk = ((size_t)d->seg_xoff[k] + (size_t)d->batch * d->seg_h[k] * d->seg_w[k]) * d->x_ld * 2;
        const size_t yk = ((size_t)d->seg_yoff[k] + (size_t)d->batch * d->seg_ho[k] * d->seg_wo[k]) * dy_ld * 2;
        xb = xk > xb ? xk : xb; yb = yk > yb ? yk : yb;
    }
    if (xb >= (1ull << 31) || yb >= (1ull << 31)) return LLV_E_UNSUPPORTED;
    static const int knock = getenv("LLV_WGRAD_KNOCK") ? atoi(getenv("LLV_WGRAD_KNOCK")) : 0;
    static const int use3 = getenv("LLV_WGRAD3") ? atoi(getenv("LLV_WGRAD3")) : 1;
    bool all3 = use3 && d->kh == 3 && d->kw == 3 && d->stride == 1 && d->pad == 1 && d->dilation <= 1;
    for (int k = 0; k < d->n_seg && all3; ++k) all3 = d->seg_ho[k] == d->seg_h[k] && d->seg_wo[k] == d->seg_w[k];
    if (all3) {
        Wg3Args b;
        memset(&b, 0, sizeof(b));
        b.x = (const _Float16*)d->x[0]; b.dy = (const _Float16*)dy; b.dw = dw; b.db = db;
        b.x_ld = d->x_ld; b.dy_ld = dy_ld; b.cin = d->cin; b.cout = d->cout; b.k_pad = d->k_pad; b.w_rows = d->w_rows; b.groups = d->groups;
        b.batch = d->batch; b.nseg = d->n_seg;
        int np = 0;
        for (int k = 0; k < LLV_MAX_SEG; ++k) {
            const bool on = k < d->n_seg;
            b.H[k] = on ? d->seg_h[k] : 1; b.W[k] = on ? d->seg_w[k] : 1;
            b.xoff[k] = on ? (int)d->seg_xoff[k] : 0; b.yoff[k] = on ? (int)d->seg_yoff[k] : 0;
            b.py_n[k] = (b.H[k] + 7) / 8; b.px_n[k] = (b.W[k] + 7) / 8;
            b.pstart[k] = np;
            if (on) np += d->batch * b.py_n[k] * b.px_n[k];
        }
        b.pstart[LLV_MAX_SEG] = np;
        b.patches = np;
        b.x_bytes = (unsigned)xb; b.dy_bytes = (unsigned)yb;
        const size_t dwg = (size_t)d->cout * d->k_pad * 4;
        if (dwg >= (1ull << 31)) return LLV_E_UNSUPPORTED;
        b.dwg_bytes = (unsigned)dwg;
        b.tiles_co = (d->cout + 63) / 64; b.tiles_ci = (d->cin + 63) / 64;
        b.tiles = b.tiles_co * b.tiles_ci * d->groups;
        // Pixel splits. A workgroup (one wave per SIMD) needs ~0.75 us per patch and ends in 9 x 64 x 64 fp32 atomics (147 KB; the
        // atomics of a launch retire at ~1.5-2.7 TB/s), more than `share` x 256 workgroups share matrix cores with the rest of the
        // launch: the split count with the lowest estimate among 1, 2, 4 and the multiples of 8 (whole XCDs per split).
        // LLV_WGRAD3_SPLITS forces a count.
        static const long forced = getenv("LLV_WGRAD3_SPLITS") ? atol(getenv("LLV_WGRAD3_SPLITS")) : 0;
        const double dw_mb = (double)d->groups * dwg * 1e-6, room = 256.0 * share;
        long sp = 1;
        double best = 1e30;
        static const long cand[] = {1, 2, 4, 8, 16, 24, 32, 48, 64};
        for (long c : cand) {
            if (c > np) break;
            const double wg = (double)b.tiles * c;
            const double per = (double)((np + c - 1) / c);
            const double t = per * 0.75 * (wg > room ? wg / room : 1.0) + c * dw_mb / 2.0;
            if (t < best) { best = t; sp = c; }
        }
        if (forced > 0) sp = forced < np ? forced : np;
        b.per_split = (int)((np + sp - 1) / sp);
        sp = (np + b.per_split - 1) / b.per_split;
        b.knock = knock;
        const long wgs = set_grid(b, sp);
        jobs[0].kind = 2; jobs[0].wg_begin = 0; jobs[0].n_wg = (int)wgs; jobs[0].reserved = 0;
        memcpy(jobs[0].args, &b, sizeof(b));
        *n_jobs = 1;
        return LLV_OK;
    }
    // every other geometry: one job per segment
    for (int k = 0; k < d->n_seg; ++k) {
        const long m = (long)d->batch * d->seg_ho[k] * d->seg_wo[k];
        if (m > 0x7fffffffL) return LLV_E_UNSUPPORTED;
        WgArgs a;
        memset(&a, 0, sizeof(a));
        a.x = (const _Float16*)d->x[0]; a.dy = (const _Float16*)dy; a.dw = dw; a.db = db;
        a.x_ld = d->x_ld; a.dy_ld = dy_ld; a.cin = d->cin; a.cout = d->cout; a.k_pad = d->k_pad; a.w_rows = d->w_rows;
        a.kh = d->kh; a.kw = d->kw; a.stride = d->stride; a.pad = d->pad; a.groups = d->groups;
        a.dil = d->dilation > 1 ? d->dilation : 1;
        a.H = d->seg_h[k]; a.W = d->seg_w[k]; a.Ho = d->seg_ho[k]; a.Wo = d->seg_wo[k];
        a.xoff = (int)d->seg_xoff[k]; a.yoff = (int)d->seg_yoff[k];
        a.m_total = (int)m; a.x_bytes = (unsigned)xb; a.dy_bytes = (unsigned)yb;
        // 64 x 64 tiles unless both channel counts exceed 512: a workgroup's pixel split ends in fp32 atomics over its whole dW tile, 16 KiB
        // for a 64 x 64 tile against 64 KiB for 128 x 128, and on single-image maps the launches are bound by those atomics, not by the MFMA
        // operand traffic the larger tile saves (same-box sweep on the training step: 128 x 128 tiles 177, 64 x 64 tiles 182 images/s)
        static const long small_max = getenv("LLV_WGRAD_SMALLMAX") ? atol(getenv("LLV_WGRAD_SMALLMAX")) : 512;
        const bool big = d->cout > small_max && d->cin > small_max;
        const int bco = big ? 128 : 64, bci = big ? 128 : 64;
        a.tiles_co = (d->cout + bco - 1) / bco; a.tiles_ci = (d->cin + bci - 1) / bci;
        a.tiles = a.tiles_co * a.tiles_ci * d->kh * d->kw * d->groups;
        // Every pixel split adds a whole dW tile of fp32 atomics: as few as fill this job's share of the chip (1024 workgroups for a
        // launch of its own; LLV_WGRAD_WGS), and at least LLV_WGRAD_MINPX pixels each (512: 768 and more lose again on the small maps)
        static const long target_wgs = getenv("LLV_WGRAD_WGS") && atol(getenv("LLV_WGRAD_WGS")) > 0 ? atol(getenv("LLV_WGRAD_WGS")) : 1024;
        static const long min_px = getenv("LLV_WGRAD_MINPX") && atol(getenv("LLV_WGRAD_MINPX")) > 0 ? atol(getenv("LLV_WGRAD_MINPX")) : 512;
        long want = (long)(target_wgs * (double)share + 0.5);
        if (want < 1) want = 1;
        long splits = (want + a.tiles - 1) / a.tiles;
        const long max_splits = (m + min_px - 1) / min_px;
        if (splits > max_splits) splits = max_splits;
        if (splits < 1) splits = 1;
        // whole XCDs per split: 1, 2, 4 or a multiple of 8 splits (3 or 5..7 would leave XCDs without work)
        if (splits >= 6) { splits = (splits + 7) / 8 * 8; if (splits > max_splits && splits > 8) splits -= 8; }
        else if (splits == 3) splits = max_splits >= 4 ? 4 : 2;
        else if (splits == 5) splits = 4;
        long per = (m + splits - 1) / splits;
        per = (per + 31) & ~31L;
        splits = (m + per - 1) / per;
        a.m_per_split = (int)per;
        a.knock = knock;
        const long wgs = set_grid(a, splits);
        jobs[k].kind = big ? 1 : 0; jobs[k].wg_begin = 0; jobs[k].n_wg = (int)wgs; jobs[k].reserved = 0;
        memcpy(jobs[k].args, &a, sizeof(a));
    }
    *n_jobs = d->n_seg;
    return LLV_OK;
}

extern "C" int llv_conv2d_nhwc_wgrad(const llv_conv_desc* d, const void* dy, int dy_ld, float* dw, float* db, void* stream) {
    llv_wgrad_job jobs[LLV_MAX_SEG];
    int n = 0;
    const int rc = llv_conv2d_nhwc_wgrad_plan(d, dy, dy_ld, dw, db, 1.f, jobs, &n);
    if (rc != LLV_OK) return rc;
    hipStream_t st = (hipStream_t)stream;
    for (int k = 0; k < n; ++k) {
        const dim3 grid((unsigned)jobs[k].n_wg), block(256);
        if (jobs[k].kind == 2) {
            Wg3Args b;
            memcpy(&b, jobs[k].args, sizeof(b));
            hipLaunchKernelGGL(conv_wgrad3x3_kernel, grid, block, 0, st, b);
        } else {
            WgArgs a;
            memcpy(&a, jobs[k].args, sizeof(a));
            if (jobs[k].kind == 1) hipLaunchKernelGGL((conv_wgrad_kernel<128, 128>), grid, block, 0, st, a);
            else hipLaunchKernelGGL((conv_wgrad_kernel<64, 64>), grid, block, 0, st, a);
        }
        if (hipGetLastError() != hipSuccess) return LLV_E_LAUNCH;
    }
    return LLV_OK;
}

extern "C" int llv_conv2d_nhwc_wgrad_jobs(const llv_wgrad_job* jobs, const uint16_t* wg2job, int kind, int total_wgs, void* stream) {
    if (!jobs || !wg2job || kind < 0 || kind > 2 || total_wgs < 1) return LLV_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)total_wgs), block(256);
    if (kind == 2) hipLaunchKernelGGL(conv_wgrad3x3_jobs_kernel, grid, block, 0, st, jobs, (const unsigned short*)wg2job);
    else if (kind == 1) hipLaunchKernelGGL((conv_wgrad_jobs_kernel<128, 128>), grid, block, 0, st, jobs, (const unsigned short*)wg2job);
    else hipLaunchKernelGGL((conv_wgrad_jobs_kernel<64, 64>), grid, block, 0, st, jobs, (const unsigned short*)wg2job);
    return hipGetLastError() == hipSuccess ? LLV_OK : LLV_E_LAUNCH;
}

extern "C" int llv_relu_bwd(const void* dy, const void* dy2, const void* y, void* dx, int64_t n, void* stream) {
    if (!dy || !dx || n < 0 || (n & 7)) return LLV_E_BADARG;
    if (n == 0) return LLV_OK;
    const long nvec = n >> 3;
    long grid = (nvec + 255) / 256;
    grid = grid > 4096 ? 4096 : grid;
    hipLaunchKernelGGL(relu_bwd_kernel, dim3((int)grid), dim3(256), 0, (hipStream_t)stream, (const _Float16*)dy, (const _Float16*)dy2,
                       (const _Float16*)y, (_Float16*)dx, nvec);
    return hipGetLastError() == hipSuccess ? LLV_OK : LLV_E_LAUNCH;
}

extern "C" int llv_relu_res_bwd(void* dy, const void* y, void* gres, int64_t n, void* stream) {
    if (!dy || !y || !gres || n < 0 || (n & 7)) return LLV_E_BADARG;
    if (n == 0) return LLV_OK;
    const long nvec = n >> 3;
    long grid = (nvec + 255) / 256;
    grid = grid > 4096 ? 4096 : grid;
    hipLaunchKernelGGL(relu_res_bwd_kernel, dim3((int)grid), dim3(256), 0, (hipStream_t)stream, (_Float16*)dy, (const _Float16*)y, (_Float16*)gres, nvec);
    return hipGetLastError() == hipSuccess ? LLV_OK : LLV_E_LAUNCH;
}

extern "C" int llv_bias_grad(const void* dy, int64_t m, int c, int ld, float* out, void* stream) {
    if (!dy || !out || m < 0 || c < 1 || ld < c) return LLV_E_BADARG;
    if (m == 0) return LLV_OK;
    long splits = (m + 2047) / 2048;
    splits = splits > 256 ? 256 : splits;
    const long per = (m + splits - 1) / splits;
    dim3 grid((c + 63) / 64, (int)((m + per - 1) / per));
    hipLaunchKernelGGL(bias_grad_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const _Float16*)dy, (long)m, c, ld, out, per);
    return hipGetLastError() == hipSuccess ? LLV_OK : LLV_E_LAUNCH;
}

extern "C" int llv_upsample_add_bwd(const void* dfine, void* dcoarse, int n, int H, int W, int h, int w, int c, int accumulate, void* stream) {
    if (!dfine || !dcoarse || n < 1 || H < 1 || W < 1 || h < 1 || w < 1 || c < 8 || (c & 7)) return LLV_E_BADARG;
    const long total = (long)n * h * w * (c >> 3);
    long grid = (total + 255) / 256;
    grid = grid > 4096 ? 4096 : grid;
    hipLaunchKernelGGL(upsample_add_bwd_kernel, dim3((int)grid), dim3(256), 0, (hipStream_t)stream, (const _Float16*)dfine, (_Float16*)dcoarse,
                       n, H, W, h, w, c >> 3, accumulate);
    return hipGetLastError() == hipSuccess ? LLV_OK : LLV_E_LAUNCH;
}

extern "C" int llv_sgd_momentum(float* w, const float* g, float* mom, int64_t n, float lr, float momentum, float weight_decay,
                                float grad_scale, int first_step, void* stream) {
    if (!w || !g || !mom || n < 0) return LLV_E_BADARG;
    if (n == 0) return LLV_OK;
    long grid = (n + 255) / 256;
    grid = grid > 4096 ? 4096 : grid;
    hipLaunchKernelGGL(sgd_momentum_kernel, dim3((int)grid), dim3(256), 0, (hipStream_t)stream, w, g, mom, (long)n, lr, momentum, weight_decay,
                       grad_scale, first_step);
    return hipGetLastError() == hipSuccess ? LLV_OK : LLV_E_LAUNCH;
}

extern "C" int llv_pack_conv_weights(const float* w, const float* gamma, const float* var, float eps, void* dst, int cout, int cin,
                                     int kh, int kw, int cpad, int k_pad, int row0, int mode, int pad, int KH, int KW, int padp,
                                     int rows, void* stream) {
    if (!w || !dst || cout < 1 || cin < 1 || kh < 1 || kw < 1 || cpad < 1 || k_pad < 1 || row0 < 0 || mode < 0 || mode > 2) return LLV_E_BADARG;
    if ((gamma == nullptr) != (var == nullptr)) return LLV_E_BADARG;
    if (mode == 0 && (cpad < cin || k_pad < kh * kw * cpad)) return LLV_E_BADARG;
    if (mode != 0 && (KH < 1 || KW < 1 || rows < 1 || cpad < row0 + cout || k_pad < KH * KW * cpad)) return LLV_E_BADARG;
    PackArgs a{w, gamma, var, eps, (_Float16*)dst, cout, cin, kh, kw, cpad, k_pad, row0, mode, pad, KH, KW, padp, rows};
    const long total = mode == 0 ? (long)cout * kh * kw * cpad : (long)rows * KH * KW * cout;
    if (total >= (1L << 31)) return LLV_E_UNSUPPORTED;
    long grid = (total + 255) / 256;
    grid = grid > 4096 ? 4096 : grid;
    hipLaunchKernelGGL(pack_weights_kernel, dim3((int)grid), dim3(256), 0, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? LLV_OK : LLV_E_LAUNCH;
}

extern "C" int llv_fold_bias(const float* b, const float* gamma, const float* beta, const float* mean, const float* var, float eps,
                             float* out, int n, void* stream) {
    if (!out || n < 1 || (gamma && (!beta || !mean || !var))) return LLV_E_BADARG;
    hipLaunchKernelGGL(fold_bias_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, b, gamma, beta, mean, var, eps, out, n);
    return hipGetLastError() == hipSuccess ? LLV_OK : LLV_E_LAUNCH;
}

extern "C" int llv_unpack_wgrad(const float* dw, const float* db, const float* w, const float* gamma, const float* mean, const float* var,
                                float eps, float* gw, float* ggamma, float* gbeta, float* gbias, int cout, int cin, int kh, int kw,
                                int cpad, int k_pad, void* stream) {
    if (!dw || cout < 1 || cin < 1 || kh < 1 || kw < 1 || cpad < cin || k_pad < kh * kw * cpad) return LLV_E_BADARG;
    if (gamma && (!mean || !var)) return LLV_E_BADARG;
    if (ggamma && (!gamma || !w)) return LLV_E_BADARG;
    UnpackArgs a{dw, db, w, gamma, mean, var, eps, gw, ggamma, gbeta, gbias, cin, kh, kw, cpad, k_pad};
    hipLaunchKernelGGL(unpack_wgrad_kernel, dim3(cout), dim3(256), 0, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? LLV_OK : LLV_E_LAUNCH;
}

// Data gradient of a convolution = llv_conv2d_nhwc_fwd over dY with the re-packed filter (llv_pack_conv_weights mode 1 / 2).
// This entry point only builds that launch from the FORWARD descriptor: segments swap roles (dY is the input, dX the
// output), a stride-2 convolution becomes the KHxKW stride-1 window + PixelShuffle cropped to the forward input size.
extern "C" int llv_conv2d_nhwc_dgrad(const llv_conv_desc* fwd, const void* w_dgrad, const float* zero_bias, int wd_rows, int wd_k_pad,
                                     int kh, int kw, int pad, const void* dy, int dy_ld, void* dx, int dx_ld, int accumulate,
                                     void* workspace, size_t workspace_bytes, void* stream) {
    if (!fwd || !w_dgrad || !zero_bias || !dy || !dx || fwd->n_src != 1 || fwd->n_seg < 1 || fwd->n_seg > LLV_MAX_SEG) return LLV_E_BADARG;
    if (fwd->stride != 1 && fwd->stride != 2) return LLV_E_UNSUPPORTED;
    const bool s2 = fwd->stride == 2;
    if (s2 && fwd->n_seg != 1) return LLV_E_UNSUPPORTED;
    if (s2 && fwd->dilation > 1) return LLV_E_UNSUPPORTED;        // dilated stride-2 convs do not occur (DC5 dilates stride-1 stages)
    llv_conv_desc d = {};
    d.dilation = fwd->dilation > 1 ? fwd->dilation : 0;          // the flipped filter keeps the forward conv's dilation
    d.x[0] = dy; d.w = w_dgrad; d.bias = zero_bias; d.y = dx;
    // fwd->res (fp16, laid out like dX, res_ld channels per pixel): dX = gradient + res, written over dX instead of accumulated into it
    // (a residual block's first conv: res = the gradient of the block output, whose identity path ends at the same tensor)
    const bool plus = fwd->res != nullptr;
    if (plus && (accumulate || (fwd->res_ld & 3) || fwd->stride != 1)) return LLV_E_BADARG;
    d.res = plus ? fwd->res : (accumulate ? dx : nullptr);
    d.n_src = 1; d.x_ld = dy_ld; d.y_ld = dx_ld; d.res_ld = plus ? fwd->res_ld : (accumulate ? dx_ld : 0);
    d.cin = dy_ld / fwd->groups;                     // channel pitch of one group inside dY
    d.cout = s2 ? 4 * fwd->cin : fwd->cin;
    d.w_rows = wd_rows; d.k_pad = wd_k_pad;
    d.kh = kh; d.kw = kw; d.stride = 1; d.pad = pad; d.groups = fwd->groups; d.batch = fwd->batch;
    d.n_seg = fwd->n_seg;
    for (int s = 0; s < fwd->n_seg; ++s) {
        d.seg_h[s] = fwd->seg_ho[s]; d.seg_w[s] = fwd->seg_wo[s];
        d.seg_ho[s] = s2 ? (fwd->seg_h[s] + 1) / 2 : fwd->seg_h[s];
        d.seg_wo[s] = s2 ? (fwd->seg_w[s] + 1) / 2 : fwd->seg_w[s];
        d.seg_xoff[s] = fwd->seg_yoff[s]; d.seg_yoff[s] = fwd->seg_xoff[s];
    }
    d.flags = (s2 ? LLV_CONV_PIXSHUF : 0) | ((accumulate || plus) ? LLV_CONV_RES_ADD : 0);
    if (fwd->flags & LLV_CONV_MASK_POS) {
        // dX also takes the ReLU mask of the forward INPUT (x[0], x_ld: the activation dX is the gradient of), so that the launch
        // that consumes it needs no mask pass of its own; mask((sum of contributions)) = sum of masked contributions
        if (s2 || !fwd->x[0] || fwd->x_ld != dx_ld) return LLV_E_BADARG;
        d.mask = fwd->x[0]; d.mask_ld = fwd->x_ld; d.flags |= LLV_CONV_MASK_POS;
    }
    if (s2) { d.ps_out_h = fwd->seg_h[0]; d.ps_out_w = fwd->seg_w[0]; }
    d.workspace = workspace; d.workspace_bytes = workspace_bytes;
    return llv_conv2d_nhwc_fwd(&d, stream);
}

extern "C" int llv_pack_conv_weights_batched(const llv_pack_job* jobs, int n_jobs, int blocks_per_job, void* stream) {
    if (!jobs || n_jobs < 1 || n_jobs > 65535 || blocks_per_job < 1) return LLV_E_BADARG;
    hipLaunchKernelGGL(pack_weights_batched_kernel, dim3(blocks_per_job > 4096 ? 4096 : blocks_per_job, n_jobs), dim3(256), 0, (hipStream_t)stream, jobs);
    return hipGetLastError() == hipSuccess ? LLV_OK : LLV_E_LAUNCH;
}

extern "C" int llv_pack_dgrad_weights_batched(const llv_pack_job* jobs, int n_jobs, int blocks_per_job, void* stream) {
    if (!jobs || n_jobs < 1 || n_jobs > 65535 || blocks_per_job < 1) return LLV_E_BADARG;
    hipLaunchKernelGGL(pack_dgrad_tiled_kernel, dim3(blocks_per_job > 1024 ? 1024 : blocks_per_job, n_jobs), dim3(256), 0, (hipStream_t)stream, jobs);
    return hipGetLastError() == hipSuccess ? LLV_OK : LLV_E_LAUNCH;
}

extern "C" int llv_fold_bias_batched(const llv_fold_job* jobs, int n_jobs, int blocks_per_job, void* stream) {
    if (!jobs || n_jobs < 1 || n_jobs > 65535 || blocks_per_job < 1) return LLV_E_BADARG;
    hipLaunchKernelGGL(fold_bias_batched_kernel, dim3(blocks_per_job > 64 ? 64 : blocks_per_job, n_jobs), dim3(256), 0, (hipStream_t)stream, jobs);
    return hipGetLastError() == hipSuccess ? LLV_OK : LLV_E_LAUNCH;
}

extern "C" int llv_unpack_wgrad_batched(const llv_unpack_job* jobs, int n_jobs, int blocks_per_job, void* stream) {
    if (!jobs || n_jobs < 1 || n_jobs > 65535 || blocks_per_job < 1) return LLV_E_BADARG;
    hipLaunchKernelGGL(unpack_wgrad_batched_kernel, dim3(blocks_per_job > 2048 ? 2048 : blocks_per_job, n_jobs), dim3(256), 0, (hipStream_t)stream, jobs);
    return hipGetLastError() == hipSuccess ? LLV_OK : LLV_E_LAUNCH;
}

extern "C" int llv_act_bwd(void* dy, const void* y, const void* res, void* gres, int64_t rows, int c, int dy_ld, int y_ld, int res_ld, int gres_ld,
                           float slope, void* stream) {
    if (!dy || !y || rows < 0 || c < 8 || (c & 7) || dy_ld < c || y_ld < c || ((dy_ld | y_ld) & 7)) return LLV_E_BADARG;
    if ((res && (res_ld < c || (res_ld & 7))) || (gres && (gres_ld < c || (gres_ld & 7)))) return LLV_E_BADARG;
    if (rows == 0) return LLV_OK;
    const long total = rows * (c / 8);
    long grid = (total + 255) / 256;
    grid = grid > 8192 ? 8192 : grid;
    hipLaunchKernelGGL(act_bwd_kernel, dim3((int)grid), dim3(256), 0, (hipStream_t)stream, (_Float16*)dy, (const _Float16*)y, (const _Float16*)res,
                       (_Float16*)gres, (long)rows, c / 8, dy_ld, y_ld, res_ld, gres_ld, slope);
    return hipGetLastError() == hipSuccess ? LLV_OK : LLV_E_LAUNCH;
}

extern "C" int llv_upsample2x_nhwc_bwd(const void* gdst, void* gsrc, int n, int h, int w, int c, int gdst_ld, int gsrc_ld, void* stream) {
    if (!gdst || !gsrc || n < 1 || h < 1 || w < 1 || c < 8 || (c & 7) || gdst_ld < c || gsrc_ld < c || ((gdst_ld | gsrc_ld) & 7)) return LLV_E_BADARG;
    const long total = (long)n * h * w * (c / 8);
    long grid = (total + 255) / 256;
    grid = grid > 8192 ? 8192 : grid;
    hipLaunchKernelGGL(upsample2x_bwd_kernel, dim3((int)grid), dim3(256), 0, (hipStream_t)stream, (const _Float16*)gdst, (_Float16*)gsrc, n, h, w,
                       c / 8, gdst_ld, gsrc_ld);
    return hipGetLastError() == hipSuccess ? LLV_OK : LLV_E_LAUNCH;
}
