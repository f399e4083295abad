// decode.hip -- RetinaNet post-processing up to NMS, for one image, fully on device (no host sync):
//   1. class_max      : per anchor max_c sigmoid(logit)            (reads every cls logit once: HBM-bound)
//   2. topk_select    : per level radix-select of nms_pre anchors, ties broken by anchor index,
//                       then LDS bitonic sort -> candidates in (score desc, anchor asc) order,
//                       exactly the order torch.topk(sorted=True) + our tie rule gives the oracle
//   3. gather_decode  : gather logits/deltas of the candidates, sigmoid, delta2bbox, clip, rescale
// Follows AnchorHead._get_bboxes (mmdetection/mmdet/models/dense_heads/anchor_head.py:636-711),
// delta2bbox (mmdetection/mmdet/core/bbox/coder/delta_xywh_bbox_coder.py:190-237) and
// AnchorGenerator.single_level_grid_anchors (mmdetection/mmdet/core/anchor/anchor_generator.py:225-270).
// Built with -ffp-contract=off: the fp32 operation order of the reference is kept.

#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include "llv.h"
#include "sort_util.h"

namespace {

LLV_GUARD_DECL

struct DecK {
    int n_levels, C, A, nms_pre;
    int h[LLV_MAX_SEG], w[LLV_MAX_SEG], sw[LLV_MAX_SEG], sh[LLV_MAX_SEG];
    long cls_off[LLV_MAX_SEG], reg_off[LLV_MAX_SEG];
    int a0[LLV_MAX_SEG + 1];  // first global anchor index of each level
    int c0[LLV_MAX_SEG + 1];  // first candidate slot of each level
    int cls_ld, reg_ld, f32;
    const void* cls;
    const void* reg;
    const float* base;
    float means[4], stds[4], max_ratio;
    int clip;
    float img_h, img_w;
    float sf[4];
    int rescale;
    float* boxes;
    float* scores;
    int* cand_anchor;
    float* ms;  // workspace: max score per anchor
    int* sel;   // workspace: selected global anchor per candidate slot
    unsigned* hist;   // [n_levels][4096] score histogram (zeroed by the launcher)
    int* meta;        // [n_levels][4]: b*, #above b*, |A|, |T|  (zeroed by the launcher)
    uint64_t* listA;  // [n_levels][capA] keys above b*
    uint64_t* listT;  // [total anchors] keys inside b* (level l at a0[l])
    int capA;
    int blk0[LLV_MAX_SEG + 1];  // first class_max block of each level
    int batch;                  // images (blockIdx.y); level l of image n starts n*h*w pixels after cls_off[l]
    float img_h_b[LLV_MAX_BATCH], img_w_b[LLV_MAX_BATCH], sf_b[LLV_MAX_BATCH][4];
    unsigned* clsmax;           // logits from a LLV_CONV_CLSMAX launch: [rows][A] per-anchor maximum logit (ordered-uint), reset here
    const unsigned char* segw;  //                                       [rows][seg_tiles] which 128-channel segments were stored
    int seg_tiles;
};

// per-image views of the workspace / outputs (image = blockIdx.y)
#define IMG_SETUP(k) \
    const int img = blockIdx.y; \
    const int A_tot = (k).a0[(k).n_levels], N_cand = (k).c0[(k).n_levels]; \
    float* const ms_i = (k).ms + (long)img * A_tot; \
    int* const sel_i = (k).sel + (long)img * N_cand; \
    unsigned* const hist_i = (k).hist + (long)img * (k).n_levels * kBins; \
    int* const meta_i = (k).meta + (long)img * (k).n_levels * 4; \
    uint64_t* const listA_i = (k).listA + (long)img * (k).n_levels * (k).capA; \
    uint64_t* const listT_i = (k).listT + (long)img * A_tot; \
    (void)ms_i; (void)sel_i; (void)hist_i; (void)meta_i; (void)listA_i; (void)listT_i;

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

template <typename T>
__device__ __forceinline__ float ld(const void* p, long i) { return (float)reinterpret_cast<const T*>(p)[i]; }

__device__ __forceinline__ int level_of_anchor(const DecK& k, int g) {
    int l = 0;
#pragma unroll
    for (int s = 1; s < LLV_MAX_SEG; ++s)
        if (s < k.n_levels && g >= k.a0[s]) l = s;
    return l;
}

constexpr int kBins = 4096;      // linear score histogram: bin = min(4095, int(score * 4096)), monotone in the score
constexpr int kSortCap = 1024;   // ties-list size the final kernel sorts in LDS; longer lists take the radix-select path
                                 // (a bitonic sort of 8192 keys by one workgroup took 0.4 ms: 91 LDS passes)

__device__ __forceinline__ int score_bin(float m) {
    int b = (int)(m * (float)kBins);
    return b < 0 ? 0 : (b > kBins - 1 ? kBins - 1 : b);
}

// wave-aggregated atomicAdd(&base[key], 1): lanes with equal key issue one atomic
__device__ __forceinline__ void wave_hist_add(unsigned* base, unsigned key, bool valid) {
    unsigned long long active = __ballot(valid);
    const int lane = threadIdx.x & 63;
    while (active) {
        const int leader = __ffsll((long long)active) - 1;
        const unsigned kb = __shfl(key, leader);
        const unsigned long long same = __ballot(valid && key == kb);
        if (lane == leader) atomicAdd(&base[kb], (unsigned)__popcll(same));
        active &= ~same;
    }
}

// K1: per anchor max_c sigmoid(logit) (every logit read once) + per-level score histogram.
// A block works inside ONE level (blk0[] = first block of each level), so its histogram lives in LDS and is
// flushed with at most one global atomic per non-empty bin.
// The logits of consecutive anchors are contiguous in memory (NHWC, A*C per cell), so the block streams them as
// 16-byte vectors, lane after lane (whole cache lines per wave); a vector lies inside one anchor (C % VEC == 0), its
// max goes to an LDS slot, and one lane per anchor folds that anchor's C/VEC slots. The order of the max does
// not change its value, so the result is bit-identical to the one-lane-per-anchor loop (kept as VEC = 1 for
// layouts whose rows are not 16-byte multiples).
constexpr int kAnchorsPerBlock = 256;        // == the block size of the class-maximum kernels (class_max_from_conv_kernel: one anchor per lane)
constexpr int kMaxVecPerAnchor = 32;
template <typename T, int VEC>
__global__ __launch_bounds__(256) void class_max_kernel(const DecK k) {
    __shared__ unsigned lh[kBins];
    __shared__ float part[VEC > 1 ? kAnchorsPerBlock * kMaxVecPerAnchor : 1];
    int l = 0;
#pragma unroll
    for (int s = 1; s < LLV_MAX_SEG; ++s)
        if (s < k.n_levels && (int)blockIdx.x >= k.blk0[s]) l = s;
    IMG_SETUP(k)
    for (int i = threadIdx.x; i < kBins; i += 256) lh[i] = 0;
    const int n = k.a0[l + 1] - k.a0[l];
    const long cls_l = k.cls_off[l] + (long)img * k.h[l] * k.w[l];
    const int first = (blockIdx.x - k.blk0[l]) * kAnchorsPerBlock;
    const int nb = n - first < kAnchorsPerBlock ? n - first : kAnchorsPerBlock;
    if constexpr (VEC > 1) {
        const int per = k.C / VEC;
        for (int v = threadIdx.x; v < nb * per; v += 256) {
            const int a = v / per, r = v - a * per;
            const int loc = first + a;
            const int cell = loc / k.A, an = loc - cell * k.A;
            const T* p = reinterpret_cast<const T*>(k.cls) + (cls_l + cell) * (long)k.cls_ld + (long)an * k.C + r * VEC;
            const uint4 raw = *reinterpret_cast<const uint4*>(p);
            const T* e = reinterpret_cast<const T*>(&raw);
            float m = 0.f;  // sigmoid > 0
#pragma unroll
            for (int c = 0; c < VEC; ++c) m = fmaxf(m, sigmoidf_((float)e[c]));
            part[v] = m;
        }
        __syncthreads();
        for (int a = threadIdx.x; a < nb; a += 256) {
            float m = 0.f;
            for (int r = 0; r < per; ++r) m = fmaxf(m, part[a * per + r]);
            ms_i[k.a0[l] + first + a] = m;
            atomicAdd(&lh[score_bin(m)], 1u);
        }
    } else {
        __syncthreads();
        for (int a = threadIdx.x; a < nb; a += 256) {
            const int loc = first + a;
            const int cell = loc / k.A, an = loc - cell * k.A;
            const T* p = reinterpret_cast<const T*>(k.cls) + (cls_l + cell) * (long)k.cls_ld + (long)an * k.C;
            float m = 0.f;
            for (int c = 0; c < k.C; ++c) m = fmaxf(m, sigmoidf_((float)p[c]));
            ms_i[k.a0[l] + loc] = m;
            atomicAdd(&lh[score_bin(m)], 1u);
        }
    }
    __syncthreads();
    unsigned* gh = hist_i + l * kBins;
    for (int i = threadIdx.x; i < kBins; i += 256) {
        const unsigned v = lh[i];
        if (v) atomicAdd(&gh[i], v);
    }
}

// K1 for logits that come from a LLV_CONV_CLSMAX launch: the convolution's epilogue has already reduced every anchor's classes to
// their maximum LOGIT (clsmax, order-preserving uint encoding); sigmoid is monotone, so sigmoid(max logit) is the max_c sigmoid(logit_c)
// class_max_kernel computes -- from 4 bytes per anchor instead of 4 C. The entry is reset for the next step.
__device__ __forceinline__ float f32_from_ordered(unsigned key) {
    return __uint_as_float((key & 0x80000000u) ? (key & 0x7fffffffu) : ~key);
}
__global__ __launch_bounds__(256) void class_max_from_conv_kernel(const DecK k) {
    __shared__ unsigned lh[kBins];
    int l = 0;
#pragma unroll
    for (int s = 1; s < LLV_MAX_SEG; ++s)
        if (s < k.n_levels && (int)blockIdx.x >= k.blk0[s]) l = s;
    IMG_SETUP(k)
    for (int i = threadIdx.x; i < kBins; i += 256) lh[i] = 0;
    __syncthreads();
    const int n = k.a0[l + 1] - k.a0[l];
    const long cls_l = k.cls_off[l] + (long)img * k.h[l] * k.w[l];
    const int first = (blockIdx.x - k.blk0[l]) * kAnchorsPerBlock;
    const int nb = n - first < kAnchorsPerBlock ? n - first : kAnchorsPerBlock;
    {
        // one anchor per lane (kAnchorsPerBlock == blockDim.x); most scores of a frame fall into a handful of low bins, so the
        // histogram update is aggregated over the wave (one LDS atomic per distinct bin instead of 64 on the same word)
        const int a = threadIdx.x;
        const bool valid = a < nb;
        float m = 0.f;
        if (valid) {
            const int loc = first + a;                   // consecutive lanes read consecutive words: (cell, anchor) order is the array's
            unsigned* p = k.clsmax + cls_l * (long)k.A + loc;
            const unsigned key = *p;
            *p = 0u;
            m = key ? sigmoidf_(f32_from_ordered(key)) : 0.f;
            ms_i[k.a0[l] + loc] = m;
        }
        wave_hist_add(lh, (unsigned)score_bin(m), valid);
    }
    __syncthreads();
    unsigned* gh = hist_i + l * kBins;
    for (int i = threadIdx.x; i < kBins; i += 256) {
        const unsigned v = lh[i];
        if (v) atomicAdd(&gh[i], v);
    }
}

// K2: one block per level: the bin b* in which the k-th largest score falls, and how many scores lie above it
__global__ __launch_bounds__(256) void topk_threshold_kernel(const DecK k) {
    const int l = blockIdx.x;
    const int n = k.a0[l + 1] - k.a0[l];
    const int kk = k.c0[l + 1] - k.c0[l];
    IMG_SETUP(k)
    int* meta = meta_i + l * 4;
    if (kk >= n) {
        if (threadIdx.x == 0) { meta[0] = -1; meta[1] = 0; }
        return;
    }
    const unsigned* h = hist_i + l * kBins;
    constexpr int PER = kBins / 256;
    const int top = kBins - 1 - threadIdx.x * PER;   // this thread owns bins top, top-1, ..., top-PER+1
    unsigned hv[PER];
    unsigned s = 0;
#pragma unroll
    for (int i = 0; i < PER; ++i) { hv[i] = h[top - i]; s += hv[i]; }
    // scores in the bins above this thread's: a block scan instead of one thread walking the 256 partial sums (15 us per frame)
    __shared__ int scr[17];
    int total = 0;
    unsigned cum = (unsigned)llv::block_exclusive_scan((int)s, scr, &total);
    const bool owner = threadIdx.x == 255 ? cum < (unsigned)kk : (cum < (unsigned)kk && cum + s >= (unsigned)kk);
    if (owner) {                                     // exactly one thread (the last one also when fewer than kk scores exist at all)
        int b = top;
        const int lo = top - PER + 1;
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            if (b > lo && cum + hv[i] < (unsigned)kk) { cum += hv[i]; --b; }
            else break;
        }
        meta[0] = b;           // b*
        meta[1] = (int)cum;    // scores in bins above b*  (< kk)
    }
}

// K3: scores above b* go to list A, scores inside b* to list T (unordered appends; the 64-bit keys are distinct).
// The list counters are per (image, level): one returning global atomic per (wave, level, list) put ~2,000 atomics of an image's finest level
// on ONE address, and same-address device atomics retire at ~46 ns each (82 us per step for 16 images; the same effect as in the loss
// kernel of the training step, round 4). A workgroup of 16 waves now counts in LDS first and takes ONE global atomic per (level, list).
constexpr int kCompactSlots = 4;              // levels a 1024-anchor block can touch (the coarse levels hold 2160 and 540 anchors)
__global__ __launch_bounds__(1024) void topk_compact_kernel(const DecK k) {
    __shared__ int s_cnt[2][kCompactSlots], s_base[2][kCompactSlots];
    const int total = k.a0[k.n_levels];
    const int stride = gridDim.x * blockDim.x;
    const int lane = threadIdx.x & 63;
    IMG_SETUP(k)
    for (int g0 = blockIdx.x * blockDim.x; g0 < total; g0 += stride) {
        const int l0 = level_of_anchor(k, g0);                  // uniform: the level of the block's first anchor
        __syncthreads();                                        // the previous round's bases are no longer read
        if (threadIdx.x < 2 * kCompactSlots) s_cnt[threadIdx.x / kCompactSlots][threadIdx.x % kCompactSlots] = 0;
        __syncthreads();
        const int g = g0 + threadIdx.x;
        int l = l0, which = 0;   // 1 = A, 2 = T
        uint64_t key = 0;
        if (g < total) {
            l = level_of_anchor(k, g);
            const int bstar = meta_i[l * 4];
            if (bstar >= 0) {
                const float m = ms_i[g];
                const int b = score_bin(m);
                which = b > bstar ? 1 : (b == bstar ? 2 : 0);
                key = llv::make_key(m, (unsigned)(g - k.a0[l]));
            }
        }
        const int slot = l - l0;                                // < kCompactSlots by the level sizes; checked below
        int mypos = -1;                                         // position inside the workgroup's run of its (list, level)
        for (int w = 1; w <= 2; ++w) {
            unsigned long long active = __ballot(which == w);
            while (active) {
                const int leader = __ffsll((long long)active) - 1;
                const int ll = __shfl(l, leader);
                const unsigned long long same = __ballot(which == w && l == ll);
                int base = 0;
                if (lane == leader && ll - l0 < kCompactSlots) base = atomicAdd(&s_cnt[w - 1][ll - l0], (int)__popcll(same));
                base = __shfl(base, leader);
                if (which == w && l == ll) mypos = base + __popcll(same & ((1ull << lane) - 1ull));
                active &= ~same;
            }
        }
        __syncthreads();
        if (threadIdx.x < 2 * kCompactSlots) {
            const int w = threadIdx.x / kCompactSlots, sl = threadIdx.x % kCompactSlots;
            const int c = s_cnt[w][sl];
            s_base[w][sl] = (c > 0 && l0 + sl < k.n_levels) ? atomicAdd(&meta_i[(l0 + sl) * 4 + 2 + w], c) : 0;
        }
        __syncthreads();
        if (which && slot >= kCompactSlots) {
            LLV_GUARD_TRIP();
        } else if (which) {
            const int pos = s_base[which - 1][slot] + mypos;
            const int cap = which == 1 ? k.capA : k.a0[l + 1] - k.a0[l];     // list A holds < nms_pre keys, list T at most the level
            if ((unsigned)pos >= (unsigned)cap) LLV_GUARD_TRIP();
            else if (which == 1) listA_i[(long)l * k.capA + pos] = key;
            else listT_i[k.a0[l] + pos] = key;
        }
    }
}

// K4: one block per level: top (k - |A|) keys of T, merged with A, sorted -> sel (score desc, anchor asc)
__global__ __launch_bounds__(1024) void topk_final_kernel(const DecK k) {
    const int l = blockIdx.x;
    const int n = k.a0[l + 1] - k.a0[l];
    const int kk = k.c0[l + 1] - k.c0[l];
    IMG_SETUP(k)
    int* sel = sel_i + k.c0[l];
    const int tid = threadIdx.x;
    if (kk >= n) {  // no top-k on this level: anchor order is kept (anchor_head.py:657-658)
        for (int i = tid; i < n; i += blockDim.x) sel[i] = k.a0[l] + i;
        return;
    }
    extern __shared__ __attribute__((aligned(16))) uint64_t lds[];
    const int np2k = llv::next_pow2(kk);
    uint64_t* fin = lds;             // [np2k]
    uint64_t* tbuf = lds + np2k;     // [kSortCap]
    __shared__ unsigned hist[256];
    __shared__ uint64_t sh_sel[2];
    const int* meta = meta_i + l * 4;
    int nA = meta[2], nT = meta[3];
    if (nA < 0 || nA >= kk || nT < kk - nA || nT > n) {           // impossible with consistent passes (see LLV_GUARD_DECL)
        if (tid == 0) LLV_GUARD_TRIP();
        nA = nA < 0 ? 0 : (nA > kk ? kk : nA);
        nT = nT < 0 ? 0 : (nT > n ? n : nT);
    }
    const int m = kk - nA < nT ? kk - nA : nT;                    // keys still needed, all from T (nT >= kk - nA by construction)
    const uint64_t* A = listA_i + (long)l * k.capA;
    const uint64_t* Tl = listT_i + k.a0[l];
    for (int i = tid; i < np2k; i += blockDim.x) fin[i] = i < nA ? A[i] : 0ull;
    __syncthreads();
    if (nT <= kSortCap) {
        const int np2t = llv::next_pow2(nT > 1 ? nT : 1);
        for (int i = tid; i < np2t; i += blockDim.x) tbuf[i] = i < nT ? Tl[i] : 0ull;
        __syncthreads();
        llv::block_sort_desc(tbuf, np2t, nT);
        for (int i = tid; i < m; i += blockDim.x) fin[nA + i] = tbuf[i];
        __syncthreads();
    } else {
        // long list (a crowded histogram bin, or a blank frame where every score is equal): radix select of the m largest
        // of the distinct 64-bit keys straight from the global list, no sort of T at all
        if (m > 0) llv::block_select_top(Tl, nT, m, fin + nA, hist, sh_sel);
    }
    llv::block_sort_desc(fin, np2k, kk);
    for (int i = tid; i < kk; i += blockDim.x) {
        unsigned idx = llv::key_index(fin[i]);
        if (idx >= (unsigned)n) { LLV_GUARD_TRIP(); idx = 0; }     // a padding key (0) or a corrupt one
        sel[i] = k.a0[l] + (int)idx;
    }
}

template <typename T>
__global__ void gather_decode_kernel(const DecK k) {
    const int slot = blockIdx.x;
    int l = 0;
#pragma unroll
    for (int s = 1; s < LLV_MAX_SEG; ++s)
        if (s < k.n_levels && slot >= k.c0[s]) l = s;
    IMG_SETUP(k)
    int g = sel_i[slot];
    if (g < k.a0[l] || g >= k.a0[l + 1]) { if (threadIdx.x == 0) LLV_GUARD_TRIP(); g = k.a0[l]; }
    const int loc = g - k.a0[l];
    const int cell = loc / k.A, an = loc - cell * k.A;
    const long ipix = (long)img * k.h[l] * k.w[l];
    const long oslot = (long)img * N_cand + slot;
    const T* pc = reinterpret_cast<const T*>(k.cls) + (k.cls_off[l] + ipix + cell) * (long)k.cls_ld + (long)an * k.C;
    if (k.segw) {
        // a class whose 16-channel block the convolution did not store had every logit of the block <= its store threshold (below
        // score_thr): score 0, which the NMS drops exactly as it would the true value. One byte per 128 channels, bit = block.
        const unsigned char* fl = k.segw + (k.cls_off[l] + ipix + cell) * (long)k.seg_tiles;
        for (int c = threadIdx.x; c < k.C; c += blockDim.x) {
            const int ch = an * k.C + c;
            k.scores[oslot * k.C + c] = ((fl[ch >> 7] >> ((ch >> 4) & 7)) & 1) ? sigmoidf_((float)pc[c]) : 0.f;
        }
    } else {
        for (int c = threadIdx.x; c < k.C; c += blockDim.x) k.scores[oslot * k.C + c] = sigmoidf_((float)pc[c]);
    }
    if (threadIdx.x == 0) {
        k.cand_anchor[oslot] = g;
        const int cy = cell / k.w[l], cx = cell - cy * k.w[l];
        const float shx = (float)(cx * k.sw[l]), shy = (float)(cy * k.sh[l]);
        const float* b = k.base + ((long)l * k.A + an) * 4;
        const float ax1 = b[0] + shx, ay1 = b[1] + shy, ax2 = b[2] + shx, ay2 = b[3] + shy;
        const T* pr = reinterpret_cast<const T*>(k.reg) + (k.reg_off[l] + ipix + cell) * (long)k.reg_ld + (long)an * 4;
        const float dx = (float)pr[0] * k.stds[0] + k.means[0];
        const float dy = (float)pr[1] * k.stds[1] + k.means[1];
        float dw = (float)pr[2] * k.stds[2] + k.means[2];
        float dh = (float)pr[3] * k.stds[3] + k.means[3];
        dw = fminf(fmaxf(dw, -k.max_ratio), k.max_ratio);
        dh = fminf(fmaxf(dh, -k.max_ratio), k.max_ratio);
        const float px = (ax1 + ax2) * 0.5f, py = (ay1 + ay2) * 0.5f;
        const float pw = ax2 - ax1, ph = ay2 - ay1;
        const float gw = pw * expf(dw), gh = ph * expf(dh);
        const float gx = px + pw * dx, gy = py + ph * dy;
        float o[4] = {gx - gw * 0.5f, gy - gh * 0.5f, gx + gw * 0.5f, gy + gh * 0.5f};
        if (k.clip) {
            const float mx[4] = {k.img_w_b[img], k.img_h_b[img], k.img_w_b[img], k.img_h_b[img]};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                o[e] = o[e] < 0.f ? 0.f : o[e];
                o[e] = o[e] > mx[e] ? mx[e] : o[e];
            }
        }
        if (k.rescale) {
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = o[e] / k.sf_b[img][e];
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) k.boxes[oslot * 4 + e] = o[e];
    }
}

int fill(const llv_decode_desc* d, DecK& k) {
    if (!d || d->n_levels < 1 || d->n_levels > LLV_MAX_SEG || d->num_classes < 1 || d->num_anchors < 1) return LLV_E_BADARG;
    k.n_levels = d->n_levels; k.C = d->num_classes; k.A = d->num_anchors; k.nms_pre = d->nms_pre;
    long a = 0, c = 0;
    for (int l = 0; l < LLV_MAX_SEG; ++l) {
        const int ll = l < d->n_levels ? l : 0;
        k.h[l] = d->level_h[ll]; k.w[l] = d->level_w[ll]; k.sw[l] = d->stride_w[ll]; k.sh[l] = d->stride_h[ll];
        k.cls_off[l] = d->cls_off[ll]; k.reg_off[l] = d->reg_off[ll];
        k.a0[l] = (int)a; k.c0[l] = (int)c;
        if (l < d->n_levels) {
            if (k.h[l] < 1 || k.w[l] < 1) return LLV_E_BADARG;
            const long n = (long)k.h[l] * k.w[l] * k.A;
            a += n;
            c += (d->nms_pre > 0 && n > d->nms_pre) ? d->nms_pre : n;
            if (a > 0x7fffffffL) return LLV_E_BADARG;
        }
    }
    for (int l = d->n_levels; l <= LLV_MAX_SEG; ++l) { k.a0[l] = (int)a; k.c0[l] = (int)c; }
    return LLV_OK;
}

}  // namespace

extern "C" int llv_retina_decode_ncand(const llv_decode_desc* d) {
    DecK k;
    const int rc = fill(d, k);
    return rc < 0 ? rc : k.c0[d->n_levels];
}

namespace {
struct WsLayout { size_t ms, sel, hist, meta, listA, listT, total; };
WsLayout ws_layout(const DecK& k, int n_levels, int nms_pre, int batch) {
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    WsLayout w;
    size_t o = 0;
    const size_t b = (size_t)(batch > 0 ? batch : 1);
    w.hist = o; o += al(b * n_levels * kBins * 4);
    w.meta = o; o += al(b * n_levels * 4 * 4);               // hist + meta are contiguous: one memset
    w.ms = o; o += al(b * k.a0[n_levels] * 4);
    w.sel = o; o += al(b * k.c0[n_levels] * 4);
    w.listA = o; o += al(b * n_levels * (size_t)(nms_pre > 0 ? nms_pre : 1) * 8);
    w.listT = o; o += al(b * k.a0[n_levels] * 8);
    w.total = o;
    return w;
}
}  // namespace

extern "C" size_t llv_retina_decode_workspace_bytes(const llv_decode_desc* d) {
    DecK k;
    if (fill(d, k) < 0) return 0;
    return ws_layout(k, d->n_levels, d->nms_pre, d->batch).total;
}

extern "C" int llv_retina_decode_topk(const llv_decode_desc* d, void* stream) {
    DecK k;
    int rc = fill(d, k);
    if (rc < 0) return rc;
    if (!d->cls || !d->reg || !d->base_anchors || !d->boxes || !d->scores || !d->cand_anchor || !d->workspace) return LLV_E_BADARG;
    if (d->workspace_bytes < llv_retina_decode_workspace_bytes(d)) return LLV_E_WORKSPACE;
    if (d->cls_ld < k.A * k.C || d->reg_ld < k.A * 4) return LLV_E_BADARG;
    if (d->nms_pre > 8192) return LLV_E_UNSUPPORTED;   // top-k list sorted in LDS: (8192 + kSortCap) keys = 72 KiB (RPN: nms_pre 6000)
    if (d->n_cand != k.c0[d->n_levels]) return LLV_E_BADARG;
    k.cls_ld = d->cls_ld; k.reg_ld = d->reg_ld; k.f32 = d->logits_f32;
    k.cls = d->cls; k.reg = d->reg; k.base = d->base_anchors;
    for (int e = 0; e < 4; ++e) { k.means[e] = d->means[e]; k.stds[e] = d->stds[e]; k.sf[e] = d->scale_factor[e]; }
    k.max_ratio = d->max_ratio; k.clip = d->clip; k.img_h = d->img_h; k.img_w = d->img_w; k.rescale = d->rescale;
    k.boxes = d->boxes; k.scores = d->scores; k.cand_anchor = d->cand_anchor;
    const int batch = d->batch > 0 ? d->batch : 1;
    if (batch > LLV_MAX_BATCH) return LLV_E_UNSUPPORTED;
    k.batch = batch;
    for (int n = 0; n < batch; ++n) {
        const bool per = d->batch > 1;
        k.img_h_b[n] = per ? d->img_h_b[n] : d->img_h;
        k.img_w_b[n] = per ? d->img_w_b[n] : d->img_w;
        for (int e = 0; e < 4; ++e) k.sf_b[n][e] = per ? d->scale_factor_b[n][e] : d->scale_factor[e];
    }
    const WsLayout w = ws_layout(k, d->n_levels, d->nms_pre, batch);
    char* ws = (char*)d->workspace;
    k.hist = (unsigned*)(ws + w.hist); k.meta = (int*)(ws + w.meta);
    k.ms = (float*)(ws + w.ms); k.sel = (int*)(ws + w.sel);
    k.listA = (uint64_t*)(ws + w.listA); k.listT = (uint64_t*)(ws + w.listT);
    k.capA = d->nms_pre > 0 ? d->nms_pre : 1;
    k.clsmax = d->clsmax; k.segw = d->seg_written; k.seg_tiles = d->seg_tiles;
    if ((d->clsmax == nullptr) != (d->seg_written == nullptr)) return LLV_E_BADARG;
    if (d->clsmax && (!d->logits_f32 || d->seg_tiles != (k.A * k.C + 127) / 128 || d->cls_ld < k.A * k.C)) return LLV_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(topk_final_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
        attr_done = true;
    }
    // histogram + counters: zeroed by a KERNEL (zero_words), so that a captured step consists of kernel nodes only
    llv::zero_words(ws + w.hist, w.ms - w.hist, st);
    const int total = k.a0[d->n_levels];
    int grid = (total + 255) / 256;
    grid = grid > 2048 ? 2048 : grid;
    int nb = 0;
    for (int l = 0; l <= LLV_MAX_SEG; ++l) {
        k.blk0[l] = nb;
        if (l < d->n_levels) nb += (k.a0[l + 1] - k.a0[l] + kAnchorsPerBlock - 1) / kAnchorsPerBlock;
    }
    if (k.clsmax) {
        hipLaunchKernelGGL(class_max_from_conv_kernel, dim3(nb, batch), dim3(256), 0, st, k);
    } else {
        // 16-byte vector path: rows, anchors and the base pointer must be 16-byte multiples
        const int vec = k.f32 ? 4 : 8, esz = k.f32 ? 4 : 2;
        bool ok = (k.C % vec) == 0 && (k.cls_ld % vec) == 0 && k.C / vec <= kMaxVecPerAnchor && ((uintptr_t)k.cls & 15) == 0;
        for (int l = 0; l < d->n_levels; ++l) ok = ok && ((k.cls_off[l] * (long)k.cls_ld * esz) & 15) == 0;
        if (k.f32) {
            if (ok) hipLaunchKernelGGL((class_max_kernel<float, 4>), dim3(nb, batch), dim3(256), 0, st, k);
            else hipLaunchKernelGGL((class_max_kernel<float, 1>), dim3(nb, batch), dim3(256), 0, st, k);
        } else {
            if (ok) hipLaunchKernelGGL((class_max_kernel<_Float16, 8>), dim3(nb, batch), dim3(256), 0, st, k);
            else hipLaunchKernelGGL((class_max_kernel<_Float16, 1>), dim3(nb, batch), dim3(256), 0, st, k);
        }
    }
    hipLaunchKernelGGL(topk_threshold_kernel, dim3(d->n_levels, batch), dim3(256), 0, st, k);
    hipLaunchKernelGGL(topk_compact_kernel, dim3((total + 1023) / 1024 > 2048 ? 2048 : (total + 1023) / 1024, batch), dim3(1024), 0, st, k);
    int np2 = 1;
    while (np2 < (d->nms_pre > 0 ? d->nms_pre : 1)) np2 <<= 1;
    hipLaunchKernelGGL(topk_final_kernel, dim3(d->n_levels, batch), dim3(1024), (size_t)(np2 + kSortCap) * 8, st, k);
    if (k.f32) hipLaunchKernelGGL((gather_decode_kernel<float>), dim3(d->n_cand, batch), dim3(128), 0, st, k);
    else hipLaunchKernelGGL((gather_decode_kernel<_Float16>), dim3(d->n_cand, batch), dim3(128), 0, st, k);
    return hipGetLastError() == hipSuccess ? LLV_OK : LLV_E_LAUNCH;
}

// ---- capacity-guard counter (sort_util.h): decode + NMS modules
unsigned int llv_nms_guard_trips_(int reset);
unsigned int llv_conv_guard_trips_(int reset);
extern "C" int llv_guard_trips(uint64_t* count_host, int reset) {
    if (!count_host) return LLV_E_BADARG;
    unsigned int v = 0;
    if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_guard_trips), sizeof(v)) != hipSuccess) return LLV_E_LAUNCH;
    if (reset) {
        const unsigned int z = 0;
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_guard_trips), &z, sizeof(z)) != hipSuccess) return LLV_E_LAUNCH;
    }
    *count_host = (uint64_t)v + (uint64_t)llv_nms_guard_trips_(reset) + (uint64_t)llv_conv_guard_trips_(reset);
    return LLV_OK;
}

// The same sum as ONE kernel node at the end of a step: *dst (device-visible, e.g. pinned host memory next to the packed results) =
// guards fired since the last reset. The host reads it after the synchronisation it does anyway, instead of three synchronous symbol
// reads per hand-over (60-80 us of a 2 ms single-frame step).
unsigned int* llv_nms_guard_addr_();
unsigned int* llv_conv_guard_addr_();
namespace {
__global__ void guard_sum_kernel(const unsigned* a, const unsigned* b, const unsigned* c, unsigned* dst) { *dst = *a + *b + *c; }
}
extern "C" int llv_guard_trips_to(uint32_t* dst, void* stream) {
    if (!dst) return LLV_E_BADARG;
    static unsigned* addr[3] = {nullptr, nullptr, nullptr};
    if (!addr[0]) {                                          // first call outside any capture (the plans' warm-up run)
        void* p = nullptr;
        if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_guard_trips)) != hipSuccess) return LLV_E_LAUNCH;
        unsigned* n_ = llv_nms_guard_addr_();
        unsigned* c_ = llv_conv_guard_addr_();
        if (!p || !n_ || !c_) return LLV_E_LAUNCH;
        addr[1] = n_; addr[2] = c_; addr[0] = (unsigned*)p;
    }
    hipLaunchKernelGGL(guard_sum_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, addr[0], addr[1], addr[2], dst);
    return hipGetLastError() == hipSuccess ? LLV_OK : LLV_E_LAUNCH;
}

// delta2bbox for (n, 4) boxes and (n, 4k) deltas: the bbox heads' class-specific decode (delta_xywh_bbox_coder.py:134-237), the same
// fp32 operation order as gather_decode_kernel above; optional clip to (img_h, img_w) and division by a scale factor
namespace {
__global__ __launch_bounds__(256) void delta2bbox_kernel(const float* __restrict__ rois, const float* __restrict__ deltas, float* __restrict__ out,
                                                         long n, int k, float m0, float m1, float m2, float m3, float s0, float s1, float s2,
                                                         float s3, float max_ratio, float img_h, float img_w, int clip, float f0, float f1,
                                                         float f2, float f3, int rescale) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * k) return;
    const long r = i / k;
    const float4 b = *reinterpret_cast<const float4*>(rois + r * 4);
    const float4 d = *reinterpret_cast<const float4*>(deltas + i * 4);
    const float dx = d.x * s0 + m0, dy = d.y * s1 + m1;
    float dw = d.z * s2 + m2, dh = d.w * s3 + m3;
    dw = fminf(fmaxf(dw, -max_ratio), max_ratio);
    dh = fminf(fmaxf(dh, -max_ratio), max_ratio);
    const float px = (b.x + b.z) * 0.5f, py = (b.y + b.w) * 0.5f;
    const float pw = b.z - b.x, ph = b.w - b.y;
    const float gw = pw * expf(dw), gh = ph * expf(dh);
    const float gx = px + pw * dx, gy = py + ph * dy;
    float o[4] = {gx - gw * 0.5f, gy - gh * 0.5f, gx + gw * 0.5f, gy + gh * 0.5f};
    if (clip) {
        const float mx[4] = {img_w, img_h, img_w, img_h};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            o[e] = o[e] < 0.f ? 0.f : o[e];
            o[e] = o[e] > mx[e] ? mx[e] : o[e];
        }
    }
    if (rescale) { o[0] = o[0] / f0; o[1] = o[1] / f1; o[2] = o[2] / f2; o[3] = o[3] / f3; }
    *reinterpret_cast<float4*>(out + i * 4) = make_float4(o[0], o[1], o[2], o[3]);
}
}  // namespace

extern "C" int llv_delta2bbox(const float* rois, const float* deltas, float* out, int64_t n, int k, const float* means, const float* stds,
                              float wh_ratio_clip, float img_h, float img_w, int clip, const float* scale_factor, void* stream) {
    if (n < 0 || k < 1 || !means || !stds || wh_ratio_clip <= 0.f) return LLV_E_BADARG;
    if (n == 0) return LLV_OK;
    if (!rois || !deltas || !out) return LLV_E_BADARG;
    if (n * k > 0x7fffffffL) return LLV_E_UNSUPPORTED;
    const float mr = fabsf(logf(wh_ratio_clip));
    const long tot = n * k;
    hipLaunchKernelGGL(delta2bbox_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream, rois, deltas, out, (long)n, k,
                       means[0], means[1], means[2], means[3], stds[0], stds[1], stds[2], stds[3], mr, img_h, img_w, clip,
                       scale_factor ? scale_factor[0] : 1.f, scale_factor ? scale_factor[1] : 1.f, scale_factor ? scale_factor[2] : 1.f,
                       scale_factor ? scale_factor[3] : 1.f, scale_factor ? 1 : 0);
    return hipGetLastError() == hipSuccess ? LLV_OK : LLV_E_LAUNCH;
}
