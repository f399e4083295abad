// nms.hip -- per-image multiclass NMS on device, index-exact against the oracle.
//
// Reference semantics (mmdetection/mmdet/core/post_processing/bbox_nms.py:7-93 -> mmcv 1.3.0
// batched_nms -> nms, un-vendored; restated in oracle/nms.py):
//   - candidates = every (box i, class c) with score[i][c] > score_thr, flat index i*C + c
//   - boxes_for_nms = box + c * (max_coordinate + 1), max over the candidates' boxes (fp32 arithmetic)
//   - greedy NMS in (score desc) order, suppress when IoU > iou_thr, offset 0 areas
//   - output sorted by score desc, truncated to max_num
// Ties in score are ordered by flat index ascending (the oracle's total order).
//
// Because the class offset makes classes independent, each class is one workgroup:
//   compact (ordered) -> LDS bitonic sort of 64-bit (score, index) keys -> wave 0 runs the greedy
//   scan 64 boxes at a time: every lane first tests its box against the LDS-resident kept boxes,
//   then the 64 lanes resolve each other with __ballot / ffs / __shfl broadcasts.
// A class can contribute at most max_num boxes to the final top-max_num, so each class stops at
// max_num kept and the merge kernel sorts at most C*max_num keys. Exact, no truncation heuristics.
// Built with -ffp-contract=off so the IoU arithmetic is bit-identical to IEEE fp32 on the host.

#include <hip/hip_runtime.h>
#include "llv.h"
#include "sort_util.h"

namespace {

LLV_GUARD_DECL

constexpr int kMaxN = 16384;     // boxes per class list (LDS keys: 128 KiB of the CU's 160; test_cfg without nms_pre keeps every anchor: the reference's own
                                 // get_bboxes fixture has 12,276 of them, mmdetection/tests/test_onnx/test_head.py:86-131)
constexpr int kMaxMerge = 8192;  // C * max_num
constexpr size_t kLdsLimit = 160 * 1024 - 8 * 1024;   // dynamic LDS the kernels opt in to (allow_big_lds)

struct NmsWs {
    float* maxc;        // [1] max coordinate over candidate boxes (ordered-uint encoded)
    unsigned* maxc_u;
    int* any;           // [1] any candidate
    int* cls_cnt;       // [C]
    uint64_t* cls_keys; // [C][max_num]  (score, flat index) of kept per class
};

__device__ __forceinline__ bool iou_gt(const float4 a, const float4 b, float thr) {
    const float left = fmaxf(a.x, b.x), right = fminf(a.z, b.z);
    const float top = fmaxf(a.y, b.y), bottom = fminf(a.w, b.w);
    const float width = fmaxf(right - left, 0.f), height = fmaxf(bottom - top, 0.f);
    const float inter = width * height;
    const float sa = (a.z - a.x) * (a.w - a.y);
    const float sb = (b.z - b.x) * (b.w - b.y);
    return inter / (sa + sb - inter) > thr;
}

// Pass 1 (multi-block, coalesced over the n x C score matrix): every (box i, class c) with score > thr is
// appended to class c's candidate list as a 64-bit (score, flat index) key, and the maximum coordinate of
// the boxes that own a candidate is reduced (boxes.max() of batched_nms). Appends are unordered; the keys are
// distinct, so the per-class sort that follows is deterministic.
__global__ __launch_bounds__(256) void nms_scan_kernel(const float* __restrict__ boxes, const float* __restrict__ scores,
                                                      const float* __restrict__ factors,
                                                      int n, int C, int c_ld, float thr, unsigned* maxc_u,
                                                      int* cnt_in, uint64_t* cand, long ws_stride, int cs, int single_list) {
    const long total = (long)n * C;
    {   // image = blockIdx.y: inputs are [batch][n][...], the workspace repeats every ws_stride bytes
        const int img = blockIdx.y;
        boxes += (long)img * n * 4 * (cs ? C : 1);      // cs: class-specific boxes [n][C][4] (bbox_nms.py:31-35)
        scores += (long)img * n * c_ld;
        if (factors) factors += (long)img * n;
        maxc_u = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(maxc_u) + img * ws_stride);
        cnt_in = reinterpret_cast<int*>(reinterpret_cast<char*>(cnt_in) + img * ws_stride);
        cand = reinterpret_cast<uint64_t*>(reinterpret_cast<char*>(cand) + img * ws_stride);
    }
    if (single_list && reinterpret_cast<const int*>(maxc_u)[1] == 0) return;     // nms_presorted_kernel found the list in order: nothing to collect
    unsigned m = 0;
    const unsigned utotal = (unsigned)total, uC = (unsigned)C;        // n * C < 2^31 (checked by the caller)
    const int lane = threadIdx.x & 63;
    for (unsigned base = blockIdx.x * blockDim.x; base < utotal; base += gridDim.x * blockDim.x) {      // block-uniform trip count
        const unsigned e = base + threadIdx.x;
        const unsigned i = e / uC, c = e - i * uC;
        const float s = e < utotal ? scores[(long)i * c_ld + c] : 0.f;
        const bool hit = e < utotal && s > thr;
        int slot = 0;
        if (C == 1) {
            // one list: a wavefront takes its slots with ONE atomic (6000 lanes on one word serialise in L2: 71 us of the RPN's step)
            const unsigned long long m64 = __ballot(hit);
            if (m64) {
                const int leader = __ffsll((long long)m64) - 1;
                int b0 = 0;
                if (lane == leader) b0 = atomicAdd(&cnt_in[0], __popcll(m64));
                slot = __shfl(b0, leader) + __popcll(m64 & ((1ull << lane) - 1ull));
            }
        } else if (hit) {
            slot = atomicAdd(&cnt_in[c], 1);
        }
        if (hit) {
            if ((unsigned)slot >= (unsigned)n) { LLV_GUARD_TRIP(); continue; }   // a class list holds at most n keys (counter not zeroed?)
            // score_factors multiply AFTER the threshold test (bbox_nms.py:52-62)
            cand[(long)c * n + slot] = llv::make_key(factors ? s * factors[i] : s, e);
            const float4 b = reinterpret_cast<const float4*>(boxes)[cs ? e : i];
            const unsigned u = llv::f32_ordered(fmaxf(fmaxf(b.x, b.y), fmaxf(b.z, b.w)));
            m = u > m ? u : m;
        }
    }
    // boxes.max(): wave -> block -> ONE atomic per block (thousands of waves hitting one address serialise in L2:
    // that was most of this kernel's 138 us on a single image)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned t = __shfl_xor(m, o);
        m = t > m ? t : m;
    }
    __shared__ unsigned wmax[4];
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned a = wmax[0] > wmax[1] ? wmax[0] : wmax[1], b = wmax[2] > wmax[3] ? wmax[2] : wmax[3];
        const unsigned mm = a > b ? a : b;
        if (mm) atomicMax(maxc_u, mm);
    }
}

// Greedy scan over `cnt` sorted keys (LDS) by wave 0. box_of(idx) gives the (offset) box of
// candidate idx. Kept keys are appended to out_keys (global) and kept boxes to kept (LDS/global).
template <typename BoxFn>
__device__ inline int greedy_scan(const uint64_t* keys, int cnt, float thr, int max_keep, float4* kept,
                                  uint64_t* out_keys, BoxFn box_of) {
    const int lane = threadIdx.x & 63;
    int nkept = 0;
    for (int base = 0; base < cnt && nkept < max_keep; base += 64) {
        const int i = base + lane;
        const bool valid = i < cnt;
        uint64_t key = 0;
        float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
        if (valid) {
            key = keys[i];
            b = box_of(llv::key_index(key));
        }
        bool alive = valid;
        for (int j = 0; j < nkept; ++j) {
            if (alive && iou_gt(kept[j], b, thr)) alive = false;
        }
        unsigned long long mask = __ballot(alive);
        int start = 0;
        while (nkept < max_keep) {
            const unsigned long long rem = start >= 64 ? 0ull : (mask >> start) << start;
            if (!rem) break;
            const int t = __ffsll((long long)rem) - 1;
            float4 kb;
            kb.x = __shfl(b.x, t); kb.y = __shfl(b.y, t); kb.z = __shfl(b.z, t); kb.w = __shfl(b.w, t);
            const uint64_t kk = __shfl((unsigned long long)key, t);
            if (lane == 0) { kept[nkept] = kb; out_keys[nkept] = kk; }
            ++nkept;
            if (alive && lane > t && iou_gt(kb, b, thr)) alive = false;
            mask = __ballot(alive);
            start = t + 1;
        }
        __builtin_amdgcn_wave_barrier();
    }
    return nkept;
}

// The same greedy scan by the whole workgroup (blockDim = 64 * NW threads; every thread calls it). A batch is blockDim candidates:
// all of them are first tested against the boxes kept before the batch (the long loop, now blockDim-wide instead of 64-wide), then
// the batch's waves resolve their 64 candidates one after the other exactly as greedy_scan does, and after each wave the later
// waves test their candidates against that wave's newly kept boxes. A candidate is suppressed by exactly the kept boxes that precede
// it in score order: the result equals the one-wave scan's.
template <typename BoxFn>
__device__ inline int greedy_scan_block(const uint64_t* keys, int cnt, float thr, int max_keep, float4* kept,
                                        uint64_t* out_keys, BoxFn box_of, int* s_nkept) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    // The kept count travels in a register that every wave updates from the word the resolving wave published; two words are
    // used alternately, so the wave that resolves next (and publishes into the other word) can never overwrite a count that a
    // slower wave has not read yet: every wave sees the same nk1 and the `break`s below stay uniform.
    int nk0 = 0, slot = 0;
    for (int base = 0; base < cnt; base += blockDim.x) {
        if (nk0 >= max_keep) break;                          // uniform: nk0 is the same in every wave
        const int i = base + threadIdx.x;
        const bool valid = i < cnt;
        uint64_t key = 0;
        float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
        if (valid) {
            key = keys[i];
            b = box_of(llv::key_index(key));
        }
        bool alive = valid;
        for (int j = 0; j < nk0; ++j)
            if (alive && iou_gt(kept[j], b, thr)) alive = false;
        for (int w = 0; w < nw; ++w) {
            if (wave == w) {
                int nkept = nk0;
                unsigned long long mask = __ballot(alive);
                int start = 0;
                while (nkept < max_keep) {
                    const unsigned long long rem = start >= 64 ? 0ull : (mask >> start) << start;
                    if (!rem) break;
                    const int t = __ffsll((long long)rem) - 1;
                    float4 kb;
                    kb.x = __shfl(b.x, t); kb.y = __shfl(b.y, t); kb.z = __shfl(b.z, t); kb.w = __shfl(b.w, t);
                    const uint64_t kk = __shfl((unsigned long long)key, t);
                    if (lane == 0) { kept[nkept] = kb; out_keys[nkept] = kk; }
                    ++nkept;
                    if (alive && lane > t && iou_gt(kb, b, thr)) alive = false;
                    mask = __ballot(alive);
                    start = t + 1;
                }
                if (lane == 0) s_nkept[slot] = nkept;
            }
            __syncthreads();
            const int nk1 = s_nkept[slot];
            slot ^= 1;
            if (wave > w)
                for (int j = nk0; j < nk1; ++j)
                    if (alive && iou_gt(kept[j], b, thr)) alive = false;
            nk0 = nk1;
            if (nk1 >= max_keep) break;                      // uniform
        }
        __syncthreads();
    }
    __syncthreads();
    return nk0;
}

// one workgroup per class: sort its candidate list in LDS, greedy scan by the whole workgroup
__global__ __launch_bounds__(1024) void nms_class_kernel(const float* __restrict__ boxes, int n, int C, float iou_thr,
                                                        int max_num, const unsigned* maxc_u, const int* cnt_in,
                                                        const uint64_t* cand, int* cls_cnt, uint64_t* cls_keys, long ws_stride, int cs) {
    const int c = blockIdx.x;
    {
        const int img = blockIdx.y;
        boxes += (long)img * n * 4 * (cs ? C : 1);
        maxc_u = reinterpret_cast<const unsigned*>(reinterpret_cast<const char*>(maxc_u) + img * ws_stride);
        cnt_in = reinterpret_cast<const int*>(reinterpret_cast<const char*>(cnt_in) + img * ws_stride);
        cand = reinterpret_cast<const uint64_t*>(reinterpret_cast<const char*>(cand) + img * ws_stride);
        cls_cnt = reinterpret_cast<int*>(reinterpret_cast<char*>(cls_cnt) + img * ws_stride);
        cls_keys = reinterpret_cast<uint64_t*>(reinterpret_cast<char*>(cls_keys) + img * ws_stride);
    }
    int cnt = cnt_in[c];
    if (cnt < 0 || cnt > n) { if (threadIdx.x == 0) LLV_GUARD_TRIP(); cnt = cnt < 0 ? 0 : n; }
    if (cnt == 0) {
        if (threadIdx.x == 0) cls_cnt[c] = 0;
        return;
    }
    extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
    uint64_t* keys = reinterpret_cast<uint64_t*>(sm);                     // [np2]
    const int np2 = llv::next_pow2(cnt);
    float4* kept = reinterpret_cast<float4*>(sm + (size_t)np2 * 8);       // [max_num]
    const uint64_t* src = cand + (long)c * n;
    for (int i = threadIdx.x; i < np2; i += blockDim.x) keys[i] = i < cnt ? src[i] : 0ull;
    __syncthreads();
    llv::block_sort_desc(keys, np2, cnt);
    {
        // boxes_for_nms = boxes + idxs.to(boxes) * (max_coordinate + 1)   (mmcv batched_nms)
        __shared__ int s_nkept[2];
        const float off = (float)c * (llv::f32_unordered(*maxc_u) + 1.0f);
        const unsigned nflat = (unsigned)n * (unsigned)C;
        auto box_of = [&](unsigned flat) {
            if (flat >= nflat) { LLV_GUARD_TRIP(); flat = 0; }
            float4 b = reinterpret_cast<const float4*>(boxes)[cs ? flat : flat / (unsigned)C];
            b.x += off; b.y += off; b.z += off; b.w += off;
            return b;
        };
        const int nk = greedy_scan_block(keys, cnt, iou_thr, max_num, kept, cls_keys + (long)c * max_num, box_of, s_nkept);
        if (threadIdx.x == 0) cls_cnt[c] = nk;
    }
}

// ---- one long list (num_classes == 1, n >= kMaskMinN: the RPN's 6000 proposals of a single-level neck) ---------------------------------
// nms_class_kernel tests every candidate against every box kept so far on ONE compute unit: cnt * kept IoUs (1.8 M for 6000 x 300), 190 us.
// Here the IoUs go to the whole chip and the serial part only combines bits:
//   nms_sort_kernel    the class kernel's load + sort; sorted keys and their boxes go to the workspace
//   nms_mask_kernel    mask[i][q] = bits j of { 64 q + j > i and IoU(box i, box 64 q + j) > thr }, a 64 x 64 tile per wavefront, upper triangle
//   nms_reduce_kernel  one workgroup of 8 waves walks the list 512 boxes at a time: wave w resolves its 64 boxes against the removed word
//                      of its chunk (a scalar loop over the alive bits, the tile's diagonal word by readlane), ORs the kept rows' words for
//                      the rest of the 512 into the removed words, and after the eight waves all threads OR the kept rows' remaining words
//                      (one coalesced pass, LDS atomics). A box is removed by exactly the kept boxes before it: the class kernel's result.
constexpr int kMaskMinN = 2048;

__host__ __device__ inline int mask_row_words(int n) { return (((n + 63) >> 6) + 7) & ~7; }     // 64-byte multiples: a thread reads 8 words in a row

// The RPN hands its proposals over in top-k order (score descending, ties by anchor: decode.hip), which is the order the sort would
// produce: this pass writes the keys and boxes in place as the sorted list, counts them, and raises header word 1 ("unsorted") when it
// sees a key smaller than its successor. Only then do nms_scan_kernel / nms_sort_kernel do their work (they run either way: the launch
// list of a captured step is fixed).
__global__ __launch_bounds__(256) void nms_presorted_kernel(const float* __restrict__ boxes, const float* __restrict__ scores,
                                                           const float* __restrict__ factors, int n, int c_ld, float thr, int* hdr,
                                                           uint64_t* skeys, float4* sbox, long ws_stride) {
    {
        const int img = blockIdx.y;
        boxes += (long)img * n * 4;
        scores += (long)img * n * c_ld;
        if (factors) factors += (long)img * n;
        hdr = reinterpret_cast<int*>(reinterpret_cast<char*>(hdr) + img * ws_stride);
        skeys = reinterpret_cast<uint64_t*>(reinterpret_cast<char*>(skeys) + img * ws_stride);
        sbox = reinterpret_cast<float4*>(reinterpret_cast<char*>(sbox) + img * ws_stride);
    }
    const int i = blockIdx.x * 256 + threadIdx.x;
    auto key_of = [&](int j) -> uint64_t {
        if (j >= n) return 0ull;
        const float s = scores[(long)j * c_ld];
        return s > thr ? llv::make_key(factors ? s * factors[j] : s, (unsigned)j) : 0ull;
    };
    const uint64_t key = key_of(i);
    const bool bad = i < n && key < key_of(i + 1);
    if (i < n) {
        skeys[i] = key;
        sbox[i] = reinterpret_cast<const float4*>(boxes)[i];
    }
    const unsigned long long hits = __ballot(key != 0ull), bads = __ballot(bad);
    if ((threadIdx.x & 63) == 0) {
        if (hits) atomicAdd(&hdr[2], __popcll(hits));
        if (bads) atomicOr(&hdr[1], 1);
    }
}

__global__ __launch_bounds__(1024) void nms_sort_kernel(const float* __restrict__ boxes, int n, const unsigned* maxc_u, const int* cnt_in,
                                                       const uint64_t* cand, uint64_t* skeys, float4* sbox, long ws_stride) {
    {
        const int img = blockIdx.y;
        boxes += (long)img * n * 4;
        cnt_in = reinterpret_cast<const int*>(reinterpret_cast<const char*>(cnt_in) + img * ws_stride);
        cand = reinterpret_cast<const uint64_t*>(reinterpret_cast<const char*>(cand) + img * ws_stride);
        skeys = reinterpret_cast<uint64_t*>(reinterpret_cast<char*>(skeys) + img * ws_stride);
        sbox = reinterpret_cast<float4*>(reinterpret_cast<char*>(sbox) + img * ws_stride);
    }
    // class 0: the offset c * (max_coordinate + 1) is 0. Header: word 0 the max coordinate, 1 "unsorted", 2 the presorted pass's count
    const int* hdr = reinterpret_cast<const int*>(reinterpret_cast<const char*>(maxc_u) + blockIdx.y * ws_stride);
    if (hdr[1] == 0) {                                       // the list arrived sorted: skeys / sbox are in place
        if (threadIdx.x == 0) const_cast<int*>(cnt_in)[0] = hdr[2];
        return;
    }
    int cnt = cnt_in[0];
    if (cnt < 0 || cnt > n) { if (threadIdx.x == 0) LLV_GUARD_TRIP(); cnt = cnt < 0 ? 0 : n; }
    if (cnt == 0) return;
    extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
    uint64_t* keys = reinterpret_cast<uint64_t*>(sm);
    const int np2 = llv::next_pow2(cnt);
    for (int i = threadIdx.x; i < np2; i += blockDim.x) keys[i] = i < cnt ? cand[i] : 0ull;
    __syncthreads();
    llv::block_sort_desc(keys, np2, cnt);
    for (int i = threadIdx.x; i < cnt; i += blockDim.x) {
        const uint64_t key = keys[i];
        unsigned flat = llv::key_index(key);
        if (flat >= (unsigned)n) { LLV_GUARD_TRIP(); flat = 0; }
        skeys[i] = key;
        sbox[i] = reinterpret_cast<const float4*>(boxes)[flat];
    }
}

__global__ __launch_bounds__(64) void nms_mask_kernel(const float4* __restrict__ sbox, const int* cnt_in, uint64_t* mask, int n, int nw,
                                                     float thr, long ws_stride) {
    const int q = blockIdx.x, r = blockIdx.y, lane = threadIdx.x;
    if (q < r) return;
    {
        const int img = blockIdx.z;
        cnt_in = reinterpret_cast<const int*>(reinterpret_cast<const char*>(cnt_in) + img * ws_stride);
        sbox = reinterpret_cast<const float4*>(reinterpret_cast<const char*>(sbox) + img * ws_stride);
        mask = reinterpret_cast<uint64_t*>(reinterpret_cast<char*>(mask) + img * ws_stride);
    }
    int cnt = cnt_in[0];
    cnt = cnt < 0 ? 0 : (cnt > n ? n : cnt);
    if (64 * q >= cnt) return;                               // r <= q: the row chunk holds candidates too
    __shared__ float4 cb[64];
    const int j0 = 64 * q;
    cb[lane] = j0 + lane < cnt ? sbox[j0 + lane] : make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();
    const int i = 64 * r + lane;
    if (i >= cnt) return;
    const float4 bi = sbox[i];
    const int jn = cnt - j0 < 64 ? cnt - j0 : 64;
    unsigned long long bits = 0ull;
    for (int jj = 0; jj < jn; ++jj)
        if (j0 + jj > i && iou_gt(bi, cb[jj], thr)) bits |= 1ull << jj;
    mask[(long)i * nw + q] = bits;
}

__device__ __forceinline__ unsigned long long uniform64(unsigned long long v) {
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
    return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ unsigned long long readlane64(unsigned long long v, int t) {
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, t), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), t);
    return ((unsigned long long)hi << 32) | lo;
}

constexpr int kRedWaves = 8;                                 // boxes per step of the reduction = 64 * kRedWaves

__global__ __launch_bounds__(64 * kRedWaves) void nms_reduce_kernel(const uint64_t* __restrict__ skeys, const int* cnt_in,
                                                                   const uint64_t* __restrict__ mask, int n, int nw, int max_num,
                                                                   int* cls_cnt, uint64_t* cls_keys, long ws_stride) {
    constexpr int NW = kRedWaves, NT = 64 * NW;
    {
        const int img = blockIdx.y;
        cnt_in = reinterpret_cast<const int*>(reinterpret_cast<const char*>(cnt_in) + img * ws_stride);
        skeys = reinterpret_cast<const uint64_t*>(reinterpret_cast<const char*>(skeys) + img * ws_stride);
        mask = reinterpret_cast<const uint64_t*>(reinterpret_cast<const char*>(mask) + img * ws_stride);
        cls_cnt = reinterpret_cast<int*>(reinterpret_cast<char*>(cls_cnt) + img * ws_stride);
        cls_keys = reinterpret_cast<uint64_t*>(reinterpret_cast<char*>(cls_keys) + img * ws_stride);
    }
    int cnt = cnt_in[0];
    cnt = cnt < 0 ? 0 : (cnt > n ? n : cnt);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    __shared__ unsigned long long rem[kMaxN / 64 + NW];      // removed bits, one word per chunk of 64
    __shared__ int s_list[NT];                               // rows kept in the current step
    __shared__ int s_nk, s_k;
    for (int i = tid; i < kMaxN / 64 + NW; i += NT) rem[i] = 0ull;
    if (tid == 0) { s_nk = 0; s_k = 0; }
    __syncthreads();
    const int nchunk = (cnt + 63) >> 6;
    // row i = 64 * (c0 + wave) + lane: its words c0 + w2 >= its own chunk (chunks past the list were never written) and its key
    auto load_row = [&](int c0, unsigned long long (&dw)[NW], uint64_t& key) {
        const int i = 64 * (c0 + wave) + lane;
#pragma unroll
        for (int w2 = 0; w2 < NW; ++w2)
            dw[w2] = (i < cnt && w2 >= wave && c0 + w2 < nchunk) ? mask[(long)i * nw + c0 + w2] : 0ull;
        key = i < cnt ? skeys[i] : 0ull;
    };
    unsigned long long dw[NW], dwn[NW];
#pragma unroll
    for (int w2 = 0; w2 < NW; ++w2) dwn[w2] = 0ull;
    uint64_t key, keyn = 0ull;
    load_row(0, dw, key);
    for (int c0 = 0; c0 < nchunk; c0 += NW) {
        const int q = c0 + wave, i = 64 * q + lane;
        const bool valid = i < cnt;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            if (wave == w) {
                int nk = s_nk;
                const int k0 = s_k;
                unsigned long long a = uniform64(~rem[q] & __ballot(valid));
                unsigned long long kept = 0ull;
                const int nk0 = nk;
                while (a && nk < max_num) {
                    const int t = __ffsll((long long)a) - 1;
                    kept |= 1ull << t;
                    ++nk;
                    a &= ~readlane64(dw[w], t);
                    a &= ~(1ull << t);
                }
                if ((kept >> lane) & 1ull) {
#pragma unroll
                    for (int w2 = w + 1; w2 < NW; ++w2)      // the kept rows' words for the later chunks of this step
                        if (dw[w2]) atomicOr(&rem[c0 + w2], dw[w2]);
                    const int rank = __popcll(kept & ((1ull << lane) - 1ull));
                    cls_keys[nk0 + rank] = key;
                    s_list[k0 + rank] = i;
                }
                if (lane == 0) { s_nk = nk; s_k = k0 + __popcll(kept); }
            }
            __syncthreads();
        }
        const int k = s_k, nk_all = s_nk;
        const int wlo = c0 + NW, wf = nchunk - wlo;
        if (wf > 0) load_row(wlo, dwn, keyn);                // in flight together with the far words below: one wait for both
        if (wf > 0 && k > 0 && nk_all < max_num) {
            const int tot = k * wf;
            for (int e0 = tid; e0 < tot; e0 += 4 * NT) {     // 4 independent loads in flight per thread, then the ORs
                unsigned long long v[4];
                int wd[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int e = e0 + NT * u;
                    const int r = e < tot ? e / wf : 0;
                    wd[u] = wlo + (e < tot ? e - r * wf : 0);
                    v[u] = e < tot ? mask[(long)s_list[r] * nw + wd[u]] : 0ull;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (v[u]) atomicOr(&rem[wd[u]], v[u]);
            }
        }
        __syncthreads();
        if (tid == 0) s_k = 0;
        if (nk_all >= max_num) break;                        // uniform
        __syncthreads();
#pragma unroll
        for (int w2 = 0; w2 < NW; ++w2) {
            asm volatile("" : "+v"(dwn[w2]));                // keeps the wait for the prefetched row behind the far words' loads
            dw[w2] = dwn[w2];
        }
        key = keyn;
    }
    if (tid == 0) cls_cnt[0] = s_nk;
}

// one workgroup: merge the per-class kept lists, global (score desc, flat asc) order, top max_num
__global__ __launch_bounds__(1024) void nms_merge_kernel(const float* __restrict__ boxes, int n, int C, int max_num,
                                                         const int* cls_cnt,
                                                         const uint64_t* cls_keys, float* dets, int64_t* labels,
                                                         int* count, int* keep_flat, long ws_stride, int cs) {
    {
        const int img = blockIdx.y;
        boxes += (long)img * n * 4 * (cs ? C : 1);
        cls_cnt = reinterpret_cast<const int*>(reinterpret_cast<const char*>(cls_cnt) + img * ws_stride);
        cls_keys = reinterpret_cast<const uint64_t*>(reinterpret_cast<const char*>(cls_keys) + img * ws_stride);
        dets += (long)img * max_num * 5;
        labels += (long)img * max_num;
        keep_flat += (long)img * max_num;
        count += img;
    }
    extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
    uint64_t* keys = reinterpret_cast<uint64_t*>(sm);
    __shared__ int s_off[1025];
    __shared__ int s_scr[17];
    __shared__ unsigned s_hist[256];
    __shared__ uint64_t s_sel[2];
    // offsets of the per-class lists: block-wide exclusive scan of cls_cnt (C <= 1024 = blockDim)
    int mine = (int)threadIdx.x < C ? cls_cnt[threadIdx.x] : 0;
    if (mine < 0 || mine > max_num) { LLV_GUARD_TRIP(); mine = mine < 0 ? 0 : max_num; }
    int total = 0;
    const int excl = llv::block_exclusive_scan(mine, s_scr, &total);
    if ((int)threadIdx.x < C) s_off[threadIdx.x] = excl;
    if (threadIdx.x == 0) s_off[C] = total;
    __syncthreads();
    // gather the kept lists: entry e of the concatenation belongs to the class c with s_off[c] <= e < s_off[c+1]
    for (int e = threadIdx.x; e < total; e += blockDim.x) {
        int lo = 0, hi = C;                                  // invariant: s_off[lo] <= e < s_off[hi]
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (s_off[mid] <= e) lo = mid; else hi = mid;
        }
        keys[e] = cls_keys[(long)lo * max_num + (e - s_off[lo])];
    }
    __syncthreads();
    const int k = total < max_num ? total : max_num;
    const int kp2 = llv::next_pow2(k > 1 ? k : 1);
    if (total > 2 * kp2 && total > 1024) {
        // only the k best of `total` kept boxes are wanted: radix-select them, then sort k keys instead of `total`
        uint64_t* top = keys + llv::next_pow2(total);
        llv::block_select_top(keys, total, k, top, s_hist, s_sel);
        for (int i = k + threadIdx.x; i < kp2; i += blockDim.x) top[i] = 0;
        __syncthreads();
        llv::block_sort_desc(top, kp2, k);
        keys = top;
    } else {
        const int np2 = llv::next_pow2(total > 1 ? total : 1);
        for (int i = total + threadIdx.x; i < np2; i += blockDim.x) keys[i] = 0;
        __syncthreads();
        llv::block_sort_desc(keys, np2, total);
    }
    for (int i = threadIdx.x; i < k; i += blockDim.x) {
        const uint64_t key = keys[i];
        unsigned flat = llv::key_index(key);
        if (flat >= (unsigned)n * (unsigned)C) { LLV_GUARD_TRIP(); flat = 0; }
        const unsigned bi = flat / (unsigned)C, c = flat - bi * (unsigned)C;
        const float4 b = reinterpret_cast<const float4*>(boxes)[cs ? flat : bi];
        dets[i * 5 + 0] = b.x; dets[i * 5 + 1] = b.y; dets[i * 5 + 2] = b.z; dets[i * 5 + 3] = b.w;
        dets[i * 5 + 4] = llv::key_score(key);
        labels[i] = (int64_t)c;
        keep_flat[i] = (int)flat;
    }
    if (threadIdx.x == 0) *count = k;
}

// plain single-class NMS (mmcv.ops.nms), n <= kMaxN; kept boxes live in global workspace
__global__ __launch_bounds__(256) void nms_plain_kernel(const float* __restrict__ boxes,
                                                        const float* __restrict__ scores, int n, float iou_thr,
                                                        int64_t* keep, int* count, float4* kept_ws,
                                                        uint64_t* keys_ws) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
    uint64_t* keys = reinterpret_cast<uint64_t*>(sm);
    const int np2 = llv::next_pow2(n);
    for (int i = threadIdx.x; i < np2; i += blockDim.x)
        keys[i] = i < n ? llv::make_key(scores[i], (unsigned)i) : 0ull;
    __syncthreads();
    llv::block_sort_desc(keys, np2, n);
    __shared__ int s_nk;
    if (threadIdx.x < 64) {
        auto box_of = [&](unsigned idx) { return reinterpret_cast<const float4*>(boxes)[idx]; };
        const int nk = greedy_scan(keys, n, iou_thr, n, kept_ws, keys_ws, box_of);
        if (threadIdx.x == 0) s_nk = nk;
    }
    __syncthreads();
    const int nk = s_nk;
    for (int i = threadIdx.x; i < nk; i += blockDim.x) keep[i] = (int64_t)llv::key_index(keys_ws[i]);
    if (threadIdx.x == 0) *count = nk;
}

inline size_t al(size_t v) { return (v + 255) & ~(size_t)255; }

// gfx950 has 160 KiB of LDS per CU; HIP caps dynamic LDS at 64 KiB unless the kernel opts in.
template <typename K>
inline void allow_big_lds(K kernel) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsLimit);
}
inline void nms_init_once() {
    static bool done = false;
    if (done) return;
    allow_big_lds(nms_class_kernel);
    allow_big_lds(nms_sort_kernel);
    allow_big_lds(nms_merge_kernel);
    allow_big_lds(nms_plain_kernel);
    done = true;
}

}  // namespace

extern "C" size_t llv_multiclass_nms_lds_bytes(int n, int max_num) {
    // nms_class_kernel: the class's candidate keys padded to a power of two + the kept boxes
    int np2 = 1;
    while (np2 < n) np2 <<= 1;
    return (size_t)np2 * 8 + (size_t)(max_num > 0 ? max_num : 0) * 16;
}
extern "C" size_t llv_multiclass_nms_lds_limit(void) { return kLdsLimit; }

static bool mask_path(int n, int num_classes) { return num_classes == 1 && n >= kMaskMinN; }

extern "C" size_t llv_multiclass_nms_lds_bytes_ex(int n, int max_num, int num_classes) {
    // one class and a long list: nms_sort_kernel (keys only) + the bit-mask kernels, which keep nothing of size max_num in LDS
    if (mask_path(n, num_classes)) {
        int np2 = 1;
        while (np2 < n) np2 <<= 1;
        return (size_t)np2 * 8;
    }
    return llv_multiclass_nms_lds_bytes(n, max_num);
}

static size_t mnms_ws_one(int n, int num_classes, int max_num) {
    // [maxc | cnt_in[C]] | cls_cnt[C] (memset together) | cls_keys[C][max_num] | cand[C][n]
    size_t b = al(16 + (size_t)num_classes * 4) + al((size_t)num_classes * 4) + al((size_t)num_classes * max_num * 8) +
               al((size_t)num_classes * (size_t)(n > 0 ? n : 1) * 8);
    if (mask_path(n, num_classes))                           // | sorted keys[n] | sorted boxes[n] | mask[n][row words]
        b += al((size_t)n * 8) + al((size_t)n * 16) + al((size_t)n * mask_row_words(n) * 8);
    return b;
}

extern "C" size_t llv_multiclass_nms_batched_workspace_bytes(int batch, int n, int num_classes, int max_num) {
    if (batch < 1 || n < 0 || num_classes < 1 || max_num < 1) return 0;
    return (size_t)batch * mnms_ws_one(n, num_classes, max_num);
}

extern "C" size_t llv_multiclass_nms_workspace_bytes(int n, int num_classes, int max_num) {
    return llv_multiclass_nms_batched_workspace_bytes(1, n, num_classes, max_num);
}

extern "C" int llv_multiclass_nms_batched_sf(const float* boxes, const float* scores, const float* score_factors, int batch, int n,
                                             int num_classes, int c_ld, float score_thr, float iou_thr, int max_num, float* dets,
                                             int64_t* labels, int32_t* count, int32_t* keep_flat, void* workspace,
                                             size_t workspace_bytes, void* stream) {
    return llv_multiclass_nms_batched_ex(boxes, scores, score_factors, 0, batch, n, num_classes, c_ld, score_thr, iou_thr, max_num, dets,
                                         labels, count, keep_flat, workspace, workspace_bytes, stream);
}

extern "C" int llv_multiclass_nms_batched_ex(const float* boxes, const float* scores, const float* score_factors, int class_specific,
                                             int batch, int n, int num_classes, int c_ld, float score_thr, float iou_thr, int max_num,
                                             float* dets, int64_t* labels, int32_t* count, int32_t* keep_flat, void* workspace,
                                             size_t workspace_bytes, void* stream) {
    const int cs = class_specific ? 1 : 0;
    if (!dets || !labels || !count || !keep_flat || !workspace || num_classes < 1 || num_classes > 1024 || max_num < 1 ||
        n < 0 || c_ld < num_classes || batch < 1)
        return LLV_E_BADARG;
    if (n > 0 && (!boxes || !scores)) return LLV_E_BADARG;
    if (workspace_bytes < llv_multiclass_nms_batched_workspace_bytes(batch, n, num_classes, max_num)) return LLV_E_WORKSPACE;
    if (n > kMaxN || (long)num_classes * max_num > kMaxMerge || (long)n * num_classes > 0x7fffffffL || batch > 65535) return LLV_E_UNSUPPORTED;
    if (llv_multiclass_nms_lds_bytes_ex(n, max_num, num_classes) > kLdsLimit) return LLV_E_UNSUPPORTED;   // refused here, not as a failed launch
    hipStream_t st = (hipStream_t)stream;
    nms_init_once();
    char* ws = (char*)workspace;
    const long stride = (long)mnms_ws_one(n, num_classes, max_num);
    const size_t o_cnt = al(16 + (size_t)num_classes * 4);
    const size_t o_keys = o_cnt + al((size_t)num_classes * 4);
    const size_t o_cand = o_keys + al((size_t)num_classes * max_num * 8);
    unsigned* maxc_u = (unsigned*)ws;
    int* cnt_in = (int*)(ws + 16);
    int* cls_cnt = (int*)(ws + o_cnt);
    uint64_t* cls_keys = (uint64_t*)(ws + o_keys);
    uint64_t* cand = (uint64_t*)(ws + o_cand);
    // counters of every image (small, contiguous per image): zeroed by ONE kernel, not memset nodes (and not one launch per image)
    llv::zero_words_strided(ws, o_keys, (size_t)stride, batch, st);
    if (n > 0) {
        long g = ((long)n * num_classes + 255) / 256;
        g = g > 256 ? 256 : g;
        const bool single = mask_path(n, num_classes);
        const size_t o_sk = o_cand + al((size_t)n * 8), o_sb = o_sk + al((size_t)n * 8), o_mask = o_sb + al((size_t)n * 16);
        if (single)
            hipLaunchKernelGGL(nms_presorted_kernel, dim3((n + 255) / 256, batch), dim3(256), 0, st, boxes, scores, score_factors, n, c_ld,
                               score_thr, (int*)ws, (uint64_t*)(ws + o_sk), (float4*)(ws + o_sb), stride);
        hipLaunchKernelGGL(nms_scan_kernel, dim3((int)g, batch), dim3(256), 0, st, boxes, scores, score_factors, n, num_classes, c_ld, score_thr,
                           maxc_u, cnt_in, cand, stride, cs, single ? 1 : 0);
        int np2 = 1;
        while (np2 < n) np2 <<= 1;
        const size_t lds = (size_t)np2 * 8 + (size_t)max_num * 16;
        // a class list of a few hundred candidates (the 80-class heads) gains nothing from 16 waves; the RPN's single 6000-box list does
        const int cls_threads = n >= 2048 && num_classes <= 8 ? 1024 : 256;
        if (single) {
            uint64_t* skeys = (uint64_t*)(ws + o_sk);
            float4* sbox = (float4*)(ws + o_sb);
            uint64_t* mask = (uint64_t*)(ws + o_mask);
            const int nw = mask_row_words(n), nch = (n + 63) / 64;
            hipLaunchKernelGGL(nms_sort_kernel, dim3(1, batch), dim3(1024), (size_t)np2 * 8, st, boxes, n, maxc_u, cnt_in, cand, skeys, sbox, stride);
            hipLaunchKernelGGL(nms_mask_kernel, dim3(nch, nch, batch), dim3(64), 0, st, sbox, cnt_in, mask, n, nw, iou_thr, stride);
            hipLaunchKernelGGL(nms_reduce_kernel, dim3(1, batch), dim3(64 * kRedWaves), 0, st, skeys, cnt_in, mask, n, nw, max_num, cls_cnt, cls_keys, stride);
        } else {
            hipLaunchKernelGGL(nms_class_kernel, dim3(num_classes, batch), dim3(cls_threads), lds, st, boxes, n, num_classes, iou_thr, max_num,
                               maxc_u, cnt_in, cand, cls_cnt, cls_keys, stride, cs);
        }
    }
    int mp2 = 1;
    while (mp2 < num_classes * max_num) mp2 <<= 1;
    int kp2 = 1;
    while (kp2 < max_num) kp2 <<= 1;
    hipLaunchKernelGGL(nms_merge_kernel, dim3(1, batch), dim3(1024), (size_t)(mp2 + kp2) * 8, st, boxes, n, num_classes, max_num,
                       cls_cnt, cls_keys, dets, labels, count, keep_flat, stride, cs);
    return hipGetLastError() == hipSuccess ? LLV_OK : LLV_E_LAUNCH;
}

extern "C" int llv_multiclass_nms(const float* boxes, const float* scores, int n, int num_classes, int c_ld,
                                  float score_thr, float iou_thr, int max_num, float* dets, int64_t* labels,
                                  int32_t* count, int32_t* keep_flat, void* workspace, size_t workspace_bytes,
                                  void* stream) {
    return llv_multiclass_nms_batched(boxes, scores, 1, n, num_classes, c_ld, score_thr, iou_thr, max_num, dets, labels, count,
                                      keep_flat, workspace, workspace_bytes, stream);
}

extern "C" int llv_multiclass_nms_batched(const float* boxes, const float* scores, int batch, int n, int num_classes, int c_ld,
                                          float score_thr, float iou_thr, int max_num, float* dets, int64_t* labels,
                                          int32_t* count, int32_t* keep_flat, void* workspace, size_t workspace_bytes,
                                          void* stream) {
    return llv_multiclass_nms_batched_sf(boxes, scores, nullptr, batch, n, num_classes, c_ld, score_thr, iou_thr, max_num, dets, labels,
                                         count, keep_flat, workspace, workspace_bytes, stream);
}

extern "C" size_t llv_nms_workspace_bytes(int n) {
    if (n < 0) return 0;
    return al((size_t)(n > 0 ? n : 1) * 16) + al((size_t)(n > 0 ? n : 1) * 8);
}

extern "C" int llv_nms(const float* boxes, const float* scores, int n, float iou_thr, int64_t* keep,
                       int32_t* count, void* workspace, size_t workspace_bytes, void* stream) {
    if (!keep || !count || !workspace || n < 0) return LLV_E_BADARG;
    if (n > kMaxN) return LLV_E_UNSUPPORTED;
    if (workspace_bytes < llv_nms_workspace_bytes(n)) return LLV_E_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    if (n == 0) { llv::zero_words(count, 4, st); return hipGetLastError() == hipSuccess ? LLV_OK : LLV_E_LAUNCH; }
    if (!boxes || !scores) return LLV_E_BADARG;
    nms_init_once();
    char* ws = (char*)workspace;
    int np2 = 1;
    while (np2 < n) np2 <<= 1;
    hipLaunchKernelGGL(nms_plain_kernel, dim3(1), dim3(256), (size_t)np2 * 8, st, boxes, scores, n, iou_thr, keep,
                       count, (float4*)ws, (uint64_t*)(ws + al((size_t)n * 16)));
    return hipGetLastError() == hipSuccess ? LLV_OK : LLV_E_LAUNCH;
}

unsigned int* llv_nms_guard_addr_() {
    void* p = nullptr;
    return hipGetSymbolAddress(&p, HIP_SYMBOL(g_guard_trips)) == hipSuccess ? (unsigned int*)p : nullptr;
}

unsigned int llv_nms_guard_trips_(int reset) {
    unsigned int v = 0;
    (void)hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_guard_trips), sizeof(v));
    if (reset) {
        const unsigned int z = 0;
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_guard_trips), &z, sizeof(z));
    }
    return v;
}

// ------------------------------------------------------------------------------------------------
// YOLOV3Head._get_bboxes between the per-anchor decode and the NMS (yolo_head.py:281-335), one workgroup per (level, image):
// the nms_pre anchors with the highest objectness (conf.topk(nms_pre), :283-292) by an 8-pass radix select over (conf, index) keys and a
// bitonic sort of the selected ones (torch.topk returns them best first), then the gather of their boxes, objectness and class
// scores = sigmoid(class logits) into the concatenated candidate lists the NMS reads, with the conf_thr rule (:320-327: a row whose
// objectness is below conf_thr is dropped == all its class scores are zero) and the rescale by scale_factor (:316-318) folded in.
// A level with at most nms_pre anchors keeps them all, in anchor order.
// ------------------------------------------------------------------------------------------------
namespace {

struct YoloSelArgs {
    const float* pred;      // [batch][h*w][ld] fp32, channel a*na + attr
    const float* boxes;     // [batch][n][4] (llv_yolo_decode)
    const float* conf;      // [batch][n]
    int n, A, na, nc, ld, m;            // n = h*w*A anchors of the level, m = min(nms_pre, n) kept
    long out0;              // first row of this level in the candidate lists
    long n_total;           // rows of the candidate lists per image
    float* out_boxes; float* out_conf; float* out_scores;
    uint64_t* keys;         // workspace: [batch][n]
    float conf_thr;
    const float* inv_scale; // [batch][4] or null
};

__global__ __launch_bounds__(1024) void yolo_select_kernel(const YoloSelArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
    const int img = blockIdx.x, tid = threadIdx.x;
    const int mp = llv::next_pow2(a.m);
    uint64_t* sel = reinterpret_cast<uint64_t*>(sm);                       // [mp]
    unsigned* hist = reinterpret_cast<unsigned*>(sel + mp);                // [256]
    uint64_t* sh = reinterpret_cast<uint64_t*>(hist + 256);                // [2]
    const float* conf = a.conf + (long)img * a.n;
    const bool all = a.m >= a.n;
    if (!all) {
        uint64_t* keys = a.keys + (long)img * a.n;
        for (int i = tid; i < a.n; i += blockDim.x) keys[i] = llv::make_key(conf[i], (uint32_t)i);
        __syncthreads();
        llv::block_select_top(keys, a.n, a.m, sel, hist, sh);
        for (int i = a.m + tid; i < mp; i += blockDim.x) sel[i] = 0;
        __syncthreads();
        llv::block_sort_desc(sel, mp, a.m);
    }
    const float* boxes = a.boxes + (long)img * a.n * 4;
    const float* pred = a.pred + (long)img * (a.n / a.A) * a.ld;
    float* ob = a.out_boxes + ((long)img * a.n_total + a.out0) * 4;
    float* oc = a.out_conf + (long)img * a.n_total + a.out0;
    float* os = a.out_scores + ((long)img * a.n_total + a.out0) * a.nc;
    float4 inv = make_float4(1.f, 1.f, 1.f, 1.f);
    if (a.inv_scale) inv = reinterpret_cast<const float4*>(a.inv_scale)[img];
    // boxes + objectness: one thread per kept row; class scores: one thread per (row, class)
    for (int j = tid; j < a.m; j += blockDim.x) {
        int idx = all ? j : (int)llv::key_index(sel[j]);
        if ((unsigned)idx >= (unsigned)a.n) { LLV_GUARD_TRIP(); idx = 0; }
        const float4 b = reinterpret_cast<const float4*>(boxes)[idx];
        reinterpret_cast<float4*>(ob)[j] = a.inv_scale ? make_float4(b.x / inv.x, b.y / inv.y, b.z / inv.z, b.w / inv.w) : b;
        oc[j] = conf[idx];
    }
    const int total = a.m * a.nc;
    for (int e = tid; e < total; e += blockDim.x) {
        const int j = e / a.nc, c = e - j * a.nc;
        int idx = all ? j : (int)llv::key_index(sel[j]);
        if ((unsigned)idx >= (unsigned)a.n) idx = 0;
        const int cell = idx / a.A, an = idx - cell * a.A;
        const float x = pred[(long)cell * a.ld + an * a.na + 5 + c];
        const float sc = 1.f / (1.f + expf(-x));
        os[(long)j * a.nc + c] = (a.conf_thr > 0.f && !(conf[idx] >= a.conf_thr)) ? 0.f : sc;
    }
}

}  // namespace

extern "C" size_t llv_yolo_select_workspace_bytes(int batch, int n) { return batch > 0 && n > 0 ? (size_t)batch * n * 8 : 0; }

extern "C" int llv_yolo_select(const float* pred, const float* boxes, const float* conf, int batch, int n, int num_anchors, int num_attrib,
                               int num_classes, int ld, int nms_pre, float conf_thr, const float* inv_scale, int64_t out_row0, int64_t n_total,
                               float* out_boxes, float* out_conf, float* out_scores, void* workspace, size_t workspace_bytes, void* stream) {
    if (!pred || !boxes || !conf || !out_boxes || !out_conf || !out_scores || batch < 1 || n < 1 || num_anchors < 1 || n % num_anchors) return LLV_E_BADARG;
    if (num_attrib < 5 + num_classes || num_classes < 1 || ld < num_anchors * num_attrib || out_row0 < 0) return LLV_E_BADARG;
    const int m = (nms_pre > 0 && nms_pre < n) ? nms_pre : n;
    if (out_row0 + m > n_total) return LLV_E_BADARG;
    YoloSelArgs a{pred, boxes, conf, n, num_anchors, num_attrib, num_classes, ld, m, (long)out_row0, (long)n_total, out_boxes, out_conf, out_scores,
                  (uint64_t*)workspace, conf_thr, inv_scale};
    size_t lds = 0;
    if (m < n) {
        if (!workspace || workspace_bytes < llv_yolo_select_workspace_bytes(batch, n)) return LLV_E_WORKSPACE;
        int mp = 1;
        while (mp < m) mp <<= 1;
        lds = (size_t)mp * 8 + 256 * 4 + 16;
        if (lds > kLdsLimit) return LLV_E_UNSUPPORTED;
        if (lds > 48 * 1024 &&
            hipFuncSetAttribute((const void*)yolo_select_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsLimit) != hipSuccess)
            return LLV_E_LAUNCH;
    }
    hipLaunchKernelGGL(yolo_select_kernel, dim3(batch), dim3(1024), lds, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? LLV_OK : LLV_E_LAUNCH;
}
