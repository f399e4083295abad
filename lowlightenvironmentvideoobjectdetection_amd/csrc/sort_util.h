// sort_util.h -- block-level helpers shared by decode.hip and nms.hip (LDS bitonic sort, scans).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

namespace llv {

// Monotone map float -> uint32 (larger float => larger uint), total over all non-NaN floats.
__device__ __forceinline__ uint32_t f32_ordered(float f) {
    const uint32_t b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float f32_unordered(uint32_t u) {
    const uint32_t b = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
    return __uint_as_float(b);
}

// (score desc, index asc) as one descending 64-bit key.
__device__ __forceinline__ uint64_t make_key(float score, uint32_t idx) {
    return ((uint64_t)f32_ordered(score) << 32) | (uint64_t)(0xffffffffu - idx);
}
__device__ __forceinline__ uint32_t key_index(uint64_t k) { return 0xffffffffu - (uint32_t)(k & 0xffffffffu); }
__device__ __forceinline__ float key_score(uint64_t k) { return f32_unordered((uint32_t)(k >> 32)); }

// In-place bitonic sort of keys[0..n) (n a power of two) in LDS, DESCENDING. All threads of the
// block must call (blockDim.x a multiple of 64). Keys equal to 0 are padding and sink to the end.
// The compare-exchange network is the textbook one; what differs from a pass-per-(k, j) loop is where the exchanges happen: partners
// less than 64 apart sit in one wavefront (element i = 64 * m + lane), so the last min(log2 k, 6) steps of every stage run in
// registers through __shfl_xor with a single LDS round trip and ONE barrier, instead of one LDS pass and one barrier each:
// 8192 keys take 36 barriers instead of 91 (round 3; the RPN's 6000-candidate lists are sorted twice per frame).
__device__ inline void bitonic_sort_desc(uint64_t* keys, int n) {
    const int lane = threadIdx.x & 63;
    for (int k = 2; k <= n; k <<= 1) {
        int j = k >> 1;
        for (; j >= 64; j >>= 1) {
            for (int i = threadIdx.x; i < n; i += blockDim.x) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const uint64_t a = keys[i], b = keys[ixj];
                    const bool desc = (i & k) == 0;
                    if (desc ? (a < b) : (a > b)) { keys[i] = b; keys[ixj] = a; }
                }
            }
            __syncthreads();
        }
        // j < 64: the remaining steps j, j/2, .., 1 of this stage inside the wavefront
        for (int base = threadIdx.x - lane; base < n; base += blockDim.x) {      // wave-uniform trip count
            const int i = base + lane;
            uint64_t v = i < n ? keys[i] : 0ull;             // n < 64: lanes past n pair only with each other
            const bool desc = (i & k) == 0;
            for (int jj = j; jj > 0; jj >>= 1) {
                const uint64_t o = (uint64_t)__shfl_xor((unsigned long long)v, jj);
                const bool lower = (i & jj) == 0;            // the lower index of the pair keeps the larger key in a descending run
                const bool take_max = lower == desc;
                v = take_max ? (v > o ? v : o) : (v < o ? v : o);
            }
            if (i < n) keys[i] = v;
        }
        __syncthreads();
    }
}

// A merge sort for lists of 2..16 keys per thread, DESCENDING, in place in LDS: keys[0 .. IPT * blockDim.x). Every thread sorts its IPT
// consecutive keys in registers, then log2(blockDim.x) rounds merge neighbouring runs: thread t owns outputs [IPT * t, IPT * t + IPT) of
// its pair of runs, finds where they start in the two runs by a binary search along the merge path's diagonal and merges IPT keys
// serially out of LDS. O(n log n) key moves -- 8192 keys: 10 rounds, where the bitonic network above runs 91 compare-exchange passes
// over all of them (the RPN's 6000-candidate lists are sorted per frame, one workgroup each time). The first six rounds stay inside a
// wavefront's own 64 * IPT keys and need no workgroup barrier. IPT need not be a power of two: 6000 keys sort as 6 x 1024, not as 8192.
template <int IPT>
__device__ inline void merge_sort_desc_t(uint64_t* keys) {
    const int o = IPT * (int)threadIdx.x;
    const int n = IPT * (int)blockDim.x;
    uint64_t v[IPT];
#pragma unroll
    for (int s = 0; s < IPT; ++s) v[s] = keys[o + s];
#pragma unroll
    for (int r = 0; r < IPT; ++r) {
#pragma unroll
        for (int s = r & 1; s + 1 < IPT; s += 2) {
            const uint64_t a = v[s], b = v[s + 1];
            v[s] = a > b ? a : b;
            v[s + 1] = a > b ? b : a;
        }
    }
#pragma unroll
    for (int s = 0; s < IPT; ++s) keys[o + s] = v[s];
    __builtin_amdgcn_wave_barrier();
    int sh = 0;                                              // run length L = IPT << sh, threads per pair 2 << sh
    for (int L = IPT; L < n; L <<= 1, ++sh) {
        const bool in_wave = sh < 6;                         // the pair's 2 << sh threads sit in one wavefront
        const int P = (int)(threadIdx.x >> (sh + 1)) * 2 * L;
        const uint64_t* A = keys + P;
        const uint64_t* B = A + L;
        const int d = o - P;                                 // this thread's outputs start d keys into the merged pair
        int lo = d > L ? d - L : 0, hi = d < L ? d : L;      // how many of those d come from A
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (A[mid] >= B[d - 1 - mid]) lo = mid + 1; else hi = mid;
        }
        int ia = lo, ib = d - lo;
        uint64_t ka = ia < L ? A[ia] : 0ull, kb = ib < L ? B[ib] : 0ull;
#pragma unroll
        for (int s = 0; s < IPT; ++s) {
            const bool ta = ib >= L || (ia < L && ka >= kb);
            v[s] = ta ? ka : kb;
            if (ta) { ++ia; ka = ia < L ? A[ia] : 0ull; }
            else { ++ib; kb = ib < L ? B[ib] : 0ull; }
        }
        if (in_wave) __builtin_amdgcn_wave_barrier(); else __syncthreads();
#pragma unroll
        for (int s = 0; s < IPT; ++s) keys[o + s] = v[s];
        if (in_wave && sh < 5) __builtin_amdgcn_wave_barrier(); else __syncthreads();     // the next round reaches across wavefronts
    }
    __syncthreads();
}

// keys[0..n) descending; n a power of two with padding keys 0 behind the cnt <= n real ones; all threads of the block call.
// 2..16 keys per thread: the merge sort over the first ceil(cnt / blockDim) * blockDim keys (the rest is padding, already in place).
__device__ inline void block_sort_desc(uint64_t* keys, int n, int cnt) {
    const int bd = (int)blockDim.x;
    const int ipt = (cnt + bd - 1) / bd;
    if (n < 2 * bd || ipt > 16 || (bd & (bd - 1))) { bitonic_sort_desc(keys, n); return; }
    switch (ipt < 2 ? 2 : ipt) {
        case 2: merge_sort_desc_t<2>(keys); break;
        case 3: merge_sort_desc_t<3>(keys); break;
        case 4: merge_sort_desc_t<4>(keys); break;
        case 5: merge_sort_desc_t<5>(keys); break;
        case 6: merge_sort_desc_t<6>(keys); break;
        case 7: merge_sort_desc_t<7>(keys); break;
        case 8: merge_sort_desc_t<8>(keys); break;
        case 9: merge_sort_desc_t<9>(keys); break;
        case 10: merge_sort_desc_t<10>(keys); break;
        case 11: merge_sort_desc_t<11>(keys); break;
        case 12: merge_sort_desc_t<12>(keys); break;
        case 13: merge_sort_desc_t<13>(keys); break;
        case 14: merge_sort_desc_t<14>(keys); break;
        case 15: merge_sort_desc_t<15>(keys); break;
        default: merge_sort_desc_t<16>(keys); break;
    }
}
__device__ inline void block_sort_desc(uint64_t* keys, int n) { block_sort_desc(keys, n, n); }

__device__ __forceinline__ int next_pow2(int v) {
    int p = 1;
    while (p < v) p <<= 1;
    return p;
}

// Block-wide exclusive scan of one int per thread (blockDim.x <= 1024, multiple of 64).
// `scratch` needs 17 ints of LDS. Returns the exclusive prefix; *total gets the block sum.
__device__ inline int block_exclusive_scan(int v, int* scratch, int* total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    int incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(incl, o);
        if (lane >= o) incl += t;
    }
    if (lane == 63) scratch[wave] = incl;
    __syncthreads();
    if (threadIdx.x == 0) {
        int s = 0;
        for (int w = 0; w < nw; ++w) { const int t = scratch[w]; scratch[w] = s; s += t; }
        scratch[16] = s;
    }
    __syncthreads();
    const int base = scratch[wave];
    *total = scratch[16];
    __syncthreads();
    return base + incl - v;
}


// The m largest of n DISTINCT 64-bit keys src[0..n) (global or LDS) -> dst[0..m) in LDS, unordered; 1 <= m <= n.
// 8 x 8-bit radix select, most significant digit first: per pass an LDS histogram of the digit over the keys that still
// match the prefix, then wave 0 walks the 256 bins from the top (4 bins per lane, suffix sums by shuffles) to the bin that
// holds the m-th largest key. All threads of the block call; hist = 256 unsigned and sh = 2 uint64 of LDS.
// Cost is O(8 * n / blockDim) instead of the O(n log^2 n / blockDim) of a bitonic sort: the caller sorts only dst.
__device__ inline void block_select_top(const uint64_t* src, int n, int m, uint64_t* dst, unsigned* hist, uint64_t* sh) {
    const int tid = threadIdx.x, lane = tid & 63;
    uint64_t prefix = 0;
    unsigned need = (unsigned)m;
    for (int pass = 0; pass < 8; ++pass) {
        const int shift = 56 - 8 * pass;
        for (int i = tid; i < 256; i += blockDim.x) hist[i] = 0;
        __syncthreads();
        for (int i = tid; i < n; i += blockDim.x) {
            const uint64_t key = src[i];
            if (pass == 0 || (key >> (shift + 8)) == prefix) atomicAdd(&hist[(unsigned)((key >> shift) & 255ull)], 1u);
        }
        __syncthreads();
        if (tid < 64) {
            const int b0 = 255 - 4 * lane;                       // this lane's bins b0, b0-1, b0-2, b0-3 (descending)
            const unsigned h0 = hist[b0], h1 = hist[b0 - 1], h2 = hist[b0 - 2], h3 = hist[b0 - 3];
            const unsigned s = h0 + h1 + h2 + h3;
            unsigned incl = s;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const unsigned t = __shfl_up(incl, o);
                if (lane >= o) incl += t;
            }
            unsigned cum = incl - s;                             // keys in bins above this lane's
            if (cum < need && need <= incl) {                    // exactly one lane: the matching keys number >= need
                int b = b0;
                if (cum + h0 < need) { cum += h0; b = b0 - 1;
                    if (cum + h1 < need) { cum += h1; b = b0 - 2;
                        if (cum + h2 < need) { cum += h2; b = b0 - 3; } } }
                sh[0] = (prefix << 8) | (uint64_t)b;
                sh[1] = (uint64_t)(need - cum);
            }
        }
        __syncthreads();
        prefix = sh[0];
        need = (unsigned)sh[1];
        __syncthreads();
    }
    // prefix is the m-th largest key; the keys are distinct, so exactly m of them are >= it
    unsigned* cnt = hist;
    if (tid == 0) *cnt = 0;
    __syncthreads();
    for (int i = tid; i < n; i += blockDim.x) {
        const uint64_t key = src[i];
        if (key >= prefix) {
            const unsigned p = atomicAdd(cnt, 1u);
            if (p < (unsigned)m) dst[p] = key;                   // keys that are NOT distinct (corrupt input) cannot overrun dst
        }
    }
    __syncthreads();
}

// Capacity guards. Every data-dependent index of the post-processing kernels (list positions drawn from atomic counters,
// anchor / box indices read back from memory) is clamped to the capacity of the buffer it addresses, and each clamp that
// actually fires is counted in a per-module device word (llv_guard_trips reads the sum). With consistent inputs no guard
// can fire; a non-zero count means the kernels were fed counters that were not zeroed or lists that changed between two
// passes -- i.e. launches of one step ran out of order or overlapped -- which without the guards is an out-of-bounds store.
#define LLV_GUARD_DECL static __device__ unsigned int g_guard_trips = 0u;
#define LLV_GUARD_TRIP() atomicAdd(&g_guard_trips, 1u)

// zero n 32-bit words (a KERNEL node where hipMemsetAsync would put a memset node into a captured graph)
static __global__ void zero_words_kernel(unsigned* p, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = 0u;
}
static inline int zero_words(void* p, size_t bytes, hipStream_t st) {
    const size_t n = bytes / 4;
    if (n == 0) return 0;
    // LLV_GRAPH_MEMNODES=1 (diagnosis only): round 1's behaviour, a memset node inside the captured step
    static const bool memnodes = getenv("LLV_GRAPH_MEMNODES") != nullptr && atoi(getenv("LLV_GRAPH_MEMNODES")) != 0;
    if (memnodes) return hipMemsetAsync(p, 0, bytes, st) == hipSuccess ? 0 : -1;
    size_t g = (n + 255) / 256;
    g = g > 1024 ? 1024 : g;
    hipLaunchKernelGGL(zero_words_kernel, dim3((unsigned)g), dim3(256), 0, st, (unsigned*)p, n);
    return 0;
}

// the same for `count` regions of `bytes` bytes, `stride` bytes apart: one launch (blockIdx.y = region) instead of one per image
static __global__ void zero_words_strided_kernel(unsigned char* p, size_t n, size_t stride) {
    unsigned* q = reinterpret_cast<unsigned*>(p + (size_t)blockIdx.y * stride);
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) q[i] = 0u;
}
static inline int zero_words_strided(void* p, size_t bytes, size_t stride, int count, hipStream_t st) {
    const size_t n = bytes / 4;
    if (n == 0 || count < 1) return 0;
    if (count == 1) return zero_words(p, bytes, st);
    size_t g = (n + 255) / 256;
    g = g > 64 ? 64 : g;
    hipLaunchKernelGGL(zero_words_strided_kernel, dim3((unsigned)g, (unsigned)count), dim3(256), 0, st, (unsigned char*)p, n, stride);
    return 0;
}

}  // namespace llv
