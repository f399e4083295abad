// sort_util.h -- block-level helpers shared by decode.hip and nms.hip (LDS bitonic sort, scans).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

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
// block must call. Keys equal to 0 are padding and sink to the end.
__device__ inline void bitonic_sort_desc(uint64_t* keys, int n) {
    for (int k = 2; k <= n; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
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
    }
}

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

}  // namespace llv
