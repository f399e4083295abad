// ta_probe: what one CU's vector-memory path delivers for LDS-DMA pieces (buffer_load_dwordx4 ... lds, 1 KiB per wave-instruction)
// as a function of how many bytes of each piece are contiguous in memory (64 B = a 32-channel pixel chunk, 128 B = a 64-channel one,
// ... 1024 B = a fully contiguous piece), and for buffer_store_dwordx4 of the same shapes. L2-resident working set (8 MiB), one
// workgroup of 4 waves per CU (or 3 with -DWGS=3).   build: hipcc --offload-arch=gfx950 -O3 tools/ta_probe.hip -o /tmp/ta_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#ifndef WGS
#define WGS 1
#endif
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t r, unsigned char* lds, int voff, int soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds, 16, voff, soff, 0, 0);
}
template <int ROWB, bool STORE>
__global__ __launch_bounds__(256, WGS) void probe(unsigned char* buf, unsigned bytes, int iters, unsigned long long* ticks) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[16 * 1024];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(buf, (short)0, (int)bytes, 0x00020000);
    constexpr int LPR = ROWB / 16;                           // lanes per contiguous run
    // runs are 4 KiB apart (a wide NHWC pixel row), the piece index walks the buffer
    const int lane_off = (lane / LPR) * 4096 + (lane % LPR) * 16;
    const unsigned span = (64 / LPR) * 4096;                 // bytes of address space one piece spans
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 v = {1u, 2u, 3u, (unsigned)tid};
    const unsigned long long t0 = __builtin_readcyclecounter();
    unsigned pos = (blockIdx.x * 4 + wave) * span * 7;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            pos = (pos + span * 29) % (bytes - span);
            pos &= ~15u;
            if (STORE) __builtin_amdgcn_raw_buffer_store_b128(v, r, (int)pos + lane_off, (k & 3) * ROWB, 0);
            else dma16(r, smem + wave * 4096 + (k & 3) * 1024, (int)pos + lane_off, (k & 3) * ROWB);
        }
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (tid == 0) ticks[blockIdx.x] = __builtin_readcyclecounter() - t0;
}
template <int ROWB, bool STORE>
static void run(unsigned char* buf, unsigned bytes, unsigned long long* ticks, int nblk) {
    const int iters = 2000;
    hipLaunchKernelGGL((probe<ROWB, STORE>), dim3(nblk), dim3(256), 0, 0, buf, bytes, 200, ticks);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((probe<ROWB, STORE>), dim3(nblk), dim3(256), 0, 0, buf, bytes, iters, ticks);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> t(nblk);
    hipMemcpy(t.data(), ticks, nblk * 8, hipMemcpyDeviceToHost);
    double avg = 0; for (auto x : t) avg += x; avg /= nblk;
    const double pieces = (double)iters * 8 * 4 * WGS;        // per CU
    printf("%s contiguous %4d B: %7.1f cycles per 1-KiB piece per CU  (%5.1f B/cycle/CU, %5.2f TB/s chip, %.3f ms)\n", STORE ? "store" : "dma  ",
           ROWB, avg / pieces * WGS, pieces * 1024 / avg / WGS, (double)nblk * iters * 8 * 4 * 1024 / (ms * 1e-3) / 1e12, ms);
}
int main() {
    const unsigned bytes = 8u << 20;
    unsigned char* buf; unsigned long long* ticks;
    const int nblk = 256 * WGS;
    hipMalloc(&buf, bytes); hipMemset(buf, 1, bytes); hipMalloc(&ticks, nblk * 8);
    printf("workgroups per CU: %d\n", WGS);
    run<64, false>(buf, bytes, ticks, nblk); run<128, false>(buf, bytes, ticks, nblk); run<256, false>(buf, bytes, ticks, nblk); run<1024, false>(buf, bytes, ticks, nblk);
    run<64, true>(buf, bytes, ticks, nblk); run<128, true>(buf, bytes, ticks, nblk); run<256, true>(buf, bytes, ticks, nblk); run<1024, true>(buf, bytes, ticks, nblk);
    return 0;
}
