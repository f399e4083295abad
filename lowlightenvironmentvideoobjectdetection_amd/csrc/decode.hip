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
};

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

template <typename T>
__global__ void class_max_kernel(const DecK k) {
    const int total = k.a0[k.n_levels];
    for (int g = blockIdx.x * blockDim.x + threadIdx.x; g < total; g += gridDim.x * blockDim.x) {
        const int l = level_of_anchor(k, g);
        const int loc = g - k.a0[l];
        const int cell = loc / k.A, an = loc - cell * k.A;
        const long base = (k.cls_off[l] + cell) * (long)k.cls_ld + (long)an * k.C;
        float m = 0.f;  // sigmoid > 0
        const T* p = reinterpret_cast<const T*>(k.cls) + base;
        for (int c = 0; c < k.C; ++c) m = fmaxf(m, sigmoidf_((float)p[c]));
        k.ms[g] = m;
    }
}

// one block (1024 threads) per level
__global__ __launch_bounds__(1024) void topk_select_kernel(const DecK k) {
    const int l = blockIdx.x;
    const int n = k.a0[l + 1] - k.a0[l];
    const int kk = k.c0[l + 1] - k.c0[l];
    const float* ms = k.ms + k.a0[l];
    int* sel = k.sel + k.c0[l];
    const int tid = threadIdx.x;
    if (kk >= n) {  // no top-k on this level: anchor order is kept (anchor_head.py:657-658)
        for (int i = tid; i < n; i += blockDim.x) sel[i] = k.a0[l] + i;
        return;
    }
    __shared__ unsigned hist[256];
    __shared__ unsigned sh_prefix, sh_need;
    __shared__ int wcnt_gt[16], wcnt_eq[16];
    extern __shared__ __attribute__((aligned(16))) uint64_t keys[];  // next_pow2(kk)

    unsigned prefix = 0, need = (unsigned)kk;
    for (int pass = 0; pass < 4; ++pass) {
        const int shift = 24 - 8 * pass;
        for (int i = tid; i < 256; i += blockDim.x) hist[i] = 0;
        __syncthreads();
        for (int i = tid; i < n; i += blockDim.x) {
            const unsigned key = llv::f32_ordered(ms[i]);
            const bool match = pass == 0 ? true : ((key >> (shift + 8)) == prefix);
            if (match) atomicAdd(&hist[(key >> shift) & 255u], 1u);
        }
        __syncthreads();
        if (tid == 0) {
            unsigned cum = 0, nd = need;
            int b = 255;
            for (; b > 0; --b) {
                if (cum + hist[b] >= nd) break;
                cum += hist[b];
            }
            sh_need = nd - cum;
            sh_prefix = (prefix << 8) | (unsigned)b;
        }
        __syncthreads();
        prefix = sh_prefix;
        need = sh_need;
        __syncthreads();
    }
    const unsigned T = prefix;  // key of the kk-th largest score; take `need` of its ties, lowest index first

    const int np2 = llv::next_pow2(kk);
    for (int i = tid; i < np2; i += blockDim.x) keys[i] = 0;
    __syncthreads();

    const int lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
    int run_take = 0, run_eq = 0;
    for (int base = 0; base < n; base += blockDim.x) {
        const int i = base + tid;
        unsigned key = 0;
        bool gt = false, eq = false;
        if (i < n) {
            key = llv::f32_ordered(ms[i]);
            gt = key > T;
            eq = key == T;
        }
        const unsigned long long bgt = __ballot(gt), beq = __ballot(eq);
        if (lane == 0) { wcnt_gt[wave] = __popcll(bgt); wcnt_eq[wave] = __popcll(beq); }
        __syncthreads();
        int pos = run_take, eqr = run_eq, tot_take = run_take, tot_eq = run_eq;
        for (int w = 0; w < nw; ++w) {
            const int e = wcnt_eq[w], g2 = wcnt_gt[w];
            int avail = (int)need - tot_eq;
            avail = avail < 0 ? 0 : avail;
            const int te = e < avail ? e : avail;
            if (w < wave) { pos += g2 + te; eqr += e; }
            tot_take += g2 + te;
            tot_eq += e;
        }
        const unsigned long long below = (1ull << lane) - 1ull;
        const int my_eq_rank = eqr + __popcll(beq & below);
        const bool take_eq = eq && my_eq_rank < (int)need;
        // lanes of this wave taken before me: gt lanes below + taken-eq lanes below
        int avail_w = (int)need - eqr;
        avail_w = avail_w < 0 ? 0 : avail_w;
        int eq_below = __popcll(beq & below);
        eq_below = eq_below < avail_w ? eq_below : avail_w;
        const int my_pos = pos + __popcll(bgt & below) + eq_below;
        if (gt || take_eq) keys[my_pos] = ((uint64_t)key << 32) | (uint64_t)(0xffffffffu - (unsigned)i);
        run_take = tot_take;
        run_eq = tot_eq;
        __syncthreads();
    }
    llv::bitonic_sort_desc(keys, np2);
    for (int i = tid; i < kk; i += blockDim.x) sel[i] = k.a0[l] + (int)llv::key_index(keys[i]);
}

template <typename T>
__global__ void gather_decode_kernel(const DecK k) {
    const int slot = blockIdx.x;
    int l = 0;
#pragma unroll
    for (int s = 1; s < LLV_MAX_SEG; ++s)
        if (s < k.n_levels && slot >= k.c0[s]) l = s;
    const int g = k.sel[slot];
    const int loc = g - k.a0[l];
    const int cell = loc / k.A, an = loc - cell * k.A;
    const T* pc = reinterpret_cast<const T*>(k.cls) + (k.cls_off[l] + cell) * (long)k.cls_ld + (long)an * k.C;
    for (int c = threadIdx.x; c < k.C; c += blockDim.x) k.scores[(long)slot * k.C + c] = sigmoidf_((float)pc[c]);
    if (threadIdx.x == 0) {
        k.cand_anchor[slot] = g;
        const int cy = cell / k.w[l], cx = cell - cy * k.w[l];
        const float shx = (float)(cx * k.sw[l]), shy = (float)(cy * k.sh[l]);
        const float* b = k.base + ((long)l * k.A + an) * 4;
        const float ax1 = b[0] + shx, ay1 = b[1] + shy, ax2 = b[2] + shx, ay2 = b[3] + shy;
        const T* pr = reinterpret_cast<const T*>(k.reg) + (k.reg_off[l] + cell) * (long)k.reg_ld + (long)an * 4;
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
            const float mx[4] = {k.img_w, k.img_h, k.img_w, k.img_h};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                o[e] = o[e] < 0.f ? 0.f : o[e];
                o[e] = o[e] > mx[e] ? mx[e] : o[e];
            }
        }
        if (k.rescale) {
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = o[e] / k.sf[e];
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) k.boxes[(long)slot * 4 + e] = o[e];
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

extern "C" size_t llv_retina_decode_workspace_bytes(const llv_decode_desc* d) {
    DecK k;
    if (fill(d, k) < 0) return 0;
    const size_t a = ((size_t)k.a0[d->n_levels] * 4 + 255) & ~(size_t)255;
    const size_t c = ((size_t)k.c0[d->n_levels] * 4 + 255) & ~(size_t)255;
    return a + c;
}

extern "C" int llv_retina_decode_topk(const llv_decode_desc* d, void* stream) {
    DecK k;
    int rc = fill(d, k);
    if (rc < 0) return rc;
    if (!d->cls || !d->reg || !d->base_anchors || !d->boxes || !d->scores || !d->cand_anchor || !d->workspace) return LLV_E_BADARG;
    if (d->workspace_bytes < llv_retina_decode_workspace_bytes(d)) return LLV_E_WORKSPACE;
    if (d->cls_ld < k.A * k.C || d->reg_ld < k.A * 4) return LLV_E_BADARG;
    if (d->nms_pre > 4096) return LLV_E_UNSUPPORTED;
    if (d->n_cand != k.c0[d->n_levels]) return LLV_E_BADARG;
    k.cls_ld = d->cls_ld; k.reg_ld = d->reg_ld; k.f32 = d->logits_f32;
    k.cls = d->cls; k.reg = d->reg; k.base = d->base_anchors;
    for (int e = 0; e < 4; ++e) { k.means[e] = d->means[e]; k.stds[e] = d->stds[e]; k.sf[e] = d->scale_factor[e]; }
    k.max_ratio = d->max_ratio; k.clip = d->clip; k.img_h = d->img_h; k.img_w = d->img_w; k.rescale = d->rescale;
    k.boxes = d->boxes; k.scores = d->scores; k.cand_anchor = d->cand_anchor;
    const size_t a = ((size_t)k.a0[d->n_levels] * 4 + 255) & ~(size_t)255;
    k.ms = (float*)d->workspace;
    k.sel = (int*)((char*)d->workspace + a);
    hipStream_t st = (hipStream_t)stream;
    const int total = k.a0[d->n_levels];
    int grid = (total + 255) / 256;
    grid = grid > 4096 ? 4096 : grid;
    if (k.f32) hipLaunchKernelGGL((class_max_kernel<float>), dim3(grid), dim3(256), 0, st, k);
    else hipLaunchKernelGGL((class_max_kernel<_Float16>), dim3(grid), dim3(256), 0, st, k);
    int np2 = 1;
    while (np2 < (d->nms_pre > 0 ? d->nms_pre : 1)) np2 <<= 1;
    hipLaunchKernelGGL(topk_select_kernel, dim3(d->n_levels), dim3(1024), (size_t)np2 * 8, st, k);
    if (k.f32) hipLaunchKernelGGL((gather_decode_kernel<float>), dim3(d->n_cand), dim3(128), 0, st, k);
    else hipLaunchKernelGGL((gather_decode_kernel<_Float16>), dim3(d->n_cand), dim3(128), 0, st, k);
    return hipGetLastError() == hipSuccess ? LLV_OK : LLV_E_LAUNCH;
}
