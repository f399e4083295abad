// train_ops.hip -- training-side kernels of the dense head: sigmoid focal loss (fwd/bwd) and the
// MaxIoUAssigner (IoU matrix + both arg-maxes + low-quality matching, no k x n matrix in HBM).
//
// Reference:
//   mmcv.ops.sigmoid_focal_loss (mmcv-full 1.3.0, un-vendored) called from
//     mmdetection/mmdet/models/losses/focal_loss.py:59-98; CPU restatement py_sigmoid_focal_loss :11-56
//   MaxIoUAssigner.assign_wrt_overlaps  mmdetection/mmdet/core/bbox/assigners/max_iou_assigner.py:134-212
//   bbox_overlaps (mode='iou', eps=1e-6) mmdetection/mmdet/core/bbox/iou_calculators/iou2d_calculator.py:96-159
// Built with -ffp-contract=off (IoU values are compared for equality in the low-quality match).

#include <hip/hip_runtime.h>
#include <float.h>
#include "llv.h"
#include "sort_util.h"

namespace {

__global__ void focal_fwd_kernel(const float* __restrict__ x, const int64_t* __restrict__ t, float* __restrict__ out,
                                 long n, int c, float gamma, float alpha) {
    const long total = n * c;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / c;
        const int cls = (int)(i - r * c);
        const bool pos = t[r] == (int64_t)cls;
        const float xi = x[i];
        const float p = 1.f / (1.f + expf(-xi));
        // log(p) and log(1 - p) as -softplus(-x) and -softplus(x): the reference's CPU path (py_sigmoid_focal_loss, focal_loss.py:11-56:
        // binary_cross_entropy_with_logits times the modulating factor) is exact for any logit, whereas the published mmcv CUDA form
        // log(max(1 - p, FLT_MIN)) saturates at -87.3 as soon as 1 - sigmoid(x) rounds to zero (x > 16.6: a negative with logit 20
        // costs 65 instead of 15). North star: match the reference CPU path.
        const float sp = log1pf(expf(-fabsf(xi)));
        const float log_p = -(fmaxf(-xi, 0.f) + sp), log_1mp = -(fmaxf(xi, 0.f) + sp);
        const float term_p = powf(1.f - p, gamma) * log_p;
        const float term_n = powf(p, gamma) * log_1mp;
        out[i] = pos ? -alpha * term_p : -(1.f - alpha) * term_n;
    }
}

__global__ void focal_bwd_kernel(const float* __restrict__ x, const int64_t* __restrict__ t, float* __restrict__ g,
                                 long n, int c, float gamma, float alpha) {
    const long total = n * c;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / c;
        const int cls = (int)(i - r * c);
        const bool pos = t[r] == (int64_t)cls;
        const float xi = x[i];
        const float p = 1.f / (1.f + expf(-xi));
        const float sp = log1pf(expf(-fabsf(xi)));
        const float log_p = -(fmaxf(-xi, 0.f) + sp), log_1mp = -(fmaxf(xi, 0.f) + sp);
        const float gp = powf(1.f - p, gamma) * (1.f - p - gamma * p * log_p);
        const float gn = powf(p, gamma) * (gamma * (1.f - p) * log_1mp - p);
        g[i] = pos ? -alpha * gp : -(1.f - alpha) * gn;
    }
}

__device__ __forceinline__ float iou_eps(const float4 g, const float4 b) {
    const float area1 = (g.z - g.x) * (g.w - g.y);
    const float area2 = (b.z - b.x) * (b.w - b.y);
    const float ltx = fmaxf(g.x, b.x), lty = fmaxf(g.y, b.y);
    const float rbx = fminf(g.z, b.z), rby = fminf(g.w, b.w);
    const float w = fmaxf(rbx - ltx, 0.f), h = fmaxf(rby - lty, 0.f);
    const float overlap = w * h;
    const float uni = fmaxf(area1 + area2 - overlap, 1e-6f);
    return overlap / uni;
}

// pass A: per anchor max/argmax over gts; per gt max over anchors. Each workgroup walks a grid-stride range of anchors, every wave
// keeps its running maximum per gt in LDS (wave-private words: no atomics inside the loop), and the workgroup publishes ONE ordered-uint
// atomicMax per gt at the end (round 3: the first version issued one atomic per wave per gt -- 27,590 atomics on ten addresses for the
// 176,580 anchors of a 736 x 1280 image, 317 us; now 256 workgroups x k).
__global__ void assign_pass_a(const float4* __restrict__ gt, int k, const float4* __restrict__ an,
                              const uint8_t* __restrict__ valid, long n, float* __restrict__ max_ov,
                              int* __restrict__ argmax, unsigned* gt_max_u) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
    float4* sgt = reinterpret_cast<float4*>(sm);
    unsigned* smax = reinterpret_cast<unsigned*>(sm + (size_t)k * 16);          // [4 waves][k]
    for (int j = threadIdx.x; j < k; j += blockDim.x) sgt[j] = gt[j];
    for (int j = threadIdx.x; j < 4 * k; j += blockDim.x) smax[j] = 0u;
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long span = (long)gridDim.x * blockDim.x;
    const long iters = (n + span - 1) / span;
    for (long it = 0; it < iters; ++it) {
        const long i = it * span + blockIdx.x * (long)blockDim.x + threadIdx.x;
        const bool ok = i < n && (!valid || valid[i]);
        float4 b = make_float4(0, 0, 0, 0);
        if (ok) b = an[i];
        float best = -1.f;
        int bi = 0;
        for (int j = 0; j < k; ++j) {
            float v = ok ? iou_eps(sgt[j], b) : -1.f;
            if (ok && v > best) { best = v; bi = j; }
            unsigned u = ok ? llv::f32_ordered(v) : 0u;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const unsigned t = __shfl_xor(u, o);
                u = t > u ? t : u;
            }
            if (lane == 0 && u > smax[wave * k + j]) smax[wave * k + j] = u;
        }
        if (i < n) {
            max_ov[i] = ok ? best : 0.f;
            argmax[i] = bi;
        }
    }
    __syncthreads();
    for (int j = threadIdx.x; j < k; j += blockDim.x) {
        unsigned u = smax[j];
        for (int w = 1; w < 4; ++w) u = smax[w * k + j] > u ? smax[w * k + j] : u;
        if (u) atomicMax(&gt_max_u[j], u);
    }
}

// pass B (gt_max_assign_all == False): first anchor index reaching each gt's max
__global__ void assign_pass_b(const float4* __restrict__ gt, int k, const float4* __restrict__ an,
                              const uint8_t* __restrict__ valid, long n, const unsigned* gt_max_u, int* gt_argmax) {
    const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (i >= n || (valid && !valid[i])) return;
    const float4 b = an[i];
    for (int j = 0; j < k; ++j) {
        const float v = iou_eps(gt[j], b);
        if (llv::f32_ordered(v) == gt_max_u[j]) atomicMin(&gt_argmax[j], (int)i);
    }
}

__global__ void assign_pass_c(const float4* __restrict__ gt, int k, const float4* __restrict__ an,
                              const uint8_t* __restrict__ valid, long n, const float* __restrict__ max_ov,
                              const int* __restrict__ argmax, const unsigned* gt_max_u, const int* gt_argmax,
                              float pos_thr, float neg_thr, float min_pos, int low_q, int assign_all,
                              int64_t* __restrict__ out) {
    const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (valid && !valid[i]) { out[i] = -1; return; }
    int64_t a = -1;
    const float mo = max_ov[i];
    if (mo >= 0.f && mo < neg_thr) a = 0;
    if (mo >= pos_thr) a = (int64_t)argmax[i] + 1;
    if (low_q) {
        const float4 b = an[i];
        for (int j = 0; j < k; ++j) {
            const float gm = llv::f32_unordered(gt_max_u[j]);
            if (!(gm >= min_pos)) continue;
            if (assign_all) {
                if (iou_eps(gt[j], b) == gm) a = j + 1;
            } else if (gt_argmax[j] == (int)i) {
                a = j + 1;
            }
        }
    }
    out[i] = a;
}

__global__ void fill_i64(int64_t* p, long n, int64_t v) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) p[i] = v;
}
__global__ void init_gt_ws(unsigned* gt_max_u, int* gt_argmax, int k) {
    for (int j = threadIdx.x; j < k; j += blockDim.x) { gt_max_u[j] = 0u; gt_argmax[j] = 0x7fffffff; }
}

inline size_t al(size_t v) { return (v + 255) & ~(size_t)255; }
inline int grid1d(long n, int b) {
    long g = (n + b - 1) / b;
    return (int)(g < 1 ? 1 : g);
}


// softmax cross entropy (F.cross_entropy(pred, label, weight=class_weight, reduction='none'), cross_entropy_loss.py:9-40):
// one wavefront per row; labels outside [0, c) (F.cross_entropy's ignore_index = -100) give loss 0 / gradient 0.
__global__ void softmax_ce_fwd_kernel(const float* __restrict__ x, const int64_t* __restrict__ t, const float* __restrict__ cw,
                                      float* __restrict__ loss, long n, int c, int ld) {
    const long r = blockIdx.x * (long)(blockDim.x >> 6) + (threadIdx.x >> 6);
    if (r >= n) return;
    const int lane = threadIdx.x & 63;
    const float* row = x + r * ld;
    float m = -FLT_MAX;
    for (int j = lane; j < c; j += 64) m = fmaxf(m, row[j]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    float s = 0.f;
    for (int j = lane; j < c; j += 64) s += expf(row[j] - m);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) {
        const int64_t y = t[r];
        float v = 0.f;
        if (y >= 0 && y < c) {
            v = (m + logf(s)) - row[y];
            if (cw) v *= cw[y];
        }
        loss[r] = v;
    }
}

// grad[r][j] = gout[r] * cw[y] * (softmax(x[r])[j] - [j == y])
__global__ void softmax_ce_bwd_kernel(const float* __restrict__ x, const int64_t* __restrict__ t, const float* __restrict__ cw,
                                      const float* __restrict__ gout, float* __restrict__ g, long n, int c, int ld) {
    const long r = blockIdx.x * (long)(blockDim.x >> 6) + (threadIdx.x >> 6);
    if (r >= n) return;
    const int lane = threadIdx.x & 63;
    const float* row = x + r * ld;
    float* grow = g + r * ld;
    const int64_t y = t[r];
    const bool live = y >= 0 && y < c;
    float m = -FLT_MAX;
    for (int j = lane; j < c; j += 64) m = fmaxf(m, row[j]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    float s = 0.f;
    for (int j = lane; j < c; j += 64) s += expf(row[j] - m);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float k = live ? gout[r] * (cw ? cw[y] : 1.f) : 0.f;
    const float inv = 1.f / s;
    for (int j = lane; j < c; j += 64) grow[j] = k * (expf(row[j] - m) * inv - (j == (int)y ? 1.f : 0.f));
}

// ------------------------------------------------------------------------------------------------
// The dense head's loss as ONE pass over the logits (round 3): AnchorHead.loss of a RetinaHead (anchor_head.py:375-493) with
// FocalLoss (focal_loss.py:160-181) and L1Loss / SmoothL1Loss (smooth_l1_loss.py:8-45) for EVERY anchor of EVERY level and image:
// label and weight from llv_iou_assign's gt indices (anchor_head.py:229-260 with PseudoSampler), the focal term and its derivative,
// the box target (delta_xywh_bbox_coder.py:88-130) of the positives and the L1 term and its derivative. The loss values go to
// per-level sums (fp32 atomics, one per workgroup), the derivatives -- times loss_weight / avg_factor * grad_scale -- straight into
// the fp16 gradient twins of the logit buffers the backward pass starts from. avg_factor = sum over images of max(#positives, 1)
// (anchor_head.py:468-470) is read from device memory (retina_count_pos_kernel): no host round trip anywhere in the step.
__global__ void retina_count_pos_kernel(const int64_t* __restrict__ gt_inds, long a_total, int* __restrict__ counts) {
    // blockIdx.y = image; every workgroup counts a grid-stride slice and adds it to the image's counter (zero on entry: the finish
    // kernel of the previous call leaves it so, the first call finds the allocation's zeros)
    const int n = blockIdx.y;
    int c = 0;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < a_total; i += (long)gridDim.x * blockDim.x)
        c += gt_inds[(long)n * a_total + i] > 0 ? 1 : 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(&counts[n], c);
}

// One workgroup walks pixels of its level (grid-stride), all 256 threads over the pixel's A*C logits: the channel -> (anchor, class)
// split comes from an LDS table built once, the pixel's A assignment words are fetched once per pixel into LDS, and the sigmoid /
// softplus pair shares one exponential (e = exp(-|x|): p = 1/(1+e) or e/(1+e), softplus(-|x|) = log1p(e)). The first version did
// three 32-bit divisions, a 64-bit gather and three transcendentals per logit: 293 us for the 14.1 M logits of one 736 x 1280 image.
constexpr int kRlMaxAC = 2048, kRlMaxA = 32, kRlPix = 16;
// G2: gamma == 2 (every config of the reference): the general powf() path is compiled out of that instance -- inlined twice per logit it
// tripled the code of the unrolled loop and pushed scalar registers into VGPR lanes
template <bool G2>
__global__ __launch_bounds__(256) void retina_loss_kernel(const llv_retina_loss_desc d, const int* __restrict__ counts) {
    const int l = blockIdx.y;
    const int A = d.num_anchors, C = d.num_classes, AC = A * C;
    const int hw = d.level_h[l] * d.level_w[l];
    const int npix = d.batch * hw;
    const long a_total = d.anchors_per_image;
    __shared__ unsigned short s_anchor[kRlMaxAC], s_cls[kRlMaxAC];
    __shared__ int s_label[kRlPix * kRlMaxA];     // per (pixel of the group, anchor): class of its gt, C for background
    __shared__ float s_w[kRlPix * kRlMaxA];
    __shared__ int s_gi[kRlPix * kRlMaxA];
    for (int ch = threadIdx.x; ch < AC; ch += blockDim.x) {
        const int a = ch / C;
        s_anchor[ch] = (unsigned short)a;
        s_cls[ch] = (unsigned short)(ch - a * C);
    }
    float avg = 0.f;
    for (int n = 0; n < d.batch; ++n) avg += (float)(counts[n] > 1 ? counts[n] : 1);
    const float inv_avg = 1.f / avg;
    const float gs_cls = d.cls_weight * inv_avg * d.grad_scale, gs_reg = d.reg_weight * inv_avg * d.grad_scale;
    constexpr bool g2 = G2;
    const float posw = d.pos_weight > 0.f ? d.pos_weight : 1.f;
    _Float16* const dcls = reinterpret_cast<_Float16*>(d.dcls);
    _Float16* const dreg = reinterpret_cast<_Float16*>(d.dreg);
    float lc = 0.f, lb = 0.f;
    // groups of kRlPix consecutive pixels of the level's (batch, h, w) block: the assignment words of the whole group are fetched at once
    const int ngroups = (npix + kRlPix - 1) / kRlPix;
    for (int grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
        const int pn0 = grp * kRlPix;
        const int np = npix - pn0 < kRlPix ? npix - pn0 : kRlPix;
        __syncthreads();                                         // the previous group's words are no longer read (and the tables are built)
        for (int t = threadIdx.x; t < np * A; t += blockDim.x) {
            const int pp = t / A, a = t - pp * A;
            const int pn = pn0 + pp, n = pn / hw, pix = pn - n * hw;
            const int64_t gi = d.gt_inds[(long)n * a_total + d.level_anchor0[l] + (long)pix * A + a];
            s_gi[pp * kRlMaxA + a] = (int)gi;
            // labels / weights: positives carry their gt's class, negatives the background class C, ignored anchors weight 0
            s_label[pp * kRlMaxA + a] = gi > 0 ? (int)d.gt_labels[(long)n * d.max_gt + (gi - 1)] : C;
            s_w[pp * kRlMaxA + a] = gi > 0 ? posw : (gi == 0 ? 1.f : 0.f);
        }
        __syncthreads();
        int pp = 0, ch = threadIdx.x;
        while (ch >= AC) { ch -= AC; ++pp; }
        while (pp < np) {
            // eight logits of this thread in flight (one was a chain of exposed load latencies: 45 per thread on the finest level;
            // four still left eleven round trips per group: 156 us per image, round 4)
            constexpr int U = 8;
            int pps[U], chs[U];
            float xs[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                pps[u] = pp; chs[u] = ch;
                xs[u] = pp < np ? d.cls[(d.level_off[l] + pn0 + pp) * (long)d.cls_ld + ch] : 0.f;
                ch += blockDim.x;
                while (ch >= AC) { ch -= AC; ++pp; }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (pps[u] >= np) break;
                const int pq = pps[u], cq = chs[u];
                const long row = d.level_off[l] + pn0 + pq;      // row of the pixel-major buffers
                const int a = s_anchor[cq];
                const bool pos = s_label[pq * kRlMaxA + a] == (int)s_cls[cq];
                const float w = s_w[pq * kRlMaxA + a];
                const float xi = xs[u];
                // hardware exp2 / log2 / rcp (1 ulp each) instead of the library's expf / log1pf / division: with those the kernel was
                // bound by ~150 ALU instructions per logit (154 us for the 14.1 M logits of an image). log1p(e), e in (0, 1]: the series
                // below 1/64 (error < e^4/4 = 1.5e-8 relative to e), log(1 + e) above it, where 1 + e is exact to 6e-8 of e / 1.6e-2
                const float e = __expf(-fabsf(xi));
                const float r = __builtin_amdgcn_rcpf(1.f + e);
                const float p = xi >= 0.f ? r : e * r;           // sigmoid(x)
                const float sp = e < 0.015625f ? e * (1.f - e * (0.5f - e * 0.33333334f)) : __logf(1.f + e);
                const float log_p = -(fmaxf(-xi, 0.f) + sp), log_1mp = -(fmaxf(xi, 0.f) + sp);
                float loss, grad;
                if (pos) {
                    const float m = g2 ? (1.f - p) * (1.f - p) : powf(1.f - p, d.gamma);
                    loss = -d.alpha * (m * log_p);
                    grad = -d.alpha * (m * (1.f - p - d.gamma * p * log_p));
                } else {
                    const float m = g2 ? p * p : powf(p, d.gamma);
                    loss = -(1.f - d.alpha) * (m * log_1mp);
                    grad = -(1.f - d.alpha) * (m * (d.gamma * (1.f - p) * log_1mp - p));
                }
                lc += loss * w;
                dcls[row * d.dcls_ld + cq] = (_Float16)(grad * w * gs_cls);
            }
        }
        for (int t = threadIdx.x; t < np * A * 4; t += blockDim.x) {      // the group's box terms: t = (pixel, anchor, coordinate)
            const int e = t & 3, pa = t >> 2;
            const int pq = pa / A, a = pa - pq * A;
            const int pn = pn0 + pq, n = pn / hw, pix = pn - n * hw;
            const long row = d.level_off[l] + pn;
            const int gi = s_gi[pq * kRlMaxA + a];
            float g = 0.f;
            if (gi > 0) {
                const float4 an = reinterpret_cast<const float4*>(d.anchors)[d.level_anchor0[l] + (long)pix * A + a];
                const float4 gt = reinterpret_cast<const float4*>(d.gt_boxes)[(long)n * d.max_gt + (gi - 1)];
                const float px = (an.x + an.z) * 0.5f, py = (an.y + an.w) * 0.5f, pw = an.z - an.x, ph = an.w - an.y;
                const float gx = (gt.x + gt.z) * 0.5f, gy = (gt.y + gt.w) * 0.5f, gw = gt.z - gt.x, gh = gt.w - gt.y;
                const float tg = e == 0 ? ((gx - px) / pw - d.means[0]) / d.stds[0]
                               : e == 1 ? ((gy - py) / ph - d.means[1]) / d.stds[1]
                               : e == 2 ? (logf(gw / pw) - d.means[2]) / d.stds[2] : (logf(gh / ph) - d.means[3]) / d.stds[3];
                const float diff = d.reg[row * d.reg_ld + a * 4 + e] - tg;
                const float ad = fabsf(diff);
                if (d.beta > 0.f && ad < d.beta) {               // smooth L1: 0.5 d^2 / beta inside, |d| - 0.5 beta outside
                    lb += 0.5f * diff * diff / d.beta;
                    g = diff / d.beta;
                } else {
                    lb += d.beta > 0.f ? ad - 0.5f * d.beta : ad;
                    g = diff > 0.f ? 1.f : (diff < 0.f ? -1.f : 0.f);
                }
            }
            dreg[row * d.dreg_ld + a * 4 + e] = (_Float16)(g * gs_reg);
        }
    }
    // workgroup sums -> one atomic per workgroup and term
    __shared__ float red[2][256];
    red[0][threadIdx.x] = lc;
    red[1][threadIdx.x] = lb;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) {
            red[0][threadIdx.x] += red[0][threadIdx.x + st];
            red[1][threadIdx.x] += red[1][threadIdx.x + st];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        if (d.partials) {
            // one slot per workgroup and term, summed by retina_loss_finish_kernel in a fixed order. 920 workgroups adding to ONE address
            // with device-scope fp32 atomics serialised at ~46 ns each: 85 of the kernel's 129 us (round 4: 44.7 us without them)
            d.partials[(long)l * gridDim.x + blockIdx.x] = red[0][0];
            d.partials[(long)(d.n_levels + l) * gridDim.x + blockIdx.x] = red[1][0];
        } else {
            atomicAdd(&d.sums[l], red[0][0]);
            atomicAdd(&d.sums[d.n_levels + l], red[1][0]);
        }
    }
}

// sums -> losses: [2][n_levels] * loss_weight / avg_factor; leaves sums and counters zero for the next call
__global__ void retina_loss_finish_kernel(float* __restrict__ sums, int* __restrict__ counts, int batch, float* __restrict__ avg_factor,
                                          float* __restrict__ out, int n_levels, float cls_weight, float reg_weight,
                                          const float* __restrict__ partials, int slots) {
    float avg = 0.f;
    for (int n = 0; n < batch; ++n) avg += (float)(counts[n] > 1 ? counts[n] : 1);
    const int i = threadIdx.x;
    if (partials) {                                   // wave t adds the slots of (term, level) pair t, lane k its slots k, k + 64, ..
        const int t = i >> 6, lane = i & 63;
        if (t < 2 * n_levels) {
            float v = 0.f;
            for (int k = lane; k < slots; k += 64) v += partials[(long)t * slots + k];
            for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
            if (lane == 0) sums[t] = v;
        }
        __syncthreads();
    }
    if (i < 2 * n_levels) out[i] = sums[i] * (i < n_levels ? cls_weight : reg_weight) / avg;
    __syncthreads();
    if (i < 2 * n_levels) sums[i] = 0.f;
    if (i < batch) counts[i] = 0;
    if (i == 0) *avg_factor = avg;
}

}  // namespace

extern "C" int llv_sigmoid_focal_loss_fwd(const float* logits, const int64_t* labels, float* loss, int64_t n, int c,
                                          float gamma, float alpha, void* stream) {
    if (n < 0 || c < 1) return LLV_E_BADARG;
    if (n == 0) return LLV_OK;
    if (!logits || !labels || !loss) return LLV_E_BADARG;
    long g = (n * c + 255) / 256;
    g = g > 8192 ? 8192 : g;
    hipLaunchKernelGGL(focal_fwd_kernel, dim3((int)g), dim3(256), 0, (hipStream_t)stream, logits, labels, loss,
                       (long)n, c, gamma, alpha);
    return hipGetLastError() == hipSuccess ? LLV_OK : LLV_E_LAUNCH;
}

extern "C" int llv_sigmoid_focal_loss_bwd(const float* logits, const int64_t* labels, float* grad, int64_t n, int c,
                                          float gamma, float alpha, void* stream) {
    if (n < 0 || c < 1) return LLV_E_BADARG;
    if (n == 0) return LLV_OK;
    if (!logits || !labels || !grad) return LLV_E_BADARG;
    long g = (n * c + 255) / 256;
    g = g > 8192 ? 8192 : g;
    hipLaunchKernelGGL(focal_bwd_kernel, dim3((int)g), dim3(256), 0, (hipStream_t)stream, logits, labels, grad,
                       (long)n, c, gamma, alpha);
    return hipGetLastError() == hipSuccess ? LLV_OK : LLV_E_LAUNCH;
}

extern "C" int llv_softmax_ce_fwd(const float* logits, const int64_t* labels, const float* class_weight, float* loss, int64_t n,
                                  int c, int ld, void* stream) {
    if (n < 0 || c < 1 || ld < c) return LLV_E_BADARG;
    if (n == 0) return LLV_OK;
    if (!logits || !labels || !loss) return LLV_E_BADARG;
    if ((n + 3) / 4 > 0x7fffffffL) return LLV_E_UNSUPPORTED;
    hipLaunchKernelGGL(softmax_ce_fwd_kernel, dim3((int)((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, logits, labels,
                       class_weight, loss, (long)n, c, ld);
    return hipGetLastError() == hipSuccess ? LLV_OK : LLV_E_LAUNCH;
}

extern "C" int llv_softmax_ce_bwd(const float* logits, const int64_t* labels, const float* class_weight, const float* grad_loss,
                                  float* grad, int64_t n, int c, int ld, void* stream) {
    if (n < 0 || c < 1 || ld < c) return LLV_E_BADARG;
    if (n == 0) return LLV_OK;
    if (!logits || !labels || !grad_loss || !grad) return LLV_E_BADARG;
    if ((n + 3) / 4 > 0x7fffffffL) return LLV_E_UNSUPPORTED;
    hipLaunchKernelGGL(softmax_ce_bwd_kernel, dim3((int)((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, logits, labels,
                       class_weight, grad_loss, grad, (long)n, c, ld);
    return hipGetLastError() == hipSuccess ? LLV_OK : LLV_E_LAUNCH;
}

extern "C" size_t llv_iou_assign_workspace_bytes(int k, int64_t n) {
    if (k < 0 || n < 0) return 0;
    return al((size_t)(k > 0 ? k : 1) * 4) * 2 + al((size_t)(n > 0 ? n : 1) * 4);
}

extern "C" int llv_iou_assign(const float* gt, int k, const float* anchors, const uint8_t* valid, int64_t n,
                              float pos_iou_thr, float neg_iou_thr, float min_pos_iou, int match_low_quality,
                              int gt_max_assign_all, int64_t* assigned_gt_inds, float* max_overlaps, void* workspace,
                              size_t workspace_bytes, void* stream) {
    if (k < 0 || n < 0 || k > 4096) return LLV_E_BADARG;
    if (n == 0) return LLV_OK;
    if (!anchors || !assigned_gt_inds || !max_overlaps || !workspace) return LLV_E_BADARG;
    if (workspace_bytes < llv_iou_assign_workspace_bytes(k, n)) return LLV_E_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    if (k == 0) {  // max_iou_assigner.py:150-153: everything is background
        hipLaunchKernelGGL(fill_i64, dim3(grid1d(n, 256) > 4096 ? 4096 : grid1d(n, 256)), dim3(256), 0, st,
                           assigned_gt_inds, (long)n, (int64_t)0);
        if (hipMemsetAsync(max_overlaps, 0, (size_t)n * 4, st) != hipSuccess) return LLV_E_LAUNCH;
        return hipGetLastError() == hipSuccess ? LLV_OK : LLV_E_LAUNCH;
    }
    if (!gt) return LLV_E_BADARG;
    char* ws = (char*)workspace;
    unsigned* gt_max_u = (unsigned*)ws;
    int* gt_argmax = (int*)(ws + al((size_t)k * 4));
    int* argmax = (int*)(ws + 2 * al((size_t)k * 4));
    hipLaunchKernelGGL(init_gt_ws, dim3(1), dim3(256), 0, st, gt_max_u, gt_argmax, k);
    // pass A keeps the gt boxes and a [4 waves][k] table of running maxima in dynamic LDS: k * 32 bytes, 128 KiB at the documented
    // limit k = 4096 -- above HIP's 64 KiB default, inside gfx950's 160 KiB once the kernel has opted in (once per process)
    static bool big_lds = false;
    if (!big_lds) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(assign_pass_a), hipFuncAttributeMaxDynamicSharedMemorySize, 4096 * 32);
        big_lds = true;
    }
    const int g = grid1d(n, 256);
    hipLaunchKernelGGL(assign_pass_a, dim3(g > 256 ? 256 : g), dim3(256), (size_t)k * 32, st, (const float4*)gt, k,
                       (const float4*)anchors, valid, (long)n, max_overlaps, argmax, gt_max_u);
    if (match_low_quality && !gt_max_assign_all)
        hipLaunchKernelGGL(assign_pass_b, dim3(g), dim3(256), 0, st, (const float4*)gt, k, (const float4*)anchors, valid,
                           (long)n, gt_max_u, gt_argmax);
    hipLaunchKernelGGL(assign_pass_c, dim3(g), dim3(256), 0, st, (const float4*)gt, k, (const float4*)anchors, valid,
                       (long)n, max_overlaps, argmax, gt_max_u, gt_argmax, pos_iou_thr, neg_iou_thr, min_pos_iou,
                       match_low_quality, gt_max_assign_all, assigned_gt_inds);
    return hipGetLastError() == hipSuccess ? LLV_OK : LLV_E_LAUNCH;
}

extern "C" int llv_retina_loss(const llv_retina_loss_desc* d, void* stream) {
    if (!d || !d->cls || !d->reg || !d->dcls || !d->dreg || !d->anchors || !d->gt_inds || !d->sums || !d->avg_factor || !d->losses || !d->counts)
        return LLV_E_BADARG;
    if (d->batch < 1 || d->batch > 64 || d->n_levels < 1 || d->n_levels > LLV_MAX_SEG || d->num_anchors < 1 || d->num_classes < 1 || d->max_gt < 0)
        return LLV_E_BADARG;
    if (d->max_gt > 0 && (!d->gt_boxes || !d->gt_labels)) return LLV_E_BADARG;
    const int AC = d->num_anchors * d->num_classes;
    if (d->cls_ld < AC || d->dcls_ld < AC || d->reg_ld < 4 * d->num_anchors || d->dreg_ld < 4 * d->num_anchors) return LLV_E_BADARG;
    long a_sum = 0, max_work = 0;
    for (int l = 0; l < d->n_levels; ++l) {
        if (d->level_h[l] < 1 || d->level_w[l] < 1 || d->level_off[l] < 0 || d->level_anchor0[l] != a_sum) return LLV_E_BADARG;
        const long hw = (long)d->level_h[l] * d->level_w[l];
        a_sum += hw * d->num_anchors;
        const long work = (long)d->batch * hw * AC;
        max_work = work > max_work ? work : max_work;
    }
    if (a_sum != d->anchors_per_image) return LLV_E_BADARG;
    if (max_work >= (1L << 31)) return LLV_E_UNSUPPORTED;        // the kernel indexes a level's logits with 32-bit arithmetic
    hipStream_t st = (hipStream_t)stream;
    long gc = (a_sum + 255) / 256;
    gc = gc > 64 ? 64 : gc;
    hipLaunchKernelGGL(retina_count_pos_kernel, dim3((int)gc, d->batch), dim3(256), 0, st, d->gt_inds, (long)d->anchors_per_image, d->counts);
    if (AC > kRlMaxAC || d->num_anchors > kRlMaxA || d->max_gt > (1 << 30)) return LLV_E_UNSUPPORTED;
    long gx = (max_work / AC + kRlPix - 1) / kRlPix;             // pixel groups of the largest level
    gx = gx > 2048 ? 2048 : (gx < 1 ? 1 : gx);
    if (d->partials && d->partials_cap < 2L * d->n_levels * gx) return LLV_E_BADARG;
    if (d->gamma == 2.f) hipLaunchKernelGGL(retina_loss_kernel<true>, dim3((int)gx, d->n_levels), dim3(256), 0, st, *d, (const int*)d->counts);
    else hipLaunchKernelGGL(retina_loss_kernel<false>, dim3((int)gx, d->n_levels), dim3(256), 0, st, *d, (const int*)d->counts);
    hipLaunchKernelGGL(retina_loss_finish_kernel, dim3(1), dim3(d->partials ? 64 * 2 * d->n_levels : 64), 0, st, d->sums, d->counts, d->batch, d->avg_factor, d->losses, d->n_levels,
                       d->cls_weight, d->reg_weight, (const float*)d->partials, (int)gx);
    return hipGetLastError() == hipSuccess ? LLV_OK : LLV_E_LAUNCH;
}
