// eval.hip -- true/false-positive marking of the reference's VOC / ImageNet-style mAP evaluation, on device.
//
// Reference (mmdetection/mmdet/core/evaluation/mean_ap.py): eval_map calls, per class and per image, tpfp_default
// (:154-244) or tpfp_imagenet (:60-151, datasets 'det' / 'vid') through a 4-process pool; each call is a small greedy
// matching of the image's detections of that class (score-descending) against its ground-truth boxes. An evaluation
// is num_classes x num_images such cells (ImageNet VID val: 30 x 176 k), all independent: here one launch marks them
// all, one wavefront per cell.
//
//   cell = one (class, image): dets[det_off[cell] .. det_off[cell+1]) (x1,y1,x2,y2,score), gts[gt_off[cell] ..) with
//   gt_ignore[j] = 1 for boxes that came from `bboxes_ignore`.
//   tp/fp: [nscale][total_dets] uint8, nscale = max(n_ranges, 1).
//
// Arithmetic is the reference's float32 numpy arithmetic in the same operation order (bbox_overlaps.py:4-48: overlap /
// max(area_det + area_gt - overlap, 1e-6); imagenet: IoU against gt - 1, per-gt threshold min(w*h / ((w+10)*(h+10)), thr)),
// built with -ffp-contract=off; the marks are bit-exact against oracle/evaluation.py. Ties in score: lower index first
// (the reference's np.argsort(-score) leaves them unspecified).
#include <hip/hip_runtime.h>
#include "llv.h"
#include "sort_util.h"

namespace {

constexpr int kMaxDet = 2048;   // detections per cell (LDS: 16 KiB keys + 16 KiB best match)
constexpr int kMaxGt = 4096;    // ground-truth boxes per cell (LDS covered flags)

__device__ __forceinline__ float iou_np(const float4 d, const float4 g) {
    const float xs = fmaxf(d.x, g.x), ys = fmaxf(d.y, g.y);
    const float xe = fminf(d.z, g.z), ye = fminf(d.w, g.w);
    const float overlap = fmaxf(xe - xs, 0.f) * fmaxf(ye - ys, 0.f);
    const float area_d = (d.z - d.x) * (d.w - d.y);
    const float area_g = (g.z - g.x) * (g.w - g.y);
    const float uni = fmaxf(area_d + area_g - overlap, 1e-6f);
    return overlap / uni;
}

__device__ __forceinline__ bool in_range(float area, const double* rg) {   // min_area <= area < max_area
    return (double)area >= rg[0] && (double)area < rg[1];
}

template <int MODE>   // 0: tpfp_default, 1: tpfp_imagenet
__global__ __launch_bounds__(64) void eval_tpfp_kernel(const float* __restrict__ dets, const int* __restrict__ det_off,
                                                      const float* __restrict__ gts, const int* __restrict__ gt_off,
                                                      const uint8_t* __restrict__ gt_ignore, const double* __restrict__ ranges,
                                                      int n_ranges, float iou_thr, long total_dets,
                                                      uint8_t* __restrict__ tp, uint8_t* __restrict__ fp) {
    __shared__ uint64_t keys[kMaxDet];
    __shared__ float best_iou[kMaxDet];
    __shared__ int best_j[kMaxDet];
    __shared__ uint8_t covered[kMaxGt];
    const int cell = blockIdx.x, lane = threadIdx.x;
    const int d0 = det_off[cell], m = det_off[cell + 1] - d0;
    const int g0 = gt_off[cell], n = gt_off[cell + 1] - g0;
    if (m <= 0) return;
    const int nscale = n_ranges > 0 ? n_ranges : 1;
    auto det_box = [&](int i) {
        const float* p = dets + (long)(d0 + i) * 5;
        return make_float4(p[0], p[1], p[2], p[3]);
    };
    auto gt_box = [&](int j) {
        const float* p = gts + (long)(g0 + j) * 4;
        return make_float4(p[0], p[1], p[2], p[3]);
    };
    // ---- no ground truth in this cell: every detection inside the area range is a false positive (mean_ap.py:189-199)
    if (n == 0) {
        for (int i = lane; i < m; i += 64) {
            const float4 b = det_box(i);
            const float area = (b.z - b.x) * (b.w - b.y);
            for (int k = 0; k < nscale; ++k) {
                tp[(long)k * total_dets + d0 + i] = 0;
                fp[(long)k * total_dets + d0 + i] = (n_ranges == 0 || in_range(area, ranges + 2 * k)) ? 1 : 0;
            }
        }
        return;
    }
    // ---- score-descending order (ties: lower index first)
    const int P = llv::next_pow2(m);
    for (int i = lane; i < P; i += 64) keys[i] = i < m ? llv::make_key(dets[(long)(d0 + i) * 5 + 4], (uint32_t)i) : 0ull;
    __syncthreads();
    llv::block_sort_desc(keys, P, m);
    if (MODE == 0) {
        // the detection's best gt does not depend on the matching state: ious.max(axis=1) / argmax (first maximum)
        for (int i = lane; i < m; i += 64) {
            const float4 b = det_box(i);
            float bi = -1.f;
            int bj = 0;
            for (int j = 0; j < n; ++j) {
                const float v = iou_np(b, gt_box(j));
                if (v > bi) { bi = v; bj = j; }
            }
            best_iou[i] = bi;
            best_j[i] = bj;
        }
        __syncthreads();
    }
    for (int k = 0; k < nscale; ++k) {
        const double* rg = ranges + 2 * k;
        for (int j = lane; j < n; j += 64) covered[j] = 0;
        __syncthreads();
        for (int s = 0; s < m; ++s) {
            const int i = (int)llv::key_index(keys[s]);
            int matched = -1;                       // wave-uniform
            if (MODE == 0) {
                if (best_iou[i] >= iou_thr) matched = best_j[i];
            } else {
                // best *uncovered* gt with IoU >= its own threshold; strict '>' while scanning j upwards: lowest j among equals
                const float4 b = det_box(i);
                float bi = -1.f;
                int bj = -1;
                for (int j = lane; j < n; j += 64) {
                    if (covered[j]) continue;
                    const float4 g = gt_box(j);
                    const float gw = g.z - g.x, gh = g.w - g.y;
                    const float thr = fminf((gw * gh) / ((gw + 10.0f) * (gh + 10.0f)), iou_thr);
                    const float4 g1 = make_float4(g.x - 1.f, g.y - 1.f, g.z - 1.f, g.w - 1.f);
                    const float v = iou_np(b, g1);
                    if (v >= thr && v > bi) { bi = v; bj = j; }
                }
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) {
                    const float oi = __shfl_xor(bi, o);
                    const int oj = __shfl_xor(bj, o);
                    if (oj >= 0 && (bj < 0 || oi > bi || (oi == bi && oj < bj))) { bi = oi; bj = oj; }
                }
                matched = bj;
            }
            uint8_t t = 0, f = 0;
            if (matched >= 0) {
                const float4 g = gt_box(matched);
                const float ga = (g.z - g.x) * (g.w - g.y);
                const bool ign = (gt_ignore && gt_ignore[g0 + matched] != 0) || (n_ranges > 0 && !in_range(ga, rg));
                if (MODE == 0) {
                    // matched to an ignored / out-of-range gt: neither tp nor fp; to a covered one: fp (mean_ap.py:224-232)
                    if (!ign) {
                        if (!covered[matched]) t = 1; else f = 1;
                    }
                } else {
                    if (!ign) t = 1;                // imagenet: the gt is consumed even when ignored (mean_ap.py:138-142)
                }
            } else {
                const float4 b = det_box(i);
                const float area = (b.z - b.x) * (b.w - b.y);
                f = (n_ranges == 0 || in_range(area, rg)) ? 1 : 0;
            }
            __syncthreads();                        // every lane has read covered[] for this detection
            if (lane == 0) {
                if (matched >= 0 && (MODE == 1 || t)) covered[matched] = 1;
                tp[(long)k * total_dets + d0 + i] = t;
                fp[(long)k * total_dets + d0 + i] = f;
            }
            __syncthreads();
        }
    }
}

}  // namespace

extern "C" int llv_eval_tpfp(const float* dets, const int* det_off, const float* gts, const int* gt_off,
                             const uint8_t* gt_ignore, int n_cells, int max_det_per_cell, int max_gt_per_cell,
                             const double* area_ranges, int n_ranges, float iou_thr, int mode, int64_t total_dets,
                             uint8_t* tp, uint8_t* fp, void* stream) {
    if (n_cells < 0 || !det_off || !gt_off || (mode != 0 && mode != 1) || n_ranges < 0 || (n_ranges > 0 && !area_ranges))
        return LLV_E_BADARG;
    if (max_det_per_cell > kMaxDet || max_gt_per_cell > kMaxGt) return LLV_E_UNSUPPORTED;
    if (n_cells == 0 || total_dets == 0) return LLV_OK;
    if (!dets || !tp || !fp) return LLV_E_BADARG;
    if (mode == 0)
        hipLaunchKernelGGL(eval_tpfp_kernel<0>, dim3(n_cells), dim3(64), 0, (hipStream_t)stream, dets, det_off, gts, gt_off, gt_ignore,
                           area_ranges, n_ranges, iou_thr, (long)total_dets, tp, fp);
    else
        hipLaunchKernelGGL(eval_tpfp_kernel<1>, dim3(n_cells), dim3(64), 0, (hipStream_t)stream, dets, det_off, gts, gt_off, gt_ignore,
                           area_ranges, n_ranges, iou_thr, (long)total_dets, tp, fp);
    return hipGetLastError() == hipSuccess ? LLV_OK : LLV_E_LAUNCH;
}

// ------------------------------------------------------------------------------------------------
// COCO-protocol matching (what CocoVideoDataset.evaluate / CocoDataset.evaluate run through pycocotools' COCOeval.evaluateImg;
// pycocotools is un-vendored, its published algorithm is restated in oracle/coco_eval.py):
// one wavefront per (category, image) cell, one LANE per (area range, IoU threshold) pair (4 x 10 = 40 of the 64 lanes for the
// standard parameters): every lane runs the sequential greedy matching of its own pair over the cell's detections (already in
// score-descending order, truncated to the largest maxDets) and ground truths, IoUs in double precision as maskUtils.iou
// computes them (crowd: intersection over the detection's area).
//   gt order: evaluateImg stably sorts the gts by their ignore flag for the area range; walking the non-ignored ones first and
//   the ignored ones second, each in the original order, is the same order.
//   outputs: dt_matched / dt_ignore [A][T][total_dets] uint8, gt_valid [A][n_cells] int32 (ground truths not ignored).
// ------------------------------------------------------------------------------------------------
namespace {

constexpr int kCocoMaxGt = 256;

__device__ __forceinline__ double coco_iou(const double* d, const double* g, bool crowd) {   // boxes are (x, y, w, h)
    const double w = fmin(d[0] + d[2], g[0] + g[2]) - fmax(d[0], g[0]);
    const double h = fmin(d[1] + d[3], g[1] + g[3]) - fmax(d[1], g[1]);
    if (w <= 0.0 || h <= 0.0) return 0.0;
    const double inter = w * h;
    const double uni = crowd ? d[2] * d[3] : d[2] * d[3] + g[2] * g[3] - inter;
    return inter / uni;
}

__global__ __launch_bounds__(64) void coco_match_kernel(const double* __restrict__ dets, const int* __restrict__ det_off,
                                                       const double* __restrict__ gts, const double* __restrict__ gt_area,
                                                       const uint8_t* __restrict__ gt_crowd, const uint8_t* __restrict__ gt_ign,
                                                       const int* __restrict__ gt_off, const double* __restrict__ area_rng, int A,
                                                       const double* __restrict__ iou_thrs, int T, long total_dets, int n_cells,
                                                       uint8_t* __restrict__ dt_matched, uint8_t* __restrict__ dt_ignore,
                                                       int* __restrict__ gt_valid) {
    __shared__ uint8_t gtm[64][kCocoMaxGt];          // per lane: gt already matched at this (area range, threshold)
    const int cell = blockIdx.x, lane = threadIdx.x;
    const int d0 = det_off[cell], m = det_off[cell + 1] - d0;
    const int g0 = gt_off[cell], n = gt_off[cell + 1] - g0;
    if (lane >= A * T) return;
    const int a = lane / T, t = lane - a * T;
    const double lo = area_rng[2 * a], hi = area_rng[2 * a + 1];
    const double thr = fmin(iou_thrs[t], 1.0 - 1e-10);
    for (int j = 0; j < n; ++j) gtm[lane][j] = 0;
    if (t == 0) {
        int valid = 0;
        for (int j = 0; j < n; ++j) {
            const double ar = gt_area[g0 + j];
            if (!(gt_ign[g0 + j] || ar < lo || ar > hi)) ++valid;
        }
        gt_valid[(long)a * n_cells + cell] = valid;
    }
    uint8_t* om = dt_matched + ((long)a * T + t) * total_dets + d0;
    uint8_t* oi = dt_ignore + ((long)a * T + t) * total_dets + d0;
    for (int i = 0; i < m; ++i) {
        const double* db = dets + (long)(d0 + i) * 4;
        double best = thr;
        int mj = -1;
        bool m_ign = false;
        for (int pass = 0; pass < 2; ++pass) {           // non-ignored gts first, then the ignored ones
            if (pass == 1 && mj >= 0) break;             // matched a non-ignored gt: the ignored section is not entered
            for (int j = 0; j < n; ++j) {
                const double ar = gt_area[g0 + j];
                const bool ig = gt_ign[g0 + j] || ar < lo || ar > hi;
                if ((int)ig != pass) continue;
                const bool crowd = gt_crowd[g0 + j] != 0;
                if (gtm[lane][j] && !crowd) continue;
                const double v = coco_iou(db, gts + (long)(g0 + j) * 4, crowd);
                if (v < best) continue;
                best = v;
                mj = j;
                m_ign = ig;
            }
        }
        uint8_t matched = 0, ignore = 0;
        if (mj >= 0) {
            matched = 1;
            ignore = m_ign ? 1 : 0;
            gtm[lane][mj] = 1;
        } else {
            const double ar = db[2] * db[3];
            ignore = (ar < lo || ar > hi) ? 1 : 0;       // unmatched detections outside the area range are ignored
        }
        om[i] = matched;
        oi[i] = ignore;
    }
}

}  // namespace

extern "C" int llv_coco_match(const double* dets, const int* det_off, const double* gts, const double* gt_area,
                              const uint8_t* gt_crowd, const uint8_t* gt_ignore, const int* gt_off, int n_cells,
                              int max_gt_per_cell, const double* area_ranges, int n_ranges, const double* iou_thrs, int n_thrs,
                              int64_t total_dets, uint8_t* dt_matched, uint8_t* dt_ignore, int32_t* gt_valid, void* stream) {
    if (n_cells < 0 || !det_off || !gt_off || !area_ranges || !iou_thrs || n_ranges < 1 || n_thrs < 1 || !gt_valid) return LLV_E_BADARG;
    if (n_ranges * n_thrs > 64 || max_gt_per_cell > kCocoMaxGt) return LLV_E_UNSUPPORTED;
    if (n_cells == 0) return LLV_OK;
    if (total_dets > 0 && (!dets || !dt_matched || !dt_ignore)) return LLV_E_BADARG;
    hipLaunchKernelGGL(coco_match_kernel, dim3(n_cells), dim3(64), 0, (hipStream_t)stream, dets, det_off, gts, gt_area, gt_crowd,
                       gt_ignore, gt_off, area_ranges, n_ranges, iou_thrs, n_thrs, (long)total_dets, n_cells, dt_matched, dt_ignore,
                       gt_valid);
    return hipGetLastError() == hipSuccess ? LLV_OK : LLV_E_LAUNCH;
}
