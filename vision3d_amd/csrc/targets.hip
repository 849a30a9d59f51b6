// targets.hip -- anchor <-> ground-truth target assignment fused around the rotated-IoU core.
//
// Replaces ProposalTargetAssigner.forward (vision3d/core/proposal_targets.py:10-88) + Matcher.__call__ /
// set_low_quality_matches_ (ops/matcher.py:55-130) + box_encode.encode (core/box_encode.py:26-36) for the BEV
// anchor grid: the reference materialises an (n_gt x 70 400) IoU matrix per class (box_iou_rotated), takes column
// max / row max over it in torch, and encodes the positives in ~20 more launches.  Here the matrix never exists:
//   pass 1  thread = anchor: IoU against the class's ground truths (staged in LDS, prepped once), running best /
//           first arg-best (torch.max semantics), per-ground-truth maximum by atomicMax on the float bits (IoU >= 0);
//   pass 2  thread = anchor: band label from its best IoU, the "low quality" rule (positive if it attains some
//           ground truth's maximum -- recomputed IoU compared to the stored maximum, bit for bit), class targets
//           and the VoxelNet box encoding of the positives.
// The BEV box is (x, y, w, l, yaw) with yaw in RADIANS fed to the degree-based IoU, as the reference does (H1).
// box_ignore is not consulted: the reference defines apply_ignore_mask but never calls it (proposal_targets.py:36-49).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/vision3d_hip.h"
#include "rotated_iou.h"
#include "v3d_common.h"

#define TA_MAX_GT 128  // ground truths of ONE class staged per workgroup

struct TaParams {
  int n_gt, n_cls, A;
  int allow_low_quality;
  float lo[16], hi[16];  // per class: IoU < lo -> 0, lo <= IoU < hi -> -1 (ignored), IoU >= hi -> +1
};

__device__ __forceinline__ v3d::BoxPrep ta_prep7(const float* b) {  // (x, y, z, w, l, h, yaw) -> BEV (x, y, w, l, yaw)
  const float bev[5] = {b[0], b[1], b[3], b[4], b[6]};
  return v3d::prep_box(bev);
}

// stage the ground truths of class c (in input order) into LDS; returns their count (block-uniform)
__device__ __forceinline__ int ta_stage(const float* gt, const long long* gt_class, int n_gt, int c, v3d::BoxPrep* sp,
                                        int* sidx, int* s_count) {
  if (threadIdx.x == 0) {
    int m = 0;
    for (int g = 0; g < n_gt && m < TA_MAX_GT; g++)
      if ((int)gt_class[g] == c) sidx[m++] = g;
    *s_count = m;
  }
  __syncthreads();
  const int m = *s_count;
  for (int i = threadIdx.x; i < m; i += blockDim.x) sp[i] = ta_prep7(gt + 7 * (size_t)sidx[i]);
  __syncthreads();
  return m;
}

__global__ __launch_bounds__(V3D_BLOCK) void ta_match_kernel(const float* __restrict__ gt, const long long* __restrict__ gt_class,
                                                             const float* __restrict__ anchors, const TaParams p,
                                                             float* __restrict__ best_iou, int* __restrict__ best_gt,
                                                             unsigned* __restrict__ gt_max /*(n_gt) float bits, pre-zeroed*/) {
  __shared__ v3d::BoxPrep sp[TA_MAX_GT];
  __shared__ int sidx[TA_MAX_GT];
  __shared__ int s_count;
  __shared__ v3d::P2 clip_pts[V3D_BLOCK / V3D_WAVE][24 * 64];  // the clipper's work arrays: LDS, not scratch (rotated_iou.h)
  __shared__ float clip_dist[V3D_BLOCK / V3D_WAVE][24 * 64];
  const int c = blockIdx.y;
  const int m = ta_stage(gt, gt_class, p.n_gt, c, sp, sidx, &s_count);
  const int a = blockIdx.x * V3D_BLOCK + threadIdx.x;
  if (a >= p.A) return;
  v3d::P2* pts = clip_pts[threadIdx.x >> 6] + (threadIdx.x & 63);
  float* dist = clip_dist[threadIdx.x >> 6] + (threadIdx.x & 63);
  const v3d::BoxPrep ba = ta_prep7(anchors + 7 * ((size_t)c * p.A + a));
  float best = -1.f;
  int arg = 0;
  for (int i = 0; i < m; i++) {
    const float q = v3d::iou_prepped_lds(sp[i], ba, pts, dist);  // box_iou_rotated(gt, anchors)[i][a]
    if (q > best) {  // strict: the FIRST maximal ground truth, as torch.max(dim=0) returns
      best = q;
      arg = i;
    }
    if (q > 0.f) atomicMax(&gt_max[sidx[i]], __float_as_uint(q));  // zero needs no write: the buffer starts at +0.0
  }
  best_iou[(size_t)c * p.A + a] = best;
  best_gt[(size_t)c * p.A + a] = m ? sidx[arg] : 0;
}

__device__ __forceinline__ float ta_remainder(float x, float m) {  // torch.remainder: result takes the sign of m
  float r = fmodf(x, m);
  if (r != 0.f && ((m < 0.f) != (r < 0.f))) r += m;
  return r;
}

__global__ __launch_bounds__(V3D_BLOCK) void ta_label_kernel(const float* __restrict__ gt, const long long* __restrict__ gt_class,
                                                             const float* __restrict__ anchors, const TaParams p,
                                                             const float* __restrict__ best_iou, const int* __restrict__ best_gt,
                                                             const unsigned* __restrict__ gt_max, signed char* __restrict__ G_cls,
                                                             unsigned char* __restrict__ M_cls, float* __restrict__ G_reg,
                                                             unsigned char* __restrict__ M_reg, long long* __restrict__ matches) {
  __shared__ v3d::BoxPrep sp[TA_MAX_GT];
  __shared__ int sidx[TA_MAX_GT];
  __shared__ int s_count;
  __shared__ v3d::P2 clip_pts[V3D_BLOCK / V3D_WAVE][24 * 64];  // the clipper's work arrays: LDS, not scratch (rotated_iou.h)
  __shared__ float clip_dist[V3D_BLOCK / V3D_WAVE][24 * 64];
  const int c = blockIdx.y;
  const int m = ta_stage(gt, gt_class, p.n_gt, c, sp, sidx, &s_count);
  const int a = blockIdx.x * V3D_BLOCK + threadIdx.x;
  if (a >= p.A) return;
  v3d::P2* pts = clip_pts[threadIdx.x >> 6] + (threadIdx.x & 63);
  float* dist = clip_dist[threadIdx.x >> 6] + (threadIdx.x & 63);
  const size_t ia = (size_t)c * p.A + a;
  const float* an = anchors + 7 * ia;
  int label = 0;  // no ground truth of this class: the lowest band (matcher.py:69-79)
  if (m) {
    const float best = best_iou[ia];
    label = best < p.lo[c] ? 0 : (best < p.hi[c] ? -1 : 1);
    if (p.allow_low_quality) {
      const v3d::BoxPrep ba = ta_prep7(an);
      for (int i = 0; i < m; i++)
        if (__float_as_uint(v3d::iou_prepped_lds(sp[i], ba, pts, dist)) == gt_max[sidx[i]]) label = 1;  // ties included (matcher.py:98-130)
    }
  }
  const int g = best_gt[ia];
  if (matches) matches[ia] = g;
  G_cls[ia] = (signed char)(label < 0 ? 0 : label);
  M_cls[ia] = label != -1;
  const bool pos = label == 1;
  M_reg[ia] = pos;
  float t[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (pos) {  // box_encode.encode
    const float* b = gt + 7 * (size_t)g;
    const float diag = sqrtf(an[3] * an[3] + an[4] * an[4]);
    t[0] = (b[0] - an[0]) / diag;
    t[1] = (b[1] - an[1]) / diag;
    t[2] = (b[2] - an[2]) / an[5];
    t[3] = logf(b[3] / an[3]);
    t[4] = logf(b[4] / an[4]);
    t[5] = logf(b[5] / an[5]);
    t[6] = ta_remainder(b[6] - an[6], 3.14159274101257324f);
  }
#pragma unroll
  for (int q = 0; q < 7; q++) G_reg[7 * ia + q] = t[q];
}

extern "C" size_t v3d_assign_targets_workspace(int n_gt, int n_cls, int anchors_per_class) {
  const size_t NA = (size_t)(n_cls > 0 ? n_cls : 1) * (size_t)(anchors_per_class > 0 ? anchors_per_class : 1);
  return v3d_align(NA * 4) * 2 + v3d_align((size_t)(n_gt > 0 ? n_gt : 1) * 4) + 256;
}

extern "C" int v3d_assign_targets(const float* gt_boxes, const int64_t* gt_class, int n_gt, const float* anchors, int n_cls,
                                  int anchors_per_class, const float* iou_thresh_host, int allow_low_quality, int8_t* G_cls,
                                  uint8_t* M_cls, float* G_reg, uint8_t* M_reg, int64_t* matches, void* workspace,
                                  size_t workspace_bytes, v3d_stream_t stream) {
  hipStream_t st = (hipStream_t)stream;
  if (n_gt < 0 || n_cls < 1 || n_cls > 16 || anchors_per_class < 1 || !anchors || !iou_thresh_host || !G_cls || !M_cls || !G_reg ||
      !M_reg || !workspace)
    return V3D_EINVAL;
  if (n_gt > 0 && (!gt_boxes || !gt_class)) return V3D_EINVAL;
  if (workspace_bytes < v3d_assign_targets_workspace(n_gt, n_cls, anchors_per_class)) return V3D_EWORKSPACE;
  TaParams p;
  p.n_gt = n_gt; p.n_cls = n_cls; p.A = anchors_per_class; p.allow_low_quality = allow_low_quality;
  for (int c = 0; c < 16; c++) {
    p.lo[c] = c < n_cls ? iou_thresh_host[2 * c] : 0.f;
    p.hi[c] = c < n_cls ? iou_thresh_host[2 * c + 1] : 0.f;
  }
  V3dArena ar(workspace, workspace_bytes);
  const size_t NA = (size_t)n_cls * anchors_per_class;
  float* best_iou = ar.take<float>(NA);
  int* best_gt = ar.take<int>(NA);
  unsigned* gt_max = ar.take<unsigned>((size_t)(n_gt > 0 ? n_gt : 1));
  if (!ar.ok()) return V3D_EWORKSPACE;
  V3D_CHECK_HIP(v3d_fill_async(gt_max, 0, (size_t)(n_gt > 0 ? n_gt : 1) * 4, st));
  dim3 grid(v3d_ceil_div(anchors_per_class, V3D_BLOCK), n_cls);
  hipLaunchKernelGGL(ta_match_kernel, grid, dim3(V3D_BLOCK), 0, st, gt_boxes, (const long long*)gt_class, anchors, p, best_iou,
                     best_gt, gt_max);
  hipLaunchKernelGGL(ta_label_kernel, grid, dim3(V3D_BLOCK), 0, st, gt_boxes, (const long long*)gt_class, anchors, p, best_iou,
                     best_gt, gt_max, (signed char*)G_cls, M_cls, G_reg, M_reg, (long long*)matches);
  V3D_CHECK_LAUNCH();
  return V3D_OK;
}
