// Fused proposal stage of the BEV head: sigmoid -> top-k per (frame, class) -> VoxelNet decode -> batched rotated
// NMS -> per-class score cut, entirely on the device, in 8 launches, no host synchronisation.
//
// Reference: vision3d/detector/proposal.py:39-80 (ProposalLayer.inference / _multiclass_batch_nms),
// core/box_encode.py:13-21 (decode), ops/iou_nms.py:90-134 (coordinate-offset batched NMS).  The torch statement
// of the same stage lives in vision3d_amd/detector/proposal.py (proposals_padded / finalize) and is the
// on-device cross-check of this file (tests/test_gpu_proposal.py); at KITTI size it costs ~55 small launches
// (~290 us inside a HIP graph) plus ~10 eager index kernels, this file ~45 us.
//
// Order semantics (the repository's spec where torch.topk leaves ties open): candidates of a group are ordered
// by (sigmoid score descending, anchor index ascending), selected on the fp32 SIGMOID value exactly as the
// reference selects (two logits that round to the same score are a tie).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/vision3d_hip.h"
#include "v3d_common.h"

#define PROP_THREADS 1024
#define PROP_WAVES (PROP_THREADS / 64)
#define PROP_MAX_TOPK 1024
#define PROP_MAX_CLS 16

struct PropGeom {
  int B, n_cls, n_yaw, HW, topk, ctot;  // ctot = n_cls*n_yaw*(1+7) channels of the fused head map
  float thresh[PROP_MAX_CLS];
};

__device__ __forceinline__ float prop_sigmoid(float x) { return 1.f / (1.f + expf(-x)); }  // torch: 1 / (1 + exp(-x))
__device__ __forceinline__ unsigned prop_logit_key(float x) {                               // order-preserving
  const unsigned u = __float_as_uint(x);
  return u ^ ((u >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}

// One workgroup per (frame, class) group.  Radix select (4 x 8 bits) of the topk-th largest LOGIT -- integer
// work only; the sigmoid is evaluated once per element in the collection pass, where membership is decided on
// the score itself: every element with score > S_T, then the lowest-index elements with score == S_T.
__global__ __launch_bounds__(PROP_THREADS) void prop_topk_kernel(const float* __restrict__ maps, PropGeom g,
                                                                 float* __restrict__ cand_score,
                                                                 int* __restrict__ cand_anchor) {
  __shared__ int hist[256];
  __shared__ unsigned sh_prefix;
  __shared__ int sh_need;
  __shared__ int sh_count;               // number of collected candidates
  __shared__ int sh_eq[PROP_WAVES + 1];  // score == S_T per wave region, then exclusive prefix
  __shared__ unsigned long long cand[PROP_MAX_TOPK];
  const int grp = blockIdx.x, b = grp / g.n_cls, c = grp % g.n_cls;
  const int n = g.n_yaw * g.HW;
  const float* x = maps + ((size_t)b * g.ctot + (size_t)c * g.n_yaw) * g.HW;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int K = g.topk;

  unsigned prefix = 0, pmask = 0;
  int need = K;
  for (int pass = 0; pass < 4; pass++) {
    const int shift = 24 - 8 * pass;
    if (tid < 256) hist[tid] = 0;
    __syncthreads();
    for (int i0 = 0; i0 < n; i0 += PROP_THREADS) {
      const int i = i0 + tid;
      const unsigned key = i < n ? prop_logit_key(x[i]) : 0u;
      bool active = i < n && (key & pmask) == prefix;
      const int digit = (key >> shift) & 255;
      // head logits cluster around the focal prior: one LDS atomic per distinct digit per wave, not per lane
      unsigned long long todo = __ballot(active);
      while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const int d0 = __shfl(digit, leader);
        const unsigned long long same = __ballot(active && digit == d0);
        if (lane == leader) atomicAdd(&hist[d0], __popcll(same));
        if (active && digit == d0) active = false;
        todo &= ~same;
      }
    }
    __syncthreads();
    if (tid == 0) {
      int cum = 0, sel = 0;
      for (int bin = 255; bin >= 0; bin--) {
        const int h = hist[bin];
        if (cum + h >= need) { sel = bin; break; }
        cum += h;
      }
      sh_prefix = prefix | ((unsigned)sel << shift);
      sh_need = need - cum;
    }
    __syncthreads();
    prefix = sh_prefix;
    need = sh_need;
    pmask |= 255u << shift;
    __syncthreads();
  }
  // prefix = key of the K-th largest logit; its score is the membership threshold
  const unsigned tkey = prefix;
  const unsigned tbits = (tkey & 0x80000000u) ? (tkey ^ 0x80000000u) : ~tkey;
  const float s_t = prop_sigmoid(__uint_as_float(tbits));

  if (tid == 0) sh_count = 0;
  __syncthreads();
  // wave w owns the contiguous index region [w*R, (w+1)*R): "lowest index first" among ties is then a per-wave
  // running count plus a prefix over the 16 regions
  const int R = ((n + PROP_WAVES - 1) / PROP_WAVES + 63) & ~63;
  const int lo = wave * R, hi = min(n, lo + R);
  int eq_here = 0;
  for (int i0 = lo; i0 < hi; i0 += 64) {
    const int i = i0 + lane;
    const float s = i < hi ? prop_sigmoid(x[i]) : -1.f;
    if (s > s_t) {
      const int slot = atomicAdd(&sh_count, 1);
      if (slot < PROP_MAX_TOPK) cand[slot] = ((unsigned long long)__float_as_uint(s) << 32) | (unsigned)(0xFFFFFFFFu - (unsigned)i);
    }
    eq_here += __popcll(__ballot(s == s_t));
  }
  if (lane == 0) sh_eq[wave] = eq_here;
  __syncthreads();
  if (tid == 0) {
    int run = 0;
    for (int w = 0; w < PROP_WAVES; w++) {
      const int e = sh_eq[w];
      sh_eq[w] = run;
      run += e;
    }
    sh_eq[PROP_WAVES] = sh_count;  // number of strictly-greater candidates (< K by construction)
  }
  __syncthreads();
  const int n_gt = sh_eq[PROP_WAVES];
  const int need_eq = K - n_gt;  // >= 1
  int seen = sh_eq[wave];        // ties before this wave's region
  for (int i0 = lo; i0 < hi && seen < need_eq; i0 += 64) {
    const int i = i0 + lane;
    const float s = i < hi ? prop_sigmoid(x[i]) : -1.f;
    const unsigned long long eq = __ballot(s == s_t);
    const int rank = seen + __popcll(eq & ((1ull << lane) - 1ull));
    if (s == s_t && rank < need_eq && n_gt + rank < PROP_MAX_TOPK)
      cand[n_gt + rank] = ((unsigned long long)__float_as_uint(s) << 32) | (unsigned)(0xFFFFFFFFu - (unsigned)i);
    seen += __popcll(eq);
  }
  __syncthreads();
  // bitonic sort of the K candidates, descending on (score bits, ~index): scores are >= 0 so bits order them
  int npad = 1;
  while (npad < K) npad <<= 1;
  for (int i = K + tid; i < npad; i += PROP_THREADS) cand[i] = 0ull;
  __syncthreads();
  for (int k = 2; k <= npad; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = tid; i < npad; i += PROP_THREADS) {
        const int p = i ^ j;
        if (p > i) {
          const unsigned long long a = cand[i], bb = cand[p];
          const bool desc = (i & k) == 0;
          if (desc ? a < bb : a > bb) { cand[i] = bb; cand[p] = a; }
        }
      }
      __syncthreads();
    }
  for (int i = tid; i < K; i += PROP_THREADS) {
    const unsigned long long v = cand[i];
    cand_score[(size_t)grp * K + i] = __uint_as_float((unsigned)(v >> 32));
    cand_anchor[(size_t)grp * K + i] = (int)(0xFFFFFFFFu - (unsigned)(v & 0xFFFFFFFFull));
  }
}

// One workgroup decodes all N = B*n_cls*topk candidates (core/box_encode.py:13-21), reduces the coordinate range
// and writes the group-shifted BEV boxes the batched NMS runs on (ops/iou_nms.py:127-133: offset = group *
// (max_coord - min_coord + 1), added to x and y).
__global__ __launch_bounds__(PROP_THREADS) void prop_decode_kernel(const float* __restrict__ maps,
                                                                   const float* __restrict__ anchors, PropGeom g,
                                                                   const int* __restrict__ cand_anchor,
                                                                   float* __restrict__ boxes /*(N,7)*/,
                                                                   long long* __restrict__ batch_idx,
                                                                   long long* __restrict__ class_idx,
                                                                   float* __restrict__ bev /*(N,5) shifted*/) {
  __shared__ float red_hi[PROP_WAVES], red_lo[PROP_WAVES];
  const int N = g.B * g.n_cls * g.topk;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n_anchor = g.n_cls * g.n_yaw;
  float hi = -INFINITY, lo = INFINITY;
  for (int t = tid; t < N; t += PROP_THREADS) {
    const int grp = t / g.topk, b = grp / g.n_cls, c = grp % g.n_cls;
    const int a = cand_anchor[t], yaw_i = a / g.HW, pix = a % g.HW;
    float d[7], an[7];
#pragma unroll
    for (int q = 0; q < 7; q++) {
      d[q] = maps[((size_t)b * g.ctot + n_anchor + (size_t)(c * 7 + q) * g.n_yaw + yaw_i) * g.HW + pix];
      an[q] = anchors[((size_t)c * g.n_yaw * g.HW + a) * 7 + q];
    }
    const float diag = sqrtf(an[3] * an[3] + an[4] * an[4]);
    float o[7];
    o[0] = d[0] * diag + an[0];
    o[1] = d[1] * diag + an[1];
    o[2] = d[2] * an[5] + an[2];
    o[3] = expf(d[3]) * an[3];
    o[4] = expf(d[4]) * an[4];
    o[5] = expf(d[5]) * an[5];
    o[6] = d[6] + an[6];
#pragma unroll
    for (int q = 0; q < 7; q++) boxes[(size_t)t * 7 + q] = o[q];
    batch_idx[t] = b;
    class_idx[t] = c;
    hi = fmaxf(hi, fmaxf(o[0], o[1]) + fmaxf(o[3], o[4]) / 2.f);
    lo = fminf(lo, fminf(o[0], o[1]) - fminf(o[3], o[4]) / 2.f);
  }
  for (int off = 32; off > 0; off >>= 1) {
    hi = fmaxf(hi, __shfl_xor(hi, off));
    lo = fminf(lo, __shfl_xor(lo, off));
  }
  if (lane == 0) { red_hi[wave] = hi; red_lo[wave] = lo; }
  __syncthreads();
  hi = red_hi[0];
  lo = red_lo[0];
  for (int w = 1; w < PROP_WAVES; w++) { hi = fmaxf(hi, red_hi[w]); lo = fminf(lo, red_lo[w]); }
  const float span = (hi - lo) + 1.f;
  for (int t = tid; t < N; t += PROP_THREADS) {  // re-reads this thread's own rows
    const int grp = t / g.topk;
    const float shift = (float)grp * span;
    const float* o = boxes + (size_t)t * 7;
    bev[(size_t)t * 5 + 0] = o[0] + shift;
    bev[(size_t)t * 5 + 1] = o[1] + shift;
    bev[(size_t)t * 5 + 2] = o[3];
    bev[(size_t)t * 5 + 3] = o[4];
    bev[(size_t)t * 5 + 4] = o[6];
  }
}

// keep (sorted by score, from the NMS) -> ordered compaction of the rows that pass their class threshold
__global__ __launch_bounds__(PROP_THREADS) void prop_finalize_kernel(const long long* __restrict__ keep,
                                                                     const int* __restrict__ n_keep, PropGeom g,
                                                                     const float* __restrict__ boxes,
                                                                     const long long* __restrict__ batch_idx,
                                                                     const long long* __restrict__ class_idx,
                                                                     const float* __restrict__ scores,
                                                                     float* __restrict__ out_boxes,
                                                                     long long* __restrict__ out_batch,
                                                                     long long* __restrict__ out_class,
                                                                     float* __restrict__ out_scores,
                                                                     int* __restrict__ n_out) {
  __shared__ int wave_cnt[PROP_WAVES];
  __shared__ int base;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nk = *n_keep;
  if (tid == 0) base = 0;
  __syncthreads();
  for (int i0 = 0; i0 < nk; i0 += PROP_THREADS) {
    const int i = i0 + tid;
    long long j = 0;
    bool pass = false;
    if (i < nk) {
      j = keep[i];
      pass = scores[j] > g.thresh[class_idx[j]];
    }
    const unsigned long long m = __ballot(pass);
    if (lane == 0) wave_cnt[wave] = __popcll(m);
    __syncthreads();
    int before = base;
    for (int w = 0; w < wave; w++) before += wave_cnt[w];
    if (pass) {
      const int r = before + __popcll(m & ((1ull << lane) - 1ull));
#pragma unroll
      for (int q = 0; q < 7; q++) out_boxes[(size_t)r * 7 + q] = boxes[(size_t)j * 7 + q];
      out_batch[r] = batch_idx[j];
      out_class[r] = class_idx[j];
      out_scores[r] = scores[j];
    }
    __syncthreads();
    if (tid == 0) {
      int tot = 0;
      for (int w = 0; w < PROP_WAVES; w++) tot += wave_cnt[w];
      base += tot;
    }
    __syncthreads();
  }
  if (tid == 0) *n_out = base;
}

static size_t prop_align(size_t x) { return (x + 255) & ~(size_t)255; }

extern "C" size_t v3d_proposals_workspace(int B, int n_cls, int topk) {
  const size_t N = (size_t)B * n_cls * topk;
  return prop_align(N * 4) * 2 /*cand score, anchor*/ + prop_align(N * 7 * 4) + prop_align(N * 8) * 3 /*batch, class, keep*/ +
         prop_align(N * 5 * 4) + 256 /*n_keep*/ + prop_align(v3d_nms_rotated_workspace((int)N)) + 1024;
}

extern "C" int v3d_proposals(const float* head_maps, const float* anchors, int B, int n_cls, int n_yaw, int H, int W,
                             int topk, const float* score_thresh_host, float iou_threshold, float* out_boxes,
                             int64_t* out_batch_idx, int64_t* out_class_idx, float* out_scores, int32_t* n_out,
                             void* workspace, size_t workspace_bytes, v3d_stream_t stream) {
  hipStream_t st = (hipStream_t)stream;
  if (!head_maps || !anchors || !score_thresh_host || !out_boxes || !out_batch_idx || !out_class_idx || !out_scores ||
      !n_out || !workspace)
    return V3D_EINVAL;
  if (B < 1 || n_cls < 1 || n_cls > PROP_MAX_CLS || n_yaw < 1 || H < 1 || W < 1 || topk < 1 || topk > PROP_MAX_TOPK)
    return V3D_EINVAL;
  if ((long long)n_yaw * H * W < topk) return V3D_EINVAL;  // torch.topk raises as well
  if (workspace_bytes < v3d_proposals_workspace(B, n_cls, topk)) return V3D_EWORKSPACE;
  PropGeom g;
  g.B = B; g.n_cls = n_cls; g.n_yaw = n_yaw; g.HW = H * W; g.topk = topk; g.ctot = n_cls * n_yaw * 8;
  for (int c = 0; c < PROP_MAX_CLS; c++) g.thresh[c] = c < n_cls ? score_thresh_host[c] : 0.f;
  const size_t N = (size_t)B * n_cls * topk;
  char* p = (char*)workspace;
  auto take = [&](size_t bytes) { char* q = p; p += prop_align(bytes); return (void*)q; };
  float* cand_score = (float*)take(N * 4);
  int* cand_anchor = (int*)take(N * 4);
  float* boxes = (float*)take(N * 7 * 4);
  long long* bidx = (long long*)take(N * 8);
  long long* cidx = (long long*)take(N * 8);
  long long* keep = (long long*)take(N * 8);
  float* bev = (float*)take(N * 5 * 4);
  int* n_keep = (int*)take(256);
  const size_t nms_bytes = v3d_nms_rotated_workspace((int)N);
  void* nms_ws = take(nms_bytes);

  hipLaunchKernelGGL(prop_topk_kernel, dim3(B * n_cls), dim3(PROP_THREADS), 0, st, head_maps, g, cand_score, cand_anchor);
  hipLaunchKernelGGL(prop_decode_kernel, dim3(1), dim3(PROP_THREADS), 0, st, head_maps, anchors, g, cand_anchor, boxes, bidx,
                     cidx, bev);
  const int rc = v3d_nms_rotated(bev, cand_score, (int)N, iou_threshold, (int64_t*)keep, n_keep, nms_ws, nms_bytes, stream);
  if (rc != V3D_OK) return rc;
  hipLaunchKernelGGL(prop_finalize_kernel, dim3(1), dim3(PROP_THREADS), 0, st, keep, n_keep, g, boxes, bidx, cidx, cand_score,
                     out_boxes, (long long*)out_batch_idx, (long long*)out_class_idx, out_scores, n_out);
  V3D_CHECK_LAUNCH();
  return V3D_OK;
}
