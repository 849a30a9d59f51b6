// Fused proposal stage of the BEV head: sigmoid -> top-k per (frame, class) -> VoxelNet decode -> batched rotated
// NMS -> per-class score cut, entirely on the device, no host synchronisation.  Launches: level-1 top-k | merge + decode + sort +
// box prep (ONE workgroup when there are at most two (frame, class) groups, else merge and decode separately) | suppression mask |
// greedy reduction + score cut = 4 at bs = 1 (6 until round 4).
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
#include "nms_device.h"
#include "rotated_iou.h"
#include "v3d_internal.h"

#define PROP_THREADS 1024
#define PROP_WAVES (PROP_THREADS / 64)
#define PROP_MAX_TOPK 1024
#define PROP_MAX_CLS 16
#define PROP_RANK_SORT_MAX 256  // candidate lists up to this (padded) length are rank-sorted, longer ones by a bitonic network
#ifndef PROP_CHUNKS
#define PROP_CHUNKS 40
#endif  // upper bound of level-1 slices per (frame, class) group (40 x topk 100 <= 4096 merge inputs)

struct PropGeom {
  int B, n_cls, n_yaw, HW, topk, ctot;  // ctot = n_cls*n_yaw*(1+7) channels of the fused head map
  float thresh[PROP_MAX_CLS];
};

__device__ __forceinline__ float prop_sigmoid(float x) { return 1.f / (1.f + expf(-x)); }  // torch: 1 / (1 + exp(-x))
__device__ __forceinline__ unsigned prop_logit_key(float x) {                               // order-preserving
  const unsigned u = __float_as_uint(x);
  return u ^ ((u >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}

// Exact top-K of one index range under the total order (score descending, index ascending), by one workgroup.
//   LEVEL 1: elements are LOGITS x[i]; radix select (4 x 8 bits) of the K-th largest logit is integer work only, the
//            sigmoid is evaluated in the collection passes, where membership is decided on the fp32 score itself:
//            every element with score > S_T, then the lowest-index elements with score == S_T.
//   LEVEL 2: elements are (score, index) candidates emitted by level 1, chunk after chunk -- position order equals
//            index order among equal scores, so the same "lowest position first" rule applies.
// Output: K rows sorted by the total order; a range shorter than K is padded with (score -1, index -1) sentinels.
// EPT elements per thread are held in registers, so one workgroup selects from <= EPT*1024 elements and reads them
// ONCE.  (256-thread workgroups with 4x the elements per thread were slower: 15.4 / 18.8 us vs 13.2 / 14.1 us.)
#define SEL_THREADS 1024
#define SEL_WAVES (SEL_THREADS / 64)
template <int LEVEL, int EPT>
__device__ void prop_select(const float* __restrict__ x, const int* __restrict__ xi, int n, int idx_base, int K,
                            float* __restrict__ out_score, int* __restrict__ out_idx) {
  __shared__ int hist[256];
  __shared__ unsigned sh_prefix;
  __shared__ int sh_need;
  __shared__ int sh_count, sh_eq_total;
  __shared__ int sh_eq[SEL_WAVES];
  __shared__ unsigned long long cand[PROP_MAX_TOPK];
  __shared__ int rank_s[PROP_RANK_SORT_MAX];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int Ke = min(K, n);  // rows that can be real
  float xv[EPT];  // element e of thread t is index t + e*256 (coalesced)
  unsigned kv[EPT];
#pragma unroll
  for (int e = 0; e < EPT; e++) {
    const int i = tid + e * SEL_THREADS;
    xv[e] = i < n ? x[i] : 0.f;
    kv[e] = prop_logit_key(xv[e]);
  }

  float s_t = 0.f;
  if (Ke > 0) {
    unsigned prefix = 0, pmask = 0;
    int need = Ke;
    for (int pass = 0; pass < 4; pass++) {
      const int shift = 24 - 8 * pass;
      if (tid < 256) hist[tid] = 0;
      __syncthreads();
#pragma unroll
      for (int e = 0; e < EPT; e++) {
        const bool active = tid + e * SEL_THREADS < n && (kv[e] & pmask) == prefix;
        const int digit = (kv[e] >> shift) & 255;
        // head logits cluster around the focal prior: the most common digit of a wave costs ONE LDS atomic, the
        // stragglers go in directly (distinct addresses do not serialise)
        const unsigned long long todo = __ballot(active);
        if (todo) {
          const int leader = __ffsll((long long)todo) - 1;
          const int d0 = __shfl(digit, leader);
          const unsigned long long same = __ballot(active && digit == d0);
          if (lane == leader) atomicAdd(&hist[d0], __popcll(same));
          if (active && digit != d0) atomicAdd(&hist[digit], 1);
        }
      }
      __syncthreads();
      {  // suffix sums S[b] = #elements with digit >= b: the digit with S[b] >= need > S[b+1] holds the K-th
        const int h = tid < 256 ? hist[tid] : 0;
        int v = h;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
          const int t = __shfl_down(v, off);
          if (lane + off < 64) v += t;
        }
        if (tid < 256 && lane == 0) sh_eq[wave] = v;
        __syncthreads();
        if (tid < 256) {
          int above = 0;
          for (int w = wave + 1; w < 4; w++) above += sh_eq[w];
          const int S = v + above, S_next = S - h;
          if (S >= need && S_next < need) {
            sh_prefix = prefix | ((unsigned)tid << shift);
            sh_need = need - S_next;
          }
        }
      }
      __syncthreads();
      prefix = sh_prefix;
      need = sh_need;
      pmask |= 255u << shift;
    }
    // prefix = key of the Ke-th largest element; its score is the membership threshold
    const unsigned tbits = (prefix & 0x80000000u) ? (prefix ^ 0x80000000u) : ~prefix;
    s_t = LEVEL == 1 ? prop_sigmoid(__uint_as_float(tbits)) : __uint_as_float(tbits);
  }
  if (tid == 0) { sh_count = 0; sh_eq_total = 0; }
  __syncthreads();
  // phase A: scores (one sigmoid per element); strictly-greater elements are appended in any order, ties counted
  float sv[EPT];
  int my_eq = 0;
#pragma unroll
  for (int e = 0; e < EPT; e++) {
    const int i = tid + e * SEL_THREADS;
    sv[e] = i < n ? (LEVEL == 1 ? prop_sigmoid(xv[e]) : xv[e]) : -2.f;
    if (sv[e] > s_t) {
      const int slot = atomicAdd(&sh_count, 1);
      if (slot < PROP_MAX_TOPK)
        cand[slot] = ((unsigned long long)prop_logit_key(sv[e]) << 32) |
                     (unsigned)(0xFFFFFFFFu - (unsigned)(LEVEL == 1 ? idx_base + i : i));
    }
    my_eq += sv[e] == s_t ? 1 : 0;
  }
  for (int off = 32; off > 0; off >>= 1) my_eq += __shfl_down(my_eq, off);
  if (lane == 0 && my_eq) atomicAdd(&sh_eq_total, my_eq);
  __syncthreads();
  const int n_gt = min(sh_count, PROP_MAX_TOPK);  // < Ke by construction
  const int need_eq = Ke - n_gt;
  if (sh_eq_total <= need_eq) {
    // the usual case: every tie is taken (one element equals the threshold) -- order is settled by the sort below
#pragma unroll
    for (int e = 0; e < EPT; e++) {
      if (sv[e] == s_t) {
        const int i = tid + e * SEL_THREADS;
        const int slot = atomicAdd(&sh_count, 1);
        if (slot < PROP_MAX_TOPK)
          cand[slot] = ((unsigned long long)prop_logit_key(sv[e]) << 32) |
                       (unsigned)(0xFFFFFFFFu - (unsigned)(LEVEL == 1 ? idx_base + i : i));
      }
    }
  } else {
    // more ties than free rows: lowest INDEX first -- slice e covers indices [e*256, (e+1)*256), threads in order
    int seen = 0;  // block-uniform
    for (int e = 0; e < EPT && seen < need_eq; e++) {
      const int i = tid + e * SEL_THREADS;
      const bool eqf = sv[e] == s_t;
      const unsigned long long eq = __ballot(eqf);
      __syncthreads();
      if (lane == 0) sh_eq[wave] = __popcll(eq);
      __syncthreads();
      int before = 0, tot = 0;
#pragma unroll
      for (int w = 0; w < SEL_WAVES; w++) {
        const int c = sh_eq[w];
        if (w < wave) before += c;
        tot += c;
      }
      const int rank = seen + before + __popcll(eq & ((1ull << lane) - 1ull));
      if (eqf && rank < need_eq && n_gt + rank < PROP_MAX_TOPK)
        cand[n_gt + rank] = ((unsigned long long)prop_logit_key(sv[e]) << 32) |
                            (unsigned)(0xFFFFFFFFu - (unsigned)(LEVEL == 1 ? idx_base + i : i));
      seen += tot;
    }
  }
  __syncthreads();
  // sort the candidates, descending on (ordered score key, ~position); zero keys (padding) sink
  int npad = 1;
  while (npad < K) npad <<= 1;
  if (npad <= PROP_RANK_SORT_MAX) {
    // RANK sort: the keys are distinct (the position is part of the key), so a candidate's place is the number of larger keys.
    // SEL_THREADS / npad threads share a candidate's count (LDS broadcast reads), one barrier pair instead of the
    // log2(npad) * (log2(npad) + 1) / 2 = 28 barrier-separated passes of a bitonic network at K = 100.
    const int i = tid & (npad - 1), part = tid / npad, parts = SEL_THREADS / npad;
    if (tid < npad) rank_s[tid] = 0;
    __syncthreads();
    unsigned long long mine = 0ull;
    if (i < Ke) {
      mine = cand[i];
      int cnt = 0;
      for (int j = part; j < Ke; j += parts) cnt += cand[j] > mine ? 1 : 0;
      if (cnt) atomicAdd(&rank_s[i], cnt);
    }
    __syncthreads();
    if (tid < Ke) mine = cand[tid];
    const int place = tid < Ke ? rank_s[tid] : 0;
    __syncthreads();
    if (tid < Ke) cand[place] = mine;
    __syncthreads();
  } else {
    for (int i = Ke + tid; i < npad; i += SEL_THREADS) cand[i] = 0ull;
    __syncthreads();
    for (int k = 2; k <= npad; k <<= 1)
      for (int j = k >> 1; j > 0; j >>= 1) {
        for (int i = tid; i < npad; i += SEL_THREADS) {
          const int p = i ^ j;
          if (p > i) {
            const unsigned long long a = cand[i], bb = cand[p];
            const bool desc = (i & k) == 0;
            if (desc ? a < bb : a > bb) { cand[i] = bb; cand[p] = a; }
          }
        }
        __syncthreads();
      }
  }
  for (int i = tid; i < K; i += SEL_THREADS) {
    if (i < Ke) {
      const unsigned long long v = cand[i];
      const unsigned kbits = (unsigned)(v >> 32);
      const unsigned pos = 0xFFFFFFFFu - (unsigned)(v & 0xFFFFFFFFull);
      out_score[i] = __uint_as_float((kbits & 0x80000000u) ? (kbits ^ 0x80000000u) : ~kbits);
      out_idx[i] = LEVEL == 1 ? (int)pos : xi[pos];
    } else {
      out_score[i] = -1.f;
      out_idx[i] = -1;
    }
  }
}

// level 1: grid (chunks, groups).  Each workgroup selects the top-K of its slice of the group's n_yaw*H*W anchors.
template <int EPT>
__global__ __launch_bounds__(SEL_THREADS) void prop_topk_chunk_kernel(const float* __restrict__ maps, PropGeom g, int chunk_len,
                                                                       float* __restrict__ part_score,
                                                                       int* __restrict__ part_idx) {
  const int chunk = blockIdx.x, chunks = gridDim.x, grp = blockIdx.y, b = grp / g.n_cls, c = grp % g.n_cls;
  const int n = g.n_yaw * g.HW;
  const float* x = maps + ((size_t)b * g.ctot + (size_t)c * g.n_yaw) * g.HW;
  const int lo = min(n, chunk * chunk_len), m = min(n, lo + chunk_len) - lo;
  const size_t o = ((size_t)grp * chunks + chunk) * g.topk;
  prop_select<1, EPT>(x + lo, nullptr, m, lo, g.topk, part_score + o, part_idx + o);
}

// level 2: one workgroup per group merges the chunks' candidates
__global__ __launch_bounds__(SEL_THREADS) void prop_topk_merge_kernel(const float* __restrict__ part_score,
                                                                       const int* __restrict__ part_idx, int chunks, int K,
                                                                       float* __restrict__ cand_score,
                                                                       int* __restrict__ cand_anchor) {
  const int grp = blockIdx.x;
  const size_t o = (size_t)grp * chunks * K;
  prop_select<2, 4>(part_score + o, part_idx + o, chunks * K, 0, K, cand_score + (size_t)grp * K, cand_anchor + (size_t)grp * K);
}

// One workgroup: decode all N = B*n_cls*topk candidates (core/box_encode.py:13-21), reduce the coordinate range,
// shift the BEV boxes by group (ops/iou_nms.py:127-133: offset = group * (max_coord - min_coord + 1), added to x and
// y), sort (score descending, index ascending) in LDS and emit the NMS inputs in sorted order -- what were four
// launches (decode, keys, bitonic sort, gather + box prep).  N <= 1024.
// REFINE: the same stage for stage-2 boxes (PV-RCNN: detector/model.py inference tail) -- residuals and the boxes they refine come
// from arrays in candidate order instead of the head maps / anchor grid, the score is the sigmoid of a confidence logit.
template <bool REFINE>
__device__ __forceinline__ void prop_decode_sort_body(const float* __restrict__ maps, const float* __restrict__ anchors,
                                                      const PropGeom& g, const int* cand_anchor, const float* cand_score,
                                                      float* __restrict__ boxes /*(N,7)*/, long long* __restrict__ batch_idx,
                                                      long long* __restrict__ class_idx, int* __restrict__ order,
                                                      v3d::BoxPrep* __restrict__ prep, const float* __restrict__ ref_deltas = nullptr,
                                                      const float* __restrict__ ref_boxes = nullptr,
                                                      const float* __restrict__ ref_conf = nullptr, float* __restrict__ ref_score = nullptr) {
  __shared__ float red_hi[PROP_WAVES], red_lo[PROP_WAVES];
  __shared__ unsigned long long keys[PROP_THREADS];
  __shared__ float sbev[PROP_THREADS][5];
  const int N = g.B * g.n_cls * g.topk;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int n_anchor = g.n_cls * g.n_yaw;
  float hi = -INFINITY, lo = INFINITY;
  float o[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  int grp = 0;
  if (t < N) {
    grp = t / g.topk;
    const int b = grp / g.n_cls, c = grp % g.n_cls;
    float d[7], an[7];
    if constexpr (REFINE) {
#pragma unroll
      for (int q = 0; q < 7; q++) {
        d[q] = ref_deltas[(size_t)t * 7 + q];
        an[q] = ref_boxes[(size_t)t * 7 + q];
      }
    } else {
      const int a = cand_anchor[t], yaw_i = a / g.HW, pix = a % g.HW;
#pragma unroll
      for (int q = 0; q < 7; q++) {
        d[q] = maps[((size_t)b * g.ctot + n_anchor + (size_t)(c * 7 + q) * g.n_yaw + yaw_i) * g.HW + pix];
        an[q] = anchors[((size_t)c * g.n_yaw * g.HW + a) * 7 + q];
      }
    }
    const float diag = sqrtf(an[3] * an[3] + an[4] * an[4]);
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
    hi = fmaxf(o[0], o[1]) + fmaxf(o[3], o[4]) / 2.f;
    lo = fminf(o[0], o[1]) - fminf(o[3], o[4]) / 2.f;
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
  const float shift = (float)grp * ((hi - lo) + 1.f);
  sbev[t][0] = o[0] + shift;
  sbev[t][1] = o[1] + shift;
  sbev[t][2] = o[3];
  sbev[t][3] = o[4];
  sbev[t][4] = o[6];
  // key = (~orderable(score) << 32) | index, ascending == the order v3d_nms_rotated sorts in; padding sorts last
  unsigned long long key = ~0ull;
  if (t < N) {
    float sc;
    if constexpr (REFINE) {
      sc = prop_sigmoid(ref_conf[t]);
      ref_score[t] = sc;
    } else {
      sc = cand_score[t];
    }
    const unsigned u = __float_as_uint(sc);
    const unsigned ord = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    key = ((unsigned long long)(~ord) << 32) | (unsigned)t;
  }
  keys[t] = key;
  __syncthreads();
  int npad = 1;
  while (npad < N) npad <<= 1;
  if (npad <= PROP_RANK_SORT_MAX) {
    // rank sort (distinct keys: the candidate index is part of the key): the place of candidate i is the number of smaller keys
    __shared__ int place_s[PROP_RANK_SORT_MAX];
    const int i = t & (npad - 1), part = t / npad, parts = PROP_THREADS / npad;
    if (t < npad) place_s[t] = 0;
    __syncthreads();
    if (i < N) {
      const unsigned long long mine = keys[i];
      int cnt = 0;
      for (int j = part; j < N; j += parts) cnt += keys[j] < mine ? 1 : 0;
      if (cnt) atomicAdd(&place_s[i], cnt);
    }
    __syncthreads();
    if (t < N) {
      const int dst = place_s[t];  // candidate t is the dst-th of the sorted order
      order[dst] = t;
      prep[dst] = v3d::prep_box(sbev[t]);
    }
    return;
  }
  for (int k = 2; k <= npad; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      if (t < npad) {
        const int p = t ^ j;
        if (p > t) {
          const unsigned long long a = keys[t], bb = keys[p];
          const bool up = (t & k) == 0;
          if ((a > bb) == up) { keys[t] = bb; keys[p] = a; }
        }
      }
      __syncthreads();
    }
  if (t < N) {
    const int src = (int)(keys[t] & 0xFFFFFFFFull);
    order[t] = src;
    prep[t] = v3d::prep_box(sbev[src]);
  }
}

__global__ __launch_bounds__(PROP_THREADS) void prop_decode_sort_kernel(const float* __restrict__ maps,
                                                                        const float* __restrict__ anchors, PropGeom g,
                                                                        const int* __restrict__ cand_anchor,
                                                                        const float* __restrict__ cand_score,
                                                                        float* __restrict__ boxes, long long* __restrict__ batch_idx,
                                                                        long long* __restrict__ class_idx, int* __restrict__ order,
                                                                        v3d::BoxPrep* __restrict__ prep) {
  prop_decode_sort_body<false>(maps, anchors, g, cand_anchor, cand_score, boxes, batch_idx, class_idx, order, prep);
}

// Level-2 merge of every group AND the decode / sort / box prep in ONE workgroup (at most PROP_FUSE_GROUPS groups: the bs = 1
// frame has one): the candidates go through global memory (a few hundred bytes, written and read by this workgroup with a
// barrier in between) because the score cut of the last launch reads them again.
#define PROP_FUSE_GROUPS 2
__global__ __launch_bounds__(PROP_THREADS) void prop_merge_decode_sort_kernel(const float* __restrict__ part_score,
                                                                              const int* __restrict__ part_idx, int chunks,
                                                                              const float* __restrict__ maps,
                                                                              const float* __restrict__ anchors, PropGeom g,
                                                                              int* cand_anchor, float* cand_score,
                                                                              float* __restrict__ boxes, long long* __restrict__ batch_idx,
                                                                              long long* __restrict__ class_idx, int* __restrict__ order,
                                                                              v3d::BoxPrep* __restrict__ prep) {
  static_assert(SEL_THREADS == PROP_THREADS, "one workgroup runs both bodies");
  const int groups = g.B * g.n_cls;
  for (int grp = 0; grp < groups; grp++) {
    const size_t o = (size_t)grp * chunks * g.topk;
    prop_select<2, 4>(part_score + o, part_idx + o, chunks * g.topk, 0, g.topk, cand_score + (size_t)grp * g.topk,
                      cand_anchor + (size_t)grp * g.topk);
    __syncthreads();  // (also: the next group reuses the selection's LDS)
  }
  prop_decode_sort_body<false>(maps, anchors, g, cand_anchor, cand_score, boxes, batch_idx, class_idx, order, prep);
}

__global__ __launch_bounds__(PROP_THREADS) void prop_refine_sort_kernel(const float* __restrict__ deltas, const float* __restrict__ props,
                                                                        const float* __restrict__ conf, PropGeom g, float* __restrict__ scores,
                                                                        float* __restrict__ boxes, long long* __restrict__ batch_idx,
                                                                        long long* __restrict__ class_idx, int* __restrict__ order,
                                                                        v3d::BoxPrep* __restrict__ prep) {
  prop_decode_sort_body<true>(nullptr, nullptr, g, nullptr, nullptr, boxes, batch_idx, class_idx, order, prep, deltas, props, conf, scores);
}

// keep (sorted by score, from the NMS) -> ordered compaction of the rows that pass their class threshold (one workgroup)
template <int THREADS>
__device__ __forceinline__ void prop_finalize_body(const long long* keep, int nk, const PropGeom& g, const float* __restrict__ boxes,
                                                   const long long* __restrict__ batch_idx, const long long* __restrict__ class_idx,
                                                   const float* __restrict__ scores, float* __restrict__ out_boxes,
                                                   long long* __restrict__ out_batch, long long* __restrict__ out_class,
                                                   float* __restrict__ out_scores, int* __restrict__ n_out,
                                                   const int* __restrict__ aux_flag) {
  __shared__ int wave_cnt[THREADS / 64];
  __shared__ int base;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) base = 0;
  __syncthreads();
  for (int i0 = 0; i0 < nk; i0 += THREADS) {
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
      for (int w = 0; w < THREADS / 64; w++) tot += wave_cnt[w];
      base += tot;
    }
    __syncthreads();
  }
  if (tid == 0) {
    n_out[0] = base;
    if (aux_flag) n_out[1] = *aux_flag;  // rides in the caller's one host read of the frame (see v3d_proposals_flag)
  }
}

// Greedy NMS reduction + score cut in ONE launch (they were two single-workgroup launches with a global round trip per 64-box
// block in between).  A mask of at most PROP_NMS_LDS_WORDS words is first copied into LDS in one parallel read -- the reduction's
// dependent look-ups (diagonal word, kept rows) then never leave the CU; larger masks are reduced from global memory.
#define PROP_NMS_THREADS 256
#define PROP_NMS_LDS_WORDS 2048
__global__ __launch_bounds__(PROP_NMS_THREADS) void prop_nms_reduce_finalize_kernel(
    const unsigned long long* __restrict__ mask, const int* __restrict__ order, int N, unsigned long long* __restrict__ remv_g,
    long long* keep, PropGeom g, const float* __restrict__ boxes, const long long* __restrict__ batch_idx,
    const long long* __restrict__ class_idx, const float* __restrict__ scores, float* __restrict__ out_boxes,
    long long* __restrict__ out_batch, long long* __restrict__ out_class, float* __restrict__ out_scores, int* __restrict__ n_out,
    const int* __restrict__ aux_flag) {
  __shared__ unsigned long long mask_s[PROP_NMS_LDS_WORDS];
  __shared__ unsigned long long remv_s[64];
  __shared__ unsigned long long kept_s;
  __shared__ int nk_s;
  const int nwords = (N + 63) / 64;
  const bool in_lds = (long long)N * nwords <= PROP_NMS_LDS_WORDS && nwords <= 64;
  if (in_lds) {
    for (int i = threadIdx.x; i < N * nwords; i += PROP_NMS_THREADS) mask_s[i] = (i % nwords) >= ((i / nwords) >> 6) ? mask[i] : 0ull;
    __syncthreads();
  }
  const int nk = v3d::nms_greedy_reduce<PROP_NMS_THREADS>(in_lds ? mask_s : mask, order, N, nwords, in_lds ? remv_s : remv_g, keep,
                                                          &kept_s, &nk_s);
  // (keep[] was written by wave 0 of this workgroup; nms_greedy_reduce ends with a barrier)
  prop_finalize_body<PROP_NMS_THREADS>(keep, nk, g, boxes, batch_idx, class_idx, scores, out_boxes, out_batch, out_class, out_scores,
                                       n_out, aux_flag);
}

static size_t prop_align(size_t x) { return (x + 255) & ~(size_t)255; }

extern "C" size_t v3d_proposals_workspace(int B, int n_cls, int topk) {
  const size_t N = (size_t)B * n_cls * topk;
  return prop_align(N * 4) * 2 /*cand score, anchor*/ + prop_align(N * PROP_CHUNKS * 4) * 2 /*level-1 lists*/ +
         prop_align(N * 7 * 4) + prop_align(N * 8) * 3 /*batch, class, keep*/ +
         prop_align(N * 5 * 4) + 256 /*n_keep*/ + prop_align(v3d_nms_rotated_workspace((int)N)) + prop_align(N * 4) /*order*/ +
         prop_align(N * sizeof(v3d::BoxPrep)) + 1024;
}

extern "C" int v3d_proposals(const float* head_maps, const float* anchors, int B, int n_cls, int n_yaw, int H, int W,
                             int topk, const float* score_thresh_host, float iou_threshold, float* out_boxes,
                             int64_t* out_batch_idx, int64_t* out_class_idx, float* out_scores, int32_t* n_out,
                             void* workspace, size_t workspace_bytes, v3d_stream_t stream) {
  return v3d_proposals_flag(head_maps, anchors, B, n_cls, n_yaw, H, W, topk, score_thresh_host, iou_threshold, out_boxes,
                            out_batch_idx, out_class_idx, out_scores, n_out, nullptr, workspace, workspace_bytes, stream);
}

// topk_boxes / topk_scores != NULL: stop behind the decode -- the (N, 7) decoded candidates and their scores in candidate order
// (group-major, score descending inside a group) are the result (v3d_proposals_topk); the NMS half does not run.
static int prop_run(const float* head_maps, const float* anchors, int B, int n_cls, int n_yaw, int H, int W, int topk,
                    const float* score_thresh_host, float iou_threshold, float* out_boxes, int64_t* out_batch_idx, int64_t* out_class_idx,
                    float* out_scores, int32_t* n_out, const int32_t* aux_flag, void* workspace, size_t workspace_bytes, v3d_stream_t stream,
                    float* topk_boxes, float* topk_scores);

extern "C" int v3d_proposals_flag(const float* head_maps, const float* anchors, int B, int n_cls, int n_yaw, int H, int W,
                                  int topk, const float* score_thresh_host, float iou_threshold, float* out_boxes,
                                  int64_t* out_batch_idx, int64_t* out_class_idx, float* out_scores, int32_t* n_out,
                                  const int32_t* aux_flag, void* workspace, size_t workspace_bytes, v3d_stream_t stream) {
  if (!score_thresh_host || !out_boxes || !out_batch_idx || !out_class_idx || !out_scores || !n_out) return V3D_EINVAL;
  return prop_run(head_maps, anchors, B, n_cls, n_yaw, H, W, topk, score_thresh_host, iou_threshold, out_boxes, out_batch_idx, out_class_idx,
                  out_scores, n_out, aux_flag, workspace, workspace_bytes, stream, nullptr, nullptr);
}

extern "C" int v3d_proposals_topk(const float* head_maps, const float* anchors, int B, int n_cls, int n_yaw, int H, int W, int topk,
                                  float* boxes, float* scores, void* workspace, size_t workspace_bytes, v3d_stream_t stream) {
  if (!boxes || !scores) return V3D_EINVAL;
  return prop_run(head_maps, anchors, B, n_cls, n_yaw, H, W, topk, nullptr, 0.f, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, workspace,
                  workspace_bytes, stream, boxes, scores);
}

static int prop_run(const float* head_maps, const float* anchors, int B, int n_cls, int n_yaw, int H, int W, int topk,
                    const float* score_thresh_host, float iou_threshold, float* out_boxes, int64_t* out_batch_idx, int64_t* out_class_idx,
                    float* out_scores, int32_t* n_out, const int32_t* aux_flag, void* workspace, size_t workspace_bytes, v3d_stream_t stream,
                    float* topk_boxes, float* topk_scores) {
  hipStream_t st = (hipStream_t)stream;
  if (!head_maps || !anchors || !workspace) return V3D_EINVAL;
  if (B < 1 || n_cls < 1 || n_cls > PROP_MAX_CLS || n_yaw < 1 || H < 1 || W < 1 || topk < 1 || topk > PROP_MAX_TOPK)
    return V3D_EINVAL;
  if ((long long)B * n_cls * topk > PROP_THREADS) return V3D_EUNSUPPORTED;  // candidates of ALL groups are sorted by one block
  if ((long long)n_yaw * H * W < topk) return V3D_EINVAL;  // torch.topk raises as well
  if (workspace_bytes < v3d_proposals_workspace(B, n_cls, topk)) return V3D_EWORKSPACE;
  PropGeom g;
  g.B = B; g.n_cls = n_cls; g.n_yaw = n_yaw; g.HW = H * W; g.topk = topk; g.ctot = n_cls * n_yaw * 8;
  for (int c = 0; c < PROP_MAX_CLS; c++) g.thresh[c] = (c < n_cls && score_thresh_host) ? score_thresh_host[c] : 0.f;
  const size_t N = (size_t)B * n_cls * topk;
  char* p = (char*)workspace;
  auto take = [&](size_t bytes) { char* q = p; p += prop_align(bytes); return (void*)q; };
  float* cand_score = (float*)take(N * 4);
  if (topk_scores) cand_score = topk_scores;
  int* cand_anchor = (int*)take(N * 4);
  float* part_score = (float*)take(N * PROP_CHUNKS * 4);
  int* part_idx = (int*)take(N * PROP_CHUNKS * 4);
  float* boxes = (float*)take(N * 7 * 4);
  if (topk_boxes) boxes = topk_boxes;
  long long* bidx = (long long*)take(N * 8);
  long long* cidx = (long long*)take(N * 8);
  long long* keep = (long long*)take(N * 8);
  float* bev = (float*)take(N * 5 * 4);
  int* n_keep = (int*)take(256);
  const size_t nms_bytes = v3d_nms_rotated_workspace((int)N);
  void* nms_ws = take(nms_bytes);

  const int n_per_group = n_yaw * H * W;
  // a level-1 workgroup holds its slice in registers (<= 8 x 1024 elements), the merge block chunks*topk (<= 4096)
  int chunks = PROP_CHUNKS;
  while (chunks > 1 && (chunks * topk > 4 * SEL_THREADS || n_per_group / chunks < 4 * topk)) chunks--;
  const int chunk_len = ((n_per_group + chunks - 1) / chunks + 63) & ~63;
  if (chunks * topk > 4 * SEL_THREADS || chunk_len > 8 * SEL_THREADS) return V3D_EUNSUPPORTED;
  if (chunk_len <= 4 * SEL_THREADS)
    hipLaunchKernelGGL(prop_topk_chunk_kernel<4>, dim3(chunks, B * n_cls), dim3(SEL_THREADS), 0, st, head_maps, g, chunk_len,
                       part_score, part_idx);
  else
    hipLaunchKernelGGL(prop_topk_chunk_kernel<8>, dim3(chunks, B * n_cls), dim3(SEL_THREADS), 0, st, head_maps, g, chunk_len,
                       part_score, part_idx);
  if (N > PROP_THREADS) return V3D_EUNSUPPORTED;  // one workgroup decodes and sorts all candidates
  int* order = (int*)take(N * 4);
  v3d::BoxPrep* prep = (v3d::BoxPrep*)take(N * sizeof(v3d::BoxPrep));
  if (B * n_cls <= PROP_FUSE_GROUPS) {
    hipLaunchKernelGGL(prop_merge_decode_sort_kernel, dim3(1), dim3(PROP_THREADS), 0, st, part_score, part_idx, chunks, head_maps,
                       anchors, g, cand_anchor, cand_score, boxes, bidx, cidx, order, prep);
  } else {
    hipLaunchKernelGGL(prop_topk_merge_kernel, dim3(B * n_cls), dim3(SEL_THREADS), 0, st, part_score, part_idx, chunks, topk,
                       cand_score, cand_anchor);
    hipLaunchKernelGGL(prop_decode_sort_kernel, dim3(1), dim3(PROP_THREADS), 0, st, head_maps, anchors, g, cand_anchor,
                       cand_score, boxes, bidx, cidx, order, prep);
  }
  if (!topk_boxes) {
    const size_t nwords = (N + 63) / 64;
    unsigned long long* mask = (unsigned long long*)nms_ws;  // N*nwords + nwords words <= v3d_nms_rotated_workspace(N)
    const int rc = v3d_i_nms_mask_sorted(prep, (int)N, iou_threshold, mask, st);
    if (rc != V3D_OK) return rc;
    hipLaunchKernelGGL(prop_nms_reduce_finalize_kernel, dim3(1), dim3(PROP_NMS_THREADS), 0, st, mask, order, (int)N, mask + N * nwords,
                       keep, g, boxes, bidx, cidx, cand_score, out_boxes, (long long*)out_batch_idx, (long long*)out_class_idx,
                       out_scores, n_out, aux_flag);
  }
  (void)bev;
  (void)n_keep;
  V3D_CHECK_LAUNCH();
  return V3D_OK;
}

// Stage-2 tail (PV-RCNN: vision3d_amd/detector/model.py inference; the reference's refinement.py:32-33 raises, SURVEY.md 8(f) rank 3
// defines it): refined box = core/box_encode.py:13-21 decode of the residuals against the proposals, score = sigmoid of the
// confidence logit, coordinate-offset batched rotated NMS per (frame, class) group (ops/iou_nms.py:90-134), per-class score cut --
// the candidates arrive in the layout v3d_proposals_topk produces ((B, n_cls, topk) group-major).  3 launches, no host sync.
extern "C" size_t v3d_refine_nms_workspace(int B, int n_cls, int topk) { return v3d_proposals_workspace(B, n_cls, topk); }

extern "C" int v3d_refine_nms(const float* deltas, const float* proposals, const float* conf, int B, int n_cls, int topk,
                              const float* score_thresh_host, float iou_threshold, float* refined, float* out_boxes, int64_t* out_batch_idx,
                              int64_t* out_class_idx, float* out_scores, int32_t* n_out, void* workspace, size_t workspace_bytes,
                              v3d_stream_t stream) {
  hipStream_t st = (hipStream_t)stream;
  if (!deltas || !proposals || !conf || !score_thresh_host || !out_boxes || !out_batch_idx || !out_class_idx || !out_scores || !n_out ||
      !workspace)
    return V3D_EINVAL;
  if (B < 1 || n_cls < 1 || n_cls > PROP_MAX_CLS || topk < 1 || topk > PROP_MAX_TOPK) return V3D_EINVAL;
  if ((long long)B * n_cls * topk > PROP_THREADS) return V3D_EUNSUPPORTED;  // one workgroup decodes and sorts all candidates
  if (workspace_bytes < v3d_refine_nms_workspace(B, n_cls, topk)) return V3D_EWORKSPACE;
  PropGeom g;
  g.B = B; g.n_cls = n_cls; g.n_yaw = 1; g.HW = 1; g.topk = topk; g.ctot = 0;
  for (int c = 0; c < PROP_MAX_CLS; c++) g.thresh[c] = c < n_cls ? score_thresh_host[c] : 0.f;
  const size_t N = (size_t)B * n_cls * topk;
  char* p = (char*)workspace;
  auto take = [&](size_t bytes) { char* q = p; p += prop_align(bytes); return (void*)q; };
  float* scores = (float*)take(N * 4);
  float* boxes = refined ? refined : (float*)take(N * 7 * 4);
  long long* bidx = (long long*)take(N * 8);
  long long* cidx = (long long*)take(N * 8);
  long long* keep = (long long*)take(N * 8);
  int* order = (int*)take(N * 4);
  v3d::BoxPrep* prep = (v3d::BoxPrep*)take(N * sizeof(v3d::BoxPrep));
  unsigned long long* mask = (unsigned long long*)take(v3d_nms_rotated_workspace((int)N));
  hipLaunchKernelGGL(prop_refine_sort_kernel, dim3(1), dim3(PROP_THREADS), 0, st, deltas, proposals, conf, g, scores, boxes, bidx, cidx, order,
                     prep);
  const size_t nwords = (N + 63) / 64;
  const int rc = v3d_i_nms_mask_sorted(prep, (int)N, iou_threshold, mask, st);
  if (rc != V3D_OK) return rc;
  hipLaunchKernelGGL(prop_nms_reduce_finalize_kernel, dim3(1), dim3(PROP_NMS_THREADS), 0, st, mask, order, (int)N, mask + N * nwords, keep, g,
                     boxes, bidx, cidx, scores, out_boxes, (long long*)out_batch_idx, (long long*)out_class_idx, out_scores, n_out,
                     (const int*)nullptr);
  V3D_CHECK_LAUNCH();
  return V3D_OK;
}
