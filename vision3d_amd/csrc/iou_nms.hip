// iou_nms.hip -- rotated BEV IoU (A2), rotated NMS (A3) and points-in-boxes (A11) for gfx950.
//
// Replaces vision3d/ops/csrc/box_iou_rotated/box_iou_rotated_cuda.cu and
// vision3d/ops/csrc/nms_rotated/nms_rotated_cuda.cu (+ the numpy core/geometry.py masks).
// Design notes (wave64):
//   * the double-precision sin/cos of a box depends on its angle only -> every box is "prepped"
//     exactly once (BoxPrep), never once per pair as in the reference kernels;
//   * NMS bitmask words ARE wave ballots: lane = column inside a 64-wide column block, one
//     __ballot() yields the 64-bit suppression word the reference builds with a 64-step loop;
//   * the greedy reduction stays on the device (the reference copies the mask to the host,
//     nms_rotated_cuda.cu:106-128): per 64-box block the diagonal word is resolved with register
//     readlanes, then the kept rows are OR-ed into the remaining words by all lanes in parallel;
//   * the descending score sort is an in-LDS bitonic network on (score, index) keys.
#include "v3d_common.h"
#include "nms_device.h"
#include "rotated_iou.h"
#include "pib_device.h"

using v3d::BoxPrep;

// ------------------------------------------------------------------------------------------------
// pairwise IoU
// ------------------------------------------------------------------------------------------------
#define IOU_ROWS 32

__global__ __launch_bounds__(V3D_BLOCK) void box_iou_rotated_kernel(const float* __restrict__ b1, int M,
                                                                    const float* __restrict__ b2, int N,
                                                                    float* __restrict__ out) {
  __shared__ BoxPrep rows[IOU_ROWS];
  __shared__ v3d::P2 clip_pts[V3D_BLOCK / V3D_WAVE][24 * 64];  // the clipper's work arrays: LDS, not scratch (rotated_iou.h)
  __shared__ float clip_dist[V3D_BLOCK / V3D_WAVE][24 * 64];
  const int row0 = blockIdx.y * IOU_ROWS;
  const int nrows = min(IOU_ROWS, M - row0);
  if ((int)threadIdx.x < nrows) rows[threadIdx.x] = v3d::prep_box(b1 + 5 * (size_t)(row0 + threadIdx.x));
  __syncthreads();
  const int j = blockIdx.x * V3D_BLOCK + threadIdx.x;
  if (j >= N) return;
  const BoxPrep bj = v3d::prep_box(b2 + 5 * (size_t)j);
  v3d::P2* pts = clip_pts[threadIdx.x >> 6] + (threadIdx.x & 63);
  float* dist = clip_dist[threadIdx.x >> 6] + (threadIdx.x & 63);
  for (int r = 0; r < nrows; r++) out[(size_t)(row0 + r) * N + j] = v3d::iou_prepped_lds(rows[r], bj, pts, dist);
}

extern "C" int v3d_box_iou_rotated(const float* boxes1, int M, const float* boxes2, int N, float* ious,
                                   v3d_stream_t stream) {
  if (M < 0 || N < 0) return V3D_EINVAL;
  if (M == 0 || N == 0) return V3D_OK;
  if (!boxes1 || !boxes2 || !ious) return V3D_EINVAL;
  dim3 grid(v3d_ceil_div(N, V3D_BLOCK), v3d_ceil_div(M, IOU_ROWS));
  if (grid.y > 65535) return V3D_EUNSUPPORTED;
  hipLaunchKernelGGL(box_iou_rotated_kernel, grid, dim3(V3D_BLOCK), 0, (hipStream_t)stream, boxes1, M, boxes2, N,
                     ious);
  V3D_CHECK_LAUNCH();
  return V3D_OK;
}

// 3-D IoU of (x, y, z, w, l, h, yaw) boxes (z = centre): BEV intersection area through the SAME operator as above on columns
// (0, 1, 3, 4, 6) x overlap of the z extents / union of the volumes.  The reference declares box_iou_rotated_3d and raises
// (ops/iou_nms.py:12-13); this is the definition SURVEY.md 8(f) rank 3 asks for, checked against oracle/ (same float32
// operation order: bit-exact), not a parity claim against the reference.
struct Box3Prep {
  BoxPrep bev;
  float zlo, zhi, vol;
};
__device__ __forceinline__ Box3Prep prep_box3(const float* b) {
  const float bev[5] = {b[0], b[1], b[3], b[4], b[6]};
  Box3Prep r;
  r.bev = v3d::prep_box(bev);
  r.zlo = b[2] - b[5] / 2.f;
  r.zhi = b[2] + b[5] / 2.f;
  r.vol = b[3] * b[4] * b[5];
  return r;
}

__global__ __launch_bounds__(V3D_BLOCK) void box_iou_rotated_3d_kernel(const float* __restrict__ b1, int M,
                                                                       const float* __restrict__ b2, int N,
                                                                       float* __restrict__ out) {
  __shared__ Box3Prep rows[IOU_ROWS];
  __shared__ v3d::P2 clip_pts[V3D_BLOCK / V3D_WAVE][24 * 64];
  __shared__ float clip_dist[V3D_BLOCK / V3D_WAVE][24 * 64];
  const int row0 = blockIdx.y * IOU_ROWS;
  const int nrows = min(IOU_ROWS, M - row0);
  if ((int)threadIdx.x < nrows) rows[threadIdx.x] = prep_box3(b1 + 7 * (size_t)(row0 + threadIdx.x));
  __syncthreads();
  const int j = blockIdx.x * V3D_BLOCK + threadIdx.x;
  if (j >= N) return;
  const Box3Prep bj = prep_box3(b2 + 7 * (size_t)j);
  v3d::P2* pts = clip_pts[threadIdx.x >> 6] + (threadIdx.x & 63);
  float* dist = clip_dist[threadIdx.x >> 6] + (threadIdx.x & 63);
  for (int r = 0; r < nrows; r++) {
    const Box3Prep& bi = rows[r];
    const float inter_bev = v3d::inter_prepped_lds(bi.bev, bj.bev, pts, dist);
    const float oh = fmaxf(fminf(bi.zhi, bj.zhi) - fmaxf(bi.zlo, bj.zlo), 0.f);
    const float inter = inter_bev * oh;
    const float den = bi.vol + bj.vol - inter;
    out[(size_t)(row0 + r) * N + j] = den > 0.f ? inter / den : 0.f;
  }
}

extern "C" int v3d_box_iou_rotated_3d(const float* boxes1, int M, const float* boxes2, int N, float* ious,
                                      v3d_stream_t stream) {
  if (M < 0 || N < 0) return V3D_EINVAL;
  if (M == 0 || N == 0) return V3D_OK;
  if (!boxes1 || !boxes2 || !ious) return V3D_EINVAL;
  dim3 grid(v3d_ceil_div(N, V3D_BLOCK), v3d_ceil_div(M, IOU_ROWS));
  if (grid.y > 65535) return V3D_EUNSUPPORTED;
  hipLaunchKernelGGL(box_iou_rotated_3d_kernel, grid, dim3(V3D_BLOCK), 0, (hipStream_t)stream, boxes1, M, boxes2, N, ious);
  V3D_CHECK_LAUNCH();
  return V3D_OK;
}

// ------------------------------------------------------------------------------------------------
// NMS step 1: sort keys.  key = (~orderable(score) << 32) | index, ascending u64 order ==
// descending score, ties by ascending index.  Padding keys are all ones (sort last).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned orderable(float f) {
  const unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__global__ void nms_make_keys_kernel(const float* __restrict__ scores, int N, int Npad,
                                     unsigned long long* __restrict__ keys) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= Npad) return;
  keys[i] = i < N ? (((unsigned long long)(~orderable(scores[i]))) << 32) | (unsigned)i : ~0ull;
}

// whole bitonic network inside one block's LDS (Npad <= 4096 -> 32 KiB)
__global__ __launch_bounds__(1024) void bitonic_lds_kernel(unsigned long long* __restrict__ keys, int Npad) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long sk[];
  for (int i = threadIdx.x; i < Npad; i += blockDim.x) sk[i] = keys[i];
  __syncthreads();
  for (int k = 2; k <= Npad; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = threadIdx.x; i < Npad; i += blockDim.x) {
        const int ixj = i ^ j;
        if (ixj > i) {
          const unsigned long long a = sk[i], b = sk[ixj];
          const bool up = (i & k) == 0;
          if ((a > b) == up) {
            sk[i] = b;
            sk[ixj] = a;
          }
        }
      }
      __syncthreads();
    }
  }
  for (int i = threadIdx.x; i < Npad; i += blockDim.x) keys[i] = sk[i];
}

// one (k, j) compare-exchange pass over global memory (Npad > 4096)
__global__ void bitonic_global_step_kernel(unsigned long long* __restrict__ keys, int Npad, int j, int k) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= Npad) return;
  const int ixj = i ^ j;
  if (ixj > i) {
    const unsigned long long a = keys[i], b = keys[ixj];
    const bool up = (i & k) == 0;
    if ((a > b) == up) {
      keys[i] = b;
      keys[ixj] = a;
    }
  }
}

// NMS step 2: order + prepped boxes in sorted order
__global__ void nms_gather_kernel(const unsigned long long* __restrict__ keys, const float* __restrict__ boxes,
                                  int N, int* __restrict__ order, BoxPrep* __restrict__ prep) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const int src = (int)(keys[i] & 0xFFFFFFFFu);
  order[i] = src;
  prep[i] = v3d::prep_box(boxes + 5 * (size_t)src);
}

// NMS step 3: suppression bitmask.  mask[i*nwords + cb] bit c  <=>  IoU(sorted i, sorted cb*64+c) >= thr
// and cb*64+c > i.  One wave per (row, column block): lane = column, the 64-bit word IS the ballot.
// Every pair is evaluated by its own lane, so the latency of the inference shape (N = 100) is one IoU
// evaluation, not a 16- or 64-step loop (reference: nms_rotated_cuda.cu:53-65).
__global__ __launch_bounds__(V3D_WAVE) void nms_mask_kernel(const BoxPrep* __restrict__ prep, int N, int nwords,
                                                            float thr, unsigned long long* __restrict__ mask) {
  const int cb = blockIdx.x, row = blockIdx.y;
  if (cb < (row >> 6)) return;  // words left of the diagonal block are never read
  __shared__ v3d::P2 clip_pts[24 * 64];  // the clipper's work arrays, lane-interleaved (rotated_iou.h: LDS instead of scratch)
  __shared__ float clip_dist[24 * 64];
  const int lane = threadIdx.x;
  const int col = cb * 64 + lane;
  bool hit = false;
  if (col < N && col > row) {
    const BoxPrep br = prep[row];  // same address in every lane: one broadcast load
    const BoxPrep bc = prep[col];
    hit = v3d::iou_prepped_lds(br, bc, clip_pts + lane, clip_dist + lane) >= thr;
  }
  const unsigned long long word = __ballot(hit);
  if (lane == 0) mask[(size_t)row * nwords + cb] = word;
}

// The same mask for N > 128, one WAVE PER ROW.  In the kernel above a wave runs the polygon clipper as soon as ONE of its
// 64 column boxes is near the row box -- in score order that is almost every wave, although only a few per cent of the
// pairs are near (proposal stage, 1 000 boxes: 8 000 waves, all on the slow path, 19 us).  Here the wave walks the row's
// column blocks with the exact disjointness test only (iou_needs_clip: the pairs it rejects have IoU = +0.0f in the
// reference too), compacts the surviving columns into an LDS queue and runs the clipper on 64 QUEUED pairs at a time --
// typically once per row.  Bit-identical mask; words are assembled in LDS (queued hits arrive out of order) and written
// once.
__global__ __launch_bounds__(V3D_BLOCK) void nms_mask_rows_kernel(const BoxPrep* __restrict__ prep, int N, int nwords,
                                                                  float thr, unsigned long long* __restrict__ mask) {
  extern __shared__ unsigned long long nms_rows_sm[];  // per wave: words[nwords], queue[128] (ints)
  __shared__ v3d::P2 clip_pts[V3D_BLOCK / V3D_WAVE][24 * 64];  // the clipper's work arrays, one lane-interleaved slab per wave
  __shared__ float clip_dist[V3D_BLOCK / V3D_WAVE][24 * 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row = blockIdx.x * (V3D_BLOCK / V3D_WAVE) + wave;
  if (row >= N) return;  // (no workgroup barrier below)
  unsigned long long* words = nms_rows_sm + (size_t)wave * (nwords + 64);
  int* queue = reinterpret_cast<int*>(words + nwords);
  const BoxPrep br = prep[row];
  const bool zero_hits = 0.f >= thr;  // thr <= 0: disjoint pairs "overlap" too
  const int cb0 = row >> 6;           // words left of the diagonal block are never read
  int cnt = 0;                        // queued columns (wave-uniform)
  auto clip = [&](int m) {
    if (lane < m) {
      const int col = queue[lane];
      if (v3d::iou_prepped_lds(br, prep[col], clip_pts[wave] + lane, clip_dist[wave] + lane) >= thr)
        atomicOr(&words[col >> 6], 1ull << (col & 63));
    }
  };
  for (int cb = cb0; cb < nwords; cb++) {
    const int col = cb * 64 + lane;
    bool near = false, far_hit = false;
    if (col < N && col > row) {
      near = v3d::iou_needs_clip(br, prep[col]);
      far_hit = !near && zero_hits;
    }
    const unsigned long long nm = __ballot(near), fm = __ballot(far_hit);
    if (lane == 0) words[cb] = fm;
    if (near) queue[cnt + __popcll(nm & ((1ull << lane) - 1ull))] = col;
    cnt += __popcll(nm);
    if (cnt >= 64) {
      clip(64);
      const int rem = cnt - 64;
      const int v = lane < rem ? queue[64 + lane] : 0;
      if (lane < rem) queue[lane] = v;
      cnt = rem;
    }
  }
  clip(cnt);
  for (int w = cb0 + lane; w < nwords; w += 64) mask[(size_t)row * nwords + w] = words[w];
}

static void launch_nms_mask(const BoxPrep* prep, int N, int nwords, float thr, unsigned long long* mask, hipStream_t st) {
  if (nwords <= 2) {  // inference shape (N ~ 100): one evaluation per lane is already the whole latency
    hipLaunchKernelGGL(nms_mask_kernel, dim3(nwords, N), dim3(V3D_WAVE), 0, st, prep, N, nwords, thr, mask);
  } else {
    constexpr int RPB = V3D_BLOCK / V3D_WAVE;
    hipLaunchKernelGGL(nms_mask_rows_kernel, dim3(v3d_ceil_div(N, RPB)), dim3(V3D_BLOCK), (size_t)RPB * (nwords + 64) * 8, st, prep, N,
                       nwords, thr, mask);
  }
}

// NMS step 4: greedy reduction on the device (nms_device.h: shared with the fused tail of the proposal stage).
__global__ __launch_bounds__(V3D_BLOCK) void nms_reduce_kernel(const unsigned long long* __restrict__ mask,
                                                               const int* __restrict__ order, int N, int nwords,
                                                               unsigned long long* __restrict__ remv /*nwords*/,
                                                               long long* __restrict__ keep, int* __restrict__ n_keep) {
  __shared__ unsigned long long kept_s;
  __shared__ int nk_s;
  const int nk = v3d::nms_greedy_reduce<V3D_BLOCK>(mask, order, N, nwords, remv, keep, &kept_s, &nk_s);
  if (threadIdx.x == 0) *n_keep = nk;
}

static inline int next_pow2(int n) {
  int p = 1;
  while (p < n) p <<= 1;
  return p;
}

// Internal: mask + greedy reduction on boxes that are ALREADY sorted and prepped (csrc/proposal.hip fuses decode,
// key sort and box prep into one launch).  mask: N * ceil(N/64) words, remv: ceil(N/64) words.
// the suppression mask alone (the proposal stage reduces it in its own fused launch)
int v3d_i_nms_mask_sorted(const void* prep_sorted, int N, float iou_threshold, unsigned long long* mask, hipStream_t st) {
  if (N < 1 || N > 65535) return V3D_EUNSUPPORTED;
  launch_nms_mask((const BoxPrep*)prep_sorted, N, (N + 63) / 64, iou_threshold, mask, st);
  V3D_CHECK_LAUNCH();
  return V3D_OK;
}

int v3d_i_nms_sorted(const void* prep_sorted, const int* order, int N, float iou_threshold, int64_t* keep, int32_t* n_keep,
                     unsigned long long* mask, unsigned long long* remv, hipStream_t st) {
  if (N < 1 || N > 65535) return V3D_EUNSUPPORTED;
  const int nwords = (N + 63) / 64;
  launch_nms_mask((const BoxPrep*)prep_sorted, N, nwords, iou_threshold, mask, st);
  hipLaunchKernelGGL(nms_reduce_kernel, dim3(1), dim3(V3D_BLOCK), 0, st, mask, order, N, nwords, remv, (long long*)keep,
                     n_keep);
  V3D_CHECK_LAUNCH();
  return V3D_OK;
}

extern "C" size_t v3d_nms_rotated_workspace(int N) {
  if (N <= 0) return 256;
  const size_t npad = (size_t)next_pow2(N), nwords = (size_t)(N + 63) / 64;
  return v3d_align(npad * 8) + v3d_align((size_t)N * 4) + v3d_align((size_t)N * sizeof(BoxPrep)) +
         v3d_align((size_t)N * nwords * 8) + v3d_align(nwords * 8) + 256;
}

extern "C" int v3d_nms_rotated(const float* boxes, const float* scores, int N, float iou_threshold, int64_t* keep,
                               int32_t* n_keep, void* workspace, size_t workspace_bytes, v3d_stream_t stream) {
  hipStream_t st = (hipStream_t)stream;
  if (N < 0 || !n_keep) return V3D_EINVAL;
  if (N == 0) {
    V3D_CHECK_HIP(v3d_fill_async(n_keep, 0, sizeof(int32_t), st));
    return V3D_OK;
  }
  if (!boxes || !scores || !keep || !workspace) return V3D_EINVAL;
  const int npad = next_pow2(N), nwords = (N + 63) / 64;
  V3dArena ar(workspace, workspace_bytes);
  unsigned long long* keys = ar.take<unsigned long long>(npad);
  int* order = ar.take<int>(N);
  BoxPrep* prep = ar.take<BoxPrep>(N);
  unsigned long long* mask = ar.take<unsigned long long>((size_t)N * nwords);
  unsigned long long* remv = ar.take<unsigned long long>(nwords);
  if (!ar.ok()) return V3D_EWORKSPACE;

  hipLaunchKernelGGL(nms_make_keys_kernel, dim3(v3d_ceil_div(npad, 256)), dim3(256), 0, st, scores, N, npad, keys);
  if (npad <= 4096) {
    hipLaunchKernelGGL(bitonic_lds_kernel, dim3(1), dim3(1024), (size_t)npad * 8, st, keys, npad);
  } else {
    for (int k = 2; k <= npad; k <<= 1)
      for (int j = k >> 1; j > 0; j >>= 1)
        hipLaunchKernelGGL(bitonic_global_step_kernel, dim3(v3d_ceil_div(npad, 256)), dim3(256), 0, st, keys, npad, j, k);
  }
  hipLaunchKernelGGL(nms_gather_kernel, dim3(v3d_ceil_div(N, 256)), dim3(256), 0, st, keys, boxes, N, order, prep);
  if (N > 65535) return V3D_EUNSUPPORTED;  // grid.y limit
  launch_nms_mask(prep, N, nwords, iou_threshold, mask, st);
  hipLaunchKernelGGL(nms_reduce_kernel, dim3(1), dim3(V3D_BLOCK), 0, st, mask, order, N, nwords, remv,
                     (long long*)keep, n_keep);
  V3D_CHECK_LAUNCH();
  return V3D_OK;
}

// ------------------------------------------------------------------------------------------------
// points in boxes (core/geometry.py:4-65).  Corner arithmetic in fp64 exactly where numpy promotes
// (geometry.py:21); cos/sin evaluated on the float32 yaw.  One thread per point, boxes staged in LDS.
// ------------------------------------------------------------------------------------------------
#define PIB_BOXES 64

__global__ __launch_bounds__(V3D_BLOCK) void points_in_boxes_kernel(const float* __restrict__ pts, int N, int C,
                                                                    const float* __restrict__ boxes, int n, int use_z,
                                                                    uint8_t* __restrict__ mask) {
  __shared__ PibBox sb[PIB_BOXES];
  const int i = blockIdx.x * V3D_BLOCK + threadIdx.x;
  float px = 0.f, py = 0.f, pz = 0.f;
  if (i < N) {
    px = pts[(size_t)i * C];
    py = pts[(size_t)i * C + 1];
    pz = pts[(size_t)i * C + 2];
  }
  for (int b0 = 0; b0 < n; b0 += PIB_BOXES) {
    const int nb = min(PIB_BOXES, n - b0);
    __syncthreads();
    if ((int)threadIdx.x < nb) {
      const PibBox pb = pib_prep(boxes + 7 * (size_t)(b0 + threadIdx.x));
      sb[threadIdx.x] = pb;
    }
    __syncthreads();
    if (i < N) {
      for (int b = 0; b < nb; b++) {
        const bool in = pib_inside(sb[b], px, py, pz, use_z != 0);
        mask[(size_t)i * n + b0 + b] = in ? 1 : 0;
      }
    }
  }
}

extern "C" int v3d_points_in_boxes(const float* points, int N, int C, const float* boxes, int n, int use_z,
                                   uint8_t* mask, v3d_stream_t stream) {
  if (N < 0 || n < 0 || C < 3) return V3D_EINVAL;
  if (N == 0 || n == 0) return V3D_OK;
  if (!points || !boxes || !mask) return V3D_EINVAL;
  hipLaunchKernelGGL(points_in_boxes_kernel, dim3(v3d_ceil_div(N, V3D_BLOCK)), dim3(V3D_BLOCK), 0, (hipStream_t)stream,
                     points, N, C, boxes, n, use_z, mask);
  V3D_CHECK_LAUNCH();
  return V3D_OK;
}
