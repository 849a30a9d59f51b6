// augment.hip -- GT-sampling + global augmentation of one training frame in TWO launches.
//
// Replaces the chain vision3d/dataset/augmentation.py:31-48 (sample :117-198 -> flip :78-95 -> scale :98-114 -> rotate :51-75),
// which the reference runs in numpy inside DataLoader workers.  The random draws are scalars made on the host through the
// reference's numpy calls in the reference's order (dataset/augmentation.py of this package); none of them depends on a
// device result, so all of them are known before the first launch and the device part is a pure function of
// (scene, database, draws):
//   aug_select_kernel (one workgroup)   pasted boxes = database boxes + float64 positions (:162-166); the collision filter
//       (:140-149): a sample survives iff its BEV rectangle has IoU > 1e-2 with nothing but itself, counted over the (n + k) boxes
//       with the SAME rotated-IoU operator as box_iou_rotated (rows = samples, columns = all boxes); kept samples are ranked, their
//       point segments laid out behind each other, their rectangles prepared for the point test; the output boxes / classes
//       (scene rows, then the kept samples in draw order) are written through the global transform.
//   aug_points_kernel                   scene points that lie in no kept rectangle (:195, core/geometry.py:51-58), compacted IN
//       ORDER by a single-pass scan over 256-point chunks with published counts (the rb_scan_emit scheme, rulebook.hip), then
//       the kept samples' points gathered from the database tensor and translated -- by extra workgroups of the same launch that
//       first add up ALL chunk counts (they sit behind the scene chunks in dispatch order: every wait is on a running block).
//       Every point leaves through the global transform.
// Arithmetic follows numpy's promotions, because they decide the values: with sampling the reference's points and boxes become
// float64 at the paste (float32 + float64 positions) and stay so until the dataset casts to float32 (kitti_dataset.py:119-120);
// without sampling everything stays float32.  T below is that working type.  flip = multiply by -1 (exact); scale = one
// multiply; rotate = x * c + y * (-s), x * s + y * c with c / s the float32 cos / sin of the float32 angle -- two products and
// one sum, not contracted (-ffp-contract=off, like torch's separate multiply / add kernels and numpy's matmul on a 2-vector).
// One host read per frame stays: the sizes of the ragged result.
#include "v3d_common.h"
#include "rotated_iou.h"
#include "pib_device.h"

using v3d::BoxPrep;

struct AugDraws {
  int flip;       // 1: y <- -y, yaw <- -yaw
  double factor;  // scale (a float32 value)
  double c, s;    // cos / sin of the float32 angle (float32 values)
  double theta;   // the float32 angle
};

struct AugSample {  // one drawn database object (host-filled, 32 bytes)
  int box_row;      // row of its box in the flat database box tensor
  int pt_start;     // first of its points in the flat database point tensor
  int pt_len;
  int cls;
  double px, py;    // float64 paste position
};

// int32 work area: head[4] = {kept samples, kept sample points, kept scene points, 0}, then keep[k], rank[k], poff[k], cum[k + 1],
// chunk_counts[n_chunks]; then (8-byte aligned) BoxPrep prep[n + k] as 8 floats each, PibBox rect[k].
#define AUG_HEAD 4
struct AugWork {
  int *head, *keep, *rank, *poff, *cum, *chunk_counts;
  float* prep;
  PibBox* rect;
  size_t bytes;
};
static __host__ __device__ inline AugWork aug_work(void* base, int n, int k, int n_chunks) {
  AugWork w;
  int* p = (int*)base;
  w.head = p;
  w.keep = p + AUG_HEAD;
  w.rank = w.keep + k;
  w.poff = w.rank + k;
  w.cum = w.poff + k;
  w.chunk_counts = w.cum + k + 1;
  size_t ints = (size_t)AUG_HEAD + 4 * (size_t)k + 1 + (size_t)n_chunks;
  ints = (ints + 3) & ~(size_t)3;
  w.prep = (float*)(p + ints);
  const size_t prep_floats = 8 * (size_t)(n + k);
  w.rect = (PibBox*)(w.prep + prep_floats);
  w.bytes = ints * 4 + prep_floats * 4 + (size_t)k * sizeof(PibBox);
  return w;
}

template <typename T>
struct AugXform {
  T factor, c, s, theta;
  bool flip;
  __device__ __forceinline__ explicit AugXform(const AugDraws& d)
      : factor((T)d.factor), c((T)d.c), s((T)d.s), theta((T)d.theta), flip(d.flip != 0) {}
  __device__ __forceinline__ void xyz(T& x, T& y, T& z) const {
    if (flip) y = y * (T)-1;
    x = x * factor;
    y = y * factor;
    z = z * factor;
    const T nx = x * c + y * (-s), ny = x * s + y * c;
    x = nx;
    y = ny;
  }
  __device__ __forceinline__ void box(T (&b)[7]) const {
    if (flip) {
      b[1] = b[1] * (T)-1;
      b[6] = b[6] * (T)-1;
    }
#pragma unroll
    for (int j = 0; j < 6; j++) b[j] = b[j] * factor;
    const T nx = b[0] * c + b[1] * (-s), ny = b[0] * s + b[1] * c;
    b[0] = nx;
    b[1] = ny;
    b[6] = b[6] + theta;
  }
};

// the pasted box of a sample in float64: database box (float32, xy demeaned) + position
__device__ __forceinline__ void aug_sample_box(const float* __restrict__ db_boxes, const AugSample& sm, double (&b)[7]) {
#pragma unroll
  for (int j = 0; j < 7; j++) b[j] = (double)db_boxes[7 * (size_t)sm.box_row + j];
  b[0] = b[0] + sm.px;
  b[1] = b[1] + sm.py;
}

__global__ __launch_bounds__(V3D_BLOCK) void aug_select_kernel(const float* __restrict__ boxes, const long long* __restrict__ class_idx, int n,
                                                               const float* __restrict__ db_boxes, const AugSample* __restrict__ samples,
                                                               int k, const AugDraws draws, void* work, int n_chunks,
                                                               float* __restrict__ out_boxes, long long* __restrict__ out_class_idx) {
  __shared__ v3d::P2 clip_pts[V3D_BLOCK / V3D_WAVE][24 * 64];  // the clipper's work arrays (rotated_iou.h)
  __shared__ float clip_dist[V3D_BLOCK / V3D_WAVE][24 * 64];
  const AugWork w = aug_work(work, n, k, n_chunks);
  const int tid = threadIdx.x, nb = n + k;
  // (w.prep: 8 floats reserved per box, the 7 of a BoxPrep used)
  // ---- BEV rectangles (x, y, w, l, yaw) of the scene boxes and the pasted boxes, float32 like the reference's concatenation
  for (int i = tid; i < nb; i += V3D_BLOCK) {
    float bev[5];
    if (i < n) {
      const float* b = boxes + 7 * (size_t)i;
      bev[0] = b[0], bev[1] = b[1], bev[2] = b[3], bev[3] = b[4], bev[4] = b[6];
    } else {
      double b[7];
      aug_sample_box(db_boxes, samples[i - n], b);
      bev[0] = (float)b[0], bev[1] = (float)b[1], bev[2] = (float)b[3], bev[3] = (float)b[4], bev[4] = (float)b[6];
    }
    reinterpret_cast<BoxPrep*>(w.prep + 8 * (size_t)i)[0] = v3d::prep_box(bev);
  }
  for (int j = tid; j < k; j += V3D_BLOCK) w.keep[j] = 0;  // (overlap counts first, the keep flags below)
  for (int i = tid; i < n_chunks; i += V3D_BLOCK) w.chunk_counts[i] = -1;
  __syncthreads();
  // ---- overlaps of every sample with every box (itself included: a kept sample counts exactly one)
  {
    v3d::P2* pts = clip_pts[tid >> 6] + (tid & 63);
    float* dist = clip_dist[tid >> 6] + (tid & 63);
    const long long pairs = (long long)k * nb;
    for (long long p = tid; p < pairs; p += V3D_BLOCK) {
      const int j = (int)(p / nb), col = (int)(p % nb);
      const BoxPrep a = *reinterpret_cast<const BoxPrep*>(w.prep + 8 * (size_t)(n + j));
      const BoxPrep b = *reinterpret_cast<const BoxPrep*>(w.prep + 8 * (size_t)col);
      if (v3d::iou_prepped_lds(a, b, pts, dist) > 1e-2f) atomicAdd(&w.keep[j], 1);
    }
  }
  __syncthreads();
  if (tid == 0) {  // k is tens: ranks and point offsets of the kept samples, in draw order
    int kept = 0, pts_kept = 0, cum = 0;
    for (int j = 0; j < k; j++) {
      const bool keep = w.keep[j] == 1;
      w.keep[j] = keep ? 1 : 0;
      w.rank[j] = keep ? kept : -1;
      w.poff[j] = pts_kept;
      w.cum[j] = cum;
      cum += samples[j].pt_len;
      if (keep) {
        kept++;
        pts_kept += samples[j].pt_len;
      }
    }
    w.cum[k] = cum;
    w.head[0] = kept;
    w.head[1] = pts_kept;
    w.head[2] = 0;
    w.head[3] = 0;
  }
  __syncthreads();
  // ---- rectangles for the point test, output boxes and classes
  const AugXform<double> xf(draws);
  for (int i = tid; i < nb; i += V3D_BLOCK) {
    double b[7];
    int row = i;
    long long cls;
    if (i < n) {
#pragma unroll
      for (int j = 0; j < 7; j++) b[j] = (double)boxes[7 * (size_t)i + j];
      cls = class_idx[i];
    } else {
      const int j = i - n;
      if (!w.keep[j]) continue;
      const AugSample sm = samples[j];
      aug_sample_box(db_boxes, sm, b);
      float bf[7];
#pragma unroll
      for (int q = 0; q < 7; q++) bf[q] = (float)b[q];
      w.rect[w.rank[j]] = pib_prep(bf);
      row = n + w.rank[j];
      cls = sm.cls;
    }
    xf.box(b);
#pragma unroll
    for (int j = 0; j < 7; j++) out_boxes[7 * (size_t)row + j] = (float)b[j];
    out_class_idx[row] = cls;
  }
}

#define AUG_RECTS 64
__global__ __launch_bounds__(V3D_BLOCK) void aug_points_kernel(const float4* __restrict__ points, int N, const float4* __restrict__ db_points,
                                                               const AugSample* __restrict__ samples, int n, int k,
                                                               const AugDraws draws, void* work, int n_chunks,
                                                               float4* __restrict__ out_points) {
  __shared__ PibBox rects[AUG_RECTS];
  __shared__ int lds[8];
  const AugWork w = aug_work(work, n, k, n_chunks);
  const int tid = threadIdx.x, b = blockIdx.x;
  const AugXform<double> xf(draws);
  const int kept = w.head[0];
  if (b < n_chunks) {
    // ---- a chunk of scene points: outside every kept rectangle?
    const int i = b * V3D_BLOCK + tid;
    float4 p = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < N) p = points[i];
    bool out = i < N;
    for (int r0 = 0; r0 < kept; r0 += AUG_RECTS) {
      const int nr = min(AUG_RECTS, kept - r0);
      __syncthreads();
      if (tid < nr) rects[tid] = w.rect[r0 + tid];
      __syncthreads();
      for (int r = 0; r < nr; r++) out = out && !pib_inside(rects[r], p.x, p.y, p.z, false);
    }
    int total;
    const int rank = v3d_block_rank(out, total, lds);
    if (tid == 0) v3d_publish_count(w.chunk_counts + b, total);
    int part = 0;
    for (int q = tid; q < b; q += V3D_BLOCK) part += v3d_wait_count(w.chunk_counts + q);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off);
    __syncthreads();
    if ((tid & 63) == 0) lds[tid >> 6] = part;
    __syncthreads();
    const int prefix = lds[0] + lds[1] + lds[2] + lds[3];
    if (b == n_chunks - 1 && tid == 0) w.head[2] = prefix + total;
    if (out) {
      double x = p.x, y = p.y, z = p.z;
      xf.xyz(x, y, z);
      out_points[prefix + rank] = make_float4((float)x, (float)y, (float)z, p.w);
    }
    return;
  }
  // ---- the kept samples' points, behind ALL kept scene points
  int part = 0;
  for (int q = tid; q < n_chunks; q += V3D_BLOCK) part += v3d_wait_count(w.chunk_counts + q);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off);
  if ((tid & 63) == 0) lds[tid >> 6] = part;
  __syncthreads();
  const int base = lds[0] + lds[1] + lds[2] + lds[3];
  const int all = w.cum[k], sblocks = gridDim.x - n_chunks;
  for (int q = (b - n_chunks) * V3D_BLOCK + tid; q < all; q += sblocks * V3D_BLOCK) {
    int lo = 0, hi = k;  // the sample j with cum[j] <= q < cum[j + 1]
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (w.cum[mid] <= q) lo = mid; else hi = mid;
    }
    const int j = lo;
    if (!w.keep[j]) continue;
    const AugSample sm = samples[j];
    const int i = q - w.cum[j];
    const float4 p = db_points[(size_t)sm.pt_start + i];
    double x = (double)p.x + sm.px, y = (double)p.y + sm.py, z = p.z;
    xf.xyz(x, y, z);
    out_points[(size_t)base + w.poff[j] + i] = make_float4((float)x, (float)y, (float)z, p.w);
  }
}

// without sampling: the global transform alone, in float32 (points of C >= 3 columns, boxes, one launch)
__global__ __launch_bounds__(V3D_BLOCK) void aug_transform_f32_kernel(const float* __restrict__ points, int N, int C, const float* __restrict__ boxes,
                                                                      int n, const AugDraws draws, float* __restrict__ out_points,
                                                                      float* __restrict__ out_boxes) {
  const AugXform<float> xf(draws);
  const int i = blockIdx.x * V3D_BLOCK + threadIdx.x;
  if (i < N) {
    const float* p = points + (size_t)i * C;
    float x = p[0], y = p[1], z = p[2];
    xf.xyz(x, y, z);
    float* o = out_points + (size_t)i * C;
    o[0] = x, o[1] = y, o[2] = z;
    for (int c = 3; c < C; c++) o[c] = p[c];
  } else if (i - N < n) {
    const int r = i - N;
    float b[7];
#pragma unroll
    for (int j = 0; j < 7; j++) b[j] = boxes[7 * (size_t)r + j];
    xf.box(b);
#pragma unroll
    for (int j = 0; j < 7; j++) out_boxes[7 * (size_t)r + j] = b[j];
  }
}

extern "C" size_t v3d_augment_work_bytes(int N, int n, int k) {
  if (N < 0 || n < 0 || k < 0) return 0;
  return aug_work(nullptr, n, k, v3d_ceil_div(N, V3D_BLOCK)).bytes;
}

extern "C" int v3d_augment_frame(const float* points, int N, int C, const float* boxes, const int64_t* class_idx, int n,
                                 const float* db_points, const float* db_boxes, const void* samples, int k, int sample_points,
                                 int flip, double factor, double cos_theta, double sin_theta, double theta, float* out_points,
                                 float* out_boxes, int64_t* out_class_idx, void* work, size_t work_bytes, v3d_stream_t stream) {
  if (N < 0 || n < 0 || k < 0 || C < 3 || sample_points < 0) return V3D_EINVAL;
  if ((N && (!points || !out_points)) || (n && (!boxes || !out_boxes))) return V3D_EINVAL;
  const AugDraws draws{flip ? 1 : 0, factor, cos_theta, sin_theta, theta};
  hipStream_t st = (hipStream_t)stream;
  if (k == 0) {  // no sampling: float32 throughout (classes are the caller's)
    if (N + n == 0) return V3D_OK;
    hipLaunchKernelGGL(aug_transform_f32_kernel, dim3(v3d_ceil_div(N + n, V3D_BLOCK)), dim3(V3D_BLOCK), 0, st, points, N, C, boxes, n,
                       draws, out_points, out_boxes);
    V3D_CHECK_LAUNCH();
    return V3D_OK;
  }
  if (C != 4) return V3D_EUNSUPPORTED;  // the database's points are (x, y, z, intensity): the paste concatenates 4 columns
  if (!db_points || !db_boxes || !samples || !work || !out_class_idx || (n && !class_idx) || !out_points || !out_boxes) return V3D_EINVAL;
  const int n_chunks = v3d_ceil_div(N, V3D_BLOCK);
  if (work_bytes < aug_work(nullptr, n, k, n_chunks).bytes) return V3D_EWORKSPACE;
  if (((uintptr_t)work & 7) || ((uintptr_t)points & 15) || ((uintptr_t)out_points & 15) || ((uintptr_t)db_points & 15)) return V3D_EINVAL;
  hipLaunchKernelGGL(aug_select_kernel, dim3(1), dim3(V3D_BLOCK), 0, st, boxes, (const long long*)class_idx, n, db_boxes,
                     (const AugSample*)samples, k, draws, work, n_chunks, out_boxes, (long long*)out_class_idx);
  V3D_CHECK_LAUNCH();
  const int sblocks = min(64, max(1, v3d_ceil_div(sample_points, V3D_BLOCK)));
  hipLaunchKernelGGL(aug_points_kernel, dim3(n_chunks + sblocks), dim3(V3D_BLOCK), 0, st, (const float4*)points, N,
                     (const float4*)db_points, (const AugSample*)samples, n, k, draws, work, n_chunks, (float4*)out_points);
  V3D_CHECK_LAUNCH();
  return V3D_OK;
}
