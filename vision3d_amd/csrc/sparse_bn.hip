// sparse_bn.hip -- training-mode BatchNorm1d (+ ReLU) over the (n_rows, C) feature matrix of a sparse tensor.
//
// Reference: nn.BatchNorm1d(eps=1e-3, momentum=0.01) + nn.ReLU applied to SparseConvTensor.features by
// spconv.SparseSequential (vision3d/detector/sparse_cnn.py:15-30).  In training torch runs it as 4-6 launches per
// layer and direction (channels-last statistics, transform, ReLU, their backward twins; ~50 us each at 80 k rows);
// here: forward = statistics + merge + fused normalise/affine/ReLU, backward = sums + merge + fused input gradient.
//   statistics: each workgroup owns a contiguous row chunk, two passes over it (mean, then sum of squared deviations
//   about THAT mean -- no E[x^2]-mean^2 cancellation; the second pass re-reads L2-resident rows); the chunk
//   triples (count, mean, M2) are merged in double precision with Chan's formula in chunk order: deterministic.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/vision3d_hip.h"
#include "v3d_internal.h"

#define SBN_MAX_CHUNKS 512
#define SBN_MIN_ROWS 256  // rows per chunk at least

__host__ __device__ static inline int sbn_chunks(int n) {
  int g = (n + SBN_MIN_ROWS - 1) / SBN_MIN_ROWS;
  return g < 1 ? 1 : (g > SBN_MAX_CHUNKS ? SBN_MAX_CHUNKS : g);
}

// The row count is either a host value (n_dev == nullptr) or lives in device memory (the plan's training path: no host
// read anywhere).  Every kernel derives the SAME chunking from it -- G = sbn_chunks(n), rows_per_chunk = ceil(n / G) -- so
// both forms give bit-identical results; with a device count the grids are sized from the capacity and surplus
// workgroups leave at once.
struct SbnRows {
  const int* n_dev;
  int n_host;  // the count itself, or the capacity when n_dev is set
  __device__ int n() const { return n_dev ? min(*n_dev, n_host) : n_host; }
};

// thread = (row lane rl, channel quad c4): 16-byte loads, C / 4 threads per row, 4 rows of every thread in flight per
// iteration (the scalar version -- one float per load, one row at a time -- ran at 0.8 TB/s on L2-resident rows)
__global__ __launch_bounds__(V3D_BLOCK) void sbn_stats_kernel(const float* __restrict__ x, SbnRows rows, int C,
                                                              float* __restrict__ part /*[G][3][C]: count, mean, M2*/) {
  __shared__ float4 red[V3D_BLOCK];
  __shared__ __attribute__((aligned(16))) float mean_s[V3D_BLOCK];
  const int n = rows.n(), G = sbn_chunks(n), rows_per_chunk = (n + G - 1) / G;
  if ((int)blockIdx.x >= G) return;
  const int C4 = C >> 2, tid = threadIdx.x, c4 = tid % C4, rl = tid / C4, RL = V3D_BLOCK / C4;
  const int r0 = blockIdx.x * rows_per_chunk, r1 = min(n, r0 + rows_per_chunk);
  const int cnt = max(0, r1 - r0);
  const float4* __restrict__ x4 = reinterpret_cast<const float4*>(x);
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  int r = r0 + rl;
  for (; r + 3 * RL < r1; r += 4 * RL) {
    const float4 a = x4[(size_t)r * C4 + c4], b = x4[(size_t)(r + RL) * C4 + c4];
    const float4 c = x4[(size_t)(r + 2 * RL) * C4 + c4], d = x4[(size_t)(r + 3 * RL) * C4 + c4];
    s.x += (a.x + b.x) + (c.x + d.x); s.y += (a.y + b.y) + (c.y + d.y);
    s.z += (a.z + b.z) + (c.z + d.z); s.w += (a.w + b.w) + (c.w + d.w);
  }
  for (; r < r1; r += RL) {
    const float4 a = x4[(size_t)r * C4 + c4];
    s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
  }
  red[tid] = s;
  __syncthreads();
  if (tid < C) {  // channel tid: its quad's partials over the row lanes
    const int q4 = tid >> 2, e = tid & 3;
    float t = 0.f;
    for (int q = 0; q < RL; q++) t += reinterpret_cast<const float*>(&red[q * C4 + q4])[e];
    mean_s[tid] = cnt ? t / (float)cnt : 0.f;
  }
  __syncthreads();
  const float4 m = *reinterpret_cast<const float4*>(&mean_s[4 * c4]);
  float4 m2 = make_float4(0.f, 0.f, 0.f, 0.f);
  r = r0 + rl;
  for (; r + 3 * RL < r1; r += 4 * RL) {
    const float4 a = x4[(size_t)r * C4 + c4], b = x4[(size_t)(r + RL) * C4 + c4];
    const float4 c = x4[(size_t)(r + 2 * RL) * C4 + c4], d = x4[(size_t)(r + 3 * RL) * C4 + c4];
#define SBN_SQ(v, f) ((v.f - m.f) * (v.f - m.f))
    m2.x += (SBN_SQ(a, x) + SBN_SQ(b, x)) + (SBN_SQ(c, x) + SBN_SQ(d, x));
    m2.y += (SBN_SQ(a, y) + SBN_SQ(b, y)) + (SBN_SQ(c, y) + SBN_SQ(d, y));
    m2.z += (SBN_SQ(a, z) + SBN_SQ(b, z)) + (SBN_SQ(c, z) + SBN_SQ(d, z));
    m2.w += (SBN_SQ(a, w) + SBN_SQ(b, w)) + (SBN_SQ(c, w) + SBN_SQ(d, w));
  }
  for (; r < r1; r += RL) {
    const float4 a = x4[(size_t)r * C4 + c4];
    m2.x += SBN_SQ(a, x); m2.y += SBN_SQ(a, y); m2.z += SBN_SQ(a, z); m2.w += SBN_SQ(a, w);
#undef SBN_SQ
  }
  __syncthreads();
  red[tid] = m2;
  __syncthreads();
  if (tid < C) {
    const int q4 = tid >> 2, e = tid & 3;
    float t = 0.f;
    for (int q = 0; q < RL; q++) t += reinterpret_cast<const float*>(&red[q * C4 + q4])[e];
    float* p = part + (size_t)blockIdx.x * 3 * C;
    p[tid] = (float)cnt;
    p[C + tid] = mean_s[tid];
    p[2 * C + tid] = t;
  }
}

// merge of the chunk triples by ONE 1024-thread workgroup, thread = (group q, channel c), in double precision:
//   mean = sum_b n_b * mean_b / n ;  M2 = sum_b [ M2_b + n_b * (mean_b - mean)^2 ]        (exact regrouping of the
// deviations about the global mean).  Two sweeps over the partials with independent loads and no division inside the
// loops (a serial Chan update over 512 chunks took 166 us, a grouped one 34 us: dependent loads + fp64 divisions; 256 threads
// 12 us -- a latency chain of 2 x 128 loads per thread -- 1024 threads a quarter of that).
#define SBN_MERGE_THREADS 1024
__global__ __launch_bounds__(SBN_MERGE_THREADS) void sbn_merge_kernel(const float* __restrict__ part, SbnRows rows, int C, float eps,
                                                              float* __restrict__ save_mean, float* __restrict__ save_invstd,
                                                              float* __restrict__ var_unbiased, float* __restrict__ running_mean,
                                                              float* __restrict__ running_var, float momentum,
                                                              long long* __restrict__ num_batches_tracked) {
  __shared__ double s_a[SBN_MERGE_THREADS];
  __shared__ double s_mean[SBN_MERGE_THREADS];
  const int n = rows.n(), G = sbn_chunks(n);
  const int tid = threadIdx.x, c = tid % C, q = tid / C, Q = SBN_MERGE_THREADS / C;
  double a = 0.0;
#pragma unroll 8
  for (int g = q; g < G; g += Q) {  // independent loads: keep 8 in flight
    const float* p = part + (size_t)g * 3 * C;
    a += (double)p[c] * (double)p[C + c];
  }
  s_a[tid] = a;
  __syncthreads();
  if (tid < C) {
    double t = 0.0;
    for (int j = 0; j < Q; j++) t += s_a[j * C + tid];
    s_mean[tid] = n > 0 ? t / (double)n : 0.0;
  }
  __syncthreads();
  const double mean = s_mean[c];
  a = 0.0;
#pragma unroll 8
  for (int g = q; g < G; g += Q) {
    const float* p = part + (size_t)g * 3 * C;
    const double d = (double)p[C + c] - mean;
    a += (double)p[2 * C + c] + (double)p[c] * d * d;
  }
  __syncthreads();
  s_a[tid] = a;
  __syncthreads();
  if (tid < C) {
    double m2 = 0.0;
    for (int j = 0; j < Q; j++) m2 += s_a[j * C + tid];
    const float var_b = n > 0 ? (float)(m2 / (double)n) : 0.f;
    save_mean[tid] = (float)mean;
    save_invstd[tid] = 1.f / sqrtf(var_b + eps);
    const float var_u = n > 1 ? (float)(m2 / (double)(n - 1)) : var_b;
    var_unbiased[tid] = var_u;
    if (running_mean) {  // nn.BatchNorm1d's update: r = (1 - momentum) * r + momentum * batch statistic
      running_mean[tid] = (1.f - momentum) * running_mean[tid] + momentum * (float)mean;
      running_var[tid] = (1.f - momentum) * running_var[tid] + momentum * var_u;
    }
    if (tid == 0 && num_batches_tracked) *num_batches_tracked += 1;
  }
}

// elementwise passes: C is a power of two >= 4 here (host-checked), a thread handles 4 consecutive channels of one row
__global__ __launch_bounds__(V3D_BLOCK) void sbn_apply_kernel(const float* __restrict__ x, SbnRows rows, int C,
                                                              const float* __restrict__ mean, const float* __restrict__ invstd,
                                                              const float* __restrict__ gamma, const float* __restrict__ beta,
                                                              int relu, float* __restrict__ y) {
  const long long total4 = (long long)rows.n() * C / 4;
  const int cmask = C - 1;
  for (long long t = (long long)blockIdx.x * V3D_BLOCK + threadIdx.x; t < total4; t += (long long)gridDim.x * V3D_BLOCK) {
    const int c = (int)((t << 2) & cmask);
    const float4 xv = reinterpret_cast<const float4*>(x)[t];
    const float4 m = *reinterpret_cast<const float4*>(mean + c), is = *reinterpret_cast<const float4*>(invstd + c);
    const float4 g = *reinterpret_cast<const float4*>(gamma + c), bt = *reinterpret_cast<const float4*>(beta + c);
    float4 v;
    v.x = (xv.x - m.x) * is.x * g.x + bt.x;
    v.y = (xv.y - m.y) * is.y * g.y + bt.y;
    v.z = (xv.z - m.z) * is.z * g.z + bt.z;
    v.w = (xv.w - m.w) * is.w * g.w + bt.w;
    if (relu) {
      v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
    }
    reinterpret_cast<float4*>(y)[t] = v;
  }
}

// backward sums per chunk: sum(dz) and sum(dz * xhat), dz = dy masked by the ReLU (y > 0); thread = (row lane, channel quad)
__global__ __launch_bounds__(V3D_BLOCK) void sbn_bwd_sums_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                                 SbnRows rows, int C, const float* __restrict__ mean,
                                                                 const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                                 const float* __restrict__ beta, int relu,
                                                                 float* __restrict__ part /*[G][2][C]*/) {
  __shared__ float4 red0[V3D_BLOCK], red1[V3D_BLOCK];
  const int n = rows.n(), G = sbn_chunks(n), rows_per_chunk = (n + G - 1) / G;
  if ((int)blockIdx.x >= G) return;
  const int C4 = C >> 2, tid = threadIdx.x, c4 = tid % C4, rl = tid / C4, RL = V3D_BLOCK / C4;
  const int r0 = blockIdx.x * rows_per_chunk, r1 = min(n, r0 + rows_per_chunk);
  const float4 m = reinterpret_cast<const float4*>(mean)[c4], is = reinterpret_cast<const float4*>(invstd)[c4];
  const float4 ga = reinterpret_cast<const float4*>(gamma)[c4], be = reinterpret_cast<const float4*>(beta)[c4];
  const float4* __restrict__ x4 = reinterpret_cast<const float4*>(x);
  const float4* __restrict__ g4 = reinterpret_cast<const float4*>(dy);
  float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0;
  // the ReLU mask is recomputed from x exactly as the forward computed y (one array less to read)
#define SBN_ACC(xv, gv, f)                                                   \
  {                                                                          \
    const float xhat = (xv.f - m.f) * is.f;                                  \
    const float dz = (relu && !(xhat * ga.f + be.f > 0.f)) ? 0.f : gv.f;     \
    s0.f += dz;                                                              \
    s1.f += dz * xhat;                                                       \
  }
  int r = r0 + rl;
  for (; r + RL < r1; r += 2 * RL) {  // two rows of every thread in flight
    const float4 xa = x4[(size_t)r * C4 + c4], gaa = g4[(size_t)r * C4 + c4];
    const float4 xb = x4[(size_t)(r + RL) * C4 + c4], gb = g4[(size_t)(r + RL) * C4 + c4];
    SBN_ACC(xa, gaa, x) SBN_ACC(xa, gaa, y) SBN_ACC(xa, gaa, z) SBN_ACC(xa, gaa, w)
    SBN_ACC(xb, gb, x) SBN_ACC(xb, gb, y) SBN_ACC(xb, gb, z) SBN_ACC(xb, gb, w)
  }
  for (; r < r1; r += RL) {
    const float4 xa = x4[(size_t)r * C4 + c4], gaa = g4[(size_t)r * C4 + c4];
    SBN_ACC(xa, gaa, x) SBN_ACC(xa, gaa, y) SBN_ACC(xa, gaa, z) SBN_ACC(xa, gaa, w)
  }
#undef SBN_ACC
  red0[tid] = s0;
  red1[tid] = s1;
  __syncthreads();
  if (tid < C) {
    const int q4 = tid >> 2, e = tid & 3;
    float t0 = 0.f, t1 = 0.f;
    for (int q = 0; q < RL; q++) {
      t0 += reinterpret_cast<const float*>(&red0[q * C4 + q4])[e];
      t1 += reinterpret_cast<const float*>(&red1[q * C4 + q4])[e];
    }
    part[(size_t)blockIdx.x * 2 * C + tid] = t0;
    part[(size_t)blockIdx.x * 2 * C + C + tid] = t1;
  }
}

__global__ __launch_bounds__(SBN_MERGE_THREADS) void sbn_bwd_merge_kernel(const float* __restrict__ part, SbnRows rows, int C,
                                                                  float* __restrict__ dbeta, float* __restrict__ dgamma) {
  __shared__ double s_a[SBN_MERGE_THREADS], s_b[SBN_MERGE_THREADS];
  const int G = sbn_chunks(rows.n());
  const int tid = threadIdx.x, c = tid % C, q = tid / C, Q = SBN_MERGE_THREADS / C;
  double a = 0.0, b = 0.0;
#pragma unroll 8
  for (int g = q; g < G; g += Q) {
    a += part[(size_t)g * 2 * C + c];
    b += part[(size_t)g * 2 * C + C + c];
  }
  s_a[tid] = a;
  s_b[tid] = b;
  __syncthreads();
  if (tid < C) {
    a = 0.0; b = 0.0;
    for (int j = 0; j < Q; j++) {
      a += s_a[j * C + tid];
      b += s_b[j * C + tid];
    }
    dbeta[tid] = (float)a;
    dgamma[tid] = (float)b;
  }
}

__global__ __launch_bounds__(V3D_BLOCK) void sbn_bwd_apply_kernel(const float* __restrict__ x,
                                                                  const float* __restrict__ dy, SbnRows rows, int C,
                                                                  const float* __restrict__ mean, const float* __restrict__ invstd,
                                                                  const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                  const float* __restrict__ dbeta,
                                                                  const float* __restrict__ dgamma, int relu,
                                                                  float* __restrict__ dx) {
  const int n = rows.n();
  const long long total4 = (long long)n * C / 4;
  const float inv_n = 1.f / (float)n;
  const int cmask = C - 1;
  for (long long t = (long long)blockIdx.x * V3D_BLOCK + threadIdx.x; t < total4; t += (long long)gridDim.x * V3D_BLOCK) {
    const int c = (int)((t << 2) & cmask);
    const float4 xv = reinterpret_cast<const float4*>(x)[t];
    const float4 gv = reinterpret_cast<const float4*>(dy)[t];
    const float xs[4] = {xv.x, xv.y, xv.z, xv.w}, gs[4] = {gv.x, gv.y, gv.z, gv.w};
    float o[4];
#pragma unroll
    for (int e = 0; e < 4; e++) {
      const float is = invstd[c + e];
      const float xhat = (xs[e] - mean[c + e]) * is;
      const float dz = (relu && !(xhat * gamma[c + e] + beta[c + e] > 0.f)) ? 0.f : gs[e];
      o[e] = gamma[c + e] * is * (dz - dbeta[c + e] * inv_n - xhat * dgamma[c + e] * inv_n);
    }
    reinterpret_cast<float4*>(dx)[t] = make_float4(o[0], o[1], o[2], o[3]);
  }
}

extern "C" size_t v3d_sparse_bn_workspace(int n, int C) {
  (void)n;
  return (size_t)SBN_MAX_CHUNKS * 3 * (size_t)(C > 0 ? C : 1) * sizeof(float) + 256;
}

static bool sbn_shape_ok(int n, int C) { return n >= 1 && C >= 4 && C <= V3D_BLOCK && (C & (C - 1)) == 0; }  // power of two

// n_dev == nullptr: n rows (host count).  n_dev != nullptr: min(*n_dev, n) rows, n = capacity of the buffers.
int v3d_i_sparse_bn_relu_fwd(const float* x, int n, const int32_t* n_dev, int C, const float* gamma, const float* beta, float eps,
                             int relu, float* y, float* save_mean, float* save_invstd, float* var_unbiased, float* running_mean,
                             float* running_var, float momentum, int64_t* num_batches_tracked, void* workspace,
                             size_t workspace_bytes, hipStream_t st) {
  if (!x || !y || !gamma || !beta || !save_mean || !save_invstd || !var_unbiased || !workspace) return V3D_EINVAL;
  if (!sbn_shape_ok(n, C)) return V3D_EUNSUPPORTED;
  if ((running_mean == nullptr) != (running_var == nullptr)) return V3D_EINVAL;
  if (workspace_bytes < v3d_sparse_bn_workspace(n, C)) return V3D_EWORKSPACE;
  const SbnRows rows{n_dev, n};
  float* part = (float*)workspace;
  hipLaunchKernelGGL(sbn_stats_kernel, dim3(sbn_chunks(n)), dim3(V3D_BLOCK), 0, st, x, rows, C, part);
  hipLaunchKernelGGL(sbn_merge_kernel, dim3(1), dim3(SBN_MERGE_THREADS), 0, st, part, rows, C, eps, save_mean, save_invstd,
                     var_unbiased, running_mean, running_var, momentum, (long long*)num_batches_tracked);
  const long long total = (long long)n * C / 4;
  const int blocks = (int)((total + V3D_BLOCK - 1) / V3D_BLOCK);
  hipLaunchKernelGGL(sbn_apply_kernel, dim3(blocks > 8192 ? 8192 : blocks), dim3(V3D_BLOCK), 0, st, x, rows, C, save_mean,
                     save_invstd, gamma, beta, relu, y);
  V3D_CHECK_LAUNCH();
  return V3D_OK;
}

int v3d_i_sparse_bn_relu_bwd(const float* x, const float* dy, int n, const int32_t* n_dev, int C, const float* gamma,
                             const float* beta, const float* save_mean, const float* save_invstd, int relu, float* dx,
                             float* dgamma, float* dbeta, void* workspace, size_t workspace_bytes, hipStream_t st) {
  if (!x || !dy || !gamma || !beta || !save_mean || !save_invstd || !dx || !dgamma || !dbeta || !workspace) return V3D_EINVAL;
  if (!sbn_shape_ok(n, C)) return V3D_EUNSUPPORTED;
  if (workspace_bytes < v3d_sparse_bn_workspace(n, C)) return V3D_EWORKSPACE;
  const SbnRows rows{n_dev, n};
  float* part = (float*)workspace;
  hipLaunchKernelGGL(sbn_bwd_sums_kernel, dim3(sbn_chunks(n)), dim3(V3D_BLOCK), 0, st, x, dy, rows, C, save_mean, save_invstd,
                     gamma, beta, relu, part);
  hipLaunchKernelGGL(sbn_bwd_merge_kernel, dim3(1), dim3(SBN_MERGE_THREADS), 0, st, part, rows, C, dbeta, dgamma);
  const long long total = (long long)n * C / 4;
  const int blocks = (int)((total + V3D_BLOCK - 1) / V3D_BLOCK);
  hipLaunchKernelGGL(sbn_bwd_apply_kernel, dim3(blocks > 8192 ? 8192 : blocks), dim3(V3D_BLOCK), 0, st, x, dy, rows, C,
                     save_mean, save_invstd, gamma, beta, dbeta, dgamma, relu, dx);
  V3D_CHECK_LAUNCH();
  return V3D_OK;
}

extern "C" int v3d_sparse_bn_relu_fwd(const float* x, int n, int C, const float* gamma, const float* beta, float eps, int relu,
                                      float* y, float* save_mean, float* save_invstd, float* var_unbiased, float* running_mean,
                                      float* running_var, float momentum, int64_t* num_batches_tracked, void* workspace,
                                      size_t workspace_bytes, v3d_stream_t stream) {
  return v3d_i_sparse_bn_relu_fwd(x, n, nullptr, C, gamma, beta, eps, relu, y, save_mean, save_invstd, var_unbiased, running_mean,
                                  running_var, momentum, num_batches_tracked, workspace, workspace_bytes, (hipStream_t)stream);
}

extern "C" int v3d_sparse_bn_relu_bwd(const float* x, const float* dy, int n, int C, const float* gamma, const float* beta,
                                      const float* save_mean, const float* save_invstd, int relu, float* dx, float* dgamma,
                                      float* dbeta, void* workspace, size_t workspace_bytes, v3d_stream_t stream) {
  return v3d_i_sparse_bn_relu_bwd(x, dy, n, nullptr, C, gamma, beta, save_mean, save_invstd, relu, dx, dgamma, dbeta, workspace,
                                  workspace_bytes, (hipStream_t)stream);
}
