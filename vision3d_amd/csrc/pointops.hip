// pointops.hip -- PV-RCNN point ops (T4/T5): farthest-point sampling, gather, ball query, grouping.
//
// Replaces pointnet2_utils.{furthest_point_sample, gather_operation, ball_query, grouping_operation}
// as called at vision3d/detector/model.py:46-66 and detector/roi_grid_pool.py:64-72.
//   * FPS: one 1024-thread workgroup per frame; every thread keeps its points AND their running
//     min-distances in VGPRs for the whole K-step loop (the reference kernel re-reads global
//     memory each step: K*N*16 B = 537 MB for 16384->2048; here the compulsory N*12 B are read
//     once).  Arg-max per step = wave64 shuffle reduction on a packed (distance, ~index) key + one LDS
//     exchange between the 16 waves; ties resolve to the LOWEST index (the reference's block
//     reduction leaves ties unspecified).
//   * ball query: one WAVE per query (ballot = hit mask, popcount prefix = slot), database points streamed through
//     LDS tiles shared by the workgroup; first-nsample-in-index-order semantics with first-hit prefill.
#include "v3d_common.h"

// ------------------------------------------------------------------------------------------ FPS
#define FPS_THREADS 1024

template <int PPT>
__global__ __launch_bounds__(FPS_THREADS) void fps_kernel(const float* __restrict__ xyz, int N, int K,
                                                          int* __restrict__ idx) {
  __shared__ unsigned long long wave_best[2][FPS_THREADS / V3D_WAVE];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* p = xyz + (size_t)b * N * 3;
  int* out = idx + (size_t)b * K;
  float px[PPT], py[PPT], pz[PPT], td[PPT];
#pragma unroll
  for (int j = 0; j < PPT; j++) {
    const int n = tid + j * FPS_THREADS;  // strided ownership: coalesced initial load
    const bool ok = n < N;
    px[j] = ok ? p[3 * n] : 0.f;
    py[j] = ok ? p[3 * n + 1] : 0.f;
    pz[j] = ok ? p[3 * n + 2] : 0.f;
    td[j] = 1e10f;
  }
  if (tid == 0) out[0] = 0;
  int last = 0;
  for (int s = 1; s < K; s++) {
    const float lx = p[3 * last], ly = p[3 * last + 1], lz = p[3 * last + 2];  // uniform -> scalar loads
    // per-thread arg-max on plain floats (strict > keeps the lowest of this thread's indices on ties: n grows with j);
    // the 64-bit (distance, ~index) key is built once per thread, for the cross-lane exchange only.
    // Measured and NOT adopted (all bit-exact, none faster than this 1.9 us/step; the step is a latency chain of
    // scalar load -> update -> wave reduction -> barrier -> 16 LDS reads, not a throughput problem): float2 packed
    // distance math (v_pk_*), DPP instead of ds_bpermute reductions, winner coordinates through LDS instead of the
    // dependent global load (2.6-2.9 us/step with the 16-way coordinate select it needs).
    float bd = -1.f;
    int bn = 0;
#pragma unroll
    for (int j = 0; j < PPT; j++) {
      const int n = tid + j * FPS_THREADS;
      if (n < N) {
        const float dx = px[j] - lx, dy = py[j] - ly, dz = pz[j] - lz;
        const float d = dx * dx + dy * dy + dz * dz;
        const float d2 = fminf(d, td[j]);
        td[j] = d2;
        const bool better = d2 > bd;
        bd = better ? d2 : bd;
        bn = better ? n : bn;
      }
    }
    // d2 >= 0 -> its bit pattern is monotone; ~n makes the lowest index win ties
    unsigned long long best = bd >= 0.f ? (((unsigned long long)__float_as_uint(bd) << 32) | (unsigned)(~bn)) : 0ull;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const unsigned lo = __shfl_xor((unsigned)best, o), hi = __shfl_xor((unsigned)(best >> 32), o);
      const unsigned long long other = ((unsigned long long)hi << 32) | lo;
      best = other > best ? other : best;
    }
    if (lane == 0) wave_best[s & 1][wave] = best;
    __syncthreads();  // double-buffered slot: one barrier per step suffices
    unsigned long long all = wave_best[s & 1][0];
#pragma unroll
    for (int w = 1; w < FPS_THREADS / V3D_WAVE; w++) {
      const unsigned long long v = wave_best[s & 1][w];
      all = v > all ? v : all;
    }
    last = (int)(~(unsigned)(all & 0xFFFFFFFFu));
    if (tid == 0) out[s] = last;
  }
}

extern "C" size_t v3d_fps_workspace(int B, int N) {
  (void)B;
  (void)N;
  return 256;  // everything lives in registers/LDS; kept for ABI stability
}

extern "C" int v3d_furthest_point_sample(const float* xyz, int B, int N, int K, int32_t* idx, void* workspace,
                                         size_t workspace_bytes, v3d_stream_t stream) {
  (void)workspace;
  (void)workspace_bytes;
  hipStream_t st = (hipStream_t)stream;
  if (B < 0 || N < 1 || K < 1 || K > N) return V3D_EINVAL;
  if (B == 0) return V3D_OK;
  if (!xyz || !idx) return V3D_EINVAL;
  const int ppt = v3d_ceil_div(N, FPS_THREADS);
#define V3D_FPS(P)                                                                                   \
  if (ppt <= P) {                                                                                    \
    hipLaunchKernelGGL(fps_kernel<P>, dim3(B), dim3(FPS_THREADS), 0, st, xyz, N, K, idx);            \
    V3D_CHECK_LAUNCH();                                                                              \
    return V3D_OK;                                                                                   \
  }
  V3D_FPS(1) V3D_FPS(2) V3D_FPS(4) V3D_FPS(8) V3D_FPS(16) V3D_FPS(24) V3D_FPS(32) V3D_FPS(64)
#undef V3D_FPS
  return V3D_EUNSUPPORTED;  // N > 65536 per frame
}

// ------------------------------------------------------------------------------------------ gather
__global__ void gather_points_kernel(const float* __restrict__ feat, const int* __restrict__ idx, int C, int N, int K,
                                     long long total, float* __restrict__ out) {
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
    const int j = (int)(t % K);
    const long long bc = t / K;
    const int b = (int)(bc / C);
    out[t] = feat[bc * N + idx[(size_t)b * K + j]];
  }
}

extern "C" int v3d_gather_points(const float* feat, const int32_t* idx, int B, int C, int N, int K, float* out,
                                 v3d_stream_t stream) {
  if (B < 0 || C < 1 || N < 1 || K < 0) return V3D_EINVAL;
  const long long total = (long long)B * C * K;
  if (total == 0) return V3D_OK;
  if (!feat || !idx || !out) return V3D_EINVAL;
  const int blocks = (int)((total + 255) / 256);
  hipLaunchKernelGGL(gather_points_kernel, dim3(blocks > 4096 ? 4096 : blocks), dim3(256), 0, (hipStream_t)stream, feat,
                     idx, C, N, K, total, out);
  V3D_CHECK_LAUNCH();
  return V3D_OK;
}

// ------------------------------------------------------------------------------------------ ball query
// One WAVE per query: 64 database points are tested per step, the hit mask is a ballot, slots follow from popcount
// prefixes -- index order is lane order, so "the first nsample points inside the ball, in index order" needs no
// sorting.  (One THREAD per query left 8 workgroups on a 256-CU chip scanning 16 384 points each from LDS: 1.09 ms
// per call.)  A workgroup = 4 waves x BQ_QPW queries; the database streams through LDS tiles shared by all of them
// and the scan stops as soon as every query of the workgroup is full.
#define BQ_TILE 2048
#define BQ_QPW 2  // queries per wave

__global__ __launch_bounds__(V3D_BLOCK) void ball_query_kernel(const float* __restrict__ xyz,
                                                               const float* __restrict__ new_xyz, int N, int M,
                                                               float r2, int ns, int* __restrict__ idx) {
  __shared__ float tile[BQ_TILE * 3];
  __shared__ int open_queries;
  const int b = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int q0 = (blockIdx.x * (V3D_BLOCK / V3D_WAVE) + wave) * BQ_QPW;
  float qx[BQ_QPW], qy[BQ_QPW], qz[BQ_QPW];
  int cnt[BQ_QPW], first[BQ_QPW];
#pragma unroll
  for (int u = 0; u < BQ_QPW; u++) {
    const int j = q0 + u;
    const float* q = new_xyz + ((size_t)b * M + (j < M ? j : 0)) * 3;
    qx[u] = q[0];
    qy[u] = q[1];
    qz[u] = q[2];
    cnt[u] = j < M ? 0 : ns;  // a query beyond M is "full" from the start
    first[u] = 0;
  }
  const float* base = xyz + (size_t)b * N * 3;
  for (int n0 = 0; n0 < N; n0 += BQ_TILE) {
    const int tn = min(BQ_TILE, N - n0);
    __syncthreads();
    if (threadIdx.x == 0) open_queries = 0;
    for (int t = threadIdx.x; t < tn * 3; t += V3D_BLOCK) tile[t] = base[(size_t)n0 * 3 + t];
    __syncthreads();
#pragma unroll
    for (int u = 0; u < BQ_QPW; u++) {
      int* o = idx + ((size_t)b * M + (q0 + u < M ? q0 + u : 0)) * ns;
      for (int t0 = 0; t0 < tn && cnt[u] < ns; t0 += 64) {  // cnt is wave-uniform
        const int t = t0 + lane;
        bool hit = false;
        if (t < tn) {
          const float dx = qx[u] - tile[3 * t], dy = qy[u] - tile[3 * t + 1], dz = qz[u] - tile[3 * t + 2];
          hit = dx * dx + dy * dy + dz * dz < r2;
        }
        const unsigned long long m = __ballot(hit);
        if (m) {
          if (cnt[u] == 0) first[u] = n0 + t0 + __ffsll((long long)m) - 1;
          const int pos = cnt[u] + __popcll(m & ((1ull << lane) - 1ull));
          if (hit && pos < ns) o[pos] = n0 + t;
          cnt[u] = min(ns, cnt[u] + __popcll(m));
        }
      }
    }
    bool any_open = false;
#pragma unroll
    for (int u = 0; u < BQ_QPW; u++) any_open = any_open || cnt[u] < ns;
    if (lane == 0 && any_open) atomicOr(&open_queries, 1);
    __syncthreads();
    if (open_queries == 0) break;  // workgroup-uniform
  }
  // unfilled slots repeat the first hit (pointnet2 pre-fills all nsample slots with it); no hit at all -> index 0
#pragma unroll
  for (int u = 0; u < BQ_QPW; u++) {
    if (q0 + u < M) {
      int* o = idx + ((size_t)b * M + q0 + u) * ns;
      for (int sidx = cnt[u] + lane; sidx < ns; sidx += 64) o[sidx] = first[u];
    }
  }
}

extern "C" int v3d_ball_query(const float* xyz, const float* new_xyz, int B, int N, int M, float radius, int nsample,
                              int32_t* idx, v3d_stream_t stream) {
  if (B < 0 || N < 1 || M < 0 || nsample < 1) return V3D_EINVAL;
  if (B == 0 || M == 0) return V3D_OK;
  if (!xyz || !new_xyz || !idx) return V3D_EINVAL;
  hipLaunchKernelGGL(ball_query_kernel, dim3(v3d_ceil_div(M, (V3D_BLOCK / V3D_WAVE) * BQ_QPW), B), dim3(V3D_BLOCK), 0, (hipStream_t)stream,
                     xyz, new_xyz, N, M, radius * radius, nsample, idx);
  V3D_CHECK_LAUNCH();
  return V3D_OK;
}

// ------------------------------------------------------------------------------------------ grouping
__global__ void group_points_kernel(const float* __restrict__ feat, const int* __restrict__ idx, int C, int N, int M,
                                    int ns, long long total, float* __restrict__ out) {
  const long long per_bc = (long long)M * ns;
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
    const long long js = t % per_bc;
    const long long bc = t / per_bc;
    const int b = (int)(bc / C);
    out[t] = feat[bc * N + idx[(size_t)b * per_bc + js]];
  }
}

extern "C" int v3d_group_points(const float* feat, const int32_t* idx, int B, int C, int N, int M, int nsample,
                                float* out, v3d_stream_t stream) {
  if (B < 0 || C < 1 || N < 1 || M < 0 || nsample < 1) return V3D_EINVAL;
  const long long total = (long long)B * C * M * nsample;
  if (total == 0) return V3D_OK;
  if (!feat || !idx || !out) return V3D_EINVAL;
  const int blocks = (int)((total + 255) / 256);
  hipLaunchKernelGGL(group_points_kernel, dim3(blocks > 8192 ? 8192 : blocks), dim3(256), 0, (hipStream_t)stream, feat,
                     idx, C, N, M, nsample, total, out);
  V3D_CHECK_LAUNCH();
  return V3D_OK;
}
