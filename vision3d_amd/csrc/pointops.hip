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
//   * ball query: one thread per query, database points streamed through LDS tiles shared by the
//     workgroup; first-nsample-in-index-order semantics with first-hit prefill.
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
    unsigned long long best = 0ull;
#pragma unroll
    for (int j = 0; j < PPT; j++) {
      const int n = tid + j * FPS_THREADS;
      if (n < N) {
        const float dx = px[j] - lx, dy = py[j] - ly, dz = pz[j] - lz;
        const float d = dx * dx + dy * dy + dz * dz;
        const float d2 = fminf(d, td[j]);
        td[j] = d2;
        // d2 >= 0 -> its bit pattern is monotone; ~n makes the lowest index win ties
        const unsigned long long key = ((unsigned long long)__float_as_uint(d2) << 32) | (unsigned)(~n);
        best = key > best ? key : best;
      }
    }
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
#define BQ_TILE 1024

__global__ __launch_bounds__(V3D_BLOCK) void ball_query_kernel(const float* __restrict__ xyz,
                                                               const float* __restrict__ new_xyz, int N, int M,
                                                               float r2, int ns, int* __restrict__ idx) {
  __shared__ float tile[BQ_TILE * 3];
  const int b = blockIdx.y;
  const int j = blockIdx.x * V3D_BLOCK + threadIdx.x;
  const bool live = j < M;
  float qx = 0.f, qy = 0.f, qz = 0.f;
  if (live) {
    const float* q = new_xyz + ((size_t)b * M + j) * 3;
    qx = q[0];
    qy = q[1];
    qz = q[2];
  }
  int* o = idx + ((size_t)b * M + (live ? j : 0)) * ns;
  int cnt = 0;
  const float* base = xyz + (size_t)b * N * 3;
  for (int n0 = 0; n0 < N; n0 += BQ_TILE) {
    const int tn = min(BQ_TILE, N - n0);
    __syncthreads();
    for (int t = threadIdx.x; t < tn * 3; t += V3D_BLOCK) tile[t] = base[(size_t)n0 * 3 + t];
    __syncthreads();
    if (live && cnt < ns) {
      for (int t = 0; t < tn; t++) {
        const float dx = qx - tile[3 * t], dy = qy - tile[3 * t + 1], dz = qz - tile[3 * t + 2];
        const float d2 = dx * dx + dy * dy + dz * dz;
        if (d2 < r2) {
          if (cnt == 0)
            for (int s = 0; s < ns; s++) o[s] = n0 + t;
          o[cnt++] = n0 + t;
          if (cnt >= ns) break;
        }
      }
    }
  }
  if (live && cnt == 0)
    for (int s = 0; s < ns; s++) o[s] = 0;
}

extern "C" int v3d_ball_query(const float* xyz, const float* new_xyz, int B, int N, int M, float radius, int nsample,
                              int32_t* idx, v3d_stream_t stream) {
  if (B < 0 || N < 1 || M < 0 || nsample < 1) return V3D_EINVAL;
  if (B == 0 || M == 0) return V3D_OK;
  if (!xyz || !new_xyz || !idx) return V3D_EINVAL;
  hipLaunchKernelGGL(ball_query_kernel, dim3(v3d_ceil_div(M, V3D_BLOCK), B), dim3(V3D_BLOCK), 0, (hipStream_t)stream,
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
