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

// Wave64 / row-of-16 max and min reductions on the DPP data path (VALU latency; __shfl_xor goes through ds_bpermute,
// ~90 clocks per hop).  max/min are idempotent, so lanes whose DPP source is out of range simply combine with their
// own value.  quad swaps, row_shr:4, row_shr:8 leave lane 15 of every row with the row result; row_bcast:15 and
// row_bcast:31 carry it on to lane 63.
#define V3D_DPP_I(v, ctrl) __builtin_amdgcn_update_dpp((v), (v), (ctrl), 0xf, 0xf, false)
template <bool FULL>
__device__ __forceinline__ float v3d_dpp_max_f32(float v) {
#define STEP(ctrl) v = fmaxf(v, __int_as_float(V3D_DPP_I(__float_as_int(v), ctrl)))
  STEP(0xb1); STEP(0x4e); STEP(0x114); STEP(0x118);
  if (FULL) { STEP(0x142); STEP(0x143); }
#undef STEP
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), FULL ? 63 : 15));
}
template <bool FULL>
__device__ __forceinline__ int v3d_dpp_min_i32(int v) {
#define STEP(ctrl) v = min(v, V3D_DPP_I(v, ctrl))
  STEP(0xb1); STEP(0x4e); STEP(0x114); STEP(0x118);
  if (FULL) { STEP(0x142); STEP(0x143); }
#undef STEP
  return __builtin_amdgcn_readlane(v, FULL ? 63 : 15);
}

template <int PPT>
__global__ __launch_bounds__(FPS_THREADS) void fps_kernel(const float* __restrict__ xyz, int N, int K,
                                                          int* __restrict__ idx) {
  // One step (cycle-counter probe of the ds_bpermute / 64-bit-key version: update 2 200, wave reduce 1 050, barrier,
  // block scan 650-1 250 clocks): per-thread arg-max on plain floats, then (max distance, min index among its
  // holders) by two DPP reductions per wave, one LDS slot per wave, ONE barrier, and the same two reductions over the
  // 16 slots held by lanes 0-15.  Ties resolve to the lowest index at every level.
  __shared__ float wave_d[2][FPS_THREADS / V3D_WAVE];
  __shared__ int wave_n[2][FPS_THREADS / V3D_WAVE];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* p = xyz + (size_t)b * N * 3;
  int* out = idx + (size_t)b * K;
  // coordinates as float2 pairs: gfx950 runs v_pk_add_f32 / v_pk_mul_f32 on two points per lane and instruction
  typedef float f2 __attribute__((ext_vector_type(2)));
  constexpr int PP = (PPT + 1) / 2;
  f2 px[PP], py[PP], pz[PP];
  float td[2 * PP];
#pragma unroll
  for (int j = 0; j < 2 * PP; j++) {
    const int n = tid + j * FPS_THREADS;  // strided ownership: coalesced initial load
    const bool ok = n < N && j < PPT;
    px[j >> 1][j & 1] = ok ? p[3 * n] : 0.f;
    py[j >> 1][j & 1] = ok ? p[3 * n + 1] : 0.f;
    pz[j >> 1][j & 1] = ok ? p[3 * n + 2] : 0.f;
    td[j] = ok ? 1e10f : -1.f;  // fminf keeps -1 forever: a slot without a point can never win (distances are >= 0)
  }
  if (tid == 0) out[0] = 0;
  int last = 0;
  for (int s = 1; s < K; s++) {
    const float lx = p[3 * last], ly = p[3 * last + 1], lz = p[3 * last + 2];  // uniform -> scalar loads
    const f2 lx2 = {lx, lx}, ly2 = {ly, ly}, lz2 = {lz, lz};
    float bd = -1.f;
    int bn = 0x7FFFFFFF;
#pragma unroll
    for (int jj = 0; jj < PP; jj++) {
      const f2 dx = px[jj] - lx2, dy = py[jj] - ly2, dz = pz[jj] - lz2;
      const f2 d = dx * dx + dy * dy + dz * dz;  // (dx*dx + dy*dy) + dz*dz without contraction: the scalar form's bits
#pragma unroll
      for (int h = 0; h < 2; h++) {
        const int j = 2 * jj + h;
        const float d2 = fminf(d[h], td[j]);
        td[j] = d2;
        const bool better = d2 > bd;  // strict: the lowest of this thread's indices wins ties (n grows with j)
        bd = better ? d2 : bd;
        bn = better ? tid + j * FPS_THREADS : bn;
      }
    }
    const float wd = v3d_dpp_max_f32<true>(bd);
    const int wn = v3d_dpp_min_i32<true>(bd == wd ? bn : 0x7FFFFFFF);
    if (lane == 0) {
      wave_d[s & 1][wave] = wd;
      wave_n[s & 1][wave] = wn;
    }
    __syncthreads();  // double-buffered slots: one barrier per step suffices
    const float sd = wave_d[s & 1][lane & 15];
    const int sn = wave_n[s & 1][lane & 15];
    const float ad = v3d_dpp_max_f32<false>(sd);
    last = v3d_dpp_min_i32<false>(sd == ad ? sn : 0x7FFFFFFF);
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
