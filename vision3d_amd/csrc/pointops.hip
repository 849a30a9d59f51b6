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
//   * ball query: the scan form -- one WAVE per query (ballot = hit mask, popcount prefix = slot), database points streamed
//     through LDS tiles shared by the workgroup; first-nsample-in-index-order semantics with first-hit prefill -- and, since
//     round 6, the form the package calls: a cell grid of the database + per-wave LDS bitmaps (same indices; see further down).
#include <algorithm>
#include <cmath>

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

// ---- slab-skipping variant for 1 024 < N <= 16 384 (the PV-RCNN keypoint size) ---------------------------------
// The K steps are a dependent chain on ONE CU (an exchange between CUs costs more per step than the step itself), and the
// kernel above is bound by VALU ISSUE: 4 waves per SIMD x ~200 instructions x 4 clocks = the 3 300 clocks a step takes
// (a 256- or 512-thread launch of the same code is not faster).  So this variant cuts the instructions per step:
//   * 256 threads = one wave per SIMD, 16-64 points per lane in registers;
//   * points are binned by x inside the kernel (counting sort, 1 024 bins, LDS) and register slot j of every lane holds one
//     point of the j-th run of 256 sorted points -- an x-slab [xlo_j, xhi_j], whose bounds live in lane j;
//   * a step only touches slab j if the new sample can lower a running distance there: every running distance is <= D* (the
//     distance of the sample just chosen = the current maximum) and a point of slab j is at least gap_j = (x distance to the
//     slab) away, so  gap_j^2 >= D*  =>  min(td, d) == td in the whole slab  =>  skip.  One ballot per step gives the touched
//     set (the same in every wave); on the KITTI-shaped sweep 3.8 of 64 slabs are touched (profiles/r02_b_fps_slab_skip.txt);
//   * per lane the maximum is kept per group of 8 slots, so a touched slab costs 10 + 3 instructions, not a 64-slot rescan;
//   * the block maximum of the DISTANCE is found first (one DPP reduction per wave, one barrier); only the wave(s) holding it
//     recover the lowest original index and the point's coordinates, which travel through LDS (no global load in the loop).
// The result is the plain algorithm's bit for bit: the bound is exact in floating point (x - c is monotone, squares and sums
// of non-negative terms are monotone), ties still resolve to the lowest ORIGINAL index whatever order the bins have inside.
#define FPS_SLAB_THREADS 256
template <int PPT>
__global__ __launch_bounds__(FPS_SLAB_THREADS) void fps_slab_kernel(const float* __restrict__ xyz, int N, int K,
                                                                    int* __restrict__ idx) {
  constexpr int T = FPS_SLAB_THREADS, NW = T / V3D_WAVE, NBINS = 1024, BPT = NBINS / T, NG = PPT / 8;
  static_assert(PPT % 8 == 0 && PPT <= 64, "one slab per lane, groups of 8 slots");
  __shared__ float red_f[2][NW];
  __shared__ int bin_cnt[NBINS];
  __shared__ int wsum[NW];
  __shared__ int slab_bin[2 * PPT];
  __shared__ int perm[PPT * T];
  __shared__ float wave_d[2][NW];
  __shared__ int win_n[2][NW];
  __shared__ float win_xyz[2][NW][3];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* p = xyz + (size_t)b * N * 3;
  int* out = idx + (size_t)b * K;
  // ---- x range
  float mn = 3.4e38f, mx = -3.4e38f;
  for (int n = tid; n < N; n += T) {
    const float x = p[3 * n];
    mn = fminf(mn, x);
    mx = fmaxf(mx, x);
  }
  mx = v3d_dpp_max_f32<true>(mx);
  mn = -v3d_dpp_max_f32<true>(-mn);
  if (lane == 0) { red_f[0][wave] = mn; red_f[1][wave] = mx; }
  for (int i = tid; i < NBINS; i += T) bin_cnt[i] = 0;
  for (int i = tid; i < 2 * PPT; i += T) slab_bin[i] = 0;
  __syncthreads();
  mn = red_f[0][0];
  mx = red_f[1][0];
#pragma unroll
  for (int w = 1; w < NW; w++) { mn = fminf(mn, red_f[0][w]); mx = fmaxf(mx, red_f[1][w]); }
  const float span = fmaxf(mx - mn, 1e-20f);
  const float inv = (float)NBINS / span, width = span / (float)NBINS;
  auto bin_of = [&](float x) { return min(NBINS - 1, max(0, (int)((x - mn) * inv))); };
  // ---- counting sort of the point indices by bin: histogram, exclusive scan (thread t owns bins BPT t ..), scatter
  for (int n = tid; n < N; n += T) atomicAdd(&bin_cnt[bin_of(p[3 * n])], 1);
  __syncthreads();
  int cnt[BPT], mine = 0;
#pragma unroll
  for (int i = 0; i < BPT; i++) { cnt[i] = bin_cnt[tid * BPT + i]; mine += cnt[i]; }
  int incl = mine;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int t = __shfl_up(incl, off);
    if (lane >= off) incl += t;
  }
  if (lane == 63) wsum[wave] = incl;
  __syncthreads();
  int base = incl - mine;
  for (int w = 0; w < wave; w++) base += wsum[w];
#pragma unroll
  for (int i = 0; i < BPT; i++) {
    // slab boundaries: sorted positions 256 j (first) and 256 j + 255 (last) lie in the bins whose runs cover them
    if (cnt[i] > 0) {
      const int j0 = (base + T - 1) / T, j1 = (base + cnt[i] - 1) / T;  // slabs whose FIRST position falls in this bin's run
      for (int j = j0; j <= j1 && j < PPT; j++)
        if (j * T >= base && j * T < base + cnt[i]) slab_bin[2 * j] = tid * BPT + i;
      for (int j = base / T; j <= (base + cnt[i] - 1) / T && j < PPT; j++) {
        const int last_pos = min(N, (j + 1) * T) - 1;
        if (last_pos >= base && last_pos < base + cnt[i]) slab_bin[2 * j + 1] = tid * BPT + i;
      }
    }
    bin_cnt[tid * BPT + i] = base;  // becomes the scatter cursor (every thread rewrites only its own bins)
    base += cnt[i];
  }
  __syncthreads();
  for (int n = tid; n < N; n += T) perm[atomicAdd(&bin_cnt[bin_of(p[3 * n])], 1)] = n;
  __syncthreads();
  // ---- slots: sorted position j * 256 + tid.  Slab j's bounds (lane j of every wave), widened by two bins: bin edges are
  //      computed in floating point, the widening keeps them true bounds
  float px[PPT], py[PPT], pz[PPT], td[PPT];
  int orig[PPT];
#pragma unroll
  for (int j = 0; j < PPT; j++) {
    const int pos = j * T + tid;
    const bool ok = pos < N;
    const int n = ok ? perm[pos] : 0;
    orig[j] = ok ? n : 0x7FFFFFFF;
    px[j] = ok ? p[3 * n] : 0.f;
    py[j] = ok ? p[3 * n + 1] : 0.f;
    pz[j] = ok ? p[3 * n + 2] : 0.f;
    td[j] = ok ? 1e10f : -1.f;  // fminf keeps -1 forever: a slot without a point can never win (distances are >= 0)
  }
  const bool live_slab = lane < PPT && lane * T < N;
  const float xlo = live_slab ? mn + (float)(slab_bin[2 * (lane % PPT)] - 2) * width : 3.4e38f;  // an empty slab is never touched
  const float xhi = live_slab ? mn + (float)(slab_bin[2 * (lane % PPT) + 1] + 3) * width : -3.4e38f;
  float gmax[NG];  // per lane: maximum of each group of 8 slots
#pragma unroll
  for (int g = 0; g < NG; g++) {
    gmax[g] = td[8 * g];
#pragma unroll
    for (int e = 1; e < 8; e++) gmax[g] = fmaxf(gmax[g], td[8 * g + e]);
  }
  if (tid == 0) out[0] = 0;
  float cx = p[0], cy = p[1], cz = p[2];  // sample 0 = point 0
  float dstar = 1e10f;
  for (int s = 1; s < K; s++) {
    const float gap = fmaxf(fmaxf(xlo - cx, cx - xhi), 0.f);
    const unsigned long long touch = __ballot(gap * gap < dstar);  // bit j: slab j can change (same in every wave)
#pragma unroll
    for (int g = 0; g < NG; g++) {
      if ((touch >> (8 * g)) & 0xFFull) {  // scalar branches
#pragma unroll
        for (int e = 0; e < 8; e++) {
          const int j = 8 * g + e;
          if ((touch >> j) & 1ull) {
            const float dx = px[j] - cx, dy = py[j] - cy, dz = pz[j] - cz;
            const float d = dx * dx + dy * dy + dz * dz;  // (dx*dx + dy*dy) + dz*dz, no contraction: the oracle's bits
            td[j] = fminf(d, td[j]);
          }
        }
        gmax[g] = fmaxf(fmaxf(fmaxf(td[8 * g], td[8 * g + 1]), fmaxf(td[8 * g + 2], td[8 * g + 3])),
                        fmaxf(fmaxf(td[8 * g + 4], td[8 * g + 5]), fmaxf(td[8 * g + 6], td[8 * g + 7])));
      }
    }
    float bd = gmax[0];
#pragma unroll
    for (int g = 1; g < NG; g++) bd = fmaxf(bd, gmax[g]);
    const float wd = v3d_dpp_max_f32<true>(bd);
    if (lane == 0) wave_d[s & 1][wave] = wd;
    __syncthreads();
    float bm = wave_d[s & 1][0];
#pragma unroll
    for (int w = 1; w < NW; w++) bm = fmaxf(bm, wave_d[s & 1][w]);
    dstar = bm;
    // only the wave(s) that hold the maximum look for its lowest original index and the point behind it (a version in
    // which every wave published a full candidate before ONE barrier was slower: 2.84 vs 2.48 ms)
    int wn = 0x7FFFFFFF;
    if (wd == dstar) {
      int bn = 0x7FFFFFFF;
      float ox = 0.f, oy = 0.f, oz = 0.f;
#pragma unroll
      for (int g = 0; g < NG; g++) {
        if (__ballot(gmax[g] == dstar)) {  // scalar branch: usually one group of one wave
#pragma unroll
          for (int e = 0; e < 8; e++) {
            const int j = 8 * g + e;
            const bool hit = td[j] == dstar && orig[j] < bn;
            bn = hit ? orig[j] : bn;
            ox = hit ? px[j] : ox;
            oy = hit ? py[j] : oy;
            oz = hit ? pz[j] : oz;
          }
        }
      }
      wn = v3d_dpp_min_i32<true>(bn);
      if (bn == wn && wn != 0x7FFFFFFF) {  // the owning lane publishes the point
        win_xyz[s & 1][wave][0] = ox;
        win_xyz[s & 1][wave][1] = oy;
        win_xyz[s & 1][wave][2] = oz;
      }
    }
    if (lane == 0) win_n[s & 1][wave] = wn;
    __syncthreads();
    int last = win_n[s & 1][0], ww = 0;
#pragma unroll
    for (int w = 1; w < NW; w++) {
      const int c = win_n[s & 1][w];
      ww = c < last ? w : ww;
      last = min(last, c);
    }
    cx = win_xyz[s & 1][ww][0];
    cy = win_xyz[s & 1][ww][1];
    cz = win_xyz[s & 1][ww][2];
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
  if (N > 1024 && N <= 64 * FPS_SLAB_THREADS) {  // the PV-RCNN keypoint sizes: slab-skipping kernel
    const int slots = v3d_ceil_div(N, FPS_SLAB_THREADS);
    if (slots <= 16) hipLaunchKernelGGL(fps_slab_kernel<16>, dim3(B), dim3(FPS_SLAB_THREADS), 0, st, xyz, N, K, idx);
    else if (slots <= 32) hipLaunchKernelGGL(fps_slab_kernel<32>, dim3(B), dim3(FPS_SLAB_THREADS), 0, st, xyz, N, K, idx);
    else hipLaunchKernelGGL(fps_slab_kernel<64>, dim3(B), dim3(FPS_SLAB_THREADS), 0, st, xyz, N, K, idx);
    V3D_CHECK_LAUNCH();
    return V3D_OK;
  }
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

// Two radii around the same queries in ONE scan of the database (the multi-scale set-abstraction modules always ask for two:
// detector/model.py:58-66, roi_grid_pool.py:64-72): a step loads its 64 database points from the LDS tile once and tests them
// against BQ2_QPW queries x 2 radii, where the single-radius kernel re-read the tile per query and the caller scanned the
// database once per radius.  Per (query, radius) the result is the single kernel's: the first nsample points inside the ball in
// index order, unfilled slots repeat the first hit.  PV-RCNN stage 2: 4.08 -> 3.80 ms per frame (2 queries per wave: 4 leave half
// the chip idle at 2 048 queries -- no gain --, 1 gives 3.68 ms one at a time but less with frames in flight).
#define BQ2_QPW 2
__global__ __launch_bounds__(V3D_BLOCK) void ball_query2_kernel(const float* __restrict__ xyz, const float* __restrict__ new_xyz,
                                                                int N, int M, float r2a, int nsa, int* __restrict__ idxa,
                                                                float r2b, int nsb, int* __restrict__ idxb) {
  __shared__ float tile[BQ_TILE * 3];
  __shared__ int open_queries;
  const int b = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int q0 = (blockIdx.x * (V3D_BLOCK / V3D_WAVE) + wave) * BQ2_QPW;
  float qx[BQ2_QPW], qy[BQ2_QPW], qz[BQ2_QPW];
  int cnta[BQ2_QPW], cntb[BQ2_QPW], firsta[BQ2_QPW], firstb[BQ2_QPW];  // wave-uniform
#pragma unroll
  for (int u = 0; u < BQ2_QPW; u++) {
    const int j = q0 + u;
    const float* q = new_xyz + ((size_t)b * M + (j < M ? j : 0)) * 3;
    qx[u] = q[0];
    qy[u] = q[1];
    qz[u] = q[2];
    cnta[u] = j < M ? 0 : nsa;  // a query beyond M is "full" from the start
    cntb[u] = j < M ? 0 : nsb;
    firsta[u] = firstb[u] = 0;
  }
  const float* base = xyz + (size_t)b * N * 3;
  for (int n0 = 0; n0 < N; n0 += BQ_TILE) {
    const int tn = min(BQ_TILE, N - n0);
    __syncthreads();
    if (threadIdx.x == 0) open_queries = 0;
    for (int t = threadIdx.x; t < tn * 3; t += V3D_BLOCK) tile[t] = base[(size_t)n0 * 3 + t];
    __syncthreads();
    bool wave_open = false;
#pragma unroll
    for (int u = 0; u < BQ2_QPW; u++) wave_open = wave_open || cnta[u] < nsa || cntb[u] < nsb;
    for (int t0 = 0; t0 < tn && wave_open; t0 += 64) {
      const int t = t0 + lane;
      const bool in = t < tn;
      const float px = in ? tile[3 * t] : 0.f, py = in ? tile[3 * t + 1] : 0.f, pz = in ? tile[3 * t + 2] : 0.f;
      wave_open = false;
#pragma unroll
      for (int u = 0; u < BQ2_QPW; u++) {
        const float dx = qx[u] - px, dy = qy[u] - py, dz = qz[u] - pz;
        const float d2 = dx * dx + dy * dy + dz * dz;
        const size_t row = (size_t)b * M + (q0 + u < M ? q0 + u : 0);
        if (cnta[u] < nsa) {
          const bool hit = in && d2 < r2a;
          const unsigned long long m = __ballot(hit);
          if (m) {
            if (cnta[u] == 0) firsta[u] = n0 + t0 + __ffsll((long long)m) - 1;
            const int pos = cnta[u] + __popcll(m & ((1ull << lane) - 1ull));
            if (hit && pos < nsa) idxa[row * nsa + pos] = n0 + t;
            cnta[u] = min(nsa, cnta[u] + __popcll(m));
          }
        }
        if (cntb[u] < nsb) {
          const bool hit = in && d2 < r2b;
          const unsigned long long m = __ballot(hit);
          if (m) {
            if (cntb[u] == 0) firstb[u] = n0 + t0 + __ffsll((long long)m) - 1;
            const int pos = cntb[u] + __popcll(m & ((1ull << lane) - 1ull));
            if (hit && pos < nsb) idxb[row * nsb + pos] = n0 + t;
            cntb[u] = min(nsb, cntb[u] + __popcll(m));
          }
        }
        wave_open = wave_open || cnta[u] < nsa || cntb[u] < nsb;
      }
    }
    if (lane == 0 && wave_open) atomicOr(&open_queries, 1);
    __syncthreads();
    if (open_queries == 0) break;  // workgroup-uniform
  }
#pragma unroll
  for (int u = 0; u < BQ2_QPW; u++) {
    if (q0 + u < M) {
      const size_t row = (size_t)b * M + q0 + u;
      for (int sidx = cnta[u] + lane; sidx < nsa; sidx += 64) idxa[row * nsa + sidx] = firsta[u];
      for (int sidx = cntb[u] + lane; sidx < nsb; sidx += 64) idxb[row * nsb + sidx] = firstb[u];
    }
  }
}

// One entry point for one or two radii: idx_b == NULL -> the single-radius kernel (radius_b / nsample_b ignored); else both index
// sets from ONE scan of the database (PointnetSAModuleMSG's two scales, detector/model.py:46-66).
extern "C" int v3d_ball_query(const float* xyz, const float* new_xyz, int B, int N, int M, float radius_a, int nsample_a,
                              int32_t* idx_a, float radius_b, int nsample_b, int32_t* idx_b, v3d_stream_t stream) {
  if (B < 0 || N < 1 || M < 0 || nsample_a < 1 || (idx_b && nsample_b < 1)) return V3D_EINVAL;
  if (B == 0 || M == 0) return V3D_OK;
  if (!xyz || !new_xyz || !idx_a) return V3D_EINVAL;
  if (!idx_b)
    hipLaunchKernelGGL(ball_query_kernel, dim3(v3d_ceil_div(M, (V3D_BLOCK / V3D_WAVE) * BQ_QPW), B), dim3(V3D_BLOCK), 0, (hipStream_t)stream,
                       xyz, new_xyz, N, M, radius_a * radius_a, nsample_a, idx_a);
  else
    hipLaunchKernelGGL(ball_query2_kernel, dim3(v3d_ceil_div(M, (V3D_BLOCK / V3D_WAVE) * BQ2_QPW), B), dim3(V3D_BLOCK), 0,
                       (hipStream_t)stream, xyz, new_xyz, N, M, radius_a * radius_a, nsample_a, idx_a, radius_b * radius_b, nsample_b,
                       idx_b);
  V3D_CHECK_LAUNCH();
  return V3D_OK;
}

// ------------------------------------------------------------------------------------------ ball query over a cell grid
// The scan kernels above test EVERY database point against every query (2 048 x 16 384 distances per call, ~290 clocks per
// 64-point step on one wave per SIMD: 75 us per call, six calls per PV-RCNN frame = a third of stage 2's kernel time), and a
// cloud in training order (`kitti_dataset.py:154` shuffles the points) offers no index locality to prune by.  Here the database
// is first binned into square (x, y) cells no smaller than the larger radius (one 1 024-thread workgroup per frame: bounds,
// LDS histogram, scan, scatter of (x, y, z, index) records), and a query looks at the 3 x 3 cells around its own only:
//   * which points are inside the ball is decided by the scan kernels' own expression on the same operands (-ffp-contract=off),
//     so the hit SET is theirs bit for bit -- the cells only have to be a superset, which the 1 % margin on the cell edge
//     guarantees against the rounding of the two cell computations (see bq_cell);
//   * "the first nsample hits in INDEX order" does not depend on the order the candidates are met in: every hit sets bit
//     `index` of a per-wave LDS bitmap (N bits per radius), and the answer is read off the bitmap in ascending order --
//     popcount prefix over the lanes' word runs, the lanes below nsample emit their bits.
// Same results as v3d_ball_query for every input (tests/test_gpu_pointops.py: equality with the scan kernel at full size, with the
// CPU oracle at small sizes, non-finite coordinates, queries far outside the database).
#define BQG_MAX_KEYS 32768  // (index chunk, cell) keys of the LDS histogram: 128 KB of the build workgroup's LDS
#define BQG_MAX_CHUNKS 8
#define BQG_MAX_JOBS 8
#define BQG_BUILD_THREADS 1024
#define BQG_HEADER_BYTES 32
#define BQG_WAVES 4
#define BQG_RPL 8                  // records per lane requested before the first is looked at
#define BQG_BATCH (BQG_RPL * 64)

// INDEX CHUNKS.  A ball in a dense region holds hundreds of points but only the `nsample` lowest indices are wanted.  The records are
// therefore sorted by (index chunk, cell): chunk = index / ceil(N / nch), nch <= 8 as the histogram allows.  A query walks the chunks
// in order and stops once every radius has nsample hits among the chunks it has finished -- later chunks hold larger indices only.
struct BqGrid {  // first words of a frame's workspace, written by the build kernel
  float x0, y0, inv_c;
  int nx, ny, nch, chunk;  // cells along x / y, index chunks, points per chunk
};

__host__ __device__ static inline size_t bqg_sorted_offset() { return (BQG_HEADER_BYTES + (BQG_MAX_KEYS + 1) * 4 + 15) / 16 * 16; }
__host__ __device__ static inline size_t bqg_frame_bytes(int N) { return bqg_sorted_offset() + (size_t)N * 16; }

// Cell coordinate of x along an axis that starts at x0: the SAME expression for database points and queries.  Two values less
// than r apart differ by less than r * inv_c <= 1 / 1.01 before rounding and by at most 2 * 32768 * 2^-23 ~ 0.008 more after it
// (two roundings each, at most BQG_MAX_KEYS cells along an axis): their floors differ by at most one.
__device__ __forceinline__ float bq_cell(float x, float x0, float inv_c) { return floorf((x - x0) * inv_c); }

struct BqBuildJob {
  const float* xyz;   // (B, N, 3)
  unsigned char* ws;  // B frames of bqg_frame_bytes(N)
  int N;
  float cell_min;
};
struct BqBuildJobs {
  BqBuildJob j[BQG_MAX_JOBS];
};

// One workgroup per (database, frame).  PPT > 0: N <= PPT * BQG_BUILD_THREADS for every job and a thread keeps its points in
// registers (ONE trip to memory instead of three: a workgroup alone on its compute unit has nothing to hide a trip behind);
// PPT == 0: any N, the points are read again in every pass.
template <int PPT>
__global__ __launch_bounds__(BQG_BUILD_THREADS) void bq_grid_build_kernel(const BqBuildJobs jobs) {
  extern __shared__ int bq_cnt[];  // [BQG_MAX_KEYS]
  __shared__ float red[4][BQG_BUILD_THREADS / V3D_WAVE];
  __shared__ int wsum[BQG_BUILD_THREADS / V3D_WAVE];
  __shared__ BqGrid g;
  int* cnt = bq_cnt;
  const BqBuildJob job = jobs.j[blockIdx.x];
  const int N = job.N;
  const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* p = job.xyz + (size_t)b * N * 3;
  unsigned char* w = job.ws + (size_t)b * bqg_frame_bytes(N);
  int* cell_start = reinterpret_cast<int*>(w + BQG_HEADER_BYTES);
  float4* sorted = reinterpret_cast<float4*>(w + bqg_sorted_offset());
  const float inf = __builtin_huge_valf();
  [[maybe_unused]] float px[PPT ? PPT : 1], py[PPT ? PPT : 1], pz[PPT ? PPT : 1];
  if constexpr (PPT > 0) {
#pragma unroll
    for (int k = 0; k < PPT; k++) {
      const int i = tid + k * BQG_BUILD_THREADS;
      px[k] = i < N ? p[3 * (size_t)i] : inf;  // (a slot beyond N reads as a non-finite point: left out like one)
      py[k] = i < N ? p[3 * (size_t)i + 1] : inf;
      pz[k] = i < N ? p[3 * (size_t)i + 2] : inf;
    }
  }
  // body(i, x, y, z) for every point with finite x and y (a point with a non-finite x or y can be in no ball: it is left out of the grid)
  auto for_points = [&](auto body) {
    if constexpr (PPT > 0) {
#pragma unroll
      for (int k = 0; k < PPT; k++)
        if (fabsf(px[k]) < inf && fabsf(py[k]) < inf) body(tid + k * BQG_BUILD_THREADS, px[k], py[k], pz[k]);
    } else {
      for (int i = tid; i < N; i += BQG_BUILD_THREADS) {
        const float x = p[3 * (size_t)i], y = p[3 * (size_t)i + 1], z = p[3 * (size_t)i + 2];
        if (fabsf(x) < inf && fabsf(y) < inf) body(i, x, y, z);
      }
    }
  };
  // (a) bounds
  float xlo = inf, xhi = -inf, ylo = inf, yhi = -inf;
  for_points([&](int, float x, float y, float) {
    xlo = fminf(xlo, x), xhi = fmaxf(xhi, x);
    ylo = fminf(ylo, y), yhi = fmaxf(yhi, y);
  });
  xlo = -v3d_dpp_max_f32<true>(-xlo), xhi = v3d_dpp_max_f32<true>(xhi);
  ylo = -v3d_dpp_max_f32<true>(-ylo), yhi = v3d_dpp_max_f32<true>(yhi);
  if (lane == 0) red[0][wave] = xlo, red[1][wave] = xhi, red[2][wave] = ylo, red[3][wave] = yhi;
  __syncthreads();
  if (tid == 0) {
    for (int i = 1; i < BQG_BUILD_THREADS / V3D_WAVE; i++) {
      xlo = fminf(xlo, red[0][i]), xhi = fmaxf(xhi, red[1][i]);
      ylo = fminf(ylo, red[2][i]), yhi = fmaxf(yhi, red[3][i]);
    }
    BqGrid t;
    t.x0 = xlo, t.y0 = ylo, t.inv_c = 0.f, t.nx = 0, t.ny = 0, t.nch = 1, t.chunk = N > 0 ? N : 1;
    if (xlo <= xhi) {  // at least one finite point
      float c = job.cell_min;
      for (;;) {  // cells no smaller than asked for, grown until the grid fits the LDS histogram
        t.inv_c = 1.f / c;
        const float fx = bq_cell(xhi, xlo, t.inv_c), fy = bq_cell(yhi, ylo, t.inv_c);  // (the largest cell along each axis)
        if (fx < (float)BQG_MAX_KEYS && fy < (float)BQG_MAX_KEYS && ((long long)fx + 1) * ((long long)fy + 1) <= BQG_MAX_KEYS) {
          t.nx = (int)fx + 1, t.ny = (int)fy + 1;
          break;
        }
        c *= 1.5f;
      }
      t.nch = min(BQG_MAX_CHUNKS, BQG_MAX_KEYS / (t.nx * t.ny));
      t.chunk = (N + t.nch - 1) / t.nch;
    }
    g = t;
    *reinterpret_cast<BqGrid*>(w) = t;
  }
  __syncthreads();
  const BqGrid gg = g;
  const int ncell = gg.nx * gg.ny, nkeys = ncell * gg.nch;
  // keys per thread of the scan below: covers nkeys + 1, odd (thread t walks the words [t * kpt, (t + 1) * kpt): an odd stride is
  // conflict-free)
  const int kpt = ((nkeys + BQG_BUILD_THREADS) / BQG_BUILD_THREADS) | 1;
  for (int i = tid; i < kpt * BQG_BUILD_THREADS; i += BQG_BUILD_THREADS)
    if (i < BQG_MAX_KEYS) cnt[i] = 0;
  __syncthreads();
  auto key_of = [&](int i, float x, float y) {
    return (i / gg.chunk) * ncell + (int)bq_cell(y, gg.y0, gg.inv_c) * gg.nx + (int)bq_cell(x, gg.x0, gg.inv_c);
  };
  // (b) histogram
  for_points([&](int i, float x, float y, float) { atomicAdd(&cnt[key_of(i, x, y)], 1); });
  __syncthreads();
  // (c) exclusive scan over the keys: kpt consecutive keys per thread, in place (the counts become the keys' write cursors)
  const int k_lo = tid * kpt;
  int sum = 0;
  for (int k = 0; k < kpt; k++)
    if (k_lo + k < BQG_MAX_KEYS) sum += cnt[k_lo + k];
  int incl = sum;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int o = __shfl_up(incl, d);
    if (lane >= d) incl += o;
  }
  if (lane == 63) wsum[wave] = incl;
  __syncthreads();
  int run = incl - sum;
  for (int i = 0; i < wave; i++) run += wsum[i];
  for (int k = 0; k < kpt; k++) {
    const int c = k_lo + k;
    if (c < BQG_MAX_KEYS) {
      const int own = cnt[c];
      cnt[c] = run;
      run += own;
    }
  }
  if (tid == BQG_BUILD_THREADS - 1) wsum[0] = run;  // the number of binned points
  __syncthreads();
  // the table as the queries read it: coalesced, not a thread's kpt consecutive words (the scan's layout: one 64-byte line per lane
  // and store instruction -- 30 us of a 45 us build at 26 000 keys); entry nkeys = the number of binned points
  for (int c = tid; c <= nkeys; c += BQG_BUILD_THREADS) cell_start[c] = c < nkeys ? cnt[c] : wsum[0];
  __syncthreads();
  // (d) scatter (the order inside a key's run is whatever the atomics give: the queries do not depend on it)
  for_points([&](int i, float x, float y, float z) { sorted[atomicAdd(&cnt[key_of(i, x, y)], 1)] = make_float4(x, y, z, __int_as_float(i)); });
}

// the first `ns` set bits of a wave's bitmap in ascending order -> o[0, ns) (empty slots repeat the first hit; no hit: 0), the
// bitmap left all zero.  Lane l owns the words [l * wpl, (l + 1) * wpl) (wpl odd: conflict-free).
__device__ __forceinline__ void bq_bitmap_emit(unsigned* __restrict__ bm, int wpl, int ns, int* __restrict__ o, int lane) {
  unsigned* mine = bm + lane * wpl;
  int cnt = 0;
  for (int k = 0; k < wpl; k++) cnt += __popc(mine[k]);
  int incl = cnt;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int up = __shfl_up(incl, d);
    if (lane >= d) incl += up;
  }
  const int total = __shfl(incl, 63);
  int pos = incl - cnt, first = 0;
  const unsigned long long holders = __ballot(cnt > 0);
  if (cnt > 0 && pos < ns) {
    for (int k = 0; k < wpl && pos < ns; k++) {
      unsigned wd = mine[k];
      while (wd && pos < ns) {
        const int bit = __ffs((int)wd) - 1;
        const int id = (lane * wpl + k) * 32 + bit;
        if (pos == 0) first = id;
        o[pos++] = id;
        wd &= wd - 1;
      }
    }
  }
  for (int k = 0; k < wpl; k++) mine[k] = 0u;
  if (total < ns) {
    first = holders ? __shfl(first, __ffsll((long long)holders) - 1) : 0;
    for (int s = total + lane; s < ns; s += 64) o[s] = first;
  }
}

struct BqQueryJob {
  const unsigned char* ws;  // B frames of bqg_frame_bytes(N)
  int* idxa;
  int* idxb;                // nullable: one radius
  size_t ws_stride;
  float r2a, r2b;
  int nsa, nsb;
};
struct BqQueryJobs {
  BqQueryJob j[BQG_MAX_JOBS];
};

// grid: (query blocks, frame, job) -- the jobs share the queries (the set-abstraction modules of a PV-RCNN frame all ask around the
// same keypoints, each in its own database: five dependent ~18 us launches as one)
__global__ __launch_bounds__(BQG_WAVES * 64) void bq_grid_query_kernel(const float* __restrict__ new_xyz, int M, const BqQueryJobs jobs,
                                                                       int wpl) {
  extern __shared__ unsigned bq_bits[];  // [waves][2][64 * wpl]
  const BqQueryJob job = jobs.j[blockIdx.z];
  const float r2a = job.r2a, r2b = job.r2b;
  const int nsa = job.nsa, nsb = job.nsb;
  int* __restrict__ idxa = job.idxa;
  int* __restrict__ idxb = job.idxb;
  const int b = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6, waves = blockDim.x >> 6;
  unsigned* ba = bq_bits + (size_t)wave * 2 * 64 * wpl;
  unsigned* bb = ba + 64 * wpl;
  for (int k = lane; k < 2 * 64 * wpl; k += 64) ba[k] = 0u;
  const unsigned char* w = job.ws + (size_t)b * job.ws_stride;
  const BqGrid g = *reinterpret_cast<const BqGrid*>(w);
  const int* cell_start = reinterpret_cast<const int*>(w + BQG_HEADER_BYTES);
  const float4* sorted = reinterpret_cast<const float4*>(w + bqg_sorted_offset());
  const int ncell = g.nx * g.ny;
  const float nanf_ = __builtin_nanf("");
  for (int q = blockIdx.x * waves + wave; q < M; q += gridDim.x * waves) {
    const float* qp = new_xyz + ((size_t)b * M + q) * 3;
    const float qx = qp[0], qy = qp[1], qz = qp[2];
    const float tx = bq_cell(qx, g.x0, g.inv_c), ty = bq_cell(qy, g.y0, g.inv_c);
    // (a query more than a cell outside the grid, or with a non-finite coordinate, has no candidates; the comparisons are
    //  false for NaN)
    if (tx >= -1.f && tx <= (float)g.nx && ty >= -1.f && ty <= (float)g.ny) {
      const int cx = (int)tx, cy = (int)ty;
      const int c0 = max(cx - 1, 0), c1 = min(cx + 1, g.nx - 1);
      // lane 3 * chunk + d: the run of row cy - 1 + d in that chunk (the three cells of a row are one run of records)
      int s_run = 0, len = 0;
      if (lane < 3 * g.nch) {
        const int ch = lane / 3, row = cy - 1 + lane % 3;
        if (row >= 0 && row < g.ny && c0 <= c1) {
          s_run = cell_start[ch * ncell + row * g.nx + c0];
          len = cell_start[ch * ncell + row * g.nx + c1 + 1] - s_run;
        }
      }
      int pre = len;  // inclusive prefix over the runs: positions in the query's candidate sequence
#pragma unroll
      for (int d = 1; d < 32; d <<= 1) {
        const int up = __shfl_up(pre, d);
        if (lane >= d) pre += up;
      }
      int cnt_a = 0, cnt_b = 0;
      // (run r of the walk below is wave-uniform: v_readlane, not a trip through the LDS crossbar per value -- 72 of them per sparse query)
      auto lane_of = [](int v, int l) { return __builtin_amdgcn_readlane(v, l); };
      for (int ch0 = 0; ch0 < g.nch;) {
        // a group of whole chunks: at least BQG_BATCH candidates (BQG_RPL records per lane in flight) or all that is left
        const int base = ch0 ? lane_of(pre, 3 * ch0 - 1) : 0;
        int ch1 = ch0 + 1, gend = lane_of(pre, 3 * ch1 - 1);
        while (ch1 < g.nch && gend - base < BQG_BATCH) ch1++, gend = lane_of(pre, 3 * ch1 - 1);
        for (int t0 = base; t0 < gend; t0 += BQG_BATCH) {
          int rec[BQG_RPL];
#pragma unroll
          for (int u = 0; u < BQG_RPL; u++) rec[u] = -1;
          for (int r = 3 * ch0; r < 3 * ch1; r++) {  // which run holds candidate t (wave-uniform walk over the group's runs)
            const int pe = lane_of(pre, r), ln = lane_of(len, r), sr = lane_of(s_run, r);
            if (ln == 0) continue;
#pragma unroll
            for (int u = 0; u < BQG_RPL; u++) {
              const int t = t0 + u * 64 + lane;
              if (t < pe && t >= pe - ln) rec[u] = sr + (t - (pe - ln));
            }
          }
          float4 pt[BQG_RPL];
#pragma unroll
          for (int u = 0; u < BQG_RPL; u++) pt[u] = rec[u] >= 0 ? sorted[rec[u]] : make_float4(nanf_, nanf_, nanf_, 0.f);  // (NaN: inside no ball)
#pragma unroll
          for (int u = 0; u < BQG_RPL; u++) {
            const float dx = qx - pt[u].x, dy = qy - pt[u].y, dz = qz - pt[u].z;
            const float d2 = dx * dx + dy * dy + dz * dz;
            const int id = __float_as_int(pt[u].w);
            const bool ha = d2 < r2a, hb = idxb && d2 < r2b;
            if (ha) atomicOr(&ba[id >> 5], 1u << (id & 31));
            if (hb) atomicOr(&bb[id >> 5], 1u << (id & 31));
            cnt_a += __popcll(__ballot(ha));
            cnt_b += __popcll(__ballot(hb));
          }
        }
        if (cnt_a >= nsa && (!idxb || cnt_b >= nsb)) break;  // the lowest nsample indices lie in the chunks walked so far
        ch0 = ch1;
      }
    }
    __threadfence_block();  // (the wave's own LDS atomics before its reads)
    bq_bitmap_emit(ba, wpl, nsa, idxa + ((size_t)b * M + q) * nsa, lane);
    if (idxb) bq_bitmap_emit(bb, wpl, nsb, idxb + ((size_t)b * M + q) * nsb, lane);
  }
}

extern "C" size_t v3d_ball_query_grid_workspace(int B, int N) {
  if (B < 1 || N < 1) return 0;
  return (size_t)B * bqg_frame_bytes(N);
}

// per-wave bitmaps of the query kernel: words per lane (odd) and waves per workgroup that fit 64 KB of LDS; waves = 0: N too large
static void bqg_bitmap_shape(int N, int& wpl, int& waves) {
  wpl = v3d_ceil_div(v3d_ceil_div(N, 32), 64) | 1;
  waves = (int)std::min<size_t>(BQG_WAVES, (size_t)64 * 1024 / ((size_t)2 * 64 * wpl * sizeof(unsigned)));
}

extern "C" int v3d_ball_query_grid_build(int n_db, const float* const* xyz, const int32_t* N, const float* radius_max,
                                         void* const* workspace, const size_t* workspace_bytes, int B, v3d_stream_t stream) {
  if (n_db < 0 || n_db > BQG_MAX_JOBS || B < 0) return V3D_EINVAL;
  if (n_db == 0 || B == 0) return V3D_OK;
  if (!xyz || !N || !radius_max || !workspace || !workspace_bytes) return V3D_EINVAL;
  BqBuildJobs jobs;
  int n_max = 0;
  for (int i = 0; i < n_db; i++) {
    const float r = fabsf(radius_max[i]);
    if (N[i] < 1 || !xyz[i] || !(r > 0.f) || !(r < 1e30f)) return V3D_EINVAL;
    if (!workspace[i] || ((uintptr_t)workspace[i] & 15) || workspace_bytes[i] < v3d_ball_query_grid_workspace(B, N[i])) return V3D_EINVAL;
    jobs.j[i] = BqBuildJob{xyz[i], (unsigned char*)workspace[i], N[i], r * 1.01f};
    n_max = std::max(n_max, N[i]);
  }
  hipStream_t st = (hipStream_t)stream;
  const size_t lds = (size_t)BQG_MAX_KEYS * sizeof(int);
#define BQG_BUILD(PPT)                                                                                                        \
  {                                                                                                                           \
    V3D_CHECK_HIP(hipFuncSetAttribute((const void*)bq_grid_build_kernel<PPT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
    hipLaunchKernelGGL(bq_grid_build_kernel<PPT>, dim3(n_db, B), dim3(BQG_BUILD_THREADS), lds, st, jobs);                      \
  }
  if (n_max <= 4 * BQG_BUILD_THREADS) BQG_BUILD(4)
  else if (n_max <= 12 * BQG_BUILD_THREADS) BQG_BUILD(12)
  else if (n_max <= 20 * BQG_BUILD_THREADS) BQG_BUILD(20)
  else BQG_BUILD(0)
#undef BQG_BUILD
  V3D_CHECK_LAUNCH();
  return V3D_OK;
}

extern "C" int v3d_ball_query_grid_query_many(int n_jobs, const float* new_xyz, int B, int M, const int32_t* N, const float* radius_a,
                                              const int32_t* nsample_a, int32_t* const* idx_a, const float* radius_b,
                                              const int32_t* nsample_b, int32_t* const* idx_b, const void* const* workspace,
                                              const size_t* workspace_bytes, v3d_stream_t stream) {
  if (n_jobs < 0 || n_jobs > BQG_MAX_JOBS || B < 0 || M < 0) return V3D_EINVAL;
  if (n_jobs == 0 || B == 0 || M == 0) return V3D_OK;
  if (!new_xyz || !N || !radius_a || !nsample_a || !idx_a || !workspace || !workspace_bytes) return V3D_EINVAL;
  BqQueryJobs jobs;
  int wpl = 1, waves = BQG_WAVES;
  for (int i = 0; i < n_jobs; i++) {
    const bool two = idx_b && idx_b[i];
    if (N[i] < 1 || nsample_a[i] < 1 || !idx_a[i] || (two && (!radius_b || !nsample_b || nsample_b[i] < 1))) return V3D_EINVAL;
    if (!workspace[i] || ((uintptr_t)workspace[i] & 15) || workspace_bytes[i] < v3d_ball_query_grid_workspace(B, N[i])) return V3D_EINVAL;
    int wp, wv;
    bqg_bitmap_shape(N[i], wp, wv);
    if (wv < 1) return V3D_EUNSUPPORTED;
    wpl = std::max(wpl, wp), waves = std::min(waves, wv);  // (a bitmap of the largest database serves every job)
    jobs.j[i] = BqQueryJob{(const unsigned char*)workspace[i], idx_a[i], two ? idx_b[i] : nullptr, bqg_frame_bytes(N[i]),
                           radius_a[i] * radius_a[i], two ? radius_b[i] * radius_b[i] : 0.f, nsample_a[i], two ? nsample_b[i] : 0};
  }
  const size_t per_wave = (size_t)2 * 64 * wpl * sizeof(unsigned);
  const int blocks = std::min(v3d_ceil_div(M, waves), 4096);
  hipLaunchKernelGGL(bq_grid_query_kernel, dim3(blocks, B, n_jobs), dim3(waves * 64), waves * per_wave, (hipStream_t)stream, new_xyz, M, jobs,
                     wpl);
  V3D_CHECK_LAUNCH();
  return V3D_OK;
}

extern "C" int v3d_ball_query_grid_query(const float* new_xyz, int B, int N, int M, float radius_a, int nsample_a, int32_t* idx_a,
                                         float radius_b, int nsample_b, int32_t* idx_b, const void* workspace, size_t workspace_bytes,
                                         v3d_stream_t stream) {
  if (N < 1 || nsample_a < 1 || (idx_b && nsample_b < 1)) return V3D_EINVAL;
  return v3d_ball_query_grid_query_many(1, new_xyz, B, M, &N, &radius_a, &nsample_a, &idx_a, &radius_b, &nsample_b, &idx_b, &workspace,
                                        &workspace_bytes, stream);
}

// v3d_ball_query through a cell grid of the database (same arguments, same results; `workspace` = v3d_ball_query_grid_workspace(B, N)
// bytes, 16-byte aligned, contents need not be kept): build + query.  Databases too large for the per-wave LDS bitmaps
// (N > ~250 000) take the scan.
extern "C" int v3d_ball_query_grid(const float* xyz, const float* new_xyz, int B, int N, int M, float radius_a, int nsample_a,
                                   int32_t* idx_a, float radius_b, int nsample_b, int32_t* idx_b, void* workspace,
                                   size_t workspace_bytes, v3d_stream_t stream) {
  if (B < 0 || N < 1 || M < 0 || nsample_a < 1 || (idx_b && nsample_b < 1)) return V3D_EINVAL;
  if (B == 0 || M == 0) return V3D_OK;
  if (!xyz || !new_xyz || !idx_a) return V3D_EINVAL;
  int wpl, waves;
  bqg_bitmap_shape(N, wpl, waves);
  const float rmax = idx_b ? std::max(fabsf(radius_a), fabsf(radius_b)) : fabsf(radius_a);
  if (waves < 1 || !(rmax > 0.f) || !(rmax < 1e30f))
    return v3d_ball_query(xyz, new_xyz, B, N, M, radius_a, nsample_a, idx_a, radius_b, nsample_b, idx_b, stream);
  int rc = v3d_ball_query_grid_build(1, &xyz, &N, &rmax, &workspace, &workspace_bytes, B, stream);
  if (rc) return rc;
  return v3d_ball_query_grid_query(new_xyz, B, N, M, radius_a, nsample_a, idx_a, radius_b, nsample_b, idx_b, workspace, workspace_bytes, stream);
}

// ------------------------------------------------------------------------------------------ bilinear BEV lookup
// F.grid_sample(feature_map (B, C, H, W), grid (B, 1, K, 2) in [-1, 1], bilinear, zeros padding, align_corners=True) as the
// BEVFeatureGatherer calls it (vision3d/detector/layers.py:29-47): out (B, C, K).  torch's generic kernel walks the channels of
// a point serially in one thread (153 us for 2 048 keypoints x 128 channels at 200 x 176); here a thread = (point, channel),
// same tap order and weights (nw, ne, sw, se; weight = product of the distances to the opposite corner).
// one lookup: the four taps of (gx, gy) in plane `plane`, torch's order and weights
__device__ __forceinline__ float bev_tap4(const float* __restrict__ plane, int H, int W, float gx, float gy) {
  const float ix = ((gx + 1.f) / 2.f) * (float)(W - 1), iy = ((gy + 1.f) / 2.f) * (float)(H - 1);
  const float x0f = floorf(ix), y0f = floorf(iy);
  const int x0 = (int)x0f, y0 = (int)y0f, x1 = x0 + 1, y1 = y0 + 1;
  const float nw = ((float)x1 - ix) * ((float)y1 - iy), ne = (ix - x0f) * ((float)y1 - iy);
  const float sw = ((float)x1 - ix) * (iy - y0f), se = (ix - x0f) * (iy - y0f);
  auto tap = [&](int y, int x) { return (y >= 0 && y < H && x >= 0 && x < W) ? plane[(size_t)y * W + x] : 0.f; };
  float v = 0.f;
  v += tap(y0, x0) * nw;
  v += tap(y0, x1) * ne;
  v += tap(y1, x0) * sw;
  v += tap(y1, x1) * se;
  return v;
}

__global__ __launch_bounds__(V3D_BLOCK) void bev_bilinear_kernel(const float* __restrict__ fmap, const float* __restrict__ grid,
                                                                 int C, int H, int W, int K, long long total,
                                                                 float* __restrict__ out) {
  for (long long t = (long long)blockIdx.x * V3D_BLOCK + threadIdx.x; t < total; t += (long long)gridDim.x * V3D_BLOCK) {
    const int k = (int)(t % K);
    const long long bc = t / K;
    const int b = (int)(bc / C);
    const float gx = grid[((size_t)b * K + k) * 2], gy = grid[((size_t)b * K + k) * 2 + 1];
    out[t] = bev_tap4(fmap + (size_t)bc * H * W, H, W, gx, gy);
  }
}

extern "C" int v3d_bev_bilinear(const float* feature_map, const float* grid, int B, int C, int H, int W, int K, float* out,
                                v3d_stream_t stream) {
  if (B < 0 || C < 1 || H < 1 || W < 1 || K < 0) return V3D_EINVAL;
  const long long total = (long long)B * C * K;
  if (total == 0) return V3D_OK;
  if (!feature_map || !grid || !out) return V3D_EINVAL;
  hipLaunchKernelGGL(bev_bilinear_kernel, dim3((int)std::min<long long>(v3d_ceil_div(total, V3D_BLOCK), 8192)), dim3(V3D_BLOCK), 0,
                     (hipStream_t)stream, feature_map, grid, C, H, W, K, total, out);
  V3D_CHECK_LAUNCH();
  return V3D_OK;
}

// BEVFeatureGatherer.forward in ONE launch (vision3d/detector/layers.py:29-47): the grid coordinates are computed from the keypoints
// with the module's own fp32 statements, one IEEE operation each (-ffp-contract=off, correctly rounded division) --
//     frac = (xy - offset) / pixel;  frac = min(max(frac, 0), [W - 1, H - 1]);  g = 2 * (frac / ([W - 1, H - 1] - 1)) - 1;  g = g.flip(-1)
// (the flip and the "- 1" in the divisor are the reference's) -- and the lookup follows.  A thread = (keypoint,
// channel), channel fastest: out is POINT-major, out[(b * K + k) * ldo + c] -- a column block of the keypoint feature rows the
// RoI-grid pooling gathers.  Ten elementwise torch launches + the lookup before.
__global__ __launch_bounds__(V3D_BLOCK) void bev_gather_keypoints_kernel(const float* __restrict__ fmap, const float* __restrict__ xyz,
                                                                         int C, int H, int W, int K, long long total, float off_x,
                                                                         float off_y, float pix_x, float pix_y, float* __restrict__ out,
                                                                         int ldo) {
  const float lim_x = (float)(W - 1), lim_y = (float)(H - 1);
  for (long long t = (long long)blockIdx.x * V3D_BLOCK + threadIdx.x; t < total; t += (long long)gridDim.x * V3D_BLOCK) {
    const int c = (int)(t % C);
    const long long bk = t / C;
    const int b = (int)(bk / K);
    const float x = xyz[bk * 3], y = xyz[bk * 3 + 1];
    float fx = (x - off_x) / pix_x, fy = (y - off_y) / pix_y;
    // torch.clamp(min=0) then torch.min(., limit): both propagate NaN
    fx = fx != fx ? fx : fminf(fmaxf(fx, 0.f), lim_x);
    fy = fy != fy ? fy : fminf(fmaxf(fy, 0.f), lim_y);
    const float nx = 2.f * (fx / (lim_x - 1.f)) - 1.f, ny = 2.f * (fy / (lim_y - 1.f)) - 1.f;
    out[(size_t)bk * ldo + c] = bev_tap4(fmap + ((size_t)b * C + c) * H * W, H, W, /*flipped:*/ ny, nx);
  }
}

extern "C" int v3d_bev_gather_keypoints(const float* feature_map, const float* keypoint_xyz, int B, int C, int H, int W, int K,
                                        float offset_x, float offset_y, float pixel_x, float pixel_y, float* out, int ldo,
                                        v3d_stream_t stream) {
  if (B < 0 || C < 1 || H < 1 || W < 1 || K < 0 || ldo < C) return V3D_EINVAL;
  const long long total = (long long)B * C * K;
  if (total == 0) return V3D_OK;
  if (!feature_map || !keypoint_xyz || !out) return V3D_EINVAL;
  hipLaunchKernelGGL(bev_gather_keypoints_kernel, dim3((int)std::min<long long>(v3d_ceil_div(total, V3D_BLOCK), 8192)), dim3(V3D_BLOCK), 0,
                     (hipStream_t)stream, feature_map, keypoint_xyz, C, H, W, K, total, offset_x, offset_y, pixel_x, pixel_y, out, ldo);
  V3D_CHECK_LAUNCH();
  return V3D_OK;
}

// ------------------------------------------------------------------------------------------ voxel sites -> metric positions
// SparseCNNBase.to_global (vision3d/detector/sparse_cnn.py:91-105): xyz = indices.flip(1)[:, :3].float() * (base_voxel_size * stride) +
// voxel_offset -- an int -> float conversion, one multiply, one add per coordinate (each rounded on its own), five torch launches per
// level before.  indices (n, 4) = (b, z, y, x); `scale` = base_voxel_size * stride as the caller computed it in fp32.
__global__ __launch_bounds__(V3D_BLOCK) void voxel_centers_kernel(const int* __restrict__ indices, int n, float sx, float sy, float sz,
                                                                  float ox, float oy, float oz, float* __restrict__ out) {
  for (int i = blockIdx.x * V3D_BLOCK + threadIdx.x; i < n; i += gridDim.x * V3D_BLOCK) {
    const int4 c = reinterpret_cast<const int4*>(indices)[i];  // (b, z, y, x)
    out[3 * (size_t)i] = (float)c.w * sx + ox;
    out[3 * (size_t)i + 1] = (float)c.z * sy + oy;
    out[3 * (size_t)i + 2] = (float)c.y * sz + oz;
  }
}

extern "C" int v3d_voxel_centers(const int32_t* indices, int n, float scale_x, float scale_y, float scale_z, float offset_x,
                                 float offset_y, float offset_z, float* out, v3d_stream_t stream) {
  if (n < 0) return V3D_EINVAL;
  if (n == 0) return V3D_OK;
  if (!indices || !out || ((uintptr_t)indices & 15)) return V3D_EINVAL;
  hipLaunchKernelGGL(voxel_centers_kernel, dim3(std::min(v3d_ceil_div(n, V3D_BLOCK), 4096)), dim3(V3D_BLOCK), 0, (hipStream_t)stream,
                     indices, n, scale_x, scale_y, scale_z, offset_x, offset_y, offset_z, out);
  V3D_CHECK_LAUNCH();
  return V3D_OK;
}

// ------------------------------------------------------------------------------------------ RoI grid points
// RoiGridPool.sample_gridpoints (vision3d/detector/roi_grid_pool.py:52-62) statement by statement, every operation rounded on its own
// (-ffp-contract=off): local = size * (sample - 0.5); rotated = (cos * lx - sin * ly, sin * lx + cos * ly, lz); point = centre + rotated.
// cos / sin of the yaw come from the caller (torch's functions: the op-by-op path's values).  Nine elementwise launches and a stack before.
__global__ __launch_bounds__(V3D_BLOCK) void roi_grid_points_kernel(const float* __restrict__ boxes, const float* __restrict__ samples,
                                                                    const float* __restrict__ cs, const float* __restrict__ sn,
                                                                    long long total, int m, float* __restrict__ out) {
  for (long long t = (long long)blockIdx.x * V3D_BLOCK + threadIdx.x; t < total; t += (long long)gridDim.x * V3D_BLOCK) {
    const long long bx = t / m;
    const float* bp = boxes + bx * 7;
    const float* sp = samples + t * 3;
    const float lx = bp[3] * (sp[0] - 0.5f), ly = bp[4] * (sp[1] - 0.5f), lz = bp[5] * (sp[2] - 0.5f);
    // cs == NULL: cosf / sinf of the yaw here (the device library's functions: tests/test_gpu_pointops.py checks them against
    // torch.cos / torch.sin bit for bit -- two launches less per frame)
    const float c = cs ? cs[bx] : cosf(bp[6]), s = sn ? sn[bx] : sinf(bp[6]);
    out[t * 3] = bp[0] + (c * lx - s * ly);
    out[t * 3 + 1] = bp[1] + (s * lx + c * ly);
    out[t * 3 + 2] = bp[2] + lz;
  }
}

extern "C" int v3d_roi_grid_points(const float* boxes, const float* samples, const float* cos_yaw, const float* sin_yaw, int n_boxes,
                                   int m, float* out, v3d_stream_t stream) {
  if (n_boxes < 0 || m < 0) return V3D_EINVAL;
  const long long total = (long long)n_boxes * m;
  if (total == 0) return V3D_OK;
  if (!boxes || !samples || !out || ((cos_yaw == nullptr) != (sin_yaw == nullptr))) return V3D_EINVAL;
  hipLaunchKernelGGL(roi_grid_points_kernel, dim3((int)std::min<long long>(v3d_ceil_div(total, V3D_BLOCK), 4096)), dim3(V3D_BLOCK), 0,
                     (hipStream_t)stream, boxes, samples, cos_yaw, sin_yaw, total, m, out);
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
