// sa_mlp.hip -- set-abstraction shared MLP on gathered rows (T5): group -> 1x1 conv + BN + ReLU -> ... -> max over the samples.
//
// Replaces what pointnet2's PointnetSAModuleMSG does per scale around vision3d's call sites (detector/model.py:58-66: the
// five voxel-set-abstraction levels; detector/roi_grid_pool.py:64-72: RoI-grid pooling): grouping_operation materialises
// (B, C+3, M, ns) in NCHW, a SharedMLP of nn.Conv2d(1x1, bias=False) + BatchNorm2d + ReLU runs over it and torch.max reduces the
// sample axis.  On ROCm those 1x1 convolutions fall to MIOpen's naive_conv kernel: 89 % of the PV-RCNN stage-2 time in round 1
// (profiles/r01_g_pvrcnn_stage2_kernel_stats.csv: 160 calls, 6.5 ms average).
//
// Here one launch per MLP layer computes, for rows = (b, m, s) in that order,
//     out[row, :] = act( [xyz[i] - new_xyz[m], 0 | feat[i, :]] @ W + bias ),   i = idx[b, m, s]      (first layer)
//     out[row, :] = act( in[row, :] @ W + bias )                                                      (later layers)
// with an optional max over the ns rows of every (b, m) group in the epilogue (last layer), so the grouped tensor never
// exists and the first layer reads POINT-major features (B, N, Kf) -- one contiguous row per gathered sample instead of Kf
// strided words.  BatchNorm (eval) is folded into W / bias by the caller (vision3d_amd/pointnet2/pointnet2_modules.py).
//
// Arithmetic: exact fp32 on the matrix cores, v_mfma_f32_16x16x4_f32 (bitwise an fmaf chain per output; 157 TFLOP/s peak).
// The largest instance (RoI grid: 51 200 rows x 516 -> 192) is 10 GFLOP -- ~70 us at the fp32 MFMA peak -- so the 5x of a
// bf16 split is not worth its error budget here.
// Tiling: a wave owns two 16-row tiles and all Nout columns (accumulators in registers); a workgroup of 8 waves (256 rows)
// streams W through LDS in chunks of 64 k-rows shared by its 16 tiles.  The MFMA reduction index is permuted so that lane
// (r, q) feeds four CONTIGUOUS floats of row r per 16-wide k block (one 16-byte load; A and B agree on the permutation).
#include <algorithm>

#include "v3d_common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define SAM_WAVES 8
#define SAM_TPW 2     // 16-row tiles per wave
#define SAM_KC 64     // k-rows of W per LDS chunk

// PAIR (the first TWO layers of a scale in one launch, v3d_sa_mlp_pair): the first layer is linear in its input row
// [xyz[i] - new_xyz[m], 0 | feat[i]], and its feature part depends on the gathered point i alone: P = feat @ W1[4:] is computed ONCE per
// database point by the caller (N rows instead of M * ns: 2 048 instead of 76 800 at RoI-grid pooling -- 37x less matrix work), and
// this kernel rebuilds the first layer's output row where the second layer consumes it,
//     h[row, k] = relu( P[i, k] + ((rel.x * Wx[0, k] + rel.y * Wx[1, k]) + rel.z * Wx[2, k]) + b1[k] ),    i = idx[row]
// as the A fragments of  out[row, :] = act(h[row, :] @ W + bias)  (+ max over the samples): the (rows, K) intermediate is never
// written.  `feat` = P (B, N, Kf), Kf = the first layer's padded width, `wx` (3, Kf) = W1[0:3], `b1` (Kf).
// A SECOND scale in the same launch (gridDim.y == 2; v3d_sa_mlp_pair2): the scales of a multi-scale module have the same widths and
// differ in their neighbour lists, sample counts and weights -- one launch fills the chip where two left it half empty (RoI-grid
// pooling: 100 + 200 workgroups on 256 compute units) and the chain is one dependent launch shorter per module.
struct SamScale {
  const float* feat;  // PAIR: this scale's column block of P
  const int* idx;
  const float* W;
  const float* bias;
  const float* wx;
  const float* b1;
  float* out;
  long long rows;
  int ns, n_store;
};

// WAVES = 8 (256 rows per workgroup) or 4 (128): a launch whose 256-row workgroups number between one and ~2.5 per compute unit runs
// as long as its unluckiest unit's TWO workgroups (RoI-grid pooling: 300 on 256 units); half-size workgroups spread evenly (sam_waves).
template <int NB, bool PAIR = false, int WAVES = SAM_WAVES>
__global__ __launch_bounds__(WAVES * 64) void sa_mlp_layer_kernel(const float* feat_0, const float* __restrict__ xyz,
                                                                     const float* __restrict__ new_xyz, const int* idx_0, int N, int M,
                                                                     int ns_0, int Kf, long long rows_0, const float* W_0,
                                                                     const float* bias_0, int relu, int pool, float* out_0, int ldo,
                                                                     int n_store_0, const float* wx_0, const float* b1_0, int ldp,
                                                                     const SamScale alt) {
  const bool second = blockIdx.y == 1;
  const float* __restrict__ feat = second ? alt.feat : feat_0;
  const int* __restrict__ idx = second ? alt.idx : idx_0;
  const float* __restrict__ W = second ? alt.W : W_0;
  const float* __restrict__ bias = second ? alt.bias : bias_0;
  const float* __restrict__ wx = second ? alt.wx : wx_0;
  const float* __restrict__ b1 = second ? alt.b1 : b1_0;
  float* __restrict__ out = second ? alt.out : out_0;
  const long long rows = second ? alt.rows : rows_0;
  const int ns = second ? alt.ns : ns_0, n_store = second ? alt.n_store : n_store_0;
  if ((long long)blockIdx.x * (16 * SAM_TPW * WAVES) >= rows) return;  // (the grid is sized for the scale with more rows)
  constexpr int NOUT = NB * 16;
  constexpr int LDW = NOUT + 4;  // row stride of the LDS weight chunk: the four k-rows a wave reads at once hit disjoint banks
  __shared__ float Ws[SAM_KC * LDW];
  __shared__ __attribute__((aligned(16))) float Wx[PAIR ? 4 * 256 : 4];  // PAIR: rows x, y, z of the first layer's weight, its bias
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, q = lane >> 4;
  const bool has_xyz = !PAIR && xyz != nullptr;
  const int Kp = Kf + (has_xyz ? 4 : 0);
  if constexpr (PAIR) {
    for (int e = tid; e < 4 * Kf; e += WAVES * 64) Wx[e] = e < 3 * Kf ? wx[e] : b1[e - 3 * Kf];
    __syncthreads();
  }
  const long long tile0 = ((long long)blockIdx.x * WAVES + wave) * SAM_TPW;

  // this lane's row in each of the wave's tiles: source row pointer and the xyz offset of the first layer
  const float* frow[SAM_TPW];
  f32x4 rel[SAM_TPW];
#pragma unroll
  for (int t = 0; t < SAM_TPW; t++) {
    const long long row = (tile0 + t) * 16 + r;
    frow[t] = nullptr;
    rel[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (row < rows) {
      const long long bm = row / ns;
      const int b = (int)(bm / M);
      const long long src = idx ? (long long)b * N + idx[row] : row;
      frow[t] = feat + src * (PAIR ? ldp : Kf);
      if (has_xyz || PAIR) {
        const float* p = xyz + src * 3;
        const float* c = new_xyz + bm * 3;
        rel[t] = f32x4{p[0] - c[0], p[1] - c[1], p[2] - c[2], 0.f};
      }
    }
  }
  f32x4 acc[SAM_TPW][NB];
#pragma unroll
  for (int t = 0; t < SAM_TPW; t++)
#pragma unroll
    for (int j = 0; j < NB; j++) acc[t][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // A: four contiguous floats of this lane's row, k = kb + 4 q .. + 3 (kb a multiple of 16).  The fragments of block kb + 16 are
  // requested BEFORE the MFMAs of block kb (a gathered row is a trip to the L2); measured on the direct first layer of RoI-grid
  // pooling (51 200 rows x 516 -> 192): 134 -> 129 us with the B operands read a step ahead as well -- that layer's real fix was not
  // to compute it per grouped row at all (PAIR, docs/rounds/round6.md G.2).
  auto load_a = [&](int kb, f32x4 (&a)[SAM_TPW]) {
    const int kg = kb + 4 * q;
#pragma unroll
    for (int t = 0; t < SAM_TPW; t++) {
      a[t] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (frow[t]) {
        if constexpr (PAIR) {
          if (kg < Kf) {
            const f32x4 p4 = *reinterpret_cast<const f32x4*>(frow[t] + kg);
            const f32x4 w0 = *reinterpret_cast<const f32x4*>(Wx + kg), w1 = *reinterpret_cast<const f32x4*>(Wx + Kf + kg);
            const f32x4 w2 = *reinterpret_cast<const f32x4*>(Wx + 2 * Kf + kg), bb = *reinterpret_cast<const f32x4*>(Wx + 3 * Kf + kg);
#pragma unroll
            for (int c = 0; c < 4; c++) a[t][c] = fmaxf(p4[c] + ((rel[t].x * w0[c] + rel[t].y * w1[c]) + rel[t].z * w2[c]) + bb[c], 0.f);
          }
        } else if (has_xyz && kg == 0) a[t] = rel[t];
        else {
          const int fo = kg - (has_xyz ? 4 : 0);
          if (fo < Kf) a[t] = *reinterpret_cast<const f32x4*>(frow[t] + fo);
        }
      }
    }
  };
  f32x4 a[SAM_TPW], an[SAM_TPW];
  load_a(0, a);
  for (int k0 = 0; k0 < Kp; k0 += SAM_KC) {
    const int kc = min(SAM_KC, Kp - k0);
    __syncthreads();  // the previous chunk has been consumed
    for (int e = tid; e < SAM_KC * (NOUT / 4); e += WAVES * 64) {
      const int kr = e / (NOUT / 4), c4 = e % (NOUT / 4);
      f32x4 w = f32x4{0.f, 0.f, 0.f, 0.f};
      if (kr < kc) w = *reinterpret_cast<const f32x4*>(W + (size_t)(k0 + kr) * NOUT + c4 * 4);
      *reinterpret_cast<f32x4*>(Ws + kr * LDW + c4 * 4) = w;
    }
    __syncthreads();
    for (int kb = 0; kb < kc; kb += 16) {  // kc is a multiple of 4; rows beyond it are zero in LDS
      if (k0 + kb + 16 < Kp) load_a(k0 + kb + 16, an);
      // B[k = 4 q + s][n = 16 j + r]: the NB words of step s + 1 are read while step s multiplies (read, wait, four MFMAs, read ...
      // left the matrix pipe idle for an LDS round trip per four instructions: one wave per SIMD ran at half rate)
      const float* wq = Ws + (kb + 4 * q) * LDW + r;
      float bc[NB], bn[NB];
#pragma unroll
      for (int j = 0; j < NB; j++) bc[j] = wq[j * 16];
#pragma unroll
      for (int s = 0; s < 4; s++) {
        if (s < 3) {
#pragma unroll
          for (int j = 0; j < NB; j++) bn[j] = wq[(s + 1) * LDW + j * 16];
        }
        __builtin_amdgcn_sched_barrier(0);  // (left alone the scheduler sinks every read next to its use again)
#pragma unroll
        for (int j = 0; j < NB; j++)
#pragma unroll
          for (int t = 0; t < SAM_TPW; t++) acc[t][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t][s], bc[j], acc[t][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < NB; j++) bc[j] = bn[j];
      }
#pragma unroll
      for (int t = 0; t < SAM_TPW; t++) a[t] = an[t];
    }
  }

  // epilogue: D[row = 4 q + i][col = 16 j + r]
  float pooled[NB];
#pragma unroll
  for (int j = 0; j < NB; j++) pooled[j] = -3.402823466e38f;
#pragma unroll
  for (int t = 0; t < SAM_TPW; t++) {
    const long long rowb = (tile0 + t) * 16 + 4 * q;
#pragma unroll
    for (int j = 0; j < NB; j++) {
      const float bj = bias ? bias[j * 16 + r] : 0.f;
      float mx = -3.402823466e38f;
#pragma unroll
      for (int i = 0; i < 4; i++) {
        float v = acc[t][j][i] + bj;
        if (relu) v = fmaxf(v, 0.f);
        if (pool) mx = fmaxf(mx, v);
        else if (rowb + i < rows && j * 16 + r < n_store) out[(size_t)(rowb + i) * ldo + j * 16 + r] = v;
      }
      if (pool) {
        mx = fmaxf(mx, __shfl_xor(mx, 16));
        mx = fmaxf(mx, __shfl_xor(mx, 32));  // max over the tile's 16 rows, in every lane
        if (ns == 16) {
          const long long g = tile0 + t;  // one tile = one (b, m) group
          if (q == 0 && g * 16 < rows && j * 16 + r < n_store) out[(size_t)g * ldo + j * 16 + r] = mx;
        } else {
          pooled[j] = fmaxf(pooled[j], mx);  // ns == 32: the wave's two tiles are one group
        }
      }
    }
  }
  if (pool && ns == 32 && q == 0 && tile0 * 16 < rows) {
    const long long g = tile0 / 2;
#pragma unroll
    for (int j = 0; j < NB; j++)
      if (j * 16 + r < n_store) out[(size_t)g * ldo + j * 16 + r] = pooled[j];
  }
}

// waves per workgroup of a launch over `rows` rows (x `scales` grids): 4 when the 8-wave workgroups would number between one and 2.5
// per compute unit
static int sam_waves(long long rows, int scales) {
  const long long wgs = (rows + 16 * SAM_TPW * SAM_WAVES - 1) / (16 * SAM_TPW * SAM_WAVES) * scales;
  const int cus = std::max(v3d_device_cu_count(), 1);
  return (wgs > cus && wgs * 2 < (long long)cus * 5) ? 4 : SAM_WAVES;
}

extern "C" int v3d_sa_mlp_layer(const float* feat, const float* xyz, const float* new_xyz, const int32_t* idx, int B, int N, int M,
                                int ns, int Kf, const float* W, const float* bias, int Nout, int relu, int pool, float* out,
                                int ldo, int n_store, v3d_stream_t stream) {
  if (B < 0 || N < 1 || M < 0 || ns < 1 || Kf < 0 || (Kf & 3) || Nout < 16 || (Nout & 15) || !W || !out) return V3D_EINVAL;
  if ((xyz == nullptr) != (new_xyz == nullptr)) return V3D_EINVAL;
  if (!feat && Kf > 0) return V3D_EINVAL;
  if (pool && ns != 16 && ns != 32) return V3D_EUNSUPPORTED;
  if (n_store <= 0 || n_store > Nout) n_store = Nout;  // columns [0, n_store) are stored (the padded tail of Nout is not)
  if (ldo <= 0) ldo = Nout;                            // row stride of `out` in floats (a column block of a wider matrix)
  if (ldo < n_store) return V3D_EINVAL;
  const long long rows = (long long)B * M * ns;
  if (rows == 0) return V3D_OK;
  if (!idx && (long long)B * N != rows) return V3D_EINVAL;  // identity rows: the input IS the (B*M*ns, Kf) matrix
  if (Kf + (xyz ? 4 : 0) < 4) return V3D_EINVAL;
  const int waves = sam_waves(rows, 1);
  const int blocks = v3d_ceil_div(rows, 16 * SAM_TPW * waves);
  hipStream_t st = (hipStream_t)stream;
#define SAM_ARGS feat, xyz, new_xyz, idx, N, M, ns, Kf, rows, W, bias, relu, pool, out, ldo, n_store, nullptr, nullptr, 0, SamScale{}
#define SAM_CASE(NBV)                                                                                                       \
  if (Nout == NBV * 16) {                                                                                                   \
    if (waves == 4) hipLaunchKernelGGL((sa_mlp_layer_kernel<NBV, false, 4>), dim3(blocks), dim3(4 * 64), 0, st, SAM_ARGS);  \
    else hipLaunchKernelGGL((sa_mlp_layer_kernel<NBV, false, SAM_WAVES>), dim3(blocks), dim3(SAM_WAVES * 64), 0, st, SAM_ARGS); \
    V3D_CHECK_LAUNCH();                                                                                                     \
    return V3D_OK;                                                                                                          \
  }
  SAM_CASE(1) SAM_CASE(2) SAM_CASE(4) SAM_CASE(6) SAM_CASE(8) SAM_CASE(12) SAM_CASE(16)
#undef SAM_CASE
#undef SAM_ARGS
  return V3D_EUNSUPPORTED;
}

// The first two layers of a scale in one launch (see the PAIR note at the kernel): P (B, N, K1) = feat @ W1[4:] from the caller
// (v3d_linear_rows on the database's feature rows), wx (3, K1) = W1[0:3], b1 (K1), then layer 2: W (K1, Nout), bias, relu, pool as
// in v3d_sa_mlp_layer.  K1 % 4 == 0, K1 <= 256; ldp = row stride of P (0: K1).
extern "C" int v3d_sa_mlp_pair(const float* P, const float* xyz, const float* new_xyz, const int32_t* idx, int B, int N, int M, int ns,
                               int K1, int ldp, const float* wx, const float* b1, const float* W, const float* bias, int Nout, int relu,
                               int pool, float* out, int ldo, int n_store, v3d_stream_t stream) {
  if (B < 0 || N < 1 || M < 0 || ns < 1 || K1 < 4 || (K1 & 3) || K1 > 256 || Nout < 16 || (Nout & 15)) return V3D_EINVAL;
  if (ldp <= 0) ldp = K1;  // row stride of P in floats (a column block of a wider matrix: both scales of a module from one product)
  if (ldp < K1 || (ldp & 3)) return V3D_EINVAL;
  if (!P || !xyz || !new_xyz || !idx || !wx || !b1 || !W || !out || ((uintptr_t)P & 15)) return V3D_EINVAL;
  if (pool && ns != 16 && ns != 32) return V3D_EUNSUPPORTED;
  if (n_store <= 0 || n_store > Nout) n_store = Nout;
  if (ldo <= 0) ldo = Nout;
  if (ldo < n_store) return V3D_EINVAL;
  const long long rows = (long long)B * M * ns;
  if (rows == 0) return V3D_OK;
  const int waves = sam_waves(rows, 1);
  const int blocks = v3d_ceil_div(rows, 16 * SAM_TPW * waves);
  hipStream_t st = (hipStream_t)stream;
#define SAM_ARGS P, xyz, new_xyz, idx, N, M, ns, K1, rows, W, bias, relu, pool, out, ldo, n_store, wx, b1, ldp, SamScale{}
#define SAM_CASE(NBV)                                                                                                      \
  if (Nout == NBV * 16) {                                                                                                  \
    if (waves == 4) hipLaunchKernelGGL((sa_mlp_layer_kernel<NBV, true, 4>), dim3(blocks), dim3(4 * 64), 0, st, SAM_ARGS);  \
    else hipLaunchKernelGGL((sa_mlp_layer_kernel<NBV, true, SAM_WAVES>), dim3(blocks), dim3(SAM_WAVES * 64), 0, st, SAM_ARGS); \
    V3D_CHECK_LAUNCH();                                                                                                    \
    return V3D_OK;                                                                                                         \
  }
  SAM_CASE(1) SAM_CASE(2) SAM_CASE(4) SAM_CASE(6) SAM_CASE(8) SAM_CASE(12) SAM_CASE(16)
#undef SAM_CASE
#undef SAM_ARGS
  return V3D_EUNSUPPORTED;
}

// v3d_sa_mlp_pair for the TWO scales of a module in one launch: P_a / P_b = the scales' column blocks of one (B, N, ldp) product,
// the same K1 and Nout (multi-scale modules repeat their widths), per scale: neighbour list + sample count, wx / b1, W / bias, and
// the column block of `out` (row stride ldo) it writes.
extern "C" int v3d_sa_mlp_pair2(const float* P_a, const float* P_b, const float* xyz, const float* new_xyz, const int32_t* idx_a,
                                const int32_t* idx_b, int B, int N, int M, int ns_a, int ns_b, int K1, int ldp, const float* wx_a,
                                const float* b1_a, const float* wx_b, const float* b1_b, const float* W_a, const float* bias_a,
                                const float* W_b, const float* bias_b, int Nout, int relu, int pool, float* out_a, float* out_b, int ldo,
                                int n_store, v3d_stream_t stream) {
  if (B < 0 || N < 1 || M < 0 || ns_a < 1 || ns_b < 1 || K1 < 4 || (K1 & 3) || K1 > 256 || Nout < 16 || (Nout & 15)) return V3D_EINVAL;
  if (ldp <= 0) ldp = K1;
  if (ldp < K1 || (ldp & 3)) return V3D_EINVAL;
  if (!P_a || !P_b || !xyz || !new_xyz || !idx_a || !idx_b || !wx_a || !b1_a || !wx_b || !b1_b || !W_a || !W_b || !out_a || !out_b) return V3D_EINVAL;
  if (((uintptr_t)P_a & 15) || ((uintptr_t)P_b & 15)) return V3D_EINVAL;
  if (pool && ((ns_a != 16 && ns_a != 32) || (ns_b != 16 && ns_b != 32))) return V3D_EUNSUPPORTED;
  if (n_store <= 0 || n_store > Nout) n_store = Nout;
  if (ldo <= 0) ldo = Nout;
  if (ldo < n_store) return V3D_EINVAL;
  const long long rows_a = (long long)B * M * ns_a, rows_b = (long long)B * M * ns_b;
  if (rows_a == 0) return V3D_OK;
  const int waves = sam_waves((rows_a + rows_b) / 2, 2);
  const int blocks = v3d_ceil_div(std::max(rows_a, rows_b), 16 * SAM_TPW * waves);
  const SamScale alt{P_b, idx_b, W_b, bias_b, wx_b, b1_b, out_b, rows_b, ns_b, n_store};
  hipStream_t st = (hipStream_t)stream;
#define SAM_ARGS P_a, xyz, new_xyz, idx_a, N, M, ns_a, K1, rows_a, W_a, bias_a, relu, pool, out_a, ldo, n_store, wx_a, b1_a, ldp, alt
#define SAM_CASE(NBV)                                                                                                          \
  if (Nout == NBV * 16) {                                                                                                      \
    if (waves == 4) hipLaunchKernelGGL((sa_mlp_layer_kernel<NBV, true, 4>), dim3(blocks, 2), dim3(4 * 64), 0, st, SAM_ARGS);   \
    else hipLaunchKernelGGL((sa_mlp_layer_kernel<NBV, true, SAM_WAVES>), dim3(blocks, 2), dim3(SAM_WAVES * 64), 0, st, SAM_ARGS); \
    V3D_CHECK_LAUNCH();                                                                                                        \
    return V3D_OK;                                                                                                             \
  }
  SAM_CASE(1) SAM_CASE(2) SAM_CASE(4) SAM_CASE(6) SAM_CASE(8) SAM_CASE(12) SAM_CASE(16)
#undef SAM_CASE
#undef SAM_ARGS
  return V3D_EUNSUPPORTED;
}

// ------------------------------------------------------------------------------------------ few rows x wide K: the MLP tail
// PV-RCNN's last matrices have a hundred rows (one per proposal): RoI-grid reduction 3 072 -> 256 -> 256 (detector/roi_grid_pool.py:
// 64-72), refinement 256 -> 128 -> 8 (detector/refinement.py:47-50) -- nn.Linear upstream, i.e. four library GEMM launches (72 us per
// frame in round 5's trace: 157 MFLOP).  sa_mlp_layer_kernel splits ROWS over workgroups (one workgroup here); this kernel splits
// the 16-column blocks over workgroups and K over the 8 waves of a workgroup (contiguous K ranges, partial tiles summed in wave
// order through LDS: a fixed summation order), exact fp32 on v_mfma_f32_16x16x4_f32 like the layers above.  Next A / W fragments are
// requested before the current ones are multiplied.
//   out[r, n] = act( sum_k A[r * lda + k] * W[k * Nout + n] + bias[n] ),  r < R, n < n_store;  K % 4 == 0, Nout % 16 == 0
#define LIN_WAVES 8
#define LIN_TPW 2  // 16-row tiles per workgroup (every wave multiplies both for its K range)
#define LIN_MAX_JOBS 8
struct LinJob {
  const float* A;
  const float* W;
  const float* bias;
  float* out;
  int lda, R, K, Nout, relu, ldo, n_store;
  int ksplit;  // 1: the 8 waves share 32 rows and split K (few rows x wide K); 0: a wave per 32 rows, all of K (many rows x narrow K)
};
struct LinJobs {
  LinJob j[LIN_MAX_JOBS];
};

// grid: (column blocks, row blocks, job), sized for the largest job (a workgroup beyond its job's extent leaves): the first-layer
// products of the five set-abstraction modules of a PV-RCNN frame are one launch
__global__ __launch_bounds__(LIN_WAVES * 64) void linear_rows_kernel(const LinJobs jobs) {
  const LinJob jb = jobs.j[blockIdx.z];
  const float* __restrict__ A = jb.A;
  const float* __restrict__ W = jb.W;
  const float* __restrict__ bias = jb.bias;
  float* __restrict__ out = jb.out;
  const int lda = jb.lda, R = jb.R, K = jb.K, Nout = jb.Nout, relu = jb.relu, ldo = jb.ldo, n_store = jb.n_store;
  const bool split = jb.ksplit != 0;
  const int wg_rows = split ? 16 * LIN_TPW : 16 * LIN_TPW * LIN_WAVES;
  if ((int)blockIdx.x * 16 >= Nout || (int)blockIdx.y * wg_rows >= R) return;
  __shared__ float part[LIN_WAVES][LIN_TPW][4][64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int col0 = blockIdx.x * 16, row0 = blockIdx.y * wg_rows + (split ? 0 : wave * (16 * LIN_TPW));
  const int kper = ((K + 16 * LIN_WAVES - 1) / (16 * LIN_WAVES)) * 16;  // K range of a wave: a multiple of 16
  const int kb = split ? wave * kper : 0, ke = split ? min(K, kb + kper) : K;
  const float* arow[LIN_TPW];
#pragma unroll
  for (int t = 0; t < LIN_TPW; t++) arow[t] = row0 + t * 16 + r < R ? A + (size_t)(row0 + t * 16 + r) * lda : nullptr;
  const float* wcol = W + col0 + r;
  auto load = [&](int k0, f32x4 (&a)[LIN_TPW], float (&bv)[4]) {
    const int kg = k0 + 4 * q;
#pragma unroll
    for (int t = 0; t < LIN_TPW; t++) a[t] = (arow[t] && kg < ke) ? *reinterpret_cast<const f32x4*>(arow[t] + kg) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 4; s++) bv[s] = kg + s < ke ? wcol[(size_t)(kg + s) * Nout] : 0.f;
  };
  f32x4 acc[LIN_TPW];
#pragma unroll
  for (int t = 0; t < LIN_TPW; t++) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 a0[LIN_TPW], a1[LIN_TPW];
  float b0[4], b1[4];
  if (kb < ke) load(kb, a0, b0);
  for (int k0 = kb; k0 < ke; k0 += 32) {
    if (k0 + 16 < ke) load(k0 + 16, a1, b1);
#pragma unroll
    for (int s = 0; s < 4; s++)
#pragma unroll
      for (int t = 0; t < LIN_TPW; t++) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[t][s], b0[s], acc[t], 0, 0, 0);
    if (k0 + 16 < ke) {
      if (k0 + 32 < ke) load(k0 + 32, a0, b0);
#pragma unroll
      for (int s = 0; s < 4; s++)
#pragma unroll
        for (int t = 0; t < LIN_TPW; t++) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[t][s], b1[s], acc[t], 0, 0, 0);
    }
  }
  if (!split) {  // (workgroup-uniform) a wave's own tiles: D[row = 4 q + i][col = r]
    const int col = col0 + r;
    const float bj = bias ? bias[col] : 0.f;
#pragma unroll
    for (int t = 0; t < LIN_TPW; t++)
#pragma unroll
      for (int i = 0; i < 4; i++) {
        float v = acc[t][i] + bj;
        if (relu) v = fmaxf(v, 0.f);
        const int row = row0 + t * 16 + 4 * q + i;
        if (row < R && col < n_store) out[(size_t)row * ldo + col] = v;
      }
    return;
  }
#pragma unroll
  for (int t = 0; t < LIN_TPW; t++)
#pragma unroll
    for (int i = 0; i < 4; i++) part[wave][t][i][lane] = acc[t][i];
  __syncthreads();
  // D[row = 4 q + i][col = r]: the 2 x 4 x 64 values of the workgroup, one per thread
  {
    const int t = tid >> 8, i = (tid >> 6) & 3, l = tid & 63;
    float v = part[0][t][i][l];
#pragma unroll
    for (int w = 1; w < LIN_WAVES; w++) v += part[w][t][i][l];
    const int row = row0 + t * 16 + 4 * (l >> 4) + i, col = col0 + (l & 15);
    if (bias) v += bias[col];
    if (relu) v = fmaxf(v, 0.f);
    if (row < R && col < n_store) out[(size_t)row * ldo + col] = v;
  }
}

extern "C" int v3d_linear_rows_many(int n_jobs, const float* const* A, const int32_t* lda, const int32_t* R, const int32_t* K,
                                    const float* const* W, const float* const* bias, const int32_t* Nout, const int32_t* relu,
                                    float* const* out, const int32_t* ldo, const int32_t* n_store, v3d_stream_t stream) {
  if (n_jobs < 0 || n_jobs > LIN_MAX_JOBS) return V3D_EINVAL;
  if (n_jobs == 0) return V3D_OK;
  if (!A || !lda || !R || !K || !W || !Nout || !out) return V3D_EINVAL;
  LinJobs jobs;
  int gx = 0, gy = 0, n = 0;
  for (int i = 0; i < n_jobs; i++) {
    if (R[i] < 0 || K[i] < 4 || (K[i] & 3) || lda[i] < K[i] || (lda[i] & 3) || Nout[i] < 16 || (Nout[i] & 15)) return V3D_EINVAL;
    if (R[i] == 0) continue;
    if (!A[i] || !W[i] || !out[i] || ((uintptr_t)A[i] & 15)) return V3D_EINVAL;
    LinJob& j = jobs.j[n++];
    j = LinJob{A[i], W[i], bias ? bias[i] : nullptr, out[i], lda[i], R[i], K[i], Nout[i], relu ? relu[i] : 0, ldo ? ldo[i] : 0,
               n_store ? n_store[i] : 0, 1};
    if (j.n_store <= 0 || j.n_store > j.Nout) j.n_store = j.Nout;
    if (j.ldo <= 0) j.ldo = j.Nout;
    if (j.ldo < j.n_store) return V3D_EINVAL;
    // many rows x narrow K (the first-layer products of the set-abstraction modules: 4 600-17 800 rows, K = 4-64): K split over the
    // waves left seven of eight idle and cost eight times the workgroups
    j.ksplit = (j.K >= 256 || j.R <= 16 * LIN_TPW * 4) ? 1 : 0;
    gx = std::max(gx, j.Nout / 16), gy = std::max(gy, v3d_ceil_div(j.R, j.ksplit ? 16 * LIN_TPW : 16 * LIN_TPW * LIN_WAVES));
  }
  if (n == 0) return V3D_OK;
  static_assert(LIN_WAVES * 64 == LIN_TPW * 4 * 64, "one output value per thread in the epilogue");
  hipLaunchKernelGGL(linear_rows_kernel, dim3(gx, gy, n), dim3(LIN_WAVES * 64), 0, (hipStream_t)stream, jobs);
  V3D_CHECK_LAUNCH();
  return V3D_OK;
}

extern "C" int v3d_linear_rows(const float* A, int lda, int R, int K, const float* W, const float* bias, int Nout, int relu, float* out,
                               int ldo, int n_store, v3d_stream_t stream) {
  return v3d_linear_rows_many(1, &A, &lda, &R, &K, &W, &bias, &Nout, &relu, &out, &ldo, &n_store, stream);
}
