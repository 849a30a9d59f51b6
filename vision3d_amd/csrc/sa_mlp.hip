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
#include "v3d_common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define SAM_WAVES 8
#define SAM_TPW 2     // 16-row tiles per wave
#define SAM_KC 64     // k-rows of W per LDS chunk

template <int NB>
__global__ __launch_bounds__(SAM_WAVES * 64) void sa_mlp_layer_kernel(const float* __restrict__ feat,
                                                                     const float* __restrict__ xyz,
                                                                     const float* __restrict__ new_xyz,
                                                                     const int* __restrict__ idx, int N, int M, int ns,
                                                                     int Kf, long long rows, const float* __restrict__ W,
                                                                     const float* __restrict__ bias, int relu, int pool,
                                                                     float* __restrict__ out) {
  constexpr int NOUT = NB * 16;
  constexpr int LDW = NOUT + 4;  // row stride of the LDS weight chunk: the four k-rows a wave reads at once hit disjoint banks
  __shared__ float Ws[SAM_KC * LDW];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, q = lane >> 4;
  const bool has_xyz = xyz != nullptr;
  const int Kp = Kf + (has_xyz ? 4 : 0);
  const long long tile0 = ((long long)blockIdx.x * SAM_WAVES + wave) * SAM_TPW;

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
      frow[t] = feat + src * Kf;
      if (has_xyz) {
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

  for (int k0 = 0; k0 < Kp; k0 += SAM_KC) {
    const int kc = min(SAM_KC, Kp - k0);
    __syncthreads();  // the previous chunk has been consumed
    for (int e = tid; e < SAM_KC * (NOUT / 4); e += SAM_WAVES * 64) {
      const int kr = e / (NOUT / 4), c4 = e % (NOUT / 4);
      f32x4 w = f32x4{0.f, 0.f, 0.f, 0.f};
      if (kr < kc) w = *reinterpret_cast<const f32x4*>(W + (size_t)(k0 + kr) * NOUT + c4 * 4);
      *reinterpret_cast<f32x4*>(Ws + kr * LDW + c4 * 4) = w;
    }
    __syncthreads();
    for (int kb = 0; kb < kc; kb += 16) {  // kc is a multiple of 4; rows beyond it are zero in LDS
      // A: four contiguous floats of this lane's row, k = k0 + kb + 4 q .. + 3
      f32x4 a[SAM_TPW];
      const int kg = k0 + kb + 4 * q;
#pragma unroll
      for (int t = 0; t < SAM_TPW; t++) {
        a[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (frow[t]) {
          if (has_xyz && kg == 0) a[t] = rel[t];
          else {
            const int fo = kg - (has_xyz ? 4 : 0);
            if (fo < Kf) a[t] = *reinterpret_cast<const f32x4*>(frow[t] + fo);
          }
        }
      }
      const float* wq = Ws + (kb + 4 * q) * LDW + r;
#pragma unroll
      for (int s = 0; s < 4; s++) {
#pragma unroll
        for (int j = 0; j < NB; j++) {
          const float b = wq[s * LDW + j * 16];  // B[k = 4 q + s][n = 16 j + r]
#pragma unroll
          for (int t = 0; t < SAM_TPW; t++) acc[t][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t][s], b, acc[t][j], 0, 0, 0);
        }
      }
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
        else if (rowb + i < rows) out[(size_t)(rowb + i) * NOUT + j * 16 + r] = v;
      }
      if (pool) {
        mx = fmaxf(mx, __shfl_xor(mx, 16));
        mx = fmaxf(mx, __shfl_xor(mx, 32));  // max over the tile's 16 rows, in every lane
        if (ns == 16) {
          const long long g = tile0 + t;  // one tile = one (b, m) group
          if (q == 0 && g * 16 < rows) out[(size_t)g * NOUT + j * 16 + r] = mx;
        } else {
          pooled[j] = fmaxf(pooled[j], mx);  // ns == 32: the wave's two tiles are one group
        }
      }
    }
  }
  if (pool && ns == 32 && q == 0 && tile0 * 16 < rows) {
    const long long g = tile0 / 2;
#pragma unroll
    for (int j = 0; j < NB; j++) out[(size_t)g * NOUT + j * 16 + r] = pooled[j];
  }
}

extern "C" int v3d_sa_mlp_layer(const float* feat, const float* xyz, const float* new_xyz, const int32_t* idx, int B, int N, int M,
                                int ns, int Kf, const float* W, const float* bias, int Nout, int relu, int pool, float* out,
                                v3d_stream_t stream) {
  if (B < 0 || N < 1 || M < 0 || ns < 1 || Kf < 0 || (Kf & 3) || Nout < 16 || (Nout & 15) || !W || !out) return V3D_EINVAL;
  if ((xyz == nullptr) != (new_xyz == nullptr)) return V3D_EINVAL;
  if (!feat && Kf > 0) return V3D_EINVAL;
  if (pool && ns != 16 && ns != 32) return V3D_EUNSUPPORTED;
  const long long rows = (long long)B * M * ns;
  if (rows == 0) return V3D_OK;
  if (!idx && (long long)B * N != rows) return V3D_EINVAL;  // identity rows: the input IS the (B*M*ns, Kf) matrix
  if (Kf + (xyz ? 4 : 0) < 4) return V3D_EINVAL;
  const int blocks = v3d_ceil_div(rows, 16 * SAM_TPW * SAM_WAVES);
  hipStream_t st = (hipStream_t)stream;
#define SAM_CASE(NBV)                                                                                                      \
  if (Nout == NBV * 16) {                                                                                                  \
    hipLaunchKernelGGL(sa_mlp_layer_kernel<NBV>, dim3(blocks), dim3(SAM_WAVES * 64), 0, st, feat, xyz, new_xyz, idx, N, M, \
                       ns, Kf, rows, W, bias, relu, pool, out);                                                            \
    V3D_CHECK_LAUNCH();                                                                                                    \
    return V3D_OK;                                                                                                         \
  }
  SAM_CASE(1) SAM_CASE(2) SAM_CASE(4) SAM_CASE(6) SAM_CASE(8) SAM_CASE(12) SAM_CASE(16)
#undef SAM_CASE
  return V3D_EUNSUPPORTED;
}
