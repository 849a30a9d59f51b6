// proposal_loss.hip -- the proposal loss of a SECOND train step and its gradient in one pass over the head maps.
//
// Reference: ProposalLoss (vision3d/detector/proposal.py:100-141): sigmoid focal loss on the class logits (ops/focal_loss.py:
// alpha 0.25, gamma 2) summed over the anchors with M_cls, smooth-L1 on the 7 box residuals (the yaw term divided by pi and, as
// upstream broadcasts it over the three columns it is added to, counted three times) summed over the anchors with M_reg, both
// divided by max(#M_reg, 1).  Through torch this is ~40 elementwise / reduction launches forward and backward (0.25 ms of a
// 7.4 ms step); here: count the positives, one pass that writes the gradient of both terms with respect to the FUSED head maps
// (B, n_cls * n_yaw * (1 + 7), H, W) and per-workgroup partial sums, one merge in a fixed order.  fp32, bit-repeatable.
//   class channel of anchor (cls, yaw):            cls * n_yaw + yaw
//   box channel d of anchor (cls, yaw):  n_anchor + (cls * 7 + d) * n_yaw + yaw          (ProposalLayer.reshape_cls / reshape_reg)
#include "v3d_internal.h"

#define PL_BLOCKS 512
#define PL_DOF 7

__global__ __launch_bounds__(V3D_BLOCK) void pl_count_kernel(const unsigned char* __restrict__ m_reg, long long n, int* __restrict__ partial) {
  __shared__ int red[V3D_BLOCK / 64];
  int c = 0;
  for (long long i = (long long)blockIdx.x * V3D_BLOCK + threadIdx.x; i < n; i += (long long)gridDim.x * V3D_BLOCK) c += m_reg[i] != 0;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) {
    int t = 0;
    for (int w = 0; w < V3D_BLOCK / 64; w++) t += red[w];
    partial[blockIdx.x] = t;
  }
}

__global__ void pl_count_merge_kernel(const int* __restrict__ partial, int blocks, float* __restrict__ normalizer) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    long long t = 0;
    for (int i = 0; i < blocks; i++) t += partial[i];
    *normalizer = t > 0 ? (float)t : 1.f;
  }
}

// thread = anchor (b, a = cls * n_yaw + yaw, pixel); dmaps gets d(cls_loss)/d(map) in the class channels and d(reg_loss)/d(map) in
// the box channels (each already divided by the normalizer); partial[block] = (sum of focal terms, sum of box terms) of the block
__global__ __launch_bounds__(V3D_BLOCK) void pl_main_kernel(const float* __restrict__ maps, const signed char* __restrict__ g_cls,
                                                            const unsigned char* __restrict__ m_cls, const float* __restrict__ g_reg,
                                                            const unsigned char* __restrict__ m_reg, int B, int n_cls, int n_yaw, int HW,
                                                            float alpha, float gamma, const float* __restrict__ normalizer,
                                                            float* __restrict__ dmaps, float* __restrict__ partial) {
  __shared__ float red[2][V3D_BLOCK / 64];
  const int n_anchor = n_cls * n_yaw, O = n_anchor * (1 + PL_DOF);
  const long long total = (long long)B * n_anchor * HW;
  const float inv_n = 1.f / *normalizer;
  float s_cls = 0.f, s_reg = 0.f;
  for (long long i = (long long)blockIdx.x * V3D_BLOCK + threadIdx.x; i < total; i += (long long)gridDim.x * V3D_BLOCK) {
    const int pix = (int)(i % HW);
    const int a = (int)((i / HW) % n_anchor), b = (int)(i / ((long long)HW * n_anchor));
    const int cls = a / n_yaw, yaw = a - cls * n_yaw;
    const size_t mb = (size_t)b * O * HW;
    // ---- focal term
    const size_t ic = mb + (size_t)a * HW + pix;
    const float x = maps[ic];
    float gx = 0.f;
    if (m_cls[i]) {
      const float t = g_cls[i] > 0 ? 1.f : 0.f;
      const float e = expf(-fabsf(x));
      const float prob = x >= 0.f ? 1.f / (1.f + e) : e / (1.f + e);
      const float bce = fmaxf(x, 0.f) - x * t + log1pf(e);
      const float p_t = prob * t + (1.f - prob) * (1.f - t);
      const float q = 1.f - p_t;
      const float w = alpha >= 0.f ? alpha * t + (1.f - alpha) * (1.f - t) : 1.f;
      const float qg = gamma == 2.f ? q * q : powf(q, gamma);
      const float qg1 = gamma == 2.f ? q : powf(q, gamma - 1.f);
      s_cls += w * bce * qg;
      // d/dx: bce' = prob - t, p_t' = prob (1 - prob) (2 t - 1)
      gx = w * ((prob - t) * qg - bce * gamma * qg1 * prob * (1.f - prob) * (2.f * t - 1.f)) * inv_n;
    }
    dmaps[ic] = gx;
    // ---- box term
    const bool pos = m_reg[i] != 0;
    const float* g = g_reg + (size_t)i * PL_DOF;
#pragma unroll
    for (int d = 0; d < PL_DOF; d++) {
      const size_t ir = mb + ((size_t)n_anchor + (size_t)(cls * PL_DOF + d) * n_yaw + yaw) * HW + pix;
      float gd = 0.f;
      if (pos) {
        const float diff = maps[ir] - g[d];
        const float ad = fabsf(diff);
        const float wd = d == PL_DOF - 1 ? 3.f / 3.14159265358979323846f : 1.f;
        s_reg += wd * (ad < 1.f ? 0.5f * diff * diff : ad - 0.5f);
        gd = wd * fminf(fmaxf(diff, -1.f), 1.f) * inv_n;
      }
      dmaps[ir] = gd;
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    s_cls += __shfl_xor(s_cls, off);
    s_reg += __shfl_xor(s_reg, off);
  }
  if ((threadIdx.x & 63) == 0) {
    red[0][threadIdx.x >> 6] = s_cls;
    red[1][threadIdx.x >> 6] = s_reg;
  }
  __syncthreads();
  if (threadIdx.x < 2) {
    float t = 0.f;
    for (int w = 0; w < V3D_BLOCK / 64; w++) t += red[threadIdx.x][w];
    partial[blockIdx.x * 2 + threadIdx.x] = t;
  }
}

__global__ void pl_finalize_kernel(const float* __restrict__ partial, int blocks, const float* __restrict__ normalizer,
                                   float* __restrict__ losses) {
  if (blockIdx.x == 0 && threadIdx.x < 2) {
    double t = 0.0;
    for (int i = 0; i < blocks; i++) t += (double)partial[i * 2 + threadIdx.x];
    losses[threadIdx.x] = (float)(t / (double)*normalizer);
    if (threadIdx.x == 0) losses[2] = *normalizer;
  }
}

// dmaps[class channels] *= *g_cls, dmaps[box channels] *= *g_reg (the upstream gradients of the two loss terms, device scalars)
__global__ __launch_bounds__(V3D_BLOCK) void pl_scale_kernel(float* __restrict__ dmaps, int B, int n_anchor, int HW,
                                                             const float* __restrict__ g_cls, const float* __restrict__ g_reg) {
  const int O = n_anchor * (1 + PL_DOF);
  const long long total = (long long)B * O * HW;
  const float gc = *g_cls, gr = *g_reg;
  for (long long i = (long long)blockIdx.x * V3D_BLOCK + threadIdx.x; i < total; i += (long long)gridDim.x * V3D_BLOCK) {
    const int o = (int)((i / HW) % O);
    dmaps[i] *= o < n_anchor ? gc : gr;
  }
}

extern "C" size_t v3d_proposal_loss_workspace(void) { return (size_t)PL_BLOCKS * (2 * sizeof(float) + sizeof(int)) + 256; }

// maps: fused head maps (B, n_cls * n_yaw * 8, H, W) fp32; g_cls int8 (> 0 = positive), m_cls / m_reg bytes (nonzero = counted),
// g_reg (B, n_cls, n_yaw, H, W, 7) fp32.  losses[3] = cls_loss, reg_loss, normalizer; dmaps: same shape as maps.
extern "C" int v3d_proposal_loss_fwd_bwd(const float* maps, const int8_t* g_cls, const uint8_t* m_cls, const float* g_reg,
                                         const uint8_t* m_reg, int B, int n_cls, int n_yaw, int H, int W, float alpha, float gamma,
                                         float* losses, float* dmaps, void* workspace, size_t workspace_bytes, v3d_stream_t stream) {
  if (!maps || !g_cls || !m_cls || !g_reg || !m_reg || !losses || !dmaps || !workspace || B < 1 || n_cls < 1 || n_yaw < 1 || H < 1 || W < 1)
    return V3D_EINVAL;
  if (workspace_bytes < v3d_proposal_loss_workspace()) return V3D_EWORKSPACE;
  const long long n = (long long)B * n_cls * n_yaw * H * W;
  hipStream_t st = (hipStream_t)stream;
  float* partial = (float*)workspace;
  int* counts = (int*)(partial + 2 * PL_BLOCKS);
  float* normalizer = losses + 2;
  const int blocks = (int)std::min<long long>(PL_BLOCKS, (n + V3D_BLOCK - 1) / V3D_BLOCK);
  hipLaunchKernelGGL(pl_count_kernel, dim3(blocks), dim3(V3D_BLOCK), 0, st, m_reg, n, counts);
  hipLaunchKernelGGL(pl_count_merge_kernel, dim3(1), dim3(64), 0, st, counts, blocks, normalizer);
  hipLaunchKernelGGL(pl_main_kernel, dim3(blocks), dim3(V3D_BLOCK), 0, st, maps, (const signed char*)g_cls, m_cls, g_reg, m_reg, B, n_cls,
                     n_yaw, H * W, alpha, gamma, normalizer, dmaps, partial);
  hipLaunchKernelGGL(pl_finalize_kernel, dim3(1), dim3(64), 0, st, partial, blocks, normalizer, losses);
  V3D_CHECK_LAUNCH();
  return V3D_OK;
}

extern "C" int v3d_proposal_loss_scale(float* dmaps, int B, int n_cls, int n_yaw, int H, int W, const float* g_cls, const float* g_reg,
                                       v3d_stream_t stream) {
  if (!dmaps || !g_cls || !g_reg || B < 1 || n_cls < 1 || n_yaw < 1 || H < 1 || W < 1) return V3D_EINVAL;
  const long long total = (long long)B * n_cls * n_yaw * (1 + PL_DOF) * H * W;
  hipLaunchKernelGGL(pl_scale_kernel, dim3((int)std::min<long long>(2048, (total + V3D_BLOCK - 1) / V3D_BLOCK)), dim3(V3D_BLOCK), 0,
                     (hipStream_t)stream, dmaps, B, n_cls * n_yaw, H * W, g_cls, g_reg);
  V3D_CHECK_LAUNCH();
  return V3D_OK;
}
