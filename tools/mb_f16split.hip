// Probe (development tool, round 5): arithmetic of the split-precision products on gfx950.
//   1. does v_mfma_f32_16x16x32_f16 flush SUBNORMAL f16 inputs?
//   2. error of a conv-shaped dot product (K = 1728 = 27 offsets x 64 channels) against float64 for
//        bf16x3   activations / weights = bf16 hi + lo (RNE), 3 terms            (rounds 1-4)
//        f16x3    f16 hi + lo, power-of-two scales, 3 terms  (hi RNE / lo RNE)
//        f16x3z   same, hi by v_cvt_pkrtz (round to zero), lo RNE
//        f16x3u   same as f16x3 without scaling (what the lo piece loses to the f16 subnormal quantum)
//        f32      v_mfma_f32_16x16x4_f32 (exact fp32 fma chain)
//      reported: max and rms of |d - ref| / |ref| over outputs with |ref| > 1e-3 max|ref|, and max |d - ref| / max|ref|.
// build: hipcc --offload-arch=gfx950 -O2 -o tools/mb_f16split tools/mb_f16split.hip
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void split_bf16(const float (&x)[8], bf16x8_t& hi, bf16x8_t& lo) {
  u32x4_t h, l;
  for (int i = 0; i < 4; i++) {
    const bf16x2_t hh = __builtin_convertvector(f32x2_t{x[2 * i], x[2 * i + 1]}, bf16x2_t);
    const unsigned hb = __builtin_bit_cast(unsigned, hh);
    const float r0 = x[2 * i] - __uint_as_float(hb << 16), r1 = x[2 * i + 1] - __uint_as_float(hb & 0xFFFF0000u);
    h[i] = hb;
    l[i] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{r0, r1}, bf16x2_t));
  }
  hi = __builtin_bit_cast(bf16x8_t, h);
  lo = __builtin_bit_cast(bf16x8_t, l);
}
template <int RTZ_HI>
__device__ __forceinline__ void split_f16(const float (&x)[8], float s, f16x8_t& hi, f16x8_t& lo) {
  u32x4_t h, l;
  for (int i = 0; i < 4; i++) {
    const float a = x[2 * i] * s, b = x[2 * i + 1] * s;
    f16x2_t hh;
    if (RTZ_HI) hh = __builtin_bit_cast(f16x2_t, __builtin_amdgcn_cvt_pkrtz(a, b));
    else hh = __builtin_convertvector(f32x2_t{a, b}, f16x2_t);
    const float r0 = a - (float)hh[0], r1 = b - (float)hh[1];
    h[i] = __builtin_bit_cast(unsigned, hh);
    l[i] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{r0, r1}, f16x2_t));
  }
  hi = __builtin_bit_cast(f16x8_t, h);
  lo = __builtin_bit_cast(f16x8_t, l);
}

// the four-instruction split of csrc/v3d_common.h against the plain expressions, bit for bit
__global__ void mix_probe(const float* __restrict__ x, int n, float s, unsigned* __restrict__ mism) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (2 * i + 1 >= n) return;
  const float x0 = x[2 * i], x1 = x[2 * i + 1];
  unsigned h = 0u, l = 0u;
  asm("v_fma_mixlo_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]" : "+v"(h) : "v"(x0), "s"(s));
  asm("v_fma_mixhi_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]" : "+v"(h) : "v"(x1), "s"(s));
  asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "+v"(l) : "v"(x0), "s"(s), "v"(h));
  asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l) : "v"(x1), "s"(s), "v"(h));
  const float a = x0 * s, b = x1 * s;
  const f16x2_t hh = __builtin_convertvector(f32x2_t{a, b}, f16x2_t);
  const float r0 = a - (float)hh[0], r1 = b - (float)hh[1];
  const unsigned he = __builtin_bit_cast(unsigned, hh), le = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{r0, r1}, f16x2_t));
  if (h != he || l != le) atomicAdd(mism, 1u);
}

__global__ void denorm_probe(float* out) {
  const int lane = threadIdx.x;
  f16x8_t a = {}, b = {};
  // A[r][k = 0] = 2^-20 (f16 subnormal, bits 0x0010), B[0][c] = 1024
  if ((lane >> 4) == 0) {
    a[0] = __builtin_bit_cast(_Float16, (unsigned short)0x0010);
    b[0] = (_Float16)1024.f;
  }
  f32x4 d = {0.f, 0.f, 0.f, 0.f};
  d = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, d, 0, 0, 0);
  out[lane] = d[0];
}

// mode: 0 bf16x3, 1 f16x3 (RNE/RNE scaled), 2 f16x3z (RTZ hi), 3 f16x3 unscaled, 4 f32 mfma, 5 f16x3 + 4th term lo*lo
__global__ void dot_kernel(const float* __restrict__ A /*[T][16][K]*/, const float* __restrict__ W /*[K][16]*/, int K, int mode, float sa,
                           float sw, float* __restrict__ D /*[T][16][16]*/) {
  const int lane = threadIdx.x, r = lane & 15, kg = lane >> 4, t = blockIdx.x;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const float* a_row = A + ((size_t)t * 16 + r) * K;
  if (mode == 4) {
    for (int k0 = 0; k0 < K; k0 += 4) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a_row[k0 + kg], W[(size_t)(k0 + kg) * 16 + r], acc, 0, 0, 0);
  } else {
    for (int k0 = 0; k0 < K; k0 += 32) {
      float a[8], w[8];
      for (int e = 0; e < 8; e++) {
        a[e] = a_row[k0 + kg * 8 + e];
        w[e] = W[(size_t)(k0 + kg * 8 + e) * 16 + r];
      }
      if (mode == 0) {
        bf16x8_t ah, al, wh, wl;
        split_bf16(a, ah, al);
        split_bf16(w, wh, wl);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, wh, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, wl, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, wh, acc, 0, 0, 0);
      } else {
        f16x8_t ah, al, wh, wl;
        const float s1 = mode == 3 ? 1.f : sa, s2 = mode == 3 ? 1.f : sw;
        if (mode == 2) split_f16<1>(a, s1, ah, al); else split_f16<0>(a, s1, ah, al);
        split_f16<0>(w, s2, wh, wl);
        if (mode == 5) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, wl, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, wh, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, wl, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, wh, acc, 0, 0, 0);
      }
    }
    if (mode != 0 && mode != 3 && mode != 4) {
      const float inv = 1.f / (sa * sw);
      for (int i = 0; i < 4; i++) acc[i] *= inv;
    }
  }
  for (int rr = 0; rr < 4; rr++) D[((size_t)t * 16 + kg * 4 + rr) * 16 + r] = acc[rr];
}

static double gauss() {
  const double u = (rand() + 1.0) / (RAND_MAX + 2.0), v = (rand() + 1.0) / (RAND_MAX + 2.0);
  return sqrt(-2.0 * log(u)) * cos(6.283185307179586 * v);
}

int main() {
  float* d_out;
  hipMalloc(&d_out, 64 * 4);
  hipLaunchKernelGGL(denorm_probe, dim3(1), dim3(64), 0, 0, d_out);
  float h_out[64];
  hipMemcpy(h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost);
  printf("denorm probe: f16 subnormal 2^-20 x 1024 through v_mfma_f32_16x16x32_f16 = %g (2^-10 = %g kept, 0 = flushed)\n", h_out[0], 1.0 / 1024);

  {
    const int n = 1 << 22;
    std::vector<float> hx(n);
    srand(99);
    for (int i = 0; i < n; i++) {  // magnitudes over 40 binades, both signs, some exact zeros and f16-subnormal products
      const double m = (rand() / (double)RAND_MAX) * 2.0 - 1.0;
      hx[i] = (i % 97 == 0) ? 0.f : (float)(m * pow(2.0, (rand() % 40) - 30));
    }
    float* dx; unsigned* dm;
    hipMalloc(&dx, n * 4); hipMalloc(&dm, 4);
    hipMemcpy(dx, hx.data(), n * 4, hipMemcpyHostToDevice);
    for (float s : {1.f, 256.f, 1.f / 1024.f, 16384.f}) {
      hipMemset(dm, 0, 4);
      hipLaunchKernelGGL(mix_probe, dim3(n / 2 / 256), dim3(256), 0, 0, dx, n, s, dm);
      unsigned mm = 0;
      hipMemcpy(&mm, dm, 4, hipMemcpyDeviceToHost);
      printf("v_fma_mix split vs expression split, scale %g: %u mismatching pairs of %d\n", s, mm, n / 2);
    }
    hipFree(dx); hipFree(dm);
  }
  const int T = 512, K = 1728;
  srand(1234);
  for (int scen = 0; scen < 3; scen++) {
    // 0: unit-scale activations (post BN + ReLU, 40 % active neighbours), kaiming weights;  1: activations x 100 (range headroom);
    // 2: activations x 1e-3 (small features)
    const double amul = scen == 0 ? 1.0 : scen == 1 ? 100.0 : 1e-3;
    std::vector<float> A((size_t)T * 16 * K), W((size_t)K * 16);
    float amax = 0.f, wmax = 0.f;
    for (auto& v : W) { v = (float)(gauss() * 0.034); wmax = fmaxf(wmax, fabsf(v)); }
    for (size_t i = 0; i < A.size(); i++) {
      const size_t kk = i % K;
      const bool offset_live = ((i / K) * 31 + (kk / 64) * 17) % 5 < 2;  // a neighbour is present under ~40 % of the offsets
      const double g = gauss();
      A[i] = offset_live && g > 0 ? (float)(g * amul) : 0.f;
      amax = fmaxf(amax, fabsf(A[i]));
    }
    int ea, ew;
    frexpf(amax, &ea); frexpf(wmax, &ew);              // amax = m * 2^ea, m in [0.5, 1)
    const float sa = ldexpf(1.f, 14 - ea), sw = ldexpf(1.f, 14 - ew);  // max * s in [2^13, 2^14)
    std::vector<double> ref((size_t)T * 256);
    double refmax = 0;
    for (int t = 0; t < T; t++)
      for (int r = 0; r < 16; r++)
        for (int c = 0; c < 16; c++) {
          double s = 0;
          for (int k = 0; k < K; k++) s += (double)A[((size_t)t * 16 + r) * K + k] * (double)W[(size_t)k * 16 + c];
          ref[((size_t)t * 16 + r) * 16 + c] = s;
          refmax = fmax(refmax, fabs(s));
        }
    float *dA, *dW, *dD;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dW, W.size() * 4); hipMalloc(&dD, ref.size() * 4);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dW, W.data(), W.size() * 4, hipMemcpyHostToDevice);
    printf("scenario %d: amax %.4g (scale 2^%d) wmax %.4g (scale 2^%d) max|ref| %.4g\n", scen, amax, 14 - ea, wmax, 14 - ew, refmax);
    const char* names[6] = {"bf16x3", "f16x3", "f16x3z", "f16x3u", "f32", "f16x4"};
    for (int mode = 0; mode < 6; mode++) {
      hipLaunchKernelGGL(dot_kernel, dim3(T), dim3(64), 0, 0, dA, dW, K, mode, sa, sw, dD);
      std::vector<float> D(ref.size());
      hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
      double emax = 0, e2 = 0, nmax = 0;
      size_t cnt = 0;
      for (size_t i = 0; i < ref.size(); i++) {
        const double e = fabs((double)D[i] - ref[i]);
        nmax = fmax(nmax, e / refmax);
        if (fabs(ref[i]) > 1e-3 * refmax) {
          const double rel = e / fabs(ref[i]);
          emax = fmax(emax, rel);
          e2 += rel * rel;
          cnt++;
        }
      }
      printf("  %-7s strict rel err on |ref| > 1e-3 max: max %.3e rms %.3e   max-norm %.3e\n", names[mode], emax, sqrt(e2 / cnt), nmax);
    }
    hipFree(dA); hipFree(dW); hipFree(dD);
  }
  return 0;
}
