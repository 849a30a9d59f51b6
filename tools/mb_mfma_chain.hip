// Round 6 microbenchmark: cycles per v_mfma_f32_16x16x32_f16 as a function of (a) how many independent accumulators the stream
// rotates over (distance between two MFMAs on the SAME accumulator), (b) waves per SIMD, (c) operand values (zeros / random: the
// power-management effect), (d) whether the accumulators may live in AGPRs.  One workgroup per CU.
// build + run: hipcc --offload-arch=gfx950 -O3 tools/mb_mfma_chain.hip -o /tmp/mb_mfma_chain && /tmp/mb_mfma_chain
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int R, int GROUPS>
__global__ void chain(const f16x8* __restrict__ ab, float* __restrict__ out, unsigned long long* __restrict__ cyc, int iters) {
  f16x8 a[4], b[4];
  for (int i = 0; i < 4; i++) { a[i] = ab[(threadIdx.x & 63) * 8 + i]; b[i] = ab[(threadIdx.x & 63) * 8 + 4 + i]; }
  f32x4 acc[R];
  for (int i = 0; i < R; i++) acc[i] = f32x4{0, 0, 0, 0};
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int g = 0; g < GROUPS; g++)
#pragma unroll
      for (int i = 0; i < R; i++) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[(g + i) & 3], b[(g * 3 + i) & 3], acc[i], 0, 0, 0);
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  f32x4 s = acc[0];
  for (int i = 1; i < R; i++) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y + s.z + s.w;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int R>
void run(const char* what, const f16x8* ab, float* out, unsigned long long* cyc, int threads, int grid) {
  constexpr int GROUPS = 96 / R;
  const int iters = 200;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  chain<R, GROUPS><<<grid, threads>>>(ab, out, cyc, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  chain<R, GROUPS><<<grid, threads>>>(ab, out, cyc, iters);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(grid);
  hipMemcpy(h.data(), cyc, grid * 8, hipMemcpyDeviceToHost);
  double avg = 0; for (auto v : h) avg += v; avg /= grid;
  const double n_per_simd = (double)iters * GROUPS * R * (threads / 256.0);
  printf("%-8s R=%2d waves/SIMD=%d grid=%3d: %.1f counter ticks per MFMA per SIMD, %.1f ns per MFMA per SIMD (wall %.3f ms) -> %.0f TFLOP/s\n", what, R, threads / 256, grid,
         avg / n_per_simd, ms * 1e6 / n_per_simd, ms, (double)grid * (threads / 64) * iters * GROUPS * R * 16384.0 / (ms * 1e-3) / 1e12);
}

int main() {
  f16x8* ab; float* out; unsigned long long* cyc;
  hipMalloc(&ab, 64 * 8 * sizeof(f16x8)); hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8);
  std::vector<_Float16> h(64 * 8 * 8);
  for (int mode = 0; mode < 2; mode++) {
    for (auto& v : h) v = mode ? (_Float16)((rand() % 2001 - 1000) / 1000.0f) : (_Float16)0.f;
    hipMemcpy(ab, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    const char* what = mode ? "random" : "zeros";
    for (int grid : {256, 32}) {
      run<1>(what, ab, out, cyc, 256, grid);
      run<2>(what, ab, out, cyc, 256, grid);
      run<4>(what, ab, out, cyc, 256, grid);
      run<8>(what, ab, out, cyc, 256, grid);
      run<16>(what, ab, out, cyc, 256, grid);
      run<4>(what, ab, out, cyc, 512, grid);
      run<8>(what, ab, out, cyc, 512, grid);
    }
  }
  return 0;
}
