// Microbenchmark (development tool): what a row gather costs on one CU.  The sparse kernels gather 16 random 256-byte rows per
// (tile, offset) with four global_load_dwordx4 per lane; removing the gathers takes the KITTI ring kernel from 11.9 to 8.1 us
// and the Waymo k-outer kernel from 53 to 33 us, independent of how far ahead they are issued.  Which part of the memory
// pipe charges that?  Patterns (each = 4 wave instructions = 16 rows x 256 B, all rows L2-resident after the warm-up):
//   mfma   lane (r = lane & 15, kg = lane >> 4) reads bytes kg*32 + {0, 16, 128, 144} of row r     (what the kernels do)
//   quad   lane reads row lane >> 2, 16-byte chunk (lane & 3) + 4 i                                (a quad = one row, 64 B)
//   line   lane reads row (lane >> 3) + 8 (i >> 1), chunk (lane & 7) + 8 (i & 1)                    (8 lanes = one 128-B line)
//   zero   every lane of a quad group reads the same row 0                                          (the "absent neighbour" row)
//   seq    rows are consecutive (r, r + 1, ...): the gather of a perfectly ordered cloud
// W waves per workgroup (one workgroup per CU), N gathers per wave, DEPTH gathers in flight per wave.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/mb_gather tools/mb_gather.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <random>

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int ROWS = 8192, ROWB = 256;

template <int PAT, int DEPTH>
__global__ __launch_bounds__(1024) void gather_kernel(const unsigned char* __restrict__ base, const int* __restrict__ perm, int n_iter,
                                                      float* sink) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int gw = blockIdx.x * (blockDim.x >> 6) + wv;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  // the 16 rows of gather `it` of this wave: perm[(gw * 977 + it * 16 + r) % ROWS] -- loaded up front (index traffic is not the question)
  for (int it0 = 0; it0 < n_iter; it0 += DEPTH) {
    f32x4 v[DEPTH][4];
    int row[DEPTH];
#pragma unroll
    for (int d = 0; d < DEPTH; d++) {
      const int it = it0 + d;
      const int rsel = PAT == 1 ? (lane >> 2) : (PAT == 2 ? (lane >> 3) : (lane & 15));
      int rr = perm[(gw * 977 + it * 16 + rsel) & (ROWS - 1)];
      if (PAT == 3) rr = 0;
      if (PAT == 4) rr = (gw * 977 + it * 16 + rsel) & (ROWS - 1);
      row[d] = rr;
    }
#pragma unroll
    for (int d = 0; d < DEPTH; d++) {
      const unsigned char* p = base + (size_t)row[d] * ROWB;
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const unsigned char* q;
        if (PAT == 1) q = p + ((lane & 3) + 4 * i) * 16;
        else if (PAT == 2) {
          const int it = it0 + d;
          const int r2 = perm[(gw * 977 + it * 16 + (lane >> 3) + 8 * (i >> 1)) & (ROWS - 1)];
          q = base + (size_t)r2 * ROWB + ((lane & 7) + 8 * (i & 1)) * 16;
        } else q = p + (lane >> 4) * 32 + (i & 1) * 16 + (i >> 1) * 128;
        v[d][i] = *reinterpret_cast<const f32x4*>(q);
      }
    }
#pragma unroll
    for (int d = 0; d < DEPTH; d++)
#pragma unroll
      for (int i = 0; i < 4; i++) acc += v[d][i];
  }
  if (acc.x + acc.y + acc.z + acc.w == 12345.678f) sink[0] = 1.f;
}

template <typename F>
static float time_us(F launch, int reps) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int i = 0; i < 5; i++) launch();
  (void)hipDeviceSynchronize();
  float best = 1e30f;
  for (int t = 0; t < 3; t++) {
    (void)hipEventRecord(e0);
    for (int i = 0; i < reps; i++) launch();
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    best = ms * 1e3f / reps < best ? ms * 1e3f / reps : best;
  }
  return best;
}

int main() {
  unsigned char* base; (void)hipMalloc(&base, (size_t)ROWS * ROWB); (void)hipMemset(base, 0, (size_t)ROWS * ROWB);
  std::vector<int> perm(ROWS);
  for (int i = 0; i < ROWS; i++) perm[i] = i;
  std::mt19937 rng(3); std::shuffle(perm.begin(), perm.end(), rng);
  int* dperm; (void)hipMalloc(&dperm, ROWS * 4); (void)hipMemcpy(dperm, perm.data(), ROWS * 4, hipMemcpyHostToDevice);
  float* sink; (void)hipMalloc(&sink, 4);
  const char* names[5] = {"mfma", "quad", "line", "zero", "seq "};
  for (int waves : {8, 4, 16}) {
    const int n_iter = 432 / waves;  // 432 gathers of 16 rows per CU (the KITTI ring workgroup issues 54 x 6 = 324, k-outer 432 per pass)
    printf("grid 256 x %d waves, %d gathers (16 rows x 256 B each) per wave = %.0f KB per CU\n", waves, n_iter, waves * n_iter * 4.0);
#define RUN(P, D)                                                                                                       \
  {                                                                                                                     \
    const float t = time_us([&] { hipLaunchKernelGGL((gather_kernel<P, D>), dim3(256), dim3(waves * 64), 0, 0, base, dperm, n_iter / D * D, sink); }, 100); \
    printf("  %s depth %d : %6.2f us per launch -> %5.1f ns per gather per CU, %6.1f GB/s per CU\n", names[P], D, t,       \
           t * 1e3 / (waves * (n_iter / D * D)), waves * (n_iter / D * D) * 4096.0 / t * 1e-3);                          \
  }
    RUN(0, 2) RUN(0, 6) RUN(1, 2) RUN(1, 6) RUN(2, 2) RUN(2, 6) RUN(3, 2) RUN(3, 6) RUN(4, 2) RUN(4, 6)
  }
  return 0;
}
