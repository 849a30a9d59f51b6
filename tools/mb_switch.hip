// Microbenchmark (development tool): what does a kernel launch cost when the PREVIOUS launch on the stream was a different
// kernel?  (One frame at a time, the 64 -> 64 sparse ring kernel takes 14.1 us in the SECOND frame and 9.9 us launched back to
// back on the same layer; the first launch of a batch of identical ones shows the same 13.9 us.)  Sequences of 256 launches in a
// captured graph: AAAA..., ABAB... for pairs that differ in code only, in LDS size, in register count, in grid size.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/mb_switch tools/mb_switch.hip ; run: tools/mb_switch
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int VARIANT>  // same body, different code objects
__global__ __launch_bounds__(256) void k_small(float* p, int n) {
  extern __shared__ float lds[];
  const int i = blockIdx.x * 256 + threadIdx.x;
  float v = p[i % n] + VARIANT;
  lds[threadIdx.x] = v;
  __syncthreads();
  for (int r = 0; r < 64; r++) v = v * 1.0001f + lds[(threadIdx.x + r) & 255];
  if (v == 12345.f) p[i % n] = v;
}

template <int VARIANT>
__global__ __launch_bounds__(256) void k_regs(float* p, int n) {  // ~200 live registers
  const int i = blockIdx.x * 256 + threadIdx.x;
  float a[192];
#pragma unroll
  for (int j = 0; j < 192; j++) a[j] = p[(i + j * 64) % n] + VARIANT;
  float v = 0.f;
#pragma unroll
  for (int r = 0; r < 4; r++)
#pragma unroll
    for (int j = 0; j < 192; j++) { a[j] = a[j] * 1.0001f + v; v += a[j]; }
  if (v == 12345.f) p[i % n] = v;
}

typedef void (*kern_t)(float*, int);
struct Launch { kern_t k; int grid; int lds; const char* name; };

static float run(const std::vector<Launch>& seq, float* buf, int n, hipStream_t st) {
  hipGraph_t g; hipGraphExec_t ge;
  hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
  for (int r = 0; r < 256; r++) {
    const Launch& l = seq[r % seq.size()];
    hipLaunchKernelGGL(l.k, dim3(l.grid), dim3(256), l.lds, st, buf, n);
  }
  hipStreamEndCapture(st, &g);
  hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9f;
  for (int t = 0; t < 5; t++) {
    hipEventRecord(e0, st); hipGraphLaunch(ge, st); hipEventRecord(e1, st); hipStreamSynchronize(st);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (t && ms < best) best = ms;
  }
  hipGraphExecDestroy(ge); hipGraphDestroy(g);
  return best * 1e3f / 256;
}

int main() {
  const int n = 1 << 22;
  float* buf; CHECK(hipMalloc(&buf, n * 4)); CHECK(hipMemset(buf, 0, n * 4));
  hipStream_t st; CHECK(hipStreamCreate(&st));
  CHECK(hipFuncSetAttribute((const void*)k_small<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
  CHECK(hipFuncSetAttribute((const void*)k_small<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
  const Launch A{k_small<0>, 256, 1024, "small<0> lds 1K"}, A2{k_small<1>, 256, 1024, "small<1> lds 1K"};
  const Launch AL{k_small<0>, 256, 128 * 1024, "small<0> lds 128K"}, A2L{k_small<1>, 256, 128 * 1024, "small<1> lds 128K"};
  const Launch AG{k_small<0>, 1024, 1024, "small<0> grid 1024"};
  const Launch R{k_regs<0>, 256, 0, "regs<0>"}, R2{k_regs<1>, 256, 0, "regs<1>"};
  struct { const char* what; std::vector<Launch> seq; } tests[] = {
      {"A A A A            (same kernel, 1 KB LDS)", {A}},
      {"A A' A A'          (different code, same resources)", {A, A2}},
      {"AL AL AL           (same kernel, 128 KB LDS)", {AL}},
      {"AL AL' AL AL'      (different code, 128 KB LDS both)", {AL, A2L}},
      {"A AL A AL          (same code, LDS 1 KB <-> 128 KB)", {A, AL}},
      {"A A'L A A'L        (different code and LDS size)", {A, A2L}},
      {"A AG A AG          (same code, grid 256 <-> 1024)", {A, AG}},
      {"R R R R            (same kernel, ~200 registers)", {R}},
      {"R R' R R'          (different code, ~200 registers)", {R, R2}},
      {"A R A R            (small <-> large register file)", {A, R}},
      {"AL R AL R          (128 KB LDS <-> large register file)", {AL, R}},
  };
  for (auto& t : tests) {
    float us = run(t.seq, buf, n, st);
    float base = 0.f;
    for (auto& l : t.seq) base += run({l}, buf, n, st);
    base /= t.seq.size();
    printf("%-62s %6.2f us per launch   (mean of its kernels back to back: %6.2f)\n", t.what, us, base);
  }
  return 0;
}
