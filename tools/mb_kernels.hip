// Microbenchmark kernels (development tool, not part of the product library).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
#include <algorithm>
#include <numeric>
#include <random>

__global__ void chase_kernel(const unsigned* __restrict__ next, int steps, unsigned* out, long long* cycles) {
  unsigned p = 0;
  long long t0 = clock64();
  for (int i = 0; i < steps; i++) p = next[p];
  long long t1 = clock64();
  *out = p;
  *cycles = t1 - t0;
}

__global__ void atomic_chain_kernel(unsigned long long* slots, int steps, unsigned* out, long long* cycles) {
  unsigned acc = 0;
  long long t0 = clock64();
  for (int i = 0; i < steps; i++) {
    unsigned long long prev = atomicCAS(&slots[(acc * 977u + i * 131u) & 0xFFFF], 0xFFFFFFFFFFFFFFFFull, (unsigned long long)i);
    acc += (unsigned)prev;
  }
  long long t1 = clock64();
  *out = acc;
  *cycles = t1 - t0;
}

__global__ void empty_kernel() {}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

int main() {
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  printf("device %s, clock %d kHz, mem clock %d kHz, CUs %d\n", prop.name, prop.clockRate, prop.memoryClockRate, prop.multiProcessorCount);
  unsigned* d_out; long long* d_cyc; CK(hipMalloc(&d_out, 4)); CK(hipMalloc(&d_cyc, 8));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (size_t bytes : {size_t(64) << 10, size_t(2) << 20, size_t(64) << 20, size_t(1) << 30}) {
    size_t n = bytes / 4;
    std::vector<unsigned> h(n);
    // random cyclic permutation with a 64-element (256 B) stride granularity
    size_t lines = n / 64;
    std::vector<unsigned> perm(lines); std::iota(perm.begin(), perm.end(), 0u);
    std::mt19937 rng(1); std::shuffle(perm.begin() + 1, perm.end(), rng);
    for (size_t i = 0; i < lines; i++) h[(size_t)perm[i] * 64] = perm[(i + 1) % lines] * 64;
    unsigned* d; CK(hipMalloc(&d, bytes)); CK(hipMemcpy(d, h.data(), bytes, hipMemcpyHostToDevice));
    int steps = (int)std::min<size_t>(lines, 20000);
    for (int rep = 0; rep < 3; rep++) {
      CK(hipEventRecord(e0)); hipLaunchKernelGGL(chase_kernel, dim3(1), dim3(1), 0, 0, d, steps, d_out, d_cyc); CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      long long cyc; CK(hipMemcpy(&cyc, d_cyc, 8, hipMemcpyDeviceToHost));
      if (rep == 2) printf("chase %8zu KB: %7.1f ns/load  (%lld clock64 ticks/load, %d steps)\n", bytes >> 10, ms * 1e6 / steps, cyc / steps, steps);
    }
    CK(hipFree(d));
  }
  unsigned long long* slots; CK(hipMalloc(&slots, 65536 * 8)); CK(hipMemset(slots, 0xFF, 65536 * 8));
  for (int rep = 0; rep < 3; rep++) {
    CK(hipEventRecord(e0)); hipLaunchKernelGGL(atomic_chain_kernel, dim3(1), dim3(1), 0, 0, slots, 2000, d_out, d_cyc); CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (rep == 2) printf("dependent 64-bit atomicCAS: %7.1f ns each\n", ms * 1e6 / 2000);
  }
  // launch cadence of empty kernels
  for (int rep = 0; rep < 2; rep++) {
    CK(hipEventRecord(e0)); for (int i = 0; i < 1000; i++) hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, 0); CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (rep == 1) printf("empty kernel back-to-back: %.2f us each\n", ms * 1e3 / 1000);
  }
  return 0;
}
