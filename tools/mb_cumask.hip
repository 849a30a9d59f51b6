// Probe (development tool, round 5): which XCDs / CUs a hipExtStreamCreateWithCUMask stream runs on, as a function of the mask bits.
// Every workgroup of a 4 096-workgroup launch records its XCC id (HW_REG_XCC_ID) and its (SE, CU) from HW_REG_HW_ID.
// build: hipcc --offload-arch=gfx950 -O2 -o tools/mb_cumask tools/mb_cumask.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <vector>
__global__ void where(unsigned* out) {
  if (threadIdx.x == 0) {
    unsigned xcc, hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    out[blockIdx.x * 2] = xcc;
    out[blockIdx.x * 2 + 1] = hw;
  }
  // stay resident a little so that the launch spreads over everything the mask allows
  unsigned long long t0 = __builtin_readcyclecounter();
  while (__builtin_readcyclecounter() - t0 < 20000) {}
}
static void run(const char* name, const std::vector<unsigned>& mask) {
  hipStream_t st;
  hipError_t e = hipExtStreamCreateWithCUMask(&st, (unsigned)mask.size(), mask.data());
  if (e != hipSuccess) { printf("%s: hipExtStreamCreateWithCUMask failed: %s\n", name, hipGetErrorString(e)); return; }
  const int N = 4096;
  unsigned* d; hipMalloc(&d, N * 8);
  hipLaunchKernelGGL(where, dim3(N), dim3(64), 0, st, d);
  hipStreamSynchronize(st);
  std::vector<unsigned> h(N * 2);
  hipMemcpy(h.data(), d, N * 8, hipMemcpyDeviceToHost);
  int per_xcc[16] = {};
  bool cu_seen[16][64] = {};
  for (int i = 0; i < N; i++) {
    const unsigned xcc = h[2 * i] & 15u, hw = h[2 * i + 1];
    const unsigned cu = (hw >> 8) & 15u, sh = (hw >> 12) & 1u, se = (hw >> 13) & 7u;  // gfx9 HW_ID: cu_id [11:8], sh_id [12], se_id [15:13]
    per_xcc[xcc]++;
    cu_seen[xcc][(se * 2 + sh) * 16 + cu > 63 ? 63 : (se * 2 + sh) * 16 + cu] = true;
  }
  printf("%-28s workgroups per XCC:", name);
  for (int x = 0; x < 8; x++) printf(" %4d", per_xcc[x]);
  printf("   distinct (se,sh,cu) per XCC:");
  for (int x = 0; x < 8; x++) { int c = 0; for (int j = 0; j < 64; j++) c += cu_seen[x][j]; printf(" %2d", c); }
  printf("\n");
  hipFree(d); hipStreamDestroy(st);
}
int main() {
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  printf("device: %s, %d CUs\n", p.name, p.multiProcessorCount);
  const int words = 8;  // 256 bits
  run("all 256 bits", std::vector<unsigned>(words, 0xFFFFFFFFu));
  for (int k = 0; k < 8; k++) {  // 32 consecutive bits
    std::vector<unsigned> m(words, 0u); m[k] = 0xFFFFFFFFu;
    char nm[64]; snprintf(nm, sizeof nm, "bits %3d..%3d", 32 * k, 32 * k + 31); run(nm, m);
  }
  for (int r = 0; r < 8; r += 2) {  // bits with i % 8 in {r, r + 1}
    std::vector<unsigned> m(words, 0u);
    for (int i = 0; i < 256; i++) if (i % 8 == r || i % 8 == r + 1) m[i / 32] |= 1u << (i % 32);
    char nm[64]; snprintf(nm, sizeof nm, "bits i %% 8 in {%d,%d}", r, r + 1); run(nm, m);
  }
  return 0;
}
