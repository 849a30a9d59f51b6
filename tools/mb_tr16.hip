// Probe (development tool): what ds_read_b64_tr_b16 delivers.  LDS holds u16 value = its own element index; every lane supplies an
// 8-byte-aligned address; the kernel dumps the 4 halfwords each lane receives.
// build: hipcc --offload-arch=gfx950 -O2 -o tools/mb_tr16 tools/mb_tr16.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void probe(const int* addr_elems, unsigned short* out) {
  __shared__ unsigned short lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned short)i;
  __syncthreads();
  const unsigned a = (unsigned)(unsigned long long)(lds) + (unsigned)addr_elems[threadIdx.x] * 2u;
  unsigned long long v;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
  for (int j = 0; j < 4; j++) out[threadIdx.x * 4 + j] = (unsigned short)(v >> (16 * j));
}
int main() {
  int h_addr[64];
  unsigned short h_out[256];
  int* d_addr; unsigned short* d_out;
  hipMalloc(&d_addr, sizeof(h_addr)); hipMalloc(&d_out, sizeof(h_out));
  for (int pattern = 0; pattern < 2; pattern++) {
    // pattern 0: lane l -> element offset 100 * l (distinct, recognisable); pattern 1: row-major [pixel][16 ch]: lane (i = l & 15, kg = l >> 4)
    // reads pixel kg * 8 + (i & 3), channels (i >> 2) * 4 .. + 3 of a [32 pixels][16 channels] tile (element = pixel * 16 + channel)
    for (int l = 0; l < 64; l++) h_addr[l] = pattern == 0 ? 64 * l + 4 * (l & 1) * 0 : ((l >> 4) * 8 + (l & 3)) * 16 + ((l & 15) >> 2) * 4;
    hipMemcpy(d_addr, h_addr, sizeof(h_addr), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d_addr, d_out);
    hipMemcpy(h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost);
    printf("pattern %d\n", pattern);
    for (int l = 0; l < 64; l++) {
      printf(" lane %2d addr %4d -> %4d %4d %4d %4d", l, h_addr[l], h_out[l * 4], h_out[l * 4 + 1], h_out[l * 4 + 2], h_out[l * 4 + 3]);
      if (pattern == 1) {  // want: channel i = l & 15 of pixels kg*8 + 0..3 -> elements (kg*8 + j) * 16 + i
        bool ok = true;
        for (int j = 0; j < 4; j++) ok = ok && h_out[l * 4 + j] == ((l >> 4) * 8 + j) * 16 + (l & 15);
        printf("  %s", ok ? "= transposed fragment" : "MISMATCH");
      }
      printf("\n");
    }
  }
  return 0;
}
