// nms_device.h -- the greedy reduction of rotated NMS as a device function, shared by iou_nms.hip (nms_reduce_kernel) and the
// fused tail of the proposal stage (proposal.hip: reduction + score cut in one launch).
// Reference: the sequential scan of the suppression mask, nms_rotated_cuda.cu:106-128 (a D->H copy + host loop there).
#pragma once
#include <hip/hip_runtime.h>

namespace v3d {

__device__ __forceinline__ unsigned long long readlane64(unsigned long long v, int lane) {
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, lane);
  const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), lane);
  return ((unsigned long long)hi << 32) | lo;
}

// One workgroup of THREADS threads.  mask[i * nwords + w] bit c <=> sorted box i suppresses sorted box w * 64 + c (words left of
// the diagonal are never read); remv[nwords] is scratch.  `mask` and `remv` may live in LDS or in global memory (generic
// pointers).  Writes keep[0 .. nk) = order[i] of the kept boxes in sorted order and returns nk to EVERY thread (through *nk_s,
// one int of LDS).  Diagonal word resolved by wave 0 with readlane, the rows kept in a block OR-ed into remv in parallel.
template <int THREADS>
__device__ __forceinline__ int nms_greedy_reduce(const unsigned long long* mask, const int* __restrict__ order, int N, int nwords,
                                                 unsigned long long* remv, long long* __restrict__ keep,
                                                 unsigned long long* kept_s /*LDS, 1 word*/, int* nk_s /*LDS, 1 int*/) {
  const int tid = threadIdx.x, lane = tid & 63;
  for (int w = tid; w < nwords; w += THREADS) remv[w] = 0ull;
  __syncthreads();
  int nk = 0;  // meaningful in wave 0
  for (int b = 0; b < nwords; b++) {
    if (tid < 64) {
      const int i = b * 64 + lane;
      const unsigned long long diag = i < N ? mask[(size_t)i * nwords + b] : 0ull;
      unsigned long long r = remv[b];
      unsigned long long kept = 0ull;
      const int lim = min(64, N - b * 64);
      for (int j = 0; j < lim; j++) {
        if (!((r >> j) & 1ull)) {
          kept |= 1ull << j;
          r |= readlane64(diag, j);
        }
      }
      if ((kept >> lane) & 1ull) keep[nk + __popcll(kept & ((1ull << lane) - 1ull))] = (long long)order[i];
      nk += __popcll(kept);
      if (lane == 0) *kept_s = kept;
    }
    __syncthreads();
    const unsigned long long kept = *kept_s;
    for (int w = b + 1 + tid; w < nwords; w += THREADS) {
      unsigned long long acc = remv[w];
      unsigned long long bits = kept;
      while (bits) {
        const int j = __ffsll((long long)bits) - 1;
        bits &= bits - 1;
        acc |= mask[(size_t)(b * 64 + j) * nwords + w];
      }
      remv[w] = acc;
    }
    __syncthreads();
  }
  if (tid == 0) *nk_s = nk;
  __syncthreads();
  return *nk_s;
}

}  // namespace v3d
