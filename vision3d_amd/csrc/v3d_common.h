// v3d_common.h -- shared device helpers for libvision3d_hip (gfx950 only, wave64 throughout).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/vision3d_hip.h"

#define V3D_WAVE 64
#define V3D_BLOCK 256

#define V3D_CHECK_LAUNCH()                        \
  do {                                            \
    hipError_t e_ = hipGetLastError();            \
    if (e_ != hipSuccess) return (int)e_;         \
  } while (0)
#define V3D_CHECK_HIP(x)                          \
  do {                                            \
    hipError_t e_ = (x);                          \
    if (e_ != hipSuccess) return (int)e_;         \
  } while (0)

static inline int v3d_ceil_div(long long a, long long b) { return (int)((a + b - 1) / b); }
static inline size_t v3d_align(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }

// Byte fill as an ordinary KERNEL node.  hipMemsetAsync inside a captured graph becomes a memset node whose fill
// pattern was observed to be clobbered between replays on ROCm 7.2 when only blit copies ran in between (tables
// came back zero-filled instead of 0xFF: tools/_dbg_graph.py) -- a kernel's arguments live in the graph itself.
static __global__ __launch_bounds__(256) void v3d_fill_kernel(unsigned* __restrict__ p, size_t nwords, unsigned v) {
  const size_t tid = (size_t)blockIdx.x * 256 + threadIdx.x, nthreads = (size_t)gridDim.x * 256;
  if (((uintptr_t)p & 15) == 0) {
    const size_t nvec = nwords >> 2;
    uint4* p4 = reinterpret_cast<uint4*>(p);
    for (size_t i = tid; i < nvec; i += nthreads) p4[i] = make_uint4(v, v, v, v);
    for (size_t i = (nvec << 2) + tid; i < nwords; i += nthreads) p[i] = v;
  } else {
    for (size_t i = tid; i < nwords; i += nthreads) p[i] = v;
  }
}
static inline hipError_t v3d_fill_async(void* ptr, int byte, size_t bytes, hipStream_t st) {
  if (bytes == 0) return hipSuccess;
  if ((((uintptr_t)ptr) & 3) || (bytes & 3)) return hipMemsetAsync(ptr, byte, bytes, st);
  const unsigned b = (unsigned)byte & 0xFFu, v = b | (b << 8) | (b << 16) | (b << 24);
  const size_t nwords = bytes >> 2;
  size_t blocks = (nwords / 4 + 255) / 256;
  if (blocks < 1) blocks = 1;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(v3d_fill_kernel, dim3((unsigned)blocks), dim3(256), 0, st, (unsigned*)ptr, nwords, v);
  return hipGetLastError();
}

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) applies to the CURRENT device: remembered per device, not per process (a
// process that drives a second GPU must raise the limit there too).  A benign race at worst sets the attribute twice.
#define V3D_MAX_DEVICES 64
struct V3dPerDeviceFlag {
  bool done[V3D_MAX_DEVICES] = {};
};
static inline hipError_t v3d_set_max_lds(V3dPerDeviceFlag& f, const void* fn, int bytes) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  const bool tracked = dev >= 0 && dev < V3D_MAX_DEVICES;
  if (tracked && f.done[dev]) return hipSuccess;
  e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e == hipSuccess && tracked) f.done[dev] = true;
  return e;
}

// f16s arithmetic (csrc/spconv.hip "the split-precision product"): (x0, x1) -> packed f16 pairs hi = rne(x * s), lo = rne(x * s - hi)
// in FOUR instructions.  v_fma_mix{lo,hi}_f16 evaluate fma(a, b, c) on f32 / f16 sources chosen per operand and round ONCE to f16
// into one half of the destination: hi = fma(x, s, 0); lo = fma(x, s, -hi) with the f16 half of `hi` read in place -- x * s is exact
// (s is a power of two) and x * s - hi fits 14 bits, so both are the values of the plain expressions (checked bit for bit by
// tools/mb_f16split.hip), without the multiply, the two conversions back and the subtraction (8 VALU operations per pair; the
// bf16 split takes 6: there is no bf16 mix instruction).  `s` must be wave-uniform (an SGPR operand).
__device__ __forceinline__ void v3d_split_f16_pair(const float x0, const float x1, const float s, unsigned& hi, unsigned& lo) {
  unsigned h = 0u, l = 0u;
#if defined(__HIP_DEVICE_COMPILE__)  // (the host pass only parses the declaration)
  asm("v_fma_mixlo_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]" : "+v"(h) : "v"(x0), "s"(s));
  asm("v_fma_mixhi_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]" : "+v"(h) : "v"(x1), "s"(s));
  asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "+v"(l) : "v"(x0), "s"(s), "v"(h));
  asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l) : "v"(x1), "s"(s), "v"(h));
#endif
  hi = h;
  lo = l;
}

// f16s range bookkeeping, downward: "this frame's tensor holds a value of at least `floor`" (floor = the consumer's limit x 2^-12).
// Every lane of the wave calls this with its own largest magnitude; a wave that saw such a value sets the tensor's word to 1 with a
// PLAIN store behind a PLAIN load (the word is zero at the start of a frame; every writer stores the same 1; a stale zero only
// repeats the store) -- the word is read by a LATER launch (plan_quiet_check_kernel), i.e. behind the end-of-kernel write-back.
// Measured on the way (KITTI frame, one at a time / 4 in flight): exact maxima with atomicMax 0.68 ms / 2 840 frames/s -- the ~1 500
// waves of a launch all pass "larger than the word" at once and serialise on one address --; device-scope atomic load + store of the
// flag 0.60 ms / 2 850: every wave's tail then waits for a round trip past the XCD's L2; without any bookkeeping 0.468 / 3 770.
__device__ __forceinline__ void v3d_mark_seen(unsigned* __restrict__ word, const float vmax, const float floor) {
  if (__ballot(vmax >= floor) != 0ull && (threadIdx.x & 63) == 0) {
    unsigned cur;
    asm volatile("global_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(cur) : "v"(word) : "memory");
    if (cur == 0u) asm volatile("global_store_dword %0, %1, off" : : "v"(word), "v"(1u) : "memory");
  }
}

// per-device cache of an integer launch parameter (occupancy, CU count): a process may drive several GPUs
struct V3dPerDeviceInt {
  int v[V3D_MAX_DEVICES] = {};
  int* slot() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= V3D_MAX_DEVICES) dev = 0;
    return &v[dev];
  }
};
// compute units of the current device (cached per device); 0 on error
static inline int v3d_device_cu_count() {
  static V3dPerDeviceInt cache;
  int* c = cache.slot();
  if (!*c) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess) *c = n;
  }
  return *c;
}

// Bump allocator over a caller-provided workspace.
struct V3dArena {
  char* base;
  size_t size, off;
  V3dArena(void* p, size_t n) : base((char*)p), size(n), off(0) {}
  template <typename T>
  T* take(size_t count) {
    size_t bytes = v3d_align(count * sizeof(T));
    if (off + bytes > size) { off = size + 1; return nullptr; }
    T* r = (T*)(base + off);
    off += bytes;
    return r;
  }
  bool ok() const { return off <= size; }
};

// ------------------------------------------------------------------------------------------------
// Open-addressing hash table  key(u64) -> slot.  Per-slot payloads live in caller-owned arrays
// indexed by slot.  Capacity is a power of two >= 2x the number of keys.  EMPTY = all ones, so the
// whole table (and any 0xFF-initialised payload) is reset by ONE hipMemsetAsync.
// ------------------------------------------------------------------------------------------------
typedef unsigned long long v3d_key_t;
#define V3D_EMPTY_KEY 0xFFFFFFFFFFFFFFFFull

struct V3dHash {
  v3d_key_t* keys;
  unsigned mask;   // capacity - 1
  unsigned shift;  // 64 - log2(capacity): Fibonacci hashing keeps the TOP bits of key * phi
};

static inline V3dHash v3d_make_hash(v3d_key_t* keys, unsigned capacity) {
  unsigned lg = 0;
  while ((1u << lg) < capacity) lg++;
  return V3dHash{keys, capacity - 1, 64u - lg};
}

static inline unsigned v3d_hash_capacity(long long n_items) {
  unsigned cap = 1024;
  while ((long long)cap < 2 * n_items) cap <<= 1;
  return cap;
}

// Top bits of the 64-bit product (NOT a middle slice: on linear voxel indices the middle bits give ~12
// probes per insert, the top bits 1.2 -- measured on KITTI-shaped clouds).
__device__ __forceinline__ unsigned v3d_hash_start(v3d_key_t key, const V3dHash h) {
  return (unsigned)((key * 0x9E3779B97F4A7C15ull) >> h.shift) & h.mask;
}

// insert-or-find; returns the slot holding `key`, or -1 if the table is full (probing is bounded by
// the capacity so a mis-sized table can never spin forever)
__device__ __forceinline__ int v3d_hash_insert(const V3dHash h, v3d_key_t key) {
  unsigned s = v3d_hash_start(key, h);
  for (unsigned probes = 0; probes <= h.mask; probes++) {
    v3d_key_t prev = atomicCAS(&h.keys[s], V3D_EMPTY_KEY, key);
    if (prev == V3D_EMPTY_KEY || prev == key) return (int)s;
    s = (s + 1) & h.mask;
  }
  return -1;
}

// lookup in a table completed by an EARLIER kernel; returns slot or -1
__device__ __forceinline__ int v3d_hash_find(const V3dHash h, v3d_key_t key) {
  unsigned s = v3d_hash_start(key, h);
  for (unsigned probes = 0; probes <= h.mask; probes++) {
    v3d_key_t k = h.keys[s];
    if (k == key) return (int)s;
    if (k == V3D_EMPTY_KEY) return -1;
    s = (s + 1) & h.mask;
  }
  return -1;
}

// ------------------------------------------------------------------------------------------------
// SITE tables (the coordinate hashes of the rulebooks): the row index of a site rides in the low 24 bits of its key word,
//   word = key << 24 | row      (row = 0xFFFFFF: not numbered yet, or clipped by a capacity),
// so a neighbour look-up is ONE random access (key match and row in the same 8 bytes) instead of the key probe followed by a
// dependent read of vals[slot] -- the look-ups are the bulk of the rulebook kernels (27 per row; 2.8 M per launch on the
// Waymo-range sweep) and one memory round trip of every rulebook launch's dependent chain.  Keys are linear cell indices
// (< 2^40: batch x D x H x W), rows < 2^24 - 1 (checked where tables are sized).  EMPTY stays all ones; the per-slot payload
// arrays (vals, first_ticket) are kept for the users that reach a slot by index.
// ------------------------------------------------------------------------------------------------
#define V3D_SITE_ROW_BITS 24
#define V3D_SITE_NO_ROW 0xFFFFFFu
#define V3D_SITE_MAX_ROWS (int)(V3D_SITE_NO_ROW - 1)

// insert-or-find of `key` with a row (or V3D_SITE_NO_ROW); returns the slot, -1 if the table is full.  Concurrent inserts of
// the same key must pass the same row (in practice: all V3D_SITE_NO_ROW, or the key is inserted once).
__device__ __forceinline__ int v3d_site_insert(const V3dHash h, v3d_key_t key, unsigned row) {
  const v3d_key_t word = (key << V3D_SITE_ROW_BITS) | row;
  unsigned s = v3d_hash_start(key, h);
  for (unsigned probes = 0; probes <= h.mask; probes++) {
    const v3d_key_t prev = atomicCAS(&h.keys[s], V3D_EMPTY_KEY, word);
    if (prev == V3D_EMPTY_KEY || (prev >> V3D_SITE_ROW_BITS) == key) return (int)s;
    s = (s + 1) & h.mask;
  }
  return -1;
}

// the row of a site in a table completed by an EARLIER kernel (or numbered by v3d_site_set_row), -1 if absent / not numbered
__device__ __forceinline__ int v3d_site_find_row_from(const V3dHash h, v3d_key_t key, unsigned s) {
  for (unsigned probes = 0; probes <= h.mask; probes++) {
    const v3d_key_t w = h.keys[s];
    if ((w >> V3D_SITE_ROW_BITS) == key) {
      const unsigned row = (unsigned)w & V3D_SITE_NO_ROW;
      return row == V3D_SITE_NO_ROW ? -1 : (int)row;
    }
    if (w == V3D_EMPTY_KEY) return -1;
    s = (s + 1) & h.mask;
  }
  return -1;
}

__device__ __forceinline__ int v3d_site_find_row(const V3dHash h, v3d_key_t key) {
  return v3d_site_find_row_from(h, key, v3d_hash_start(key, h));
}

// number the site in slot `s` (its only writer at this point: no insert runs concurrently)
__device__ __forceinline__ void v3d_site_set_row(const V3dHash h, int s, v3d_key_t key, unsigned row) {
  h.keys[s] = (key << V3D_SITE_ROW_BITS) | row;
}

// ------------------------------------------------------------------------------------------------
// Wave64 ballot + popcount compaction.  One flag per thread, 256-thread blocks.
// Returns the exclusive rank of this thread's flag inside the block; `total` = flags set in the block.
// `lds` must hold >= 4 ints and is free for reuse on return.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int v3d_block_rank(bool flag, int& total, int* lds) {
  const unsigned long long m = __ballot(flag);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int r = __popcll(m & ((1ull << lane) - 1ull));
  if (lane == 0) lds[w] = __popcll(m);
  __syncthreads();
  int off = 0, tot = 0;
#pragma unroll
  for (int i = 0; i < V3D_BLOCK / V3D_WAVE; i++) {
    const int c = lds[i];
    if (i < w) off += c;
    tot += c;
  }
  __syncthreads();
  total = tot;
  return off + r;
}

// count -> scan in ONE launch without fences: counts[] is pre-set to -1 (part of the frame's 0xFF fill); every block
// PUBLISHES its count with an agent-scope atomic store, and the block with the highest index -- dispatched after all
// the others, so everything it waits for is already running or done -- reads them with agent-scope loads, spinning
// on entries that are still -1, and scans.  (A fence + arrival-counter variant cost 15 us per launch in L2
// write-backs; this one costs the same as the plain count kernel and saves the dependent scan launch.)
__device__ __forceinline__ void v3d_publish_count(int* slot, int value) {
  __hip_atomic_store(slot, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ int v3d_load_coherent(const int* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ int v3d_wait_count(const int* slot) {
  int v = v3d_load_coherent(slot);
  while (v < 0) {
    __builtin_amdgcn_s_sleep(1);
    v = v3d_load_coherent(slot);
  }
  return v;
}

// In-place exclusive scan of the published counts[0..n) by ONE 256-thread block (the highest-index block of a count
// kernel); returns the grand total to every thread.  `lds` >= 8 ints.
__device__ __forceinline__ int v3d_block_exclusive_scan_global(int* counts, int n, int* lds) {
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  int carry = 0;
  for (int c0 = 0; c0 < n; c0 += V3D_BLOCK) {
    const int idx = c0 + tid;
    const int v = idx < n ? v3d_wait_count(counts + idx) : 0;
    int incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int t = __shfl_up(incl, off);
      if (lane >= off) incl += t;
    }
    if (lane == 63) lds[w] = incl;
    __syncthreads();
    int woff = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < V3D_BLOCK / V3D_WAVE; i++) {
      const int c = lds[i];
      if (i < w) woff += c;
      tot += c;
    }
    if (idx < n) v3d_publish_count(counts + idx, carry + woff + incl - v);
    carry += tot;
    __syncthreads();
  }
  return carry;
}

// Items handled by one block of the chunked flag scans (256 threads x 8 rounds).
#define V3D_SCAN_CHUNK 2048
