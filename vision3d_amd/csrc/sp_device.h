// sp_device.h -- device helpers shared by the packed sparse-convolution kernels (spconv.hip, brick.hip): the split-precision
// product (bf16x3 / f16s pieces, three MFMA terms), scale entries, hand-issued loads with counted waits, LDS-DMA.
#pragma once
#include "v3d_internal.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned bf16_rne_bits(float f) {
  unsigned u = __float_as_uint(f);
  if ((u & 0x7F800000u) == 0x7F800000u) return u >> 16;
  return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
}
// ---- the split-precision product of the packed kernels, two arithmetics (template parameter PREC of every kernel below) ----
// Both evaluate  a * w = al*Wh + ah*Wl + ah*Wh  (three MFMA terms, fp32 accumulation, smallest terms first) on operands split into
// hi = rne(x), lo = rne(x - hi); they differ in the 16-bit format of the pieces:
//   PREC 0 "bf16x3"  bf16 pieces: 8 + 8 significant bits, 2^-17 per product, any fp32 magnitude (no scale to choose): the arithmetic
//                    of rounds 1-4, kept for the training plan (gradients span too many binades for a per-tensor scale) and as
//                    the library's `fast` inference mode.  Strict elementwise error of a SECOND layer against float64 on entries
//                    above 1e-3 of the layer maximum: 1.1e-3 ... 2.1e-3 (torch's fp32 conv3d: 2e-5 ... 1.1e-4).
//   PREC 1 "f16s"    f16 pieces of x * s with a power-of-two scale s per tensor: 11 + 11 significant bits, 2^-22 per product --
//                    the error of a 1 728-term dot product is then fp32's own accumulation noise (tools/mb_f16split.hip on MI355X:
//                    strict relative error max 1.0e-4 / rms 2.2e-6 against 1.7e-4 / 2.5e-6 for the exact-fp32 MFMA and
//                    2.1e-3 / 5.5e-5 for bf16x3), at the SAME three MFMAs.  v_mfma_f32_16x16x32_f16 keeps subnormal f16 inputs
//                    (probed), so a piece below 2^-14 degrades to the 2^-24 quantum instead of vanishing: with the tensor's
//                    maximum scaled to 2^8 ... 2^14 everything down to 2^-17 of the maximum keeps full precision.
//                    The scales: activations -- V3dActScale: {s, 1/s, limit} in device memory, chosen by the caller from the
//                    observed maximum of the tensor with headroom (runtime.py: calibration); an output beyond the CONSUMER's
//                    limit raises a device flag (the frame is then re-run after recalibration, like a capacity overflow) --;
//                    weights -- per layer from max|W| at pack time, its inverse in the image's trailer.  Scaling by powers of
//                    two is exact, so the result does not depend on the scales as long as nothing leaves the f16 range.
typedef _Float16 spr_f16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 spr_f16x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 spr_bf16x2_t __attribute__((ext_vector_type(2)));
typedef float spr_f32x2_t __attribute__((ext_vector_type(2)));

template <int PREC>
__device__ __forceinline__ f32x4 sp_mfma(const u32x4_t a, const u32x4_t b, const f32x4 c) {
#ifdef SP_EXP_F16S_BF16_MFMA  // experiment only (wrong results): the f16s kernels on the bf16 instruction -- is the f16 MFMA itself slower?
  if constexpr (true)
#else
  if constexpr (PREC == 0)
#endif
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(spr_f16x8_t, a), __builtin_bit_cast(spr_f16x8_t, b), c, 0, 0, 0);
}

// 8 fp32 activations -> packed hi / lo fragments (both RNE on the hardware converters: v_cvt_pk_bf16_f32 / v_cvt_pk_f16_f32).
// PREC 1: of x * s (s: the tensor's power-of-two scale, exact).
template <int PREC>
__device__ __forceinline__ void split_act(const float (&x)[8], const float s, u32x4_t& hi, u32x4_t& lo) {
#pragma unroll
  for (int i = 0; i < 4; i++) {
    if constexpr (PREC == 0) {
      const spr_bf16x2_t hh = __builtin_convertvector(spr_f32x2_t{x[2 * i], x[2 * i + 1]}, spr_bf16x2_t);
      const unsigned hb = __builtin_bit_cast(unsigned, hh);
      const float r0 = x[2 * i] - __uint_as_float(hb << 16), r1 = x[2 * i + 1] - __uint_as_float(hb & 0xFFFF0000u);
      hi[i] = hb;
      lo[i] = __builtin_bit_cast(unsigned, __builtin_convertvector(spr_f32x2_t{r0, r1}, spr_bf16x2_t));
#ifdef SP_EXP_F16S_BF16_SPLIT  // experiment only (wrong results): the f16s kernels with the bf16 split's instructions
    } else if constexpr (true) {
      const spr_bf16x2_t hh = __builtin_convertvector(spr_f32x2_t{x[2 * i], x[2 * i + 1]}, spr_bf16x2_t);
      const unsigned hb = __builtin_bit_cast(unsigned, hh);
      const float r0 = x[2 * i] - __uint_as_float(hb << 16), r1 = x[2 * i + 1] - __uint_as_float(hb & 0xFFFF0000u);
      hi[i] = hb;
      lo[i] = __builtin_bit_cast(unsigned, __builtin_convertvector(spr_f32x2_t{r0, r1}, spr_bf16x2_t));
#endif
    } else {
      unsigned h, l;  // (four mixed-precision fmas per pair: v3d_common.h)
      v3d_split_f16_pair(x[2 * i], x[2 * i + 1], __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, s))), h, l);
      hi[i] = h;
      lo[i] = l;
    }
  }
}

// INS = 1: the gathered rows are ALREADY split (row = [hi: C x 16 bit | lo: C x 16 bit], the bytes of the fp32 row): the producing layer
// wrote them that way under THIS layer's scale entry (sp_store_split below), so the lane's two 16-byte loads return its hi and lo
// fragments directly and the main loop has no conversion work at all.  The 8 dwords travel in the same `float[8]` registers the
// fp32 path uses: [0..3] = hi, [4..7] = lo.
template <int PREC, int INS>
__device__ __forceinline__ void split_in(const float (&x)[8], const float s, u32x4_t& hi, u32x4_t& lo) {
  if constexpr (INS) {
    hi = u32x4_t{__float_as_uint(x[0]), __float_as_uint(x[1]), __float_as_uint(x[2]), __float_as_uint(x[3])};
    lo = u32x4_t{__float_as_uint(x[4]), __float_as_uint(x[5]), __float_as_uint(x[6]), __float_as_uint(x[7])};
  } else {
    split_act<PREC>(x, s, hi, lo);
  }
}

// (v0, v1) -> packed hi pair and lo pair, PREC 1: of v * s
template <int PREC>
__device__ __forceinline__ void sp_split_pair(const float v0, const float v1, const float s, unsigned& hi, unsigned& lo) {
  if constexpr (PREC == 0) {
    const spr_bf16x2_t hh = __builtin_convertvector(spr_f32x2_t{v0, v1}, spr_bf16x2_t);
    hi = __builtin_bit_cast(unsigned, hh);
    const float r0 = v0 - __uint_as_float(hi << 16), r1 = v1 - __uint_as_float(hi & 0xFFFF0000u);
    lo = __builtin_bit_cast(unsigned, __builtin_convertvector(spr_f32x2_t{r0, r1}, spr_bf16x2_t));
  } else {
    v3d_split_f16_pair(v0, v1, __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, s))), hi, lo);
  }
}

// Epilogue, second copy of an output row for the NEXT packed layer: the row split into pieces (of v * s: the consumer's scale entry).
// A lane holds D[row][col = j * 16 + r]; lanes r and r ^ 1 hold neighbouring columns of the same row: they exchange their values
// (one DPP move), both split the pair, the even lane stores the hi pair and the odd lane the lo pair -- one 4-byte store per
// value and lane, as many store instructions as the fp32 row takes.  Every lane of a pair must call this (rows are uniform
// across the 16 lanes that share (kg, rr): the callers' `row < n` test keeps pairs together).
template <int PREC, int COUT>
__device__ __forceinline__ void sp_store_split(unsigned short* __restrict__ out_s, const int row, const int col, const int r, const float v,
                                               const float s) {
  const float pv = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1 /*quad_perm [1,0,3,2]*/, 0xF, 0xF, true));
  unsigned h, l;
  sp_split_pair<PREC>((r & 1) ? pv : v, (r & 1) ? v : pv, s, h, l);
  *reinterpret_cast<unsigned*>(out_s + (size_t)row * (2 * COUT) + ((r & 1) ? COUT : 0) + (col & ~1)) = (r & 1) ? l : h;
}

// ---- epilogue stores of a 16-row output tile, COALESCED through LDS -------------------------------------------------------------
// Measured in round 6 (profiles/r06_brick_ablation.txt): storing a tile straight from the MFMA accumulator layout -- a lane holds
// D[row = kg * 4 + rr][col = j * 16 + r], so every store instruction of the wave writes eight 32-byte (split rows) or four 64-byte
// (fp32 rows) pieces of four different rows -- costs ~300 clocks per store instruction: 11.5 us of a 45 us 64 -> 64 launch at 56 k
// rows went into the 128 store instructions per lane of its epilogue.  The rows of a tile are CONSECUTIVE rows of the output array,
// i.e. one contiguous block of 16 x ROWB bytes: the wave writes its values into a private LDS staging block (4-byte ds_writes, row
// stride ROWB + 16 so that the four kg row groups fall on different banks) and stores the block with 16-byte stores of consecutive
// lanes -- ROWB / 64 fully coalesced store instructions per tile instead of 4 * NB.  Same bytes in memory.
//   stage: LDS scratch PRIVATE to the wave, SP_STAGE_BYTES(COUT) bytes, 16-byte aligned (no barrier: a wave's LDS operations
//          execute in order); v[j][rr]: the finished values; nv: rows of the tile that exist (rows >= nv are not stored).
#define SP_STAGE_STRIDE(COUT) ((COUT) * 4 + 16)
#define SP_STAGE_BYTES(COUT) (16 * SP_STAGE_STRIDE(COUT))
// NT = threads that flush together (64: a wave's private block; a workgroup's size: a block shared behind a barrier), t = the caller's index in them
template <int COUT, int NT = 64>
__device__ __forceinline__ void sp_stage_flush(const unsigned char* stage, unsigned char* __restrict__ dst_tile, const int nv, const int t) {
  constexpr int ROWB = COUT * 4, CPR = ROWB / 16;  // bytes and 16-byte chunks per row
#pragma unroll
  for (int i = 0; i < (16 * CPR + NT - 1) / NT; i++) {
    const int ch = i * NT + t, row = ch / CPR, wc = ch % CPR;
    if (ch < 16 * CPR && row < nv)
      *reinterpret_cast<uint4*>(dst_tile + (size_t)row * ROWB + wc * 16) = *reinterpret_cast<const uint4*>(stage + row * SP_STAGE_STRIDE(COUT) + wc * 16);
  }
}
// one value D[rowi][col] into a staging block: as fp32 / as its half of a split pair (lanes r, r ^ 1 hold neighbouring columns of the
// same row: every lane of the pair must call this)
template <int COUT>
__device__ __forceinline__ void sp_stage_put_f32(unsigned char* stage, const int rowi, const int col, const float v) {
  *reinterpret_cast<float*>(stage + rowi * SP_STAGE_STRIDE(COUT) + col * 4) = v;
}
template <int PREC, int COUT>
__device__ __forceinline__ void sp_stage_put_split(unsigned char* stage, const int rowi, const int col, const int r, const float v, const float s) {
  const float pv = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1 /*quad_perm [1,0,3,2]*/, 0xF, 0xF, true));
  unsigned h, l;
  sp_split_pair<PREC>((r & 1) ? pv : v, (r & 1) ? v : pv, s, h, l);
  *reinterpret_cast<unsigned*>(stage + rowi * SP_STAGE_STRIDE(COUT) + ((r & 1) ? COUT * 2 : 0) + (col & ~1) * 2) = (r & 1) ? l : h;
}
// fp32 rows (cap, COUT): out_tile = first row of the tile
template <int COUT, int NB>
__device__ __forceinline__ void sp_tile_store_f32(unsigned char* stage, float* __restrict__ out_tile, const float (&v)[NB][4], const int nv,
                                                  const int lane) {
  const int r = lane & 15, kg = lane >> 4;
#pragma unroll
  for (int j = 0; j < NB; j++)
#pragma unroll
    for (int rr = 0; rr < 4; rr++) sp_stage_put_f32<COUT>(stage, kg * 4 + rr, j * 16 + r, v[j][rr]);
  sp_stage_flush<COUT>(stage, reinterpret_cast<unsigned char*>(out_tile), nv, lane);
}
// split rows (cap, 2 * COUT) 16-bit = [hi: COUT | lo: COUT] under the consumer's scale `s` (sp_store_split's bytes)
template <int PREC, int COUT, int NB>
__device__ __forceinline__ void sp_tile_store_split(unsigned char* stage, unsigned short* __restrict__ out_s_tile, const float (&v)[NB][4],
                                                    const float s, const int nv, const int lane) {
  const int r = lane & 15, kg = lane >> 4;
#pragma unroll
  for (int j = 0; j < NB; j++)
#pragma unroll
    for (int rr = 0; rr < 4; rr++) sp_stage_put_split<PREC, COUT>(stage, kg * 4 + rr, j * 16 + r, r, v[j][rr], s);
  sp_stage_flush<COUT>(stage, reinterpret_cast<unsigned char*>(out_s_tile), nv, lane);
}

// one value -> (hi, lo) 16-bit patterns, PREC 1: of v * s
template <int PREC>
__device__ __forceinline__ void split_one(const float v, const float s, unsigned short& hi, unsigned short& lo) {
  if constexpr (PREC == 0) {
    const unsigned h = bf16_rne_bits(v);
    hi = (unsigned short)h;
    lo = (unsigned short)bf16_rne_bits(v - __uint_as_float(h << 16));
  } else {
    const float a = v * s;
    const _Float16 h = (_Float16)a;
    const _Float16 l = (_Float16)(a - (float)h);
    hi = __builtin_bit_cast(unsigned short, h);
    lo = __builtin_bit_cast(unsigned short, l);
  }
}

// PREC 1: scale of the input rows, the factor that undoes input and weight scales, and the consumer's limit on this launch's output
struct SpScales {
  float s_in, undo, limit;
};
// trailer of a packed weight image (all precisions allocate it; PREC 1 fills it): {max|W| bits, 1/s_w, s_w, precision}
#define V3D_WIMG_TRAILER 256
template <int PREC>
__device__ __forceinline__ SpScales sp_scales(const V3dActScale& as, const unsigned short* wimg, size_t img_elems) {
  SpScales r{1.f, 1.f, 3.0e38f};
  if constexpr (PREC == 1) {
    r.s_in = as.in[0];
    r.undo = as.in[1] * (as.w_inv ? *as.w_inv : reinterpret_cast<const float*>(wimg + img_elems)[1]);
    if (as.next) r.limit = as.next[2];
  }
  return r;
}
// an output beyond the consumer's limit: the frame's summary flag (<= 0: fine, 1: a capacity was hit, 2: out of f16s range)
// (every lane of the wave calls this, at the end of the kernel)
__device__ __forceinline__ void sp_range_check(const V3dActScale& as, const float vmax, const float limit) {
  if (as.flag && vmax > limit) atomicMax(as.flag, V3D_FLAG_RANGE);
  if (as.seen) v3d_mark_seen(as.seen, vmax, limit * (1.f / (float)(1 << V3D_QUIET_BITS)));
}

static __device__ __attribute__((aligned(256))) const float spr_zero_row[128] = {};

__device__ __forceinline__ void asm_gld16(f32x4& d, const void* p) { asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(d) : "v"(p) : "memory"); }
__device__ __forceinline__ void asm_gld16_16(f32x4& d, const void* p) { asm volatile("global_load_dwordx4 %0, %1, off offset:16" : "=v"(d) : "v"(p) : "memory"); }
__device__ __forceinline__ void asm_gld16_128(f32x4& d, const void* p) { asm volatile("global_load_dwordx4 %0, %1, off offset:128" : "=v"(d) : "v"(p) : "memory"); }
__device__ __forceinline__ void asm_gld16_144(f32x4& d, const void* p) { asm volatile("global_load_dwordx4 %0, %1, off offset:144" : "=v"(d) : "v"(p) : "memory"); }
__device__ __forceinline__ void asm_gld4(int& d, const void* p) { asm volatile("global_load_dword %0, %1, off" : "=v"(d) : "v"(p) : "memory"); }
template <int N> __device__ __forceinline__ void vm_wait_tie(f32x4 (&a)[2]) { asm volatile("s_waitcnt vmcnt(%2)" : "+v"(a[0]), "+v"(a[1]) : "n"(N) : "memory"); }
template <int N> __device__ __forceinline__ void vm_wait_tie(f32x4 (&a)[4]) { asm volatile("s_waitcnt vmcnt(%4)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]) : "n"(N) : "memory"); }
template <int N> __device__ __forceinline__ void vm_wait_tie(f32x4 (&a)[8]) {
  asm volatile("s_waitcnt vmcnt(%8)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "n"(N) : "memory");
}
template <int N> __device__ __forceinline__ void vm_wait_tie(f32x4 (&a)[1]) { asm volatile("s_waitcnt vmcnt(%1)" : "+v"(a[0]) : "n"(N) : "memory"); }
// (the same, additionally ordered behind the instruction that produces `after`: e.g. the last MFMA of a step)
template <int N> __device__ __forceinline__ void vm_wait_tie_after(f32x4 (&a)[1], f32x4& after) { asm volatile("s_waitcnt vmcnt(%2)" : "+v"(a[0]), "+v"(after) : "n"(N) : "memory"); }
template <int N> __device__ __forceinline__ void vm_wait_tie_after(f32x4 (&a)[2], f32x4& after) { asm volatile("s_waitcnt vmcnt(%3)" : "+v"(a[0]), "+v"(a[1]), "+v"(after) : "n"(N) : "memory"); }
template <int N> __device__ __forceinline__ void vm_wait_tie(int (&a)[1]) { asm volatile("s_waitcnt vmcnt(%1)" : "+v"(a[0]) : "n"(N) : "memory"); }
template <int N> __device__ __forceinline__ void vm_wait_tie(int (&a)[2]) { asm volatile("s_waitcnt vmcnt(%2)" : "+v"(a[0]), "+v"(a[1]) : "n"(N) : "memory"); }
template <int N> __device__ __forceinline__ void vm_wait_tie(int (&a)[3]) { asm volatile("s_waitcnt vmcnt(%3)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]) : "n"(N) : "memory"); }

__device__ __forceinline__ void asm_gld16_64(f32x4& d, const void* p) { asm volatile("global_load_dwordx4 %0, %1, off offset:64" : "=v"(d) : "v"(p) : "memory"); }
__device__ __forceinline__ void asm_gld16_192(f32x4& d, const void* p) { asm volatile("global_load_dwordx4 %0, %1, off offset:192" : "=v"(d) : "v"(p) : "memory"); }
// LDS-DMA of one 16-byte piece per lane: global address per lane, LDS destination = wave-uniform `lds_dst` + lane * 16 (M0 is
// compiler-reserved: saved and restored inside the statement).  No register destination: completion = the wave's vmcnt.
__device__ __forceinline__ void asm_dma16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
template <int N> __device__ __forceinline__ void vm_wait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// LDS byte address of a __shared__ object: the low half of its generic address (flat aperture base in the high half)
__device__ __forceinline__ unsigned lds_addr_of(const void* p) { return (unsigned)(unsigned long long)p; }
