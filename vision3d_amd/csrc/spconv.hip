// spconv.hip -- sparse 3-D convolution forward (T3, spconv ops.indice_conv) and .dense() (T2) on gfx950.
//
// Reference path: spconv gathers rows per kernel offset, runs one cuBLAS GEMM per offset and
// scatter-adds the result (~80 tiny launches per layer, atomics, R*(Cin+Cout) traffic).  Here ONE
// launch per layer computes   out[o,:] = act((sum_k in[nbr[k][o],:] @ W[k]) * scale + shift)
// output-stationary: every output row is produced by exactly one workgroup, written once, in a fixed
// summation order (deterministic, no atomics); BatchNorm(eval)+ReLU are fused into the epilogue.
//
// Kernels: `spconv_fwd_rows*` (algo 4: packed bf16x3 product, the default for Cin >= 16), `spconv_fwd_wave` (algo 3: exact
// fp32 MFMA, the 4-channel input layer), `spconv_fwd_scalar` (algo 1: the plain VALU statement of the same sum, the
// on-device cross-check and the fallback for channel counts the MFMA tilings do not cover).  Earlier variants that lost
// their measurements (an LDS-staged fp32 kernel "algo 2", register-tile variants of algo 4, a two-offsets-per-round ring)
// live in the history of this file, not in the library.
#include "v3d_internal.h"

#include "sp_device.h"
#include "rb_device.h"

// A rulebook scan riding at the front of a sparse layer's grid (rb_device.h RbScanJob): the first rider.blocks workgroups run it on
// their first V3D_BLOCK threads and `lds` (>= RB_SCAN_LDS bytes of the kernel's LDS), the others are the layer's own, renumbered.
#define SP_RIDER_PROLOGUE(lds)                                                                        \
  if (rider.blocks && (int)blockIdx.x < rider.blocks) {                                               \
    if (threadIdx.x < V3D_BLOCK) rb_rider_run(rider, (int)blockIdx.x, (unsigned char*)(lds));         \
    return;                                                                                           \
  }                                                                                                   \
  const int bid = (int)blockIdx.x - rider.blocks;                                                     \
  [[maybe_unused]] const int nblk = (int)gridDim.x - rider.blocks;


// The library has no process-global state: kernel variants are chosen from the arguments of each call (rows_hint; a
// NEGATIVE rows_hint forces a variant, for tests and benchmarks -- see v3d_sparse_conv_fwd_packed in the header).

// ---------------------------------------------------------------------------------- algo 1: scalar
__global__ __launch_bounds__(V3D_BLOCK) void spconv_fwd_scalar(const float* __restrict__ in,
                                                               const float* __restrict__ W,
                                                               const int* __restrict__ nbr,
                                                               const int* __restrict__ n_ptr, int cap, int K, int Cin,
                                                               int Cout, const float* __restrict__ scale,
                                                               const float* __restrict__ shift, int relu,
                                                               float* __restrict__ out) {
  const int n = min(*n_ptr, cap);
  const long long total = (long long)n * Cout;
  for (long long t = (long long)blockIdx.x * V3D_BLOCK + threadIdx.x; t < total; t += (long long)gridDim.x * V3D_BLOCK) {
    const int o = (int)(t / Cout), c = (int)(t % Cout);
    float acc = 0.f;
    for (int k = 0; k < K; k++) {
      const int i = nbr[(size_t)k * cap + o];
      if (i < 0) continue;
      const float* x = in + (size_t)i * Cin;
      const float* w = W + (size_t)k * Cin * Cout + c;
      float part = 0.f;
      for (int ci = 0; ci < Cin; ci++) part = fmaf(x[ci], w[(size_t)ci * Cout], part);
      acc += part;
    }
    if (scale) acc = acc * scale[c] + shift[c];
    if (relu) acc = fmaxf(acc, 0.f);
    out[t] = acc;
  }
}

#define SPC_TM 64  // output rows per workgroup of spconv_fwd_wave == wave width (ballot compaction)

// ---------------------------------------------------------------------------------- algo 3: wave-autonomous MFMA
// One workgroup = one 64-row output tile, 8 waves.  Wave (g, nb) owns the 16 output columns
// [16 nb, 16 nb + 16) and the kernel offsets of k-group g (NB * G = 8, NB = Cout/16): inside the main
// loop a wave never synchronises with another wave -- no barrier, no shared operand staging:
//   * A (gathered input rows) and B (weights) go straight global -> VGPR.  The MFMA reduction index is
//     PERMUTED so that lane (r = lane&15, q = lane>>4) holds Cin/4 CONTIGUOUS floats of row r
//     (A[r][q*T + t], T = Cin/4: whole float4 loads, the four q-lanes cover the row's line) and B
//     lane (q, j) holds W[k][q*T + t][16 nb + j]; step t of the chain multiplies the matching slices.
//     Any permutation of the reduction index is legal as long as A and B agree.
//   * the next (offset, 16-row block) operands are in flight while the current block's MFMAs run
//     (two-deep register pipeline).
//   * results are added into the wave's private slice of the LDS accumulator acc[g][row][col]
//     (ds_add_f32; one owner per element -> deterministic); the epilogue sums the G partials in
//     fixed order, applies scale/shift/ReLU and stores coalesced float4.
#define SPW_WAVES 8

template <int CIN, int COUT>
__global__ __launch_bounds__(SPW_WAVES* V3D_WAVE) void spconv_fwd_wave(const float* __restrict__ in,
                                                                       const float* __restrict__ W,
                                                                       const int* __restrict__ nbr,
                                                                       const int* __restrict__ n_ptr, int cap, int K,
                                                                       const float* __restrict__ scale,
                                                                       const float* __restrict__ shift, int relu,
                                                                       float* __restrict__ out, const float* __restrict__ next_entry,
                                                                       int* __restrict__ range_flag, unsigned* __restrict__ seen,
                                                                       const RbScanJob rider) {
  constexpr int NB = COUT / 16;
  constexpr int G = SPW_WAVES / NB;
  constexpr int T = CIN / 4;  // MFMA steps; also floats of one row held per lane
  constexpr int NT = SPW_WAVES * V3D_WAVE;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* acc = smem;                                      // [G][64][COUT]
  int* list_in = (int*)(acc + G * SPC_TM * COUT);         // [K][64]
  int* cnt_pad = list_in + K * SPC_TM;                    // [K]
  unsigned char* list_row = (unsigned char*)(cnt_pad + K);  // [K][64]
  SP_RIDER_PROLOGUE(smem)

  const int n = min(*n_ptr, cap);
  const int row0 = bid * SPC_TM;
  if (row0 >= n) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

  for (int i = tid; i < G * SPC_TM * COUT; i += NT) acc[i] = 0.f;
  for (int k = wave; k < K; k += SPW_WAVES) {
    const int row = row0 + lane;
    const int v = row < n ? nbr[(size_t)k * cap + row] : -1;
    const unsigned long long m = __ballot(v >= 0);
    const int c = __popcll(m);
    const int cp = (c + 15) & ~15;
    const int pos = __popcll(m & ((1ull << lane) - 1ull));
    if (v >= 0) {
      list_in[k * SPC_TM + pos] = v;
      list_row[k * SPC_TM + pos] = (unsigned char)lane;
    }
    if (lane >= c && lane < cp) {
      list_in[k * SPC_TM + lane] = -1;
      list_row[k * SPC_TM + lane] = 255;
    }
    if (lane == 0) cnt_pad[k] = cp;
  }
  __syncthreads();

  // Everything that steers the main loop is made wave-uniform (SGPR) on purpose: the wave id through
  // readfirstlane, the per-offset counts through one register per lane + readlane.  The loop then runs
  // on scalar branches instead of exec-mask juggling, and needs no LDS traffic for its control.
  const int uwave = __builtin_amdgcn_readfirstlane(wave);
  const int g = uwave / NB, nb = uwave % NB;
  const int kper = (K + G - 1) / G;
  const int k_lo = g * kper, k_hi = min(K, k_lo + kper);
  const int r = lane & 15, q = lane >> 4;
  float* my_acc = acc + (size_t)g * SPC_TM * COUT + nb * 16 + r;
  const int cnt_reg = lane < K ? cnt_pad[lane] : 0;  // K <= 64: lane k holds cnt_pad[k]
  auto count_of = [&](int k) { return __builtin_amdgcn_readlane(cnt_reg, k); };

  float a0[T], a1[T], b0[T], b1[T];
  auto load_a = [&](int k, int rblk, float (&a)[T]) {
    const int src = list_in[k * SPC_TM + rblk * 16 + r];
    if (src >= 0) {
      const float* p = in + (size_t)src * CIN + q * T;
      if constexpr (T % 4 == 0) {
#pragma unroll
        for (int i = 0; i < T / 4; i++) {
          const float4 v = reinterpret_cast<const float4*>(p)[i];
          a[4 * i] = v.x;
          a[4 * i + 1] = v.y;
          a[4 * i + 2] = v.z;
          a[4 * i + 3] = v.w;
        }
      } else {
#pragma unroll
        for (int t = 0; t < T; t++) a[t] = p[t];
      }
    } else {
#pragma unroll
      for (int t = 0; t < T; t++) a[t] = 0.f;
    }
  };
  auto load_b = [&](int k, float (&b)[T]) {
    const float* p = W + ((size_t)k * CIN + q * T) * COUT + nb * 16 + r;
#pragma unroll
    for (int t = 0; t < T; t++) b[t] = p[(size_t)t * COUT];
  };
  // next non-empty (k, rblk) after (k, rblk); k == k_hi means "done".  Scalar code only.
  auto advance = [&](int& k, int& rblk) {
    rblk++;
    if (rblk * 16 < count_of(k)) return;
    rblk = 0;
    do {
      k++;
    } while (k < k_hi && count_of(k) == 0);
  };

  int k = k_lo, rblk = 0;
  while (k < k_hi && count_of(k) == 0) k++;
  if (k < k_hi) {
    load_a(k, rblk, a0);
    load_b(k, b0);
  }
  while (k < k_hi) {
    int kn = k, rn = rblk;
    advance(kn, rn);
    const bool more = kn < k_hi, newk = kn != k;
    // tile rows of this block's 4 result rows: 4 bytes, one aligned 32-bit LDS read
    const unsigned rows4 = *reinterpret_cast<const unsigned*>(list_row + k * SPC_TM + rblk * 16 + q * 4);
    if (more) {  // operands of the next block in flight during this block's MFMAs
      load_a(kn, rn, a1);
      if (newk) load_b(kn, b1);
    }
    f32x4 d0 = {0.f, 0.f, 0.f, 0.f}, d1 = {0.f, 0.f, 0.f, 0.f};
    if constexpr (T >= 2) {
#pragma unroll
      for (int t = 0; t < T; t += 2) {  // two independent accumulator chains hide the MFMA latency
        d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[t], b0[t], d0, 0, 0, 0);
        d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[t + 1], b0[t + 1], d1, 0, 0, 0);
      }
      d0 += d1;
    } else {
      d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[0], b0[0], d0, 0, 0, 0);
    }
    // accumulate: this wave is the only owner of acc[g][:, 16 nb .. 16 nb + 16) and a tile row occurs
    // at most once per offset -> plain read-modify-write (LDS float atomics are ~0.4 us each here)
    {
      float old[4];
#pragma unroll
      for (int rr = 0; rr < 4; rr++) {
        const unsigned trow = (rows4 >> (8 * rr)) & 255u;
        old[rr] = trow != 255u ? my_acc[trow * COUT] : 0.f;
      }
#pragma unroll
      for (int rr = 0; rr < 4; rr++) {
        const unsigned trow = (rows4 >> (8 * rr)) & 255u;
        if (trow != 255u) my_acc[trow * COUT] = old[rr] + d0[rr];
      }
    }
    if (more) {
#pragma unroll
      for (int t = 0; t < T; t++) a0[t] = a1[t];
      if (newk) {
#pragma unroll
        for (int t = 0; t < T; t++) b0[t] = b1[t];
      }
    }
    k = kn;
    rblk = rn;
  }
  __syncthreads();

  float vmax = 0.f;  // (an f16s plan: this exact layer feeds a scaled one -- its output is checked against that tensor's limit)
  for (int idx = tid; idx < SPC_TM * (COUT / 4); idx += NT) {
    const int rw = idx / (COUT / 4), c4 = idx % (COUT / 4);
    if (row0 + rw >= n) continue;
    float4 v = *reinterpret_cast<const float4*>(acc + rw * COUT + c4 * 4);
#pragma unroll
    for (int gg = 1; gg < G; gg++) {
      const float4 u = *reinterpret_cast<const float4*>(acc + ((size_t)gg * SPC_TM + rw) * COUT + c4 * 4);
      v.x += u.x;
      v.y += u.y;
      v.z += u.z;
      v.w += u.w;
    }
    if (scale) {
      const float4 sc = *reinterpret_cast<const float4*>(scale + c4 * 4);
      const float4 sh = *reinterpret_cast<const float4*>(shift + c4 * 4);
      v.x = v.x * sc.x + sh.x;
      v.y = v.y * sc.y + sh.y;
      v.z = v.z * sc.z + sh.z;
      v.w = v.w * sc.w + sh.w;
    }
    if (relu) {
      v.x = fmaxf(v.x, 0.f);
      v.y = fmaxf(v.y, 0.f);
      v.z = fmaxf(v.z, 0.f);
      v.w = fmaxf(v.w, 0.f);
    }
    vmax = fmaxf(fmaxf(vmax, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
    *reinterpret_cast<float4*>(out + (size_t)(row0 + rw) * COUT + c4 * 4) = v;
  }
  if (next_entry && range_flag && vmax > next_entry[2]) atomicMax(range_flag, V3D_FLAG_RANGE);
  if (seen && next_entry) v3d_mark_seen(seen, vmax, next_entry[2] * (1.f / (float)(1 << V3D_QUIET_BITS)));
}

template <int CIN, int COUT>
static int launch_wave(const float* in, const float* W, const int* nbr, const int* n_ptr, int cap, int K,
                       const float* scale, const float* shift, int relu, float* out, hipStream_t st, const float* next_entry,
                       int* range_flag, unsigned* seen, const RbScanJob* rider, bool* rider_taken) {
  constexpr int G = SPW_WAVES / (COUT / 16);
  const size_t lds = std::max((size_t)G * SPC_TM * COUT * 4 + (size_t)K * SPC_TM * 4 + (size_t)K * 4 + (size_t)K * SPC_TM + 64, (size_t)RB_SCAN_LDS);
  if (lds > 64 * 1024) return V3D_EUNSUPPORTED;
  const RbScanJob r = rider ? *rider : RbScanJob{};
  hipLaunchKernelGGL((spconv_fwd_wave<CIN, COUT>), dim3(v3d_ceil_div(cap, SPC_TM) + r.blocks), dim3(SPW_WAVES * V3D_WAVE), lds, st,
                     in, W, nbr, n_ptr, cap, K, scale, shift, relu, out, next_entry, range_flag, seen, r);
  if (rider_taken) *rider_taken = r.blocks > 0;
  V3D_CHECK_LAUNCH();
  return V3D_OK;
}

// ---------------------------------------------------------------------------------- algo 4: row-owner bf16x3 MFMA
// One WAVE owns 16 consecutive output rows x all Cout columns; accumulators stay in registers for the whole
// kernel (output-stationary in the strictest sense): no LDS accumulators, no compaction lists, no barriers
// in the loop, no atomics -- the ablation of algo 3 showed that machinery, not the MFMAs, was the cost.
// Rows without a neighbour under offset k simply contribute a zero A row; the ~3x redundant matrix work
// that causes is affordable because the products run on the 16-bit matrix pipe in split precision (operands = hi + lo
// 16-bit pieces, 3 MFMAs per product tile, fp32 accumulate: see "the split-precision product" below).  Weights are split and
// packed ONCE per layer into the exact fragment order (v3d_sparse_conv_pack_weights), activations are split in registers.
// Per offset a lane issues 2*KI float4 loads of its gathered row slice and KI*NB*2 16-byte loads of packed
// weights; operands of offset k+1 are in flight while offset k multiplies (two register sets).

// power-of-two scale that puts a tensor whose largest magnitude has the fp32 bits `amax_bits` into [2^target, 2^(target + 1)):
// only the exponent is used.  Zero / subnormal maxima give 1; the exponent is clamped so that the scale AND its inverse are normal.
__host__ __device__ static inline float v3d_pow2_scale(unsigned amax_bits, int target) {
  const int eb = (int)((amax_bits >> 23) & 0xFFu);
  if (eb == 0 || eb == 255) return 1.f;
  int sb = 127 + target - (eb - 127);
  sb = sb < 2 ? 2 : (sb > 252 ? 252 : sb);
  const unsigned bits = (unsigned)sb << 23;
#if defined(__HIP_DEVICE_COMPILE__)
  return __uint_as_float(bits);
#else
  float f;
  memcpy(&f, &bits, 4);
  return f;
#endif
}
#define V3D_F16S_WEIGHT_TARGET 13  // max|W| * s_w in [2^13, 2^14)

// max |w| of a weight tensor into word 0 of the image's trailer (zeroed by the caller): non-negative floats order like their bits
__global__ void spconv_wmax_kernel(const float* __restrict__ W, long long n, unsigned* __restrict__ trailer) {
  unsigned m = 0u;
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (long long)gridDim.x * blockDim.x)
    m = max(m, __float_as_uint(W[t]) & 0x7FFFFFFFu);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
  if ((threadIdx.x & 63) == 0 && m) atomicMax(trailer, m);
}

// packed weights: img[k][ki][nb][plane][lane][8]: lane (j = lane&15, kg = lane>>4) holds
// W[k][cin = ki*32 + kg*8 + e][cout = nb*16 + j], e < 8 (zero beyond Cin); PREC 1: of W * s_w, trailer = {max bits, 1/s_w, s_w, 1}
template <int PREC>
__global__ void spconv_pack_weights_kernel(const float* __restrict__ W, int K, int Cin, int Cout, unsigned short* __restrict__ img) {
  const int KI = (Cin + 31) / 32, NB = Cout / 16;
  const long long total = (long long)K * KI * NB * 64 * 8;
  float sw = 1.f;
  if constexpr (PREC == 1) {
    unsigned* trailer = reinterpret_cast<unsigned*>(img + total * 2);
    sw = v3d_pow2_scale(trailer[0], V3D_F16S_WEIGHT_TARGET);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      reinterpret_cast<float*>(trailer)[1] = 1.f / sw;
      reinterpret_cast<float*>(trailer)[2] = sw;
      trailer[3] = 1u;
    }
  }
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
    long long r = t;
    const int e = (int)(r % 8); r /= 8;
    const int lane = (int)(r % 64); r /= 64;
    const int nb = (int)(r % NB); r /= NB;
    const int ki = (int)(r % KI); r /= KI;
    const int k = (int)r;
    const int cin = ki * 32 + (lane >> 4) * 8 + e, cout = nb * 16 + (lane & 15);
    const float v = cin < Cin ? W[((size_t)k * Cin + cin) * Cout + cout] : 0.f;
    unsigned short h, l;
    split_one<PREC>(v, sw, h, l);
    const size_t base = ((((size_t)k * KI + ki) * NB + nb) * 2) * 512 + (size_t)lane * 8 + e;
    img[base] = h;
    img[base + 512] = l;
  }
}

// Several images in ONE launch (the training plan packs every layer's forward image and the image of its transposed weights at
// the start of a step: 26 launches of ~5 us for ~1 us of work each).  Job j = blockIdx.y; mode 0: image of W (K, Cin, Cout);
// mode 1 / 2: image of the TRANSPOSED layer Wt[k'][co][ci] = W[k][ci][co] with k' = k (1) or K - 1 - k (2: submanifold layers,
// whose transposed offset table is the forward one reversed) -- Cin / Cout below are those of the image (Cin' = cout, Cout' = cin).
__global__ void spconv_pack_batch_kernel(V3dPackJobs jobs) {
  const int j = blockIdx.y;
  const float* __restrict__ W = jobs.w[j];
  unsigned short* __restrict__ img = reinterpret_cast<unsigned short*>(jobs.img[j]);
  const int K = jobs.K[j], Cin = jobs.cin[j], Cout = jobs.cout[j], mode = jobs.mode[j];
  const int KI = (Cin + 31) / 32, NB = Cout / 16;
  const long long total = (long long)K * KI * NB * 64 * 8;
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
    long long r = t;
    const int e = (int)(r % 8); r /= 8;
    const int lane = (int)(r % 64); r /= 64;
    const int nb = (int)(r % NB); r /= NB;
    const int ki = (int)(r % KI); r /= KI;
    const int k = (int)r;
    const int cin = ki * 32 + (lane >> 4) * 8 + e, cout = nb * 16 + (lane & 15);
    float v = 0.f;
    if (cin < Cin) {
      if (mode == 0) v = W[((size_t)k * Cin + cin) * Cout + cout];
      else v = W[((size_t)(mode == 2 ? K - 1 - k : k) * Cout + cout) * Cin + cin];  // the source layer is (K, Cout, Cin)
    }
    const unsigned h = bf16_rne_bits(v);
    const unsigned l = bf16_rne_bits(v - __uint_as_float(h << 16));
    const size_t base = ((((size_t)k * KI + ki) * NB + nb) * 2) * 512 + (size_t)lane * 8 + e;
    img[base] = (unsigned short)h;
    img[base + 512] = (unsigned short)l;
  }
}

int v3d_i_sparse_conv_pack_batch(const V3dPackJobs& jobs, int n, hipStream_t stream) {
  if (n < 1 || n > V3D_PACK_JOBS_MAX) return V3D_EINVAL;
  for (int j = 0; j < n; j++)
    if (!jobs.w[j] || !jobs.img[j] || jobs.K[j] < 1 || jobs.cin[j] < 1 || jobs.cout[j] < 16 || jobs.cout[j] % 16) return V3D_EINVAL;
  hipLaunchKernelGGL(spconv_pack_batch_kernel, dim3(32, n), dim3(256), 0, stream, jobs);
  V3D_CHECK_LAUNCH();
  return V3D_OK;
}

// f16s: scale entry {s, 1/s, limit, max} of a tensor from its own rows -- the exact maximum, so the entry needs no range flag.
// One workgroup (the per-op path and calibration passes: not on the frame's critical path).
#define V3D_F16S_ACT_TARGET 13  // max|x| * s in [2^(13 - headroom), 2^(14 - headroom))
__global__ __launch_bounds__(1024) void act_scale_from_rows_kernel(const float* __restrict__ rows, const int* __restrict__ n_ptr, int cap,
                                                                    int C, int headroom, float* __restrict__ entry) {
  __shared__ unsigned wmax[16];
  const long long total = (long long)(n_ptr ? min(*n_ptr, cap) : cap) * C;
  unsigned m = 0u;
  const long long vec = (((uintptr_t)rows & 15) == 0) ? total / 4 : 0;  // (a misaligned view: scalar loads)
  for (long long t = threadIdx.x; t < vec; t += 1024) {
    const uint4 v = reinterpret_cast<const uint4*>(rows)[t];
    m = max(max(m, v.x & 0x7FFFFFFFu), max(max(v.y & 0x7FFFFFFFu, v.z & 0x7FFFFFFFu), v.w & 0x7FFFFFFFu));
  }
  for (long long t = vec * 4 + threadIdx.x; t < total; t += 1024) m = max(m, __float_as_uint(rows[t]) & 0x7FFFFFFFu);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
  if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 16; w++) m = max(m, wmax[w]);
    const float s = v3d_pow2_scale(m, V3D_F16S_ACT_TARGET - headroom);
    entry[0] = s;
    entry[1] = 1.f / s;
    entry[2] = 32768.f / s;
    entry[3] = __uint_as_float(m);
  }
}

// The same entry from a grid of workgroups (the per-op path runs this in front of every f16s layer: one workgroup reading 4 MB of rows
// took 33 us, 0.46 ms of a PV-RCNN frame).  scratch = {running maximum, arrival ticket}, zero when the launch starts: every
// workgroup folds its maximum in, fences, takes a ticket; the last one to arrive reads the maximum, writes the entry and puts both
// words back to zero -- the scratch is ready for the next launch on the same stream without a fill.
#define V3D_ACT_SCALE_ITEMS 4096  // float4 loads per workgroup
__global__ __launch_bounds__(V3D_BLOCK) void act_scale_from_rows_grid_kernel(const float* __restrict__ rows, const int* __restrict__ n_ptr,
                                                                             int cap, int C, int headroom, float* __restrict__ entry,
                                                                             unsigned* __restrict__ scratch) {
  __shared__ unsigned wmax[V3D_BLOCK / V3D_WAVE];
  const long long total = (long long)(n_ptr ? min(*n_ptr, cap) : cap) * C;
  unsigned m = 0u;
  const long long vec = (((uintptr_t)rows & 15) == 0) ? total / 4 : 0;  // (a misaligned view: scalar loads)
  for (long long t = (long long)blockIdx.x * V3D_BLOCK + threadIdx.x; t < vec; t += (long long)gridDim.x * V3D_BLOCK) {
    const uint4 v = reinterpret_cast<const uint4*>(rows)[t];
    m = max(max(m, v.x & 0x7FFFFFFFu), max(max(v.y & 0x7FFFFFFFu, v.z & 0x7FFFFFFFu), v.w & 0x7FFFFFFFu));
  }
  for (long long t = vec * 4 + (long long)blockIdx.x * V3D_BLOCK + threadIdx.x; t < total; t += (long long)gridDim.x * V3D_BLOCK)
    m = max(m, __float_as_uint(rows[t]) & 0x7FFFFFFFu);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
  if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < V3D_BLOCK / V3D_WAVE; w++) m = max(m, wmax[w]);
    atomicMax(&scratch[0], m);
    __threadfence();
    if (atomicAdd(&scratch[1], 1u) == gridDim.x - 1) {  // the last workgroup to arrive
      __threadfence();
      m = atomicExch(&scratch[0], 0u);
      atomicExch(&scratch[1], 0u);
      const float s = v3d_pow2_scale(m, V3D_F16S_ACT_TARGET - headroom);
      entry[0] = s;
      entry[1] = 1.f / s;
      entry[2] = 32768.f / s;
      entry[3] = __uint_as_float(m);
    }
  }
}

extern "C" int v3d_act_scale_from_rows(const float* rows, const int32_t* n_rows, int cap, int C, int headroom_bits, float* entry,
                                        uint32_t* scratch, v3d_stream_t stream) {
  if (!rows || !entry || cap < 1 || C < 1 || headroom_bits < 0 || headroom_bits > 12) return V3D_EINVAL;
  if (!scratch) {
    hipLaunchKernelGGL(act_scale_from_rows_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, rows, n_rows, cap, C, headroom_bits, entry);
  } else {
    const long long vec = ((long long)cap * C + 3) / 4;
    const int blocks = (int)std::min<long long>(512, std::max<long long>(1, (vec + V3D_ACT_SCALE_ITEMS - 1) / V3D_ACT_SCALE_ITEMS));
    hipLaunchKernelGGL(act_scale_from_rows_grid_kernel, dim3(blocks), dim3(V3D_BLOCK), 0, (hipStream_t)stream, rows, n_rows, cap, C,
                       headroom_bits, entry, scratch);
  }
  V3D_CHECK_LAUNCH();
  return V3D_OK;
}

extern "C" size_t v3d_sparse_conv_weight_image_bytes(int K, int Cin, int Cout) {
  return (size_t)K * ((Cin + 31) / 32) * (Cout / 16) * 2 * 512 * sizeof(unsigned short) + V3D_WIMG_TRAILER;
}

extern "C" int v3d_sparse_conv_pack_weights(const float* weight, int K, int Cin, int Cout, int prec, void* image,
                                             v3d_stream_t stream) {
  if (!weight || !image || K < 1 || Cin < 1 || Cout < 16 || Cout % 16) return V3D_EINVAL;
  if (prec != V3D_PREC_BF16X3 && prec != V3D_PREC_F16S) return V3D_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  unsigned short* img = (unsigned short*)image;
  if (prec == V3D_PREC_F16S) {
    unsigned* trailer = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(image) + v3d_sparse_conv_weight_image_bytes(K, Cin, Cout) - V3D_WIMG_TRAILER);
    V3D_CHECK_HIP(v3d_fill_async(trailer, 0, V3D_WIMG_TRAILER, st));
    hipLaunchKernelGGL(spconv_wmax_kernel, dim3(64), dim3(256), 0, st, weight, (long long)K * Cin * Cout, trailer);
    hipLaunchKernelGGL(spconv_pack_weights_kernel<1>, dim3(256), dim3(256), 0, st, weight, K, Cin, Cout, img);
  } else {
    hipLaunchKernelGGL(spconv_pack_weights_kernel<0>, dim3(256), dim3(256), 0, st, weight, K, Cin, Cout, img);
  }
  V3D_CHECK_LAUNCH();
  return V3D_OK;
}

#ifndef SPR_TIMELINE
#define SPR_TIMELINE 0  // 1: cycle-counter stamps of 4 workgroups x 4 waves (tools/mb_rows_timeline.py prints them)
#endif
#if SPR_TIMELINE
__device__ unsigned long long spr_tl[4][4][16];
extern "C" int v3d_debug_rows_timeline(unsigned long long* host_out) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(spr_tl), sizeof(spr_tl));
}
#define SPR_STAMP(idx) do { if (lane == 0 && wave < 4 && (blockIdx.x & 127) == 5 && blockIdx.x < 512) spr_tl[blockIdx.x >> 7][wave][idx] = __builtin_readcyclecounter(); } while (0)
#else
#define SPR_STAMP(idx)
#endif

template <int CIN, int COUT, int PREC, int INS>
__global__ __launch_bounds__(V3D_BLOCK) void spconv_fwd_rows(const float* __restrict__ in,
                                                             const unsigned short* __restrict__ wimg,
                                                             const int* __restrict__ nbr, const int* __restrict__ n_ptr,
                                                             int cap, int K, const float* __restrict__ scale,
                                                             const float* __restrict__ shift, int relu,
                                                             float* __restrict__ out, const V3dDensifyOut dn, const V3dActScale as,
                                                             unsigned short* __restrict__ out_s, const RbScanJob rider) {
  // workgroup = 16 output rows; its 4 waves split the K kernel offsets (wave w takes k = w, w+4, ...), keep
  // private register accumulators and meet ONCE, in the epilogue, where the 4 partial tiles are summed in a
  // fixed order (deterministic).  4x more waves in flight and a 4x shorter dependent chain per wave than one
  // wave walking all offsets -- at KITTI size the kernel is latency-, not throughput-bound.
  constexpr int KI = (CIN + 31) / 32, NB = COUT / 16;
  constexpr int NF = KI * NB * 2;  // 16-byte weight fragments per offset per lane
  constexpr int NW = V3D_BLOCK / V3D_WAVE;
  extern __shared__ __attribute__((aligned(16))) float smem_rows[];
  float* part = smem_rows;                                  // [NW][NB][4][64] partial accumulators
  int* nbr_s = (int*)(part + NW * NB * 4 * 64);             // [K][16]
  SP_RIDER_PROLOGUE(smem_rows)
  const int n = min(*n_ptr, cap);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int wg = bid;
  {
    // XCD-aware tile order: the dispatcher deals consecutive workgroups round-robin to the 8 XCDs (private L2s);
    // remap the LIVE tiles so that each XCD walks one contiguous run of rows -- the rows a tile gathers are mostly
    // its neighbours', which then sit in that XCD's L2 instead of being fetched by all eight (64->64 at 36 k rows:
    // 53 -> 48 us; neutral at 8 k).
    const int nwg = (n + 15) / 16;
    if (dn.hi && bid == 0 && tid == 0) *dn.pix_n = n;  // (before the early exit: an EMPTY frame lists no pixels)
    if (wg >= nwg) return;
    const int q = nwg / 8, rmd = nwg % 8, xcd = wg % 8, idx = wg / 8;
    wg = (xcd < rmd ? xcd * (q + 1) : rmd * (q + 1) + (xcd - rmd) * q) + idx;  // bijective on [0, nwg)
  }
  const int row0 = wg * 16;
  const int r = lane & 15, kg = lane >> 4;
  const SpScales ss = sp_scales<PREC>(as, wimg, (size_t)K * NF * 512);
  SPR_STAMP(0);
  for (int k = tid >> 4; k < K; k += V3D_BLOCK / 16)
    nbr_s[k * 16 + r] = (row0 + r < n) ? nbr[(size_t)k * cap + row0 + r] : -1;
  __syncthreads();
  SPR_STAMP(1);

  float araw[2][KI][8];
  u32x4_t braw[2][NF];
  auto load_ops = [&](int k, float (&a)[KI][8], u32x4_t (&b)[NF]) {
    const int src = nbr_s[k * 16 + r];
#pragma unroll
    for (int ki = 0; ki < KI; ki++) {
      const int c0 = ki * 32 + kg * 8;
      if (src >= 0 && c0 < CIN) {
        const float* p = in + (size_t)src * CIN + c0;
        if constexpr (INS) {  // split rows: hi fragment at byte 2 c0, lo fragment CIN halfwords further
          static_assert(!INS || CIN % 8 == 0, "split rows need whole 16-byte fragments");
          const char* q = reinterpret_cast<const char*>(in) + (size_t)src * CIN * 4 + c0 * 2;
          const float4 v0 = *reinterpret_cast<const float4*>(q), v1 = *reinterpret_cast<const float4*>(q + CIN * 2);
          a[ki][0] = v0.x; a[ki][1] = v0.y; a[ki][2] = v0.z; a[ki][3] = v0.w;
          a[ki][4] = v1.x; a[ki][5] = v1.y; a[ki][6] = v1.z; a[ki][7] = v1.w;
        } else if constexpr (CIN % 8 == 0) {
          const float4 v0 = reinterpret_cast<const float4*>(p)[0], v1 = reinterpret_cast<const float4*>(p)[1];
          a[ki][0] = v0.x; a[ki][1] = v0.y; a[ki][2] = v0.z; a[ki][3] = v0.w;
          a[ki][4] = v1.x; a[ki][5] = v1.y; a[ki][6] = v1.z; a[ki][7] = v1.w;
        } else {
#pragma unroll
          for (int e = 0; e < 8; e++) a[ki][e] = (c0 + e < CIN) ? p[e] : 0.f;
        }
      } else {
#pragma unroll
        for (int e = 0; e < 8; e++) a[ki][e] = 0.f;
      }
    }
    const u32x4_t* wp = reinterpret_cast<const u32x4_t*>(wimg) + (size_t)k * NF * 64 + lane;
#pragma unroll
    for (int f = 0; f < NF; f++) b[f] = wp[(size_t)f * 64];
  };

  f32x4 acc[NB];
#pragma unroll
  for (int j = 0; j < NB; j++) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto multiply = [&](const float (&a)[KI][8], const u32x4_t (&b)[NF]) {
#pragma unroll
    for (int ki = 0; ki < KI; ki++) {
      u32x4_t ah, al;
      split_in<PREC, INS>(a[ki], ss.s_in, ah, al);
#pragma unroll
      for (int j = 0; j < NB; j++) acc[j] = sp_mfma<PREC>(al, b[(ki * NB + j) * 2], acc[j]);  // smallest terms first
#pragma unroll
      for (int j = 0; j < NB; j++) acc[j] = sp_mfma<PREC>(ah, b[(ki * NB + j) * 2 + 1], acc[j]);
#pragma unroll
      for (int j = 0; j < NB; j++) acc[j] = sp_mfma<PREC>(ah, b[(ki * NB + j) * 2], acc[j]);
    }
  };

  // this wave's offsets: wave, wave + NW, ...   (two register sets: operands of the next offset in flight)
  SPR_STAMP(2);
  if (wave < K) load_ops(wave, araw[0], braw[0]);
  SPR_STAMP(3);
  for (int k = wave; k < K; k += 2 * NW) {
    if (k + NW < K) load_ops(k + NW, araw[1], braw[1]);
    multiply(araw[0], braw[0]);
#if SPR_TIMELINE
    __builtin_amdgcn_sched_barrier(0);
    SPR_STAMP(4 + 2 * ((k - wave) / (2 * NW)));
#endif
    if (k + NW < K) {
      if (k + 2 * NW < K) load_ops(k + 2 * NW, araw[0], braw[0]);
      multiply(araw[1], braw[1]);
#if SPR_TIMELINE
      __builtin_amdgcn_sched_barrier(0);
      SPR_STAMP(5 + 2 * ((k - wave) / (2 * NW)));
#endif
    }
  }
#pragma unroll
  for (int j = 0; j < NB; j++)
#pragma unroll
    for (int rr = 0; rr < 4; rr++) part[((wave * NB + j) * 4 + rr) * 64 + lane] = acc[j][rr];
  SPR_STAMP(12);
  __syncthreads();
  SPR_STAMP(13);

  // epilogue: wave w finishes column blocks j = w, w + NW, ...;  D[row = kg*4 + rr][col = r].  The 16 rows of the tile are one
  // contiguous block of the output arrays: the finished values meet in two staging blocks behind the partial sums (fp32 rows,
  // split rows) and leave as whole 16-byte pieces of consecutive rows (sp_device.h "epilogue stores ... COALESCED through LDS").
  float vmax = 0.f;
  // f16s: planes / split rows hold the pieces of v * (the consumer's scale)
  const float s_next = (PREC == 1 && (dn.hi || out_s)) ? as.next[0] : 1.f;
  unsigned char* stg_f = reinterpret_cast<unsigned char*>(nbr_s + K * 16);
  unsigned char* stg_s = stg_f + SP_STAGE_BYTES(COUT);
  for (int j = wave; j < NB; j += NW) {
    const int col = j * 16 + r;
    // (f16s: the power-of-two factor that undoes the operand scales rides in the BatchNorm scale: exact)
    const float sc = (scale ? scale[col] : 1.f) * ss.undo, sh = scale ? shift[col] : 0.f;
#pragma unroll
    for (int rr = 0; rr < 4; rr++) {
      float v = part[((0 * NB + j) * 4 + rr) * 64 + lane];
#pragma unroll
      for (int w = 1; w < NW; w++) v += part[((w * NB + j) * 4 + rr) * 64 + lane];
      const int row = row0 + kg * 4 + rr;
      if (scale || PREC == 1) v = v * sc + sh;
      if (relu) v = fmaxf(v, 0.f);
      if (out) sp_stage_put_f32<COUT>(stg_f, kg * 4 + rr, col, v);
      if (out_s) sp_stage_put_split<PREC, COUT>(stg_s, kg * 4 + rr, col, r, v, s_next);
      if (row < n) {
        if constexpr (PREC == 1) vmax = fmaxf(vmax, fabsf(v));
        if (dn.hi) {  // .dense() of the last layer: the row straight into the split BEV planes (see V3dDensifyOut)
          const int4 c = reinterpret_cast<const int4*>(dn.coords)[row];
          const int pixel = (c.x * dn.H + c.z) * dn.W + c.w;
          const size_t o = (size_t)pixel * ((size_t)COUT * dn.D) + (size_t)col * dn.D + c.y;
          unsigned short h, l;
          split_one<PREC>(v, s_next, h, l);
          reinterpret_cast<unsigned short*>(dn.hi)[o] = h;
          reinterpret_cast<unsigned short*>(dn.lo)[o] = l;
          if (col == 0) {
            if (dn.occ) atomicAnd(dn.occ + ((size_t)c.x * dn.H + c.z) * ((dn.W + 31) >> 5) + (c.w >> 5), ~(1u << (c.w & 31)));
            dn.pix[row] = pixel;
          }
        }
      }
    }
  }
  if (out || out_s) {  // (workgroup-uniform)
    __syncthreads();
    const int nv = min(16, n - row0);
    if (out) sp_stage_flush<COUT, V3D_BLOCK>(stg_f, reinterpret_cast<unsigned char*>(out + (size_t)row0 * COUT), nv, tid);
    if (out_s) sp_stage_flush<COUT, V3D_BLOCK>(stg_s, reinterpret_cast<unsigned char*>(out_s + (size_t)row0 * (2 * COUT)), nv, tid);
  }
  if constexpr (PREC == 1) sp_range_check(as, vmax, ss.limit);
  SPR_STAMP(14);
}

// Large-N variant: 64 output rows per workgroup, one 16-row tile per wave, every wave walks ALL K offsets (no
// K-split, accumulators never leave registers) and the four waves share ONE copy of each W[k] image through LDS
// (double buffered, fetched once per workgroup and offset).  Beyond ~30 k rows the 16-row kernel is bound by the
// L2 -> CU weight stream (3 500 workgroups x 442 KB = 1.5 GB per launch at 56 k rows ~ the 34 TB/s of the L2s); here
// that stream is 4x smaller and the kernel runs into the MFMA issue rate of the 4-term split instead.  Below that
// size the 4x longer dependent chain per wave (27 instead of 7 offsets) loses: launch_rows picks by capacity.
template <int CIN, int COUT, int PREC, int INS>
__global__ __launch_bounds__(V3D_BLOCK) void spconv_fwd_rows_big(const float* __restrict__ in,
                                                                 const unsigned short* __restrict__ wimg,
                                                                 const int* __restrict__ nbr, const int* __restrict__ n_ptr,
                                                                 int cap, int K, const float* __restrict__ scale,
                                                                 const float* __restrict__ shift, int relu,
                                                                 float* __restrict__ out, const V3dActScale as,
                                                                 unsigned short* __restrict__ out_s) {
  constexpr int KI = (CIN + 31) / 32, NB = COUT / 16;
  constexpr int NF = KI * NB * 2;          // 16-byte weight fragments per offset per lane
  constexpr int WBYTES = NF * 64 * 16;     // one W[k] image
  constexpr int WLOADS = WBYTES / (V3D_BLOCK * 16);  // 16-byte pieces per thread per offset
  static_assert(WBYTES % (V3D_BLOCK * 16) == 0, "weight image must split evenly over the workgroup");
  extern __shared__ __attribute__((aligned(16))) float smem_rows[];
  unsigned char* wbuf = reinterpret_cast<unsigned char*>(smem_rows);      // [2][WBYTES]
  int* nbr_s = reinterpret_cast<int*>(wbuf + 2 * WBYTES);                  // [K][64]
  const int n = min(*n_ptr, cap);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int row0 = blockIdx.x * 64;
  if (row0 >= n) return;
  const int r = lane & 15, kg = lane >> 4;
  const SpScales ss = sp_scales<PREC>(as, wimg, (size_t)K * NF * 512);
  for (int t = tid; t < K * 64; t += V3D_BLOCK) {
    const int k = t >> 6, rr = t & 63;
    nbr_s[t] = (row0 + rr < n) ? nbr[(size_t)k * cap + row0 + rr] : -1;
  }
  u32x4_t wreg[WLOADS];
  auto load_w = [&](int k) {
    const u32x4_t* wp = reinterpret_cast<const u32x4_t*>(wimg) + (size_t)k * NF * 64 + tid;
#pragma unroll
    for (int i = 0; i < WLOADS; i++) wreg[i] = wp[(size_t)i * V3D_BLOCK];
  };
  auto store_w = [&](int buf) {
    u32x4_t* dst = reinterpret_cast<u32x4_t*>(wbuf + buf * WBYTES) + tid;
#pragma unroll
    for (int i = 0; i < WLOADS; i++) dst[(size_t)i * V3D_BLOCK] = wreg[i];
  };
  float araw[2][KI][8];
  auto load_a = [&](int k, float (&a)[KI][8]) {
    const int src = nbr_s[k * 64 + wave * 16 + r];
#pragma unroll
    for (int ki = 0; ki < KI; ki++) {
      const int c0 = ki * 32 + kg * 8;
      if (src >= 0 && c0 < CIN) {
        const char* p = reinterpret_cast<const char*>(in) + (size_t)src * CIN * 4 + (INS ? c0 * 2 : c0 * 4);
        const float4 v0 = *reinterpret_cast<const float4*>(p), v1 = *reinterpret_cast<const float4*>(p + (INS ? CIN * 2 : 16));
        a[ki][0] = v0.x; a[ki][1] = v0.y; a[ki][2] = v0.z; a[ki][3] = v0.w;
        a[ki][4] = v1.x; a[ki][5] = v1.y; a[ki][6] = v1.z; a[ki][7] = v1.w;
      } else {
#pragma unroll
        for (int e = 0; e < 8; e++) a[ki][e] = 0.f;
      }
    }
  };
  f32x4 acc[NB];
#pragma unroll
  for (int j = 0; j < NB; j++) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto multiply = [&](const float (&a)[KI][8], int buf) {
    const u32x4_t* bw = reinterpret_cast<const u32x4_t*>(wbuf + buf * WBYTES) + lane;
#pragma unroll
    for (int ki = 0; ki < KI; ki++) {
      u32x4_t ah, al;
      split_in<PREC, INS>(a[ki], ss.s_in, ah, al);
#pragma unroll
      for (int j = 0; j < NB; j++) {
        const u32x4_t bh = bw[(size_t)((ki * NB + j) * 2) * 64], bl = bw[(size_t)((ki * NB + j) * 2 + 1) * 64];
        acc[j] = sp_mfma<PREC>(al, bh, acc[j]);  // smallest terms first
        acc[j] = sp_mfma<PREC>(ah, bl, acc[j]);
        acc[j] = sp_mfma<PREC>(ah, bh, acc[j]);
      }
    }
  };

  load_w(0);
  store_w(0);
  __syncthreads();  // nbr_s and W[0] in place
  load_a(0, araw[0]);
  for (int k = 0; k < K; k += 2) {
    if (k + 1 < K) {
      load_w(k + 1);
      load_a(k + 1, araw[1]);
    }
    multiply(araw[0], 0);
    if (k + 1 < K) store_w(1);
    __syncthreads();
    if (k + 1 < K) {
      if (k + 2 < K) {
        load_w(k + 2);
        load_a(k + 2, araw[0]);
      }
      multiply(araw[1], 1);
      if (k + 2 < K) store_w(0);
      __syncthreads();
    }
  }
  // epilogue straight from the accumulators: D[row = kg*4 + rr][col = j*16 + r] of this wave's tile; rows stored coalesced through
  // the (now dead) weight / index buffers (sp_device.h, sp_tile_store_*; launch_rows_big sizes the request for the staging blocks)
  float vmax = 0.f;
  const float s_next = (PREC == 1 && out_s) ? as.next[0] : 1.f;
  {
    unsigned char* stage = reinterpret_cast<unsigned char*>(smem_rows) + wave * SP_STAGE_BYTES(COUT);
    float v[NB][4];
#pragma unroll
    for (int j = 0; j < NB; j++) {
      const int col = j * 16 + r;
      const float sc = (scale ? scale[col] : 1.f) * ss.undo, sh = scale ? shift[col] : 0.f;
#pragma unroll
      for (int rr = 0; rr < 4; rr++) {
        float vv = acc[j][rr];
        if (scale || PREC == 1) vv = vv * sc + sh;
        if (relu) vv = fmaxf(vv, 0.f);
        if constexpr (PREC == 1)
          if (row0 + wave * 16 + kg * 4 + rr < n) vmax = fmaxf(vmax, fabsf(vv));
        v[j][rr] = vv;
      }
    }
    const int nv = min(16, n - (row0 + wave * 16));
    if (nv > 0) {  // (wave-uniform)
      if (out) sp_tile_store_f32<COUT, NB>(stage, out + (size_t)(row0 + wave * 16) * COUT, v, nv, lane);
      if (out_s) sp_tile_store_split<PREC, COUT, NB>(stage, out_s + (size_t)(row0 + wave * 16) * (2 * COUT), v, s_next, nv, lane);
    }
  }
  if constexpr (PREC == 1) sp_range_check(as, vmax, ss.limit);
}


// Large-N variant, offset-outer ("k-outer") and persistent: the answer to the weight re-stream of the kernel above (875
// workgroups x 442 KB = 387 MB per launch at 56 k rows).  A workgroup of 8 waves (two per SIMD) owns 8 * T consecutive 16-row
// tiles per pass and walks the K offsets ONCE for all of them: W[k] crosses L2 -> LDS once per pass (global -> registers ->
// ds_write, double buffered: 2 x 16 KB), every wave reads the offset's fragments from LDS once into registers and applies them
// to its T tiles (accumulators T x NB f32x4 stay in registers for the whole pass).  Neighbour indices and gathered rows are
// prefetched through registers (index two offsets ahead, rows one offset ahead); nothing but the weights touches LDS.
// All fragment reads of an offset are issued before its first MFMA and the MFMAs are ordered term-major over the NB
// accumulators (sched_group_barrier): left alone the scheduler emitted read -> wait -> three DEPENDENT MFMAs per fragment.
// Grid = min(passes, 256) workgroups, pass -> rows XCD-contiguous.
// The loads of the offset loop are HAND-ISSUED (inline asm) and hand-counted.  With plain C++ loads the compiler (ROCm 7.2)
// (a) sank the row gathers behind the step's MFMAs, (b) turned "load, then select" into a conditional load and (c) -- taking the
// conservative merge of outstanding-load counts at every branch join -- waited for loads issued IN the step: one memory round
// trip per offset (64 us at 56 k rows, 33 us with the gathers removed).  Here every step issues the same loads in the same
// order -- next weights, indices two offsets ahead, rows one offset ahead -- and every consumer sits behind
// `vm_wait_tie<N>(regs)`: s_waitcnt vmcnt(N) with the registers named as in/out operands, so no use can be scheduled above
// its wait.  Loads return in order, so "all but the newest N have landed" is exact; compiler-issued memory operations in
// between can only make a wait more conservative.  Addresses are clamped, results masked afterwards (no branch).

// STAGE = 1: the gathered rows go through LDS.  What a row gather costs is set by the CU's address/tag pipe, not by bytes
// (tools/mb_gather.hip): in the MFMA operand layout the four lanes of a quad read four DIFFERENT rows -- 64 tag look-ups per
// instruction, 113 ns per 16-row x 256-byte gather per CU whatever the rows are -- while quads that read 64 contiguous bytes of
// ONE row cost 39 ns, and quads of an absent neighbour (the zero row) next to nothing.  So the rows are fetched row-contiguous
// (lane -> row lane >> 2, 16-byte chunk (lane & 3) ^ swizzle) by LDS-DMA into a 4 KB slot per (wave, tile) and the MFMA
// fragments are read back with ds_read_b128 (the swizzle (row >> 3) & 1 makes those reads conflict-free).  One slot per tile is
// enough: the fragments of offset k are in registers before the requests of k + 1 overwrite the slot piece by piece, each piece
// behind the MFMA group that consumed it.
// The pass of T tiles per wave as a device function: the kernel below picks T per LAUNCH from the live row count (device-side).
template <int CIN, int COUT, int T, int K, int STAGE, int PREC, int INS>
__device__ __forceinline__ void spconv_kouter_body(const float* __restrict__ in, const unsigned short* __restrict__ wimg,
                                                   const int* __restrict__ nbr, const int n, int cap, const float* __restrict__ scale,
                                                   const float* __restrict__ shift, int relu, float* __restrict__ out,
                                                   unsigned char* wbuf0 /*LDS: 2 x one W[k] image*/,
                                                   unsigned char* aslot /*LDS: [wave][tile][piece][16 rows][64 B]*/,
                                                   const V3dActScale& as, unsigned short* __restrict__ out_s) {
  static_assert(CIN % 32 == 0 && CIN <= 64 && T >= 1 && T <= 3, "shape not covered by the offset-outer kernel");
  constexpr int KI = CIN / 32, NB = COUT / 16, NW = 8;
  constexpr int NF = KI * NB * 2;                 // 1 KB weight fragments per offset
  constexpr int WBYTES = NF * 1024;               // one W[k] image
  constexpr int WPT = (WBYTES + NW * 64 * 16 - 1) / (NW * 64 * 16);  // 16-byte pieces per thread per offset
  constexpr int G = T * KI * 2;                   // row-gather loads per step and lane
  static_assert(WPT == 1 || WPT == 2, "weight image / workgroup shape");
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, kg = lane >> 4;
  const int rows_per_pass = 16 * T * NW;
  const int npass = (n + rows_per_pass - 1) / rows_per_pass;
  const SpScales ss = sp_scales<PREC>(as, wimg, (size_t)K * NF * 512);
  const float s_next = (PREC == 1 && out_s) ? as.next[0] : 1.f;
  float vmax = 0.f;

  f32x4 wreg[WPT];
  auto issue_w = [&](int k) {  // (threads beyond a small image re-read its last piece: no branch)
    const f32x4* wp = reinterpret_cast<const f32x4*>(wimg) + (size_t)min(k, K - 1) * NF * 64;
#pragma unroll
    for (int i = 0; i < WPT; i++) asm_gld16(wreg[i], wp + min(i * NW * 64 + tid, NF * 64 - 1));
  };
  auto store_w = [&](int buf) {
    f32x4* dst = reinterpret_cast<f32x4*>(wbuf0 + buf * WBYTES) + tid;
#pragma unroll
    for (int i = 0; i < WPT; i++)
      if ((i * NW * 64 + tid) * 16 < WBYTES) dst[(size_t)i * NW * 64] = wreg[i];
  };

  for (int v = blockIdx.x; v < npass; v += gridDim.x) {
    int pass;
    {  // each XCD (private L2; workgroup b runs on XCD b % 8, gridDim.x is a multiple of 8) walks one contiguous run of rows
      const int q = npass / 8, rmd = npass % 8, xcd = v % 8, idx = v / 8;
      pass = (xcd < rmd ? xcd * (q + 1) : rmd * (q + 1) + (xcd - rmd) * q) + idx;  // bijective on [0, npass)
    }
    const int row0 = (pass * NW + wave) * T * 16;  // first row of this wave's T tiles
    int src[2][T];
    auto issue_idx = [&](int k, int (&dst)[T]) {
#pragma unroll
      for (int t = 0; t < T; t++)
        asm_gld4(dst[t], nbr + (size_t)min(k, K - 1) * cap + min(row0 + t * 16 + (STAGE ? (lane >> 2) : r), cap - 1));
    };
    f32x4 araw[STAGE ? 1 : 2][G];
    const float* prow[T];  // this lane's slice of the rows being requested
    auto row_ptrs = [&](int k, const int (&sidx)[T]) {  // sidx = the (landed) indices of offset k
#pragma unroll
      for (int t = 0; t < T; t++) {
        int sq = (k < K && row0 + t * 16 + (STAGE ? (lane >> 2) : r) < n) ? sidx[t] : -1;
        prow[t] = (sq >= 0 ? in + (size_t)sq * CIN : spr_zero_row) +
                  (STAGE ? ((lane & 3) ^ ((lane >> 5) & 1)) * 4 : (INS ? kg * 4 : kg * 8));  // (split rows: 16-byte hi fragment kg)
      }
    };
    const unsigned slot0 = STAGE ? __builtin_amdgcn_readfirstlane(lds_addr_of(aslot) + wave * (T * CIN * 64)) : 0u;
    auto issue_chunk = [&](int t, int ki, f32x4 (&a)[G]) {  // the two loads of (tile t, channel block ki)
      if constexpr (STAGE) {
        asm_dma16(prow[t] + (ki * 2) * 16, slot0 + t * (CIN * 64) + (ki * 2) * 1024);
        asm_dma16(prow[t] + (ki * 2 + 1) * 16, slot0 + t * (CIN * 64) + (ki * 2 + 1) * 1024);
      } else if constexpr (INS) {  // split rows: hi fragment at 64 ki (+ 16 kg, in prow), lo fragment 2 CIN bytes further
        if (ki == 0) {
          asm_gld16(a[t * KI * 2], prow[t]);
          if constexpr (CIN == 64) asm_gld16_128(a[t * KI * 2 + 1], prow[t]);
          else asm_gld16_64(a[t * KI * 2 + 1], prow[t]);
        } else {
          asm_gld16_64(a[t * KI * 2 + 2], prow[t]);
          asm_gld16_192(a[t * KI * 2 + 3], prow[t]);
        }
      } else if (ki == 0) {
        asm_gld16(a[t * KI * 2], prow[t]);
        asm_gld16_16(a[t * KI * 2 + 1], prow[t]);
      } else {
        asm_gld16_128(a[t * KI * 2 + 2], prow[t]);
        asm_gld16_144(a[t * KI * 2 + 3], prow[t]);
      }
    };
    f32x4 acc[T][NB];
#pragma unroll
    for (int t = 0; t < T; t++)
#pragma unroll
      for (int j = 0; j < NB; j++) acc[t][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // one offset: rows of k are in araw[cur] (requested during the previous step), indices of k + 1 in src[cur ^ 1].
    // The row requests of k + 1 are ISSUED BETWEEN THE MFMA GROUPS of k (two loads behind every 3 NB MFMAs): a wave whose
    // vector-memory instruction waits for a slot in the CU's address pipe cannot issue anything else, so twelve loads in a row
    // at the top of the step stalled every wave of the workgroup at the same time (measured: gather time ADDED to the MFMA time
    // instead of hiding behind it); spread out, the stall of one wave overlaps the matrix work of the other wave on its SIMD.
    auto step = [&](int k, int cur) {
      issue_idx(k + 2, src[cur]);              // [T]
      issue_w(k + 1);                          // [WPT]
      if constexpr (!STAGE) vm_wait_tie<T + WPT>(araw[cur]);  // rows of k (and, older still, the indices of k + 1)
      vm_wait_tie<T + WPT>(src[cur ^ 1]);
      row_ptrs(k + 1, src[cur ^ 1]);
      if constexpr (STAGE) {  // this wave's slots hold the rows of k: fragments into registers
        const int sw = (r >> 3) & 1;
        // (position q of a row's 64-byte piece holds source chunk q ^ sw: the DMA lane mapping above.  fp32 rows: the lane's 8
        //  floats are chunks 2 (kg & 1) + v of piece 2 ki + (kg >> 1); split rows: hi = chunk kg of piece ki, lo = of piece KI + ki)
        const unsigned char* sl = aslot + wave * (T * CIN * 64) + (INS ? r * 64 + ((kg ^ sw) * 16) : (kg >> 1) * 1024 + r * 64);
#pragma unroll
        for (int t = 0; t < T; t++)
#pragma unroll
          for (int ki = 0; ki < KI; ki++)
#pragma unroll
            for (int v = 0; v < 2; v++)
              araw[0][(t * KI + ki) * 2 + v] =
                  INS ? *reinterpret_cast<const f32x4*>(sl + t * (CIN * 64) + (v * KI + ki) * 1024)
                      : *reinterpret_cast<const f32x4*>(sl + t * (CIN * 64) + ki * 2048 + ((((kg & 1) * 2 + v) ^ sw) * 16));
      }
      const u32x4_t* bw = reinterpret_cast<const u32x4_t*>(wbuf0 + cur * WBYTES) + lane;
      u32x4_t bh[KI][NB], bl[KI][NB];
#pragma unroll
      for (int ki = 0; ki < KI; ki++)
#pragma unroll
        for (int j = 0; j < NB; j++) {
          bh[ki][j] = bw[(size_t)((ki * NB + j) * 2) * 64];
          bl[ki][j] = bw[(size_t)((ki * NB + j) * 2 + 1) * 64];
        }
#pragma unroll
      for (int t = 0; t < T; t++) {
#pragma unroll
        for (int ki = 0; ki < KI; ki++) {
          const f32x4 v0 = araw[STAGE ? 0 : cur][(t * KI + ki) * 2], v1 = araw[STAGE ? 0 : cur][(t * KI + ki) * 2 + 1];
          const float x[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
          u32x4_t ah, al;
          split_in<PREC, INS>(x, ss.s_in, ah, al);
#pragma unroll
          for (int j = 0; j < NB; j++) acc[t][j] = sp_mfma<PREC>(al, bh[ki][j], acc[t][j]);  // smallest terms first
#pragma unroll
          for (int j = 0; j < NB; j++) acc[t][j] = sp_mfma<PREC>(ah, bl[ki][j], acc[t][j]);
#pragma unroll
          for (int j = 0; j < NB; j++) acc[t][j] = sp_mfma<PREC>(ah, bh[ki][j], acc[t][j]);
          __builtin_amdgcn_sched_barrier(0);
          issue_chunk(t, ki, araw[STAGE ? 0 : cur ^ 1]);   // [2]  (STAGE: overwrites the slot piece this MFMA group consumed)
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      vm_wait_tie_after<G>(wreg, acc[T - 1][NB - 1]);  // W[k + 1]: asked for only once the step's MFMAs are issued
      store_w(cur ^ 1);
      __syncthreads();
    };

    issue_idx(0, src[0]);
    issue_idx(1, src[1]);
    issue_w(0);
    vm_wait_tie<T + WPT>(src[0]);
    row_ptrs(0, src[0]);
#pragma unroll
    for (int t = 0; t < T; t++)
#pragma unroll
      for (int ki = 0; ki < KI; ki++) issue_chunk(t, ki, araw[0]);
    vm_wait_tie<G>(wreg);
    store_w(0);
    __syncthreads();  // W[0] in place (and every wave is done with the previous pass's buffers)
    // two offsets per trip (the two register sets alternate); an odd K is padded by one offset whose rows are all absent
    // (3.7 % more MFMAs at K = 27, and not one branch in the loop)
#pragma unroll 1
    for (int k = 0; k < K; k += 2) {
      step(k, 0);
      step(k + 1, 1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the padded prefetches of the last step
    // epilogue straight from the accumulators: D[row = kg*4 + rr][col = j*16 + r] of each of this wave's tiles.  STAGE: the rows go
    // through this wave's row slots (dead until the next pass's first request) and leave as whole 16-byte pieces of consecutive
    // rows (sp_device.h, sp_tile_store_*: 47.5 -> 4x us at 56 k rows) -- the same bytes.
    if constexpr (STAGE && T * CIN * 64 >= SP_STAGE_BYTES(COUT)) {  // (the staging block fits the wave's row slots)
      unsigned char* stage = aslot + wave * (T * CIN * 64);
#pragma unroll
      for (int t = 0; t < T; t++) {
        float v[NB][4];
#pragma unroll
        for (int j = 0; j < NB; j++) {
          const int col = j * 16 + r;
          const float sc = (scale ? scale[col] : 1.f) * ss.undo, sh = scale ? shift[col] : 0.f;
#pragma unroll
          for (int rr = 0; rr < 4; rr++) {
            float vv = acc[t][j][rr];
            if (scale || PREC == 1) vv = vv * sc + sh;
            if (relu) vv = fmaxf(vv, 0.f);
            if constexpr (PREC == 1)
              if (row0 + t * 16 + kg * 4 + rr < n) vmax = fmaxf(vmax, fabsf(vv));
            v[j][rr] = vv;
          }
        }
        const int nv = min(16, n - (row0 + t * 16));
        if (nv > 0) {  // (wave-uniform)
          if (out) sp_tile_store_f32<COUT, NB>(stage, out + (size_t)(row0 + t * 16) * COUT, v, nv, lane);
          if (out_s) sp_tile_store_split<PREC, COUT, NB>(stage, out_s + (size_t)(row0 + t * 16) * (2 * COUT), v, s_next, nv, lane);
        }
      }
    } else {
#pragma unroll
      for (int t = 0; t < T; t++)
#pragma unroll
        for (int j = 0; j < NB; j++) {
          const int col = j * 16 + r;
          const float sc = (scale ? scale[col] : 1.f) * ss.undo, sh = scale ? shift[col] : 0.f;
#pragma unroll
          for (int rr = 0; rr < 4; rr++) {
            const int row = row0 + t * 16 + kg * 4 + rr;
            if (row < n) {
              float vv = acc[t][j][rr];
              if (scale || PREC == 1) vv = vv * sc + sh;
              if (relu) vv = fmaxf(vv, 0.f);
              if constexpr (PREC == 1) vmax = fmaxf(vmax, fabsf(vv));
              if (out) out[(size_t)row * COUT + col] = vv;
              if (out_s) sp_store_split<PREC, COUT>(out_s, row, col, r, vv, s_next);
            }
          }
        }
    }
  }
  if constexpr (PREC == 1) sp_range_check(as, vmax, ss.limit);
}

// T = tiles per wave and pass (rows per pass = 128 T).  TMAX = 3: the pass size is picked per LAUNCH from the live row count, which
// is device-side and moves from sweep to sweep (Waymo-range stage 2: 56 k - 75 k rows): up to 65 536 rows two tiles per wave fill ONE
// round of 256 workgroups; beyond, 256-row passes would need a second, nearly empty round (49 -> 91 us in the frame, round 4
// trace) -- three tiles per wave keep up to 98 304 rows in one round.  Both bodies live in the one kernel (registers and LDS of the
// larger); every wave of the grid reads the same count, so the choice is uniform.
template <int CIN, int COUT, int TMAX, int K, int STAGE, int PREC, int INS>
__global__ __launch_bounds__(512) void spconv_fwd_rows_kouter(const float* __restrict__ in,
                                                              const unsigned short* __restrict__ wimg,
                                                              const int* __restrict__ nbr, const int* __restrict__ n_ptr,
                                                              int cap, const float* __restrict__ scale,
                                                              const float* __restrict__ shift, int relu,
                                                              float* __restrict__ out, const V3dActScale as,
                                                              unsigned short* __restrict__ out_s) {
  constexpr int WBYTES = (CIN / 32) * (COUT / 16) * 2 * 1024;
  __shared__ __attribute__((aligned(16))) unsigned char wbuf[2 * WBYTES];
  __shared__ __attribute__((aligned(16))) unsigned char aslot[STAGE ? 8 * TMAX * CIN * 64 : 16];
  const int n = min(*n_ptr, cap);
  if constexpr (TMAX == 3) {
    if (n > 2 * 128 * 256)
      spconv_kouter_body<CIN, COUT, 3, K, STAGE, PREC, INS>(in, wimg, nbr, n, cap, scale, shift, relu, out, wbuf, aslot, as, out_s);
    else
      spconv_kouter_body<CIN, COUT, 2, K, STAGE, PREC, INS>(in, wimg, nbr, n, cap, scale, shift, relu, out, wbuf, aslot, as, out_s);
  } else {
    spconv_kouter_body<CIN, COUT, TMAX, K, STAGE, PREC, INS>(in, wimg, nbr, n, cap, scale, shift, relu, out, wbuf, aslot, as, out_s);
  }
}

// Mid-size variant for K = 27 (3x3x3), the KITTI bs = 1 operating point.  The cycle-counter timeline of the 16-row
// kernel (tools/mb_rows_timeline.py, -DSPR_TIMELINE=1) shows it bound by the CU's 64 B/clk vector-memory path:
// every wave pulls its own 16 KB W[k] image plus 4 KB of gathered rows per offset -- 1.08 MB per CU per launch at
// 64->64 -- and spends 2 500-4 000 clocks just ISSUING one offset's 20 loads, whether one or two workgroups share
// the CU.  Here a workgroup owns TWO 16-row tiles and walks the offsets in 9 rounds of THREE: the three W[k] images
// of a round cross the vector path ONCE per workgroup, by LDS-DMA (global_load_lds_dwordx4) into one of THREE
// statically separate LDS round buffers, and every multiplying wave takes its operand fragments from LDS (a separate
// 128 B/clk path).  WAVE SPECIALISATION: waves 0-5 multiply (tile t, offset group g), waves 6-7 only move weights.
// Why: global_load_lds is a FLAT-encoded instruction that writes LDS; once one is pending in a wave, the compiler's
// waitcnt pass treats that wave's vector-memory counter as out of order and emits s_waitcnt vmcnt(0) for EVERY
// later register or LDS dependency (a first version with DMA and multiply in the same waves ran them back to back:
// 3 150 clocks per 4-offset round).  A wave that never issues a DMA keeps exact vmcnt bookkeeping (its gathers stay a
// round ahead), and a wave that only issues DMAs needs no compiler-placed waits at all -- it counts its own
// (s_waitcnt vmcnt(24) = "everything but the newest round has landed").  64->64 at 8 160 rows: 14.4 -> 12.3 us.
// Round r: [barrier: round r's weights are in LDS, multiply(r-1) is finished everywhere]
//          movers: DMA(r+2) into the buffer multiply(r-1) just released, wait for DMA(r+1)
//          multipliers: gather(r+2), multiply(r) from buffer r%3.
// Two rounds of weights are in flight while one multiplies.  Absent neighbours gather a row of zeros (branch-free:
// the vmcnt bookkeeping needs a fixed number of loads per round).

// STAGE = 1 (see spconv_fwd_rows_kouter): gathered rows fetched row-contiguous by LDS-DMA into ALOOK 4 KB slots per multiplying
// wave, MFMA fragments read back from LDS.
// TILES = 16-row tiles per workgroup (2, 3 or 4 -> 32 / 48 / 64 rows, TILES * OG multiplying waves).  One 64 -> 64 workgroup
// fills a CU's LDS, so a layer runs in ROUNDS of 256 workgroups: 8 160 live rows are 255 two-tile workgroups, but 8 300 rows are
// 260 -- a second round for 4 workgroups, 20 us instead of 10 (round 4 trace: every KITTI frame but the one the kernel was tuned on
// sat just above 8 192 rows in its three stage-2 layers).  More tiles per workgroup share the same weight rounds (the movers' work
// does not grow) and put up to 16 waves on the CU, whose matrix pipes idle half of a two-tile round: the caller picks the smallest
// TILES that keeps the expected row count inside ONE round.
// (the body of one workgroup's TILES tiles; the kernel behind it loops over the live tile groups)
template <int CIN, int COUT, int OG, int NBUF, int NMV, int ALOOK, int STAGE, int TILES, int PREC, int INS>
__device__ __forceinline__ void spconv_ring_body(const float* __restrict__ in, const unsigned short* __restrict__ wimg,
                                                 const int* __restrict__ nbr, const int n, int cap, const float* __restrict__ scale,
                                                 const float* __restrict__ shift, int relu, float* __restrict__ out, const int wg_in,
                                                 const SpScales& ss, float& vmax, unsigned short* __restrict__ out_s, const float s_next,
                                                 unsigned char* __restrict__ wb0 /*round buffer 0: the kernel's (a rider block's scratch)*/) {
  // OG = offsets per round = multiplying waves per tile.  OG = 3: 9 rounds, 3 round buffers, 6 + 2 waves.
  // OG = 2: 14 rounds (the 28th offset is a zero row), 4 round buffers (three rounds of weights in flight), 4 + 2
  // waves -- one multiplying wave per SIMD.  Measured slower (64->64 at 8 160 rows: 12.3 vs 13.0 us): the cost of a round
  // is mostly its barrier and the LDS read burst behind it, not the matrix pipe, so fewer, longer rounds win.  Also
  // tried and dropped: splitting (R, ki+1) while the MFMAs of (R, ki) run, with and without sched_group_barrier
  // interleave (12.9 / 13.2 us).  Only OG = 3 is instantiated.
  constexpr int K = 27, ROUNDS = (K + OG - 1) / OG, NCW = TILES * OG;
  constexpr int LOOK = NBUF - 1;  // rounds of weights in flight ahead of the multiply
  constexpr int ABUF = ALOOK + 1;  // ALOOK rounds of gathered rows in flight ahead of the multiply (registers)
  constexpr int KI = CIN / 32, NB = COUT / 16;
  constexpr int NF = KI * NB * 2;          // 1 KB weight fragments per offset
  constexpr int WBYTES = NF * 1024;        // one W[k] image
  constexpr int RB = OG * WBYTES;          // one round
  constexpr int FPM = OG * NF / NMV;       // fragments each mover wave moves per round
  constexpr int NSTG = (ROUNDS + 3) / 4;   // neighbour-table entries a multiplier lane stages
  static_assert(CIN % 32 == 0 && CIN <= 128 && (OG * NF) % NMV == 0 && (LOOK - 1) * FPM < 64, "shape not covered by the ring kernel");
  static_assert(RB >= TILES * OG * NB * 4 * 64 * 4, "partials must fit one round buffer");
  static_assert(ALOOK >= 1 && ALOOK <= 4 && (ALOOK * KI * 2 <= 8 || ALOOK * KI * 2 == 12 || ALOOK * KI * 2 == 16), "gather look-ahead");
  __shared__ __attribute__((aligned(16))) unsigned char wb1[RB];
  __shared__ __attribute__((aligned(16))) unsigned char wb2[NBUF >= 3 ? RB : 16];
  __shared__ __attribute__((aligned(16))) unsigned char wb3[NBUF == 4 ? RB : 16];
  __shared__ int nbr_all[TILES * K * 16];
  __shared__ __attribute__((aligned(16))) unsigned char aslot[STAGE ? NCW * ALOOK * CIN * 64 : 16];  // [wave][slot][piece][16 rows][64 B]
#define SPR_RING(i) ((i) % NBUF == 0 ? wb0 : ((i) % NBUF == 1 ? wb1 : ((i) % NBUF == 2 ? wb2 : wb3)))
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  int wg = wg_in;
  {
    const int nwg = (n + 16 * TILES - 1) / (16 * TILES);  // live tile groups; each XCD (private L2) walks one contiguous run
    const int qd = nwg / 8, rmd = nwg % 8, xcd = wg % 8, idx = wg / 8;
    wg = (xcd < rmd ? xcd * (qd + 1) : rmd * (qd + 1) + (xcd - rmd) * qd) + idx;  // bijective on [0, nwg)
  }

  [[maybe_unused]] const int wave = wv == NCW ? 3 : (wv < 3 ? wv : 99);  // SPR_STAMP slot: multipliers 0-2, first mover
  SPR_STAMP(0);
  if (wv >= NCW) {
    // ---------------------------------------------------------------- movers: weights, global -> LDS, nothing else
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    const int mv = wv - NCW;
    const unsigned char* wsrc = reinterpret_cast<const unsigned char*>(wimg) + (size_t)mv * FPM * 1024 + (size_t)lane * 16;
    // (an offset slot beyond K -- OG = 2, last round -- re-reads the previous image: its rows are all zero)
#define SPR_DMA(R)                                                                                                  \
  {                                                                                                                 \
    const unsigned char* gsrc = wsrc + (size_t)(R) * RB;                                                            \
    _Pragma("unroll") for (int i = 0; i < FPM; i++) {                                                               \
      const bool ok = (R) * OG + (mv * FPM + i) / NF < K;                                                           \
      __builtin_amdgcn_global_load_lds((gptr_t)(gsrc + i * 1024 - (ok ? 0 : WBYTES)),                               \
                                       (lptr_t)(SPR_RING(R) + (mv * FPM + i) * 1024), 16, 0, 0);                    \
    }                                                                                                               \
  }
#pragma unroll
    for (int R = 0; R < LOOK; R++) SPR_DMA(R)
    SPR_STAMP(1);
#pragma unroll
    for (int R = 0; R < ROUNDS; R++) {
      // round R has landed once only the DMAs of the (up to LOOK - 1) younger rounds are outstanding
      const int younger = (ROUNDS - 1 - R) < (LOOK - 1) ? (ROUNDS - 1 - R) : (LOOK - 1);
      if (younger == 2) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(2 * FPM < 64 ? 2 * FPM : 63) : "memory");
      else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(FPM) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
      if (R + LOOK < ROUNDS) SPR_DMA(R + LOOK)
      SPR_STAMP(3 + (R < 9 ? R : 8));
    }
#undef SPR_DMA
    __syncthreads();  // the multipliers' partial-sum barrier
    return;
  }

  // -------------------------------------------------------------------- multipliers: tile t, offset group g
  const int t = wv / OG, g = wv % OG;
  int* nbr_s = nbr_all + t * K * 16;
  const int row0 = (wg * TILES + t) * 16;
  const int r = lane & 15, kg = lane >> 4;
  // Gathered rows: hand-issued loads (asm_gld16*), hand-counted waits (vm_wait_tie) -- see spconv_fwd_rows_kouter.  The
  // rows of round R + ALOOK are requested at the TOP of round R from an index that was read out of LDS one round earlier
  // (left to the compiler the gathers sank behind the round's MFMAs: the index read, its wait and the address arithmetic
  // sat in front of them).  ALOOK rounds of rows stay in flight across the round barrier.
  f32x4 araw[STAGE ? 1 : ABUF][KI * 2];
  f32x4 acc[NB];
#pragma unroll
  for (int j = 0; j < NB; j++) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int rsel = STAGE ? (lane >> 2) : r;  // the tile row whose index this lane holds: its MFMA row / its DMA row
  const float* prow = spr_zero_row;  // this lane's slice of the row being requested
  auto row_ptr = [&](int src_i) {
    const int src = src_i;
    prow = (src >= 0 ? in + (size_t)src * CIN : spr_zero_row) + (STAGE ? ((lane & 3) ^ ((lane >> 5) & 1)) * 4 : (INS ? kg * 4 : kg * 8));
  };
  const unsigned slot0 = STAGE ? __builtin_amdgcn_readfirstlane(lds_addr_of(aslot) + wv * (ALOOK * CIN * 64)) : 0u;
  auto issue_chunk = [&](int ki, int Rr) {  // the two requests of channel block ki for round Rr
    if constexpr (STAGE) {
      asm_dma16(prow + (ki * 2) * 16, slot0 + (Rr % ALOOK) * (CIN * 64) + (ki * 2) * 1024);
      asm_dma16(prow + (ki * 2 + 1) * 16, slot0 + (Rr % ALOOK) * (CIN * 64) + (ki * 2 + 1) * 1024);
    } else {
      f32x4 (&a)[KI * 2] = araw[Rr % ABUF];
      if constexpr (INS) {  // split rows: hi fragment at 64 ki (+ 16 kg, in prow), lo fragment 2 CIN bytes further
        if (ki == 0) {
          asm_gld16(a[0], prow);
          if constexpr (CIN == 64) asm_gld16_128(a[1], prow);
          else asm_gld16_64(a[1], prow);
        } else if constexpr (KI == 2) {
          asm_gld16_64(a[2], prow);
          asm_gld16_192(a[3], prow);
        }
      } else if (ki == 0) {
        asm_gld16(a[0], prow);
        asm_gld16_16(a[1], prow);
      } else if constexpr (KI == 2) {
        asm_gld16_128(a[2], prow);
        asm_gld16_144(a[3], prow);
      }
    }
  };
  int nsrc = -1;  // index of the round whose rows are requested next
  {
    // entries of the first ALOOK rounds straight from global (the first gathers do not wait for the staging), then this
    // wave's share of the tile's table for the later rounds: offsets k = g, g + OG, ... x 16 rows, a quarter of the rounds
    // per 16-lane group.  The table of (tile, offset group) is written AND read by this wave only.
    const bool live = row0 + r < n;
    int first[ALOOK], stage[NSTG];
#pragma unroll
    for (int i = 0; i < ALOOK; i++) first[i] = (row0 + rsel < n && i * OG + g < K) ? nbr[(size_t)(i * OG + g) * cap + row0 + rsel] : -1;
#pragma unroll
    for (int i = 0; i < NSTG; i++) {
      const int kk = (kg + 4 * i) * OG + g;
      stage[i] = (live && kk < K) ? nbr[(size_t)kk * cap + row0 + r] : -1;
    }
#pragma unroll
    for (int i = 0; i < ALOOK; i++) {
      row_ptr(first[i]);
#pragma unroll
      for (int ki = 0; ki < KI; ki++) issue_chunk(ki, i);
    }
#pragma unroll
    for (int i = 0; i < NSTG; i++)
      if ((kg + 4 * i) * OG + g < K) nbr_s[((kg + 4 * i) * OG + g) * 16 + r] = stage[i];
    if (ALOOK < ROUNDS && ALOOK * OG + g < K) nsrc = nbr_s[(ALOOK * OG + g) * 16 + rsel];
  }
  SPR_STAMP(1);
#pragma unroll
  for (int R = 0; R < ROUNDS; R++) {
    // (no vmcnt: the gathers stay in flight across the barrier)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    // The rows of round R + ALOOK are requested BETWEEN the MFMA groups of this round (two loads behind every 3 NB MFMAs): a
    // wave whose vector-memory instruction waits for a slot in the CU's address pipe can issue nothing else, so requests
    // bunched at the top of the round stall every multiplier at once; spread out they overlap the other waves' matrix work.
    if (R + ALOOK < ROUNDS) row_ptr(nsrc);
    if (R + ALOOK + 1 < ROUNDS) nsrc = ((R + ALOOK + 1) * OG + g < K) ? nbr_s[((R + ALOOK + 1) * OG + g) * 16 + rsel] : -1;
    {  // rows of round R have landed once only the (up to ALOOK - 1) younger rounds' loads are outstanding
      const int younger = (ROUNDS - 1 - R < ALOOK - 1 ? ROUNDS - 1 - R : ALOOK - 1) * KI * 2;  // a constant after unrolling
      if constexpr (STAGE) {
        if (younger == 0) vm_wait<0>();
        else if (younger == 2) vm_wait<2>();
        else if (younger == 4) vm_wait<4>();
        else if (younger == 6) vm_wait<6>();
        else if (younger == 8) vm_wait<8>();
        else vm_wait<12>();
        // this wave's slot holds the rows of round R: fragments into registers
        const int sw = (r >> 3) & 1;
        const unsigned char* sl = aslot + (wv * ALOOK + R % ALOOK) * (CIN * 64) + (INS ? r * 64 + ((kg ^ sw) * 16) : (kg >> 1) * 1024 + r * 64);
#pragma unroll
        for (int ki = 0; ki < KI; ki++)
#pragma unroll
          for (int v = 0; v < 2; v++)  // (split rows: hi = chunk kg of piece ki, lo = of piece KI + ki; see spconv_kouter_body)
            araw[0][ki * 2 + v] = INS ? *reinterpret_cast<const f32x4*>(sl + (v * KI + ki) * 1024)
                                      : *reinterpret_cast<const f32x4*>(sl + ki * 2048 + ((((kg & 1) * 2 + v) ^ sw) * 16));
      } else {
        f32x4 (&ar)[KI * 2] = araw[R % ABUF];
        if (younger == 0) vm_wait_tie<0>(ar);
        else if (younger == 2) vm_wait_tie<2>(ar);
        else if (younger == 4) vm_wait_tie<4>(ar);
        else if (younger == 6) vm_wait_tie<6>(ar);
        else if (younger == 8) vm_wait_tie<8>(ar);
        else if (younger == 12) vm_wait_tie<12>(ar);
        else vm_wait_tie<16>(ar);
      }
    }
    const u32x4_t* bw = reinterpret_cast<const u32x4_t*>(SPR_RING(R) + g * WBYTES) + lane;
    u32x4_t ah[KI], al[KI], bh[KI][NB], bl[KI][NB];
#pragma unroll
    for (int ki = 0; ki < KI; ki++)
#pragma unroll
      for (int j = 0; j < NB; j++) {
        bh[ki][j] = bw[(size_t)((ki * NB + j) * 2) * 64];
        bl[ki][j] = bw[(size_t)((ki * NB + j) * 2 + 1) * 64];
      }
#pragma unroll
    for (int ki = 0; ki < KI; ki++) {
      const f32x4 v0 = araw[STAGE ? 0 : R % ABUF][ki * 2], v1 = araw[STAGE ? 0 : R % ABUF][ki * 2 + 1];
      const float x[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
      split_in<PREC, INS>(x, ss.s_in, ah[ki], al[ki]);
    }
#pragma unroll
    for (int ki = 0; ki < KI; ki++) {
#pragma unroll
      for (int j = 0; j < NB; j++) acc[j] = sp_mfma<PREC>(al[ki], bh[ki][j], acc[j]);  // smallest terms first
#pragma unroll
      for (int j = 0; j < NB; j++) acc[j] = sp_mfma<PREC>(ah[ki], bl[ki][j], acc[j]);
#pragma unroll
      for (int j = 0; j < NB; j++) acc[j] = sp_mfma<PREC>(ah[ki], bh[ki][j], acc[j]);
      if (R + ALOOK < ROUNDS) {  // (STAGE: overwrites the slot piece this MFMA group consumed)
        __builtin_amdgcn_sched_barrier(0);
        issue_chunk(ki, R + ALOOK);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
#if SPR_TIMELINE
    __builtin_amdgcn_sched_barrier(0);
    SPR_STAMP(3 + (R < 9 ? R : 8));
#endif
  }
  // partial tiles of the OG offset groups meet in round buffer 0 (its last reader round is followed by at least one
  // more round barrier: every wave is past it)
  // (the buffer used is not the one the last round reads; its own last reader round lies behind a round barrier)
  float* part = reinterpret_cast<float*>((ROUNDS - 1) % NBUF != 0 ? wb0 : wb1) + t * (OG * NB * 4 * 64);  // [TILES][OG][NB][4][64]
#pragma unroll
  for (int j = 0; j < NB; j++)
#pragma unroll
    for (int rr = 0; rr < 4; rr++) part[((g * NB + j) * 4 + rr) * 64 + lane] = acc[j][rr];
  SPR_STAMP(12);
  __syncthreads();
  SPR_STAMP(13);
  for (int it = g; it < NB * 4; it += OG) {  // D[row = kg*4 + rr][col = j*16 + r]: 4 NB (j, rr) slices dealt to the tile's waves
    const int j = it >> 2, rr = it & 3;
    const int col = j * 16 + r, row = row0 + kg * 4 + rr;
    float v = part[((0 * NB + j) * 4 + rr) * 64 + lane];
#pragma unroll
    for (int w = 1; w < OG; w++) v += part[((w * NB + j) * 4 + rr) * 64 + lane];
    if (row < n) {
      if (scale) v = v * (scale[col] * ss.undo) + shift[col];
      else if constexpr (PREC == 1) v = v * ss.undo;
      if (relu) v = fmaxf(v, 0.f);
      if constexpr (PREC == 1) vmax = fmaxf(vmax, fabsf(v));
      if (out) out[(size_t)row * COUT + col] = v;
      if (out_s) sp_store_split<PREC, COUT>(out_s, row, col, r, v, s_next);
    }
  }
  SPR_STAMP(14);
#undef SPR_RING
}

// The grid is what the chip holds at once (launch_rows_ring), not the capacity's, and every workgroup strides over the live tile
// groups (normally one each).  Why: a 64 -> 64 workgroup owns a CU's LDS, so of a capacity-sized grid (1 026
// workgroups for a 32 k-row capacity, ~255 of them live) the dead tail cannot be PLACED until a live workgroup retires -- the
// dispatcher sits on this kernel for its whole duration and, with several frames in flight, no other frame's kernel starts beside
// it (round-4 overlap trace: the ring kernels ran alone 99.8 % of their time, the dense tile kernel 69 %).
template <int CIN, int COUT, int OG, int NBUF, int NMV, int ALOOK, int STAGE, int TILES, int PREC, int INS>
__global__ __launch_bounds__((TILES * OG + NMV) * 64) void spconv_fwd_rows_ring(const float* __restrict__ in,
                                                                          const unsigned short* __restrict__ wimg,
                                                                          const int* __restrict__ nbr,
                                                                          const int* __restrict__ n_ptr, int cap,
                                                                          const float* __restrict__ scale,
                                                                          const float* __restrict__ shift, int relu,
                                                                          float* __restrict__ out, const V3dActScale as,
                                                                          unsigned short* __restrict__ out_s, const RbScanJob rider) {
  // round buffer 0 of the body lives here: a rider block uses it as its scratch (a buffer of its own would take the 32 -> 32 shape
  // from three workgroups per CU to two)
  constexpr int RB0 = OG * (CIN / 32) * (COUT / 16) * 2 * 1024;
  static_assert(RB0 >= RB_SCAN_LDS, "a rider's scratch fits round buffer 0");
  __shared__ __attribute__((aligned(16))) unsigned char wb0[RB0];
  // (the 64 -> 64 shape carries none: its workgroup owns a CU, and reading the job's block count in front of everything else cost
  //  every launch 0.3-0.4 us -- 11.8 -> 12.2 us at 4 204 rows)
  constexpr bool HOST = !(CIN == 64 && COUT == 64);
  int bid = (int)blockIdx.x, nblk = (int)gridDim.x;
  if constexpr (HOST) {
    if (rider.blocks && bid < rider.blocks) {
      if (threadIdx.x < V3D_BLOCK) rb_rider_run(rider, bid, wb0);
      return;
    }
    bid -= rider.blocks;
    nblk -= rider.blocks;
  }
  const int n = min(*n_ptr, cap);
  const int nwg = (n + 16 * TILES - 1) / (16 * TILES);
  const SpScales ss = sp_scales<PREC>(as, wimg, (size_t)27 * (CIN / 32) * (COUT / 16) * 2 * 512);
  const float s_next = (PREC == 1 && out_s) ? as.next[0] : 1.f;
  float vmax = 0.f;
  // (the XCD-contiguous remap inside the body is a bijection of [0, nwg) for ANY set of indices below nwg)
  for (int g = bid; g < nwg; g += nblk) {
    spconv_ring_body<CIN, COUT, OG, NBUF, NMV, ALOOK, STAGE, TILES, PREC, INS>(in, wimg, nbr, n, cap, scale, shift, relu, out, g, ss, vmax,
                                                                              out_s, s_next, wb0);
    __syncthreads();  // the next group's weight DMA and partial sums reuse this group's LDS
  }
  if constexpr (PREC == 1) sp_range_check(as, vmax, ss.limit);
}

// how a launch is parameterised beyond the layer itself: arithmetic, its scale entries, and the split-row copies (INS: the gathered
// rows are `in_split`; out_split: also write the output rows split for the next packed layer)
struct SpLaunch {
  int prec;
  V3dActScale as;
  const void* in_split;
  unsigned short* out_split;
  const RbScanJob* rider;  // nullable: a rulebook scan the launch may carry (kernels that can: *rider_taken = true)
  bool* rider_taken;
};

// What a ring launch carries: its rider blocks take a workgroup slot each (8 waves, 50-80 KB of LDS), most of them leave at once
// (one block per chunk of the CAPACITY; the live chunks are a fraction) -- a 64 -> 64 workgroup owns a CU: none (max_blocks 0).
static RbScanJob sp_ring_rider(const SpLaunch& sl, int max_blocks) {
  if (!sl.rider || max_blocks < 1 || sl.rider->blocks > max_blocks) return RbScanJob{};
  return *sl.rider;
}

template <int CIN, int COUT, int OG, int NBUF, int NMV, int ALOOK, int STAGE, int TILES, int PREC, int INS>
static int launch_rows_ring_p(const float* in, const void* wimg, const int* nbr, const int* n_ptr, int cap,
                              const float* scale, const float* shift, int relu, float* out, hipStream_t st, const SpLaunch& sl) {
  // grid: never more workgroups than the chip can hold AT ONCE (occupancy of this instantiation x CUs, a multiple of 8 for the XCD
  // map) -- every workgroup is placed the moment the kernel is dispatched -- and never more than the capacity needs
  static V3dPerDeviceInt cache;
  int* slots = cache.slot();
  if (!*slots) {
    int dev = 0, n_cu = 0, per_cu = 0;
    V3D_CHECK_HIP(hipGetDevice(&dev));
    V3D_CHECK_HIP(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev));
    V3D_CHECK_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, spconv_fwd_rows_ring<CIN, COUT, OG, NBUF, NMV, ALOOK, STAGE, TILES, PREC, INS>,
                                                               (TILES * OG + NMV) * 64, 0));
    *slots = std::max(8, per_cu * n_cu / 8 * 8);
  }
  // (rider blocks take a workgroup slot each: a 64 -> 64 workgroup owns a CU, so few of them; the lighter shapes hold 2-3 per CU)
  const RbScanJob r = sp_ring_rider(sl, CIN == 64 && COUT == 64 ? 0 : 512);
  const int grid = std::min(v3d_ceil_div(cap, 16 * TILES), std::max(*slots - r.blocks, 8));
  hipLaunchKernelGGL((spconv_fwd_rows_ring<CIN, COUT, OG, NBUF, NMV, ALOOK, STAGE, TILES, PREC, INS>), dim3(grid + r.blocks),
                     dim3((TILES * OG + NMV) * 64), 0, st, INS ? (const float*)sl.in_split : in, (const unsigned short*)wimg, nbr, n_ptr, cap,
                     scale, shift, relu, out, sl.as, sl.out_split, r);
  if (sl.rider_taken) *sl.rider_taken = r.blocks > 0;
  V3D_CHECK_LAUNCH();
  return V3D_OK;
}
template <int CIN, int COUT, int OG, int NBUF, int NMV, int ALOOK = 2, int STAGE = 0, int TILES = 2>
static int launch_rows_ring(const float* in, const void* wimg, const int* nbr, const int* n_ptr, int cap,
                            const float* scale, const float* shift, int relu, float* out, hipStream_t st, const SpLaunch& sl) {
#define V3D_RING(P, I) launch_rows_ring_p<CIN, COUT, OG, NBUF, NMV, ALOOK, STAGE, TILES, P, I>(in, wimg, nbr, n_ptr, cap, scale, shift, relu, out, st, sl)
  if (sl.prec == V3D_PREC_F16S) return sl.in_split ? V3D_RING(1, 1) : V3D_RING(1, 0);
  return sl.in_split ? V3D_RING(0, 1) : V3D_RING(0, 0);
#undef V3D_RING
}

template <int CIN, int COUT, int PREC, int INS>
static void launch_rows_big(const float* in, const void* wimg, const int* nbr, const int* n_ptr, int cap, int K,
                            const float* scale, const float* shift, int relu, float* out, hipStream_t st, const SpLaunch& sl) {
  constexpr int NF = ((CIN + 31) / 32) * (COUT / 16) * 2;
  const size_t lds = std::max((size_t)2 * NF * 64 * 16 + (size_t)K * 64 * 4, (size_t)4 * SP_STAGE_BYTES(COUT));  // (buffers; epilogue staging)
  hipLaunchKernelGGL((spconv_fwd_rows_big<CIN, COUT, PREC, INS>), dim3(v3d_ceil_div(cap, 64)), dim3(V3D_BLOCK), lds, st,
                     INS ? (const float*)sl.in_split : in, (const unsigned short*)wimg, nbr, n_ptr, cap, K, scale, shift, relu, out, sl.as,
                     sl.out_split);
}

template <int CIN, int COUT, int T, int STAGE, int PREC, int INS>
static void launch_rows_kouter(const float* in, const void* wimg, const int* nbr, const int* n_ptr, int cap,
                               const float* scale, const float* shift, int relu, float* out, hipStream_t st, const SpLaunch& sl) {
  const int passes = v3d_ceil_div(cap, 16 * (T == 3 ? 2 : T) * 8);  // (T = 3: the kernel may walk 256-row passes)
  const int grid = passes >= 256 ? 256 : ((passes + 7) / 8) * 8;  // a multiple of 8: the pass -> XCD map assumes it
  hipLaunchKernelGGL((spconv_fwd_rows_kouter<CIN, COUT, T, 27, STAGE, PREC, INS>), dim3(grid), dim3(512), 0, st,
                     INS ? (const float*)sl.in_split : in, (const unsigned short*)wimg, nbr, n_ptr, cap, scale, shift, relu, out, sl.as,
                     sl.out_split);
}

// rows_hint > 0: expected number of LIVE rows (the live count itself is device-side): the caller's best knowledge -- the
// capacity when it is exact (per-op Python path), the counts observed on earlier frames (v3d_backbone_tune); 0 = unknown.
// rows_hint < 0 forces a kernel (tests, benchmarks): -1 the 16-row kernel, -5 the 64-row LDS-shared-weights kernel, -6 / -7 the
// offset-outer persistent kernel (rows staged through LDS / gathered into registers), -10 the LDS-ring kernel in the form chosen
// for the shape, -16 its register-gather form (each where the shape has it, else the 16-row kernel).
#define V3D_BIG_ROWS 32768
#define V3D_RING_ROWS 16384  // two full rounds of 32-row workgroups on 256 CUs; beyond, the 16-row kernel wins again (36 k rows: 47 vs 53 us)
template <int CIN, int COUT, int PREC, int INS>
static int launch_rows(const float* in, const void* wimg, const int* nbr, const int* n_ptr, int cap, int K,
                       const float* scale, const float* shift, int relu, float* out, int rows_hint, hipStream_t st,
                       const V3dDensifyOut* densify, int tiles_min, const SpLaunch& sl) {
  const int force = densify ? 1 : (rows_hint < 0 ? -rows_hint : 0);  // .dense() rides in the 16-row kernel's epilogue only
  if constexpr (CIN >= 32 && CIN <= 64 && COUT >= 32 && COUT <= 64) {
    // the offset-outer persistent kernel (3x3x3 only; 6: rows staged through LDS, 7: rows gathered into registers): 64->64 at
    // 56 k rows 58 -> 47.5 us.  It needs a full round of 256-row passes to pay: the 32-channel shapes and the mid sizes stay on
    // the kernels below (32->32 at 81 k rows: 316 passes on 256 workgroups = 2 rounds, 43 vs 30 us)
    if (K == 27 && (force == 6 || force == 7 || force == 8 || (force == 0 && CIN == 64 && COUT == 64 && rows_hint >= V3D_BIG_ROWS))) {
      if (force == 7) launch_rows_kouter<CIN, COUT, 2, 0, PREC, INS>(in, wimg, nbr, n_ptr, cap, scale, shift, relu, out, st, sl);
      else if (force == 8) launch_rows_kouter<CIN, COUT, 2, 1, PREC, INS>(in, wimg, nbr, n_ptr, cap, scale, shift, relu, out, st, sl);  // 256-row passes only
      else launch_rows_kouter<CIN, COUT, 3, 1, PREC, INS>(in, wimg, nbr, n_ptr, cap, scale, shift, relu, out, st, sl);  // 256 / 384-row passes by live count
      V3D_CHECK_LAUNCH();
      return V3D_OK;
    }
    // the 64-row LDS-shared-weights kernel: from ~32 k live rows on the 16-row kernel is bound by the L2 -> CU weight
    // stream (64->64 at 36 k rows 54 vs 53 us, at 56 k 82 vs 66 us, at 81 k 110 vs 85 us)
    if (force == 5 || (force == 0 && rows_hint >= V3D_BIG_ROWS)) {
      launch_rows_big<CIN, COUT, PREC, INS>(in, wimg, nbr, n_ptr, cap, K, scale, shift, relu, out, st, sl);
      V3D_CHECK_LAUNCH();
      return V3D_OK;
    }
    // the two-tile LDS-ring kernel (3x3x3 only, <= 16 384 rows).  10 = the form chosen per shape by measurement (same box, KITTI
    // layers): 64->64 rows staged through LDS, one slot, 2 weight buffers, 4 movers (11.2 -> 9.9 us at 8 160 rows); 32->32 staged,
    // two slots, 2 movers (8.4 -> 7.4 us at 13 731 rows); 32->64 / 64->32 register gathers, 3 weight buffers (7.2 us; staged 7.6-8.3).
    // 16 = the register-gather form for every shape (cross-check).
    if (K == 27 && (force == 10 || force == 16 || (force >= 12 && force <= 14) || (force == 0 && rows_hint <= V3D_RING_ROWS))) {
      // (measured and not kept: two offsets per round + two weight buffers + register gathers = 69 KB of LDS at 64 -> 64, so that
      // two such workgroups -- the same layer of another frame in flight -- or one and an 80-pixel dense tile could share a CU:
      // 12.9 vs 10.0 us in isolation and 3 121 vs 3 306 frames/s pipelined; measured again in round 6 beside the four-wave dense tile
      // kernel, which DOES leave it half a CU -- <64, 64, 2, 2, 4, 1, 1, 2>: 85 KB, 8 waves x 106 VGPRs -- 3 710-3 730 vs 4 075-4 120
      // frames/s, and another summation order (two partial sums per output instead of three))
      if constexpr (CIN == 64 && COUT == 64) {
        if (force != 16) {
          // one LDS-filling workgroup per CU: keep the layer inside ONE round of 256 workgroups.  The live count is device-side and
          // moves from frame to frame (KITTI stage 2: 8 100 - 8 700 rows), so the tile count is picked with 10 % of headroom over
          // the count the plan was tuned on; force 12 / 13 / 14 pin the 2 / 3 / 4-tile form (tests, microbenchmarks)
          const long long want = force ? 0 : (long long)rows_hint + rows_hint / 10;
          // tiles_min (the plan's throughput mode): with several frames in flight what counts is the CU-time of a launch, not its
          // duration -- 4 204 rows are 132 two-tile workgroups x 10.6 us or 66 four-tile workgroups x ~13 us, 39 % less CU-time that
          // another frame's kernels use (same box: 3 820 -> 3 875 / 3 970 frames/s pipelined, 0.451 -> 0.462 ms one frame at a time)
          int tiles = force == 12 ? 2 : force == 13 ? 3 : force == 14 ? 4 : (want <= 32 * 256 ? 2 : want <= 48 * 256 ? 3 : 4);
          if (!force && tiles < tiles_min) tiles = tiles_min > 4 ? 4 : tiles_min;
          if (tiles == 2) return launch_rows_ring<CIN, COUT, 3, 2, 4, 1, 1, 2>(in, wimg, nbr, n_ptr, cap, scale, shift, relu, out, st, sl);
          if (tiles == 3) return launch_rows_ring<CIN, COUT, 3, 2, 4, 1, 1, 3>(in, wimg, nbr, n_ptr, cap, scale, shift, relu, out, st, sl);
          return launch_rows_ring<CIN, COUT, 3, 2, 4, 1, 1, 4>(in, wimg, nbr, n_ptr, cap, scale, shift, relu, out, st, sl);
        }
      }
      if (force != 16 && CIN == 32 && COUT == 32) return launch_rows_ring<CIN, COUT, 3, 2, 2, 2, 1>(in, wimg, nbr, n_ptr, cap, scale, shift, relu, out, st, sl);
      return launch_rows_ring<CIN, COUT, 3, 3, 2>(in, wimg, nbr, n_ptr, cap, scale, shift, relu, out, st, sl);
    }
  }
  const size_t lds = std::max((size_t)4 * (COUT / 16) * 4 * 64 * 4 + (size_t)K * 16 * 4 + 2 * SP_STAGE_BYTES(COUT),  // partial sums | indices | staging
                              (size_t)RB_SCAN_LDS);
  const RbScanJob r = sl.rider ? *sl.rider : RbScanJob{};
  hipLaunchKernelGGL((spconv_fwd_rows<CIN, COUT, PREC, INS>), dim3(v3d_ceil_div(cap, 16) + r.blocks), dim3(V3D_BLOCK), lds, st,
                     INS ? (const float*)sl.in_split : in, (const unsigned short*)wimg, nbr, n_ptr, cap, K, scale, shift, relu, out,
                     densify ? *densify : V3dDensifyOut{}, sl.as, sl.out_split, r);
  if (sl.rider_taken) *sl.rider_taken = r.blocks > 0;
  V3D_CHECK_LAUNCH();
  return V3D_OK;
}

// Forward with PRE-PACKED split weights (v3d_sparse_conv_pack_weights): the split-precision row-owner kernels.
extern "C" int v3d_sparse_conv_fwd_packed(const float* in, const void* weight_image, const int32_t* nbr,
                                           const int32_t* n_out, int cap_out, int K, int Cin, int Cout, const float* scale,
                                           const float* shift, int relu, float* out, int rows_hint, int prec,
                                           const float* act_in, const float* act_next, int32_t* range_flag,
                                           const void* in_split, void* out_split, v3d_stream_t stream) {
  const V3dActScale as{act_in, act_next, range_flag, nullptr};
  return v3d_i_sparse_conv_fwd_packed(in ? in : (const float*)in_split, weight_image, nbr, n_out, cap_out, K, Cin, Cout, scale, shift,
                                      relu, out, rows_hint, (hipStream_t)stream, nullptr, 2, prec, &as, in_split, out_split);
}

// fp32 rows -> split rows ([hi: C x 16 bit | lo: C x 16 bit] per row, the bytes of the fp32 row) in the pieces of `prec`, f16s: of
// x * entry[0].  What a packed layer writes through out_split, as a launch of its own (callers that hold fp32 rows, tests).
template <int PREC>
__global__ __launch_bounds__(256) void rows_split_kernel(const float* __restrict__ rows, const int* __restrict__ n_ptr, int cap, int C,
                                                         const float* __restrict__ entry, unsigned short* __restrict__ out) {
  const long long total = (long long)(n_ptr ? min(*n_ptr, cap) : cap) * C;
  const float s = PREC == 1 ? entry[0] : 1.f;
  for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long long)gridDim.x * 256) {
    const long long row = t / C;
    const int c = (int)(t - row * C);
    unsigned short h, l;
    split_one<PREC>(rows[t], s, h, l);
    out[row * 2 * C + c] = h;
    out[row * 2 * C + C + c] = l;
  }
}

extern "C" int v3d_sparse_rows_split(const float* rows, const int32_t* n_rows, int cap, int C, int prec, const float* act_entry,
                                     void* out_split, v3d_stream_t stream) {
  if (!rows || !out_split || cap < 1 || C < 8 || C % 8) return V3D_EINVAL;
  if (prec == V3D_PREC_F16S ? !act_entry : prec != V3D_PREC_BF16X3) return V3D_EINVAL;
  const int blocks = (int)std::min<long long>(((long long)cap * C + 255) / 256, 4096);
  if (prec == V3D_PREC_F16S)
    hipLaunchKernelGGL(rows_split_kernel<1>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, rows, n_rows, cap, C, act_entry, (unsigned short*)out_split);
  else
    hipLaunchKernelGGL(rows_split_kernel<0>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, rows, n_rows, cap, C, nullptr, (unsigned short*)out_split);
  V3D_CHECK_LAUNCH();
  return V3D_OK;
}

// the (Cin, Cout) pairs v3d_i_sparse_conv_fwd_packed has kernels for (its V3D_TRY table): what a plan asks BEFORE it decides that a
// layer's fp32 rows need not be written because the next layer gathers the split copy
bool v3d_i_sparse_conv_packed_supported(int Cin, int Cout) {
  static const int shapes[][2] = {{4, 16}, {16, 16}, {16, 32}, {32, 32}, {32, 64}, {64, 64}, {4, 32}, {32, 16}, {64, 32}, {64, 128}, {128, 128}};
  for (const auto& s : shapes)
    if (s[0] == Cin && s[1] == Cout) return true;
  return false;
}

int v3d_i_sparse_conv_fwd_packed(const float* in, const void* weight_image, const int32_t* nbr, const int32_t* n_out,
                                 int cap_out, int K, int Cin, int Cout, const float* scale, const float* shift, int relu,
                                 float* out, int rows_hint, hipStream_t st, const V3dDensifyOut* densify, int ring_tiles_min,
                                 int prec, const V3dActScale* act, const void* in_split, void* out_split, const RbScanJob* rider,
                                 bool* rider_taken) {
  if (rider_taken) *rider_taken = false;
  if (!in || !weight_image || !nbr || !n_out || (!out && !out_split) || cap_out < 1 || K < 1) return V3D_EINVAL;  // (out may be NULL
  // when only the split copy of the rows is wanted: a plan in throughput mode, whose intermediate fp32 rows nobody reads)
  if ((scale == nullptr) != (shift == nullptr)) return V3D_EINVAL;
  if (!out && densify) return V3D_EINVAL;
  if (prec != V3D_PREC_BF16X3 && prec != V3D_PREC_F16S) return V3D_EINVAL;
  if (prec == V3D_PREC_F16S && (!act || !act->in || ((densify || out_split) && !act->next))) return V3D_EINVAL;  // no scale, no f16s
  if (in_split && Cin % 8) return V3D_EINVAL;
  SpLaunch sl;
  sl.prec = prec;
  sl.as = (prec == V3D_PREC_F16S) ? *act : V3dActScale{nullptr, nullptr, nullptr, nullptr};
  sl.in_split = in_split;
  sl.out_split = (unsigned short*)out_split;
  sl.rider = rider;
  sl.rider_taken = rider_taken;
#define V3D_GO(ci, co, P, I) \
  return launch_rows<ci, co, P, I>(in, weight_image, nbr, n_out, cap_out, K, scale, shift, relu, out, rows_hint, st, densify, ring_tiles_min, sl)
#define V3D_TRY(ci, co)                                    \
  if (Cin == ci && Cout == co) {                           \
    if constexpr ((ci) % 8 == 0) {                         \
      if (prec == V3D_PREC_F16S && in_split) V3D_GO(ci, co, 1, 1); \
      if (prec == V3D_PREC_BF16X3 && in_split) V3D_GO(ci, co, 0, 1); \
    }                                                      \
    if (prec == V3D_PREC_F16S) V3D_GO(ci, co, 1, 0);       \
    V3D_GO(ci, co, 0, 0);                                  \
  }
  V3D_TRY(4, 16)
  V3D_TRY(16, 16)
  V3D_TRY(16, 32)
  V3D_TRY(32, 32)
  V3D_TRY(32, 64)
  V3D_TRY(64, 64)
  V3D_TRY(4, 32)
  V3D_TRY(32, 16)
  V3D_TRY(64, 32)
  V3D_TRY(64, 128)
  V3D_TRY(128, 128)
#undef V3D_TRY
#undef V3D_GO
  return V3D_EUNSUPPORTED;
}

extern "C" int v3d_sparse_conv_fwd(const float* in, const float* weight, const int32_t* nbr, const int32_t* n_out,
                                   int cap_out, int K, int Cin, int Cout, const float* scale, const float* shift,
                                   int relu, float* out, int algo, v3d_stream_t stream) {
  return v3d_i_sparse_conv_fwd_exact(in, weight, nbr, n_out, cap_out, K, Cin, Cout, scale, shift, relu, out, algo, (hipStream_t)stream,
                                     nullptr, nullptr, nullptr, nullptr, nullptr);
}

int v3d_i_sparse_conv_fwd_exact(const float* in, const float* weight, const int32_t* nbr, const int32_t* n_out, int cap_out, int K,
                                int Cin, int Cout, const float* scale, const float* shift, int relu, float* out, int algo,
                                hipStream_t st, const float* next_entry, int32_t* range_flag, unsigned* seen, const RbScanJob* rider,
                                bool* rider_taken) {
  if (rider_taken) *rider_taken = false;
  if (!in || !weight || !nbr || !n_out || !out || cap_out < 1 || K < 1 || Cin < 1 || Cout < 1) return V3D_EINVAL;
  if ((scale == nullptr) != (shift == nullptr)) return V3D_EINVAL;
  if (algo != 0 && algo != 1 && algo != 3) return V3D_EINVAL;  // (2 was an LDS-staged fp32 kernel: removed)
  if (algo == 0 || algo == 3) {
    int rc = V3D_EUNSUPPORTED;
#define V3D_TRY(ci, co) \
  if (Cin == ci && Cout == co) rc = launch_wave<ci, co>(in, weight, nbr, n_out, cap_out, K, scale, shift, relu, out, st, next_entry, range_flag, seen, rider, rider_taken);
    V3D_TRY(4, 16)
    V3D_TRY(16, 16)
    V3D_TRY(16, 32)
    V3D_TRY(32, 32)
    V3D_TRY(32, 64)
    V3D_TRY(64, 64)
    V3D_TRY(4, 32)
    V3D_TRY(32, 16)
    V3D_TRY(64, 32)
    V3D_TRY(64, 128)
    V3D_TRY(128, 128)
#undef V3D_TRY
    if (rc != V3D_EUNSUPPORTED || algo == 3) return rc;  // algo 0 falls through to the scalar kernel
  }
  const long long total = (long long)cap_out * Cout;
  const int blocks = (int)((total + V3D_BLOCK - 1) / V3D_BLOCK);
  hipLaunchKernelGGL(spconv_fwd_scalar, dim3(blocks > 8192 ? 8192 : blocks), dim3(V3D_BLOCK), 0, st, in, weight, nbr,
                       n_out, cap_out, K, Cin, Cout, scale, shift, relu, out);
  V3D_CHECK_LAUNCH();
  return V3D_OK;
}

// ---------------------------------------------------------------------------------- backward: weight gradient
// dW[k][ci][co] = sum over pairs (i = nbr[k][o] >= 0) of X[i][ci] * dY[o][co]      (spconv indice_conv backward,
// SURVEY.md section 8a T3).  A reduction over pairs -> exact-fp32 MFMA v_mfma_f32_16x16x4_f32 with the PAIR index
// as the MFMA reduction dimension: lane (r = lane&15, p = lane>>4) feeds A[ci = r][p] = X[row_p][ci0 + r] and
// B[p][co = r] = dY[o_p][co0 + r]; both are 64-byte row slices.
//
// grid = (row slabs S, K offsets, channel tile groups).  Each wave owns a quarter of its block's slab: it reads
// 64 rulebook entries at a time (coalesced), COMPACTS the live pairs with ballot/popcount into a wave-private LDS
// queue, and consumes the queue 16 pairs at a time -- so the MFMAs only ever see live pairs (a 3x3x3 offset is
// populated for ~10-40 % of the rows).  The TI x TJ accumulator tiles stay in registers for the whole slab; the 4
// waves are summed through LDS in wave order and the S slab partials by a second tiny kernel in slab order:
// deterministic, no atomics.
#define BW_MAX_SPLITS 64
#define BW_Q 128

// Rows per slab from the LIVE row count (device-side): the grid always has BW_MAX_SPLITS slabs, so a plan that passes a
// capacity several times the live count still spreads the live rows over all of them.  A multiple of 256 (4 waves x 64).
__host__ __device__ static inline int bw_rows_per_block(int n) {
  const int r = ((n + BW_MAX_SPLITS - 1) / BW_MAX_SPLITS + 255) & ~255;
  return r < 256 ? 256 : r;
}

template <int TI, int TJ>
__global__ __launch_bounds__(V3D_BLOCK) void spconv_bwd_weight_tiles(const float* __restrict__ X,
                                                                     const float* __restrict__ dY,
                                                                     const int* __restrict__ nbr,
                                                                     const int* __restrict__ n_ptr, int cap, int Cin,
                                                                     int Cout, int groups_j,
                                                                     float* __restrict__ partial /*[S][K][Cin][Cout]*/) {
  __shared__ int q_src[4][BW_Q];
  __shared__ int q_out[4][BW_Q];
  __shared__ float red[TI * TJ * 256];
  const int n = min(*n_ptr, cap);
  const int rows_per_block = bw_rows_per_block(n);
  const int slab = blockIdx.x, k = blockIdx.y, K = gridDim.y;
  const int lo = slab * rows_per_block;
  if (lo >= n) return;  // the reduce kernel only reads the live slabs
  const int hi = min(n, lo + rows_per_block);
  const int ci_base = (blockIdx.z / groups_j) * TI * 16, co_base = (blockIdx.z % groups_j) * TJ * 16;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 15, p = lane >> 4;
  const int rows_per_wave = rows_per_block >> 2;  // a multiple of 64
  const int wlo = lo + wave * rows_per_wave, whi = min(hi, wlo + rows_per_wave);
  const int* __restrict__ nk = nbr + (size_t)k * cap;
  int* qs = q_src[wave];
  int* qo = q_out[wave];

  f32x4 acc[TI][TJ];
#pragma unroll
  for (int t = 0; t < TI; t++)
#pragma unroll
    for (int u = 0; u < TJ; u++) acc[t][u] = f32x4{0.f, 0.f, 0.f, 0.f};

  // 16 queue entries from `base`, the first `valid` of them live
  auto consume = [&](int base, int valid) {
    float a[4][TI], b[4][TJ];
#pragma unroll
    for (int s4 = 0; s4 < 4; s4++) {
      const int e = 4 * s4 + p;
      const bool ok = e < valid;
      const int src = ok ? qs[base + e] : 0, o = ok ? qo[base + e] : 0;
#pragma unroll
      for (int t = 0; t < TI; t++) {
        const int ci = ci_base + 16 * t + r;
        a[s4][t] = (ok && ci < Cin) ? X[(size_t)src * Cin + ci] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < TJ; u++) {
        const int co = co_base + 16 * u + r;
        b[s4][u] = (ok && co < Cout) ? dY[(size_t)o * Cout + co] : 0.f;
      }
    }
#pragma unroll
    for (int s4 = 0; s4 < 4; s4++)
#pragma unroll
      for (int t = 0; t < TI; t++)
#pragma unroll
        for (int u = 0; u < TJ; u++) acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s4][t], b[s4][u], acc[t][u], 0, 0, 0);
  };

  int cnt = 0;  // wave-uniform queue length
  int src_next = (wlo < whi && wlo + lane < whi) ? nk[wlo + lane] : -1;
  for (int o0 = wlo; o0 < whi; o0 += 64) {
    const int src = src_next;
    const int o = o0 + lane;
    src_next = (o + 64 < whi) ? nk[o + 64] : -1;  // next 64 entries in flight while this batch is consumed
    const unsigned long long live = __ballot(src >= 0);
    if (src >= 0) {
      const int pos = cnt + __popcll(live & ((1ull << lane) - 1ull));
      qs[pos] = src;
      qo[pos] = o;
    }
    cnt += __popcll(live);
    int g = 0;
    for (; cnt - g >= 16; g += 16) consume(g, 16);
    if (g) {  // move the < 16 leftovers to the front (sources >= 16 > destinations: disjoint)
      const int rem = cnt - g;
      int ms = 0, mo = 0;
      if (lane < rem) { ms = qs[g + lane]; mo = qo[g + lane]; }
      if (lane < rem) { qs[lane] = ms; qo[lane] = mo; }
      cnt = rem;
    }
  }
  if (cnt > 0) consume(0, cnt);

  // block reduction in wave order through LDS, then one coalesced store of the slab partial
  for (int w = 0; w < 4; w++) {
    if (wave == w) {
#pragma unroll
      for (int t = 0; t < TI; t++)
#pragma unroll
        for (int u = 0; u < TJ; u++)
#pragma unroll
          for (int rr = 0; rr < 4; rr++) {
            const int idx = (((t * TJ + u) * 4 + rr) << 6) + lane;
            red[idx] = w == 0 ? acc[t][u][rr] : red[idx] + acc[t][u][rr];
          }
    }
    __syncthreads();
  }
  float* outp = partial + ((size_t)slab * K + k) * Cin * Cout;
  for (int idx = threadIdx.x; idx < TI * TJ * 256; idx += V3D_BLOCK) {
    const int l = idx & 63, rr = (idx >> 6) & 3, tu = idx >> 8;
    const int ci = ci_base + 16 * (tu / TJ) + (l >> 4) * 4 + rr;  // D[row = p*4 + rr][col = r]
    const int co = co_base + 16 * (tu % TJ) + (l & 15);
    if (ci < Cin && co < Cout) outp[(size_t)ci * Cout + co] = red[idx];
  }
}

__global__ void spconv_bwd_weight_reduce_kernel(const float* __restrict__ partial, const int* __restrict__ n_ptr, int cap,
                                                long long elems, float* __restrict__ dW) {
  const int n = min(*n_ptr, cap);
  const int rows_per_block = bw_rows_per_block(n);
  const int slabs = (n + rows_per_block - 1) / rows_per_block;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < elems; i += (long long)gridDim.x * blockDim.x) {
    float s = 0.f;
    for (int sp = 0; sp < slabs; sp++) s += partial[(size_t)sp * elems + i];
    dW[i] = s;
  }
}

extern "C" size_t v3d_sparse_conv_bwd_weight_workspace(int K, int Cin, int Cout) {
  return (size_t)BW_MAX_SPLITS * K * Cin * Cout * sizeof(float) + 256;
}

template <int TI, int TJ>
static void launch_bwd_weight(const float* X, const float* dY, const int* nbr, const int* n_out, int cap, int K, int Cin,
                              int Cout, int slabs, float* partial, hipStream_t st) {
  const int gi = ((Cin + 15) / 16 + TI - 1) / TI, gj = ((Cout + 15) / 16 + TJ - 1) / TJ;
  hipLaunchKernelGGL((spconv_bwd_weight_tiles<TI, TJ>), dim3(slabs, K, gi * gj), dim3(V3D_BLOCK), 0, st, X, dY, nbr, n_out,
                     cap, Cin, Cout, gj, partial);
}

// X (>= n_in, Cin) forward input, dY (cap_out, Cout) output gradient, nbr (K, cap_out) the FORWARD rulebook.
extern "C" int v3d_sparse_conv_bwd_weight(const float* X, const float* dY, const int32_t* nbr, const int32_t* n_out,
                                          int cap_out, int K, int Cin, int Cout, float* dW, void* workspace,
                                          size_t workspace_bytes, v3d_stream_t stream) {
  hipStream_t st = (hipStream_t)stream;
  if (!X || !dY || !nbr || !n_out || !dW || !workspace || cap_out < 1 || K < 1 || K > 65535 || Cin < 1 || Cout < 1)
    return V3D_EINVAL;
  if (workspace_bytes < v3d_sparse_conv_bwd_weight_workspace(K, Cin, Cout)) return V3D_EWORKSPACE;
  const int slabs = std::min(BW_MAX_SPLITS, v3d_ceil_div(cap_out, 256));  // slabs beyond the live rows leave at once
  const int ti = Cin > 32 ? 4 : (Cin > 16 ? 2 : 1), tj = Cout > 32 ? 4 : (Cout > 16 ? 2 : 1);
  float* partial = (float*)workspace;
#define BW_CASE(A, B)                                                                                              \
  if (ti == A && tj == B) launch_bwd_weight<A, B>(X, dY, nbr, n_out, cap_out, K, Cin, Cout, slabs, partial, st)
  BW_CASE(1, 1); BW_CASE(1, 2); BW_CASE(1, 4); BW_CASE(2, 1); BW_CASE(2, 2); BW_CASE(2, 4); BW_CASE(4, 1); BW_CASE(4, 2); BW_CASE(4, 4);
#undef BW_CASE
  const long long elems = (long long)K * Cin * Cout;
  hipLaunchKernelGGL(spconv_bwd_weight_reduce_kernel, dim3((int)((elems + 255) / 256)), dim3(256), 0, st,
                     (const float*)workspace, n_out, cap_out, elems, dW);
  V3D_CHECK_LAUNCH();
  return V3D_OK;
}

// ---------------------------------------------------------------------------------- T2: .dense()
// dense (B,C,D,H,W) = 0; dense[b, :, z, y, x] = feat[i, :].  The zero fill is a memset node, the
// scatter one thread per (row, channel) with channel fastest inside a wave's 64 lanes.
__global__ __launch_bounds__(V3D_BLOCK) void densify_kernel(const float* __restrict__ feat,
                                                            const int4* __restrict__ coords,
                                                            const int* __restrict__ n_ptr, int cap, int C, int D, int H,
                                                            int Wd, float* __restrict__ dense) {
  const int n = min(*n_ptr, cap);
  const long long total = (long long)n * C;
  const size_t vol = (size_t)D * H * Wd;
  for (long long t = (long long)blockIdx.x * V3D_BLOCK + threadIdx.x; t < total; t += (long long)gridDim.x * V3D_BLOCK) {
    const int i = (int)(t / C), ch = (int)(t % C);
    const int4 c = coords[i];
    dense[((size_t)c.x * C + ch) * vol + ((size_t)c.y * H + c.z) * Wd + c.w] = feat[t];
  }
}

extern "C" int v3d_densify(const float* feat, const int32_t* coords, const int32_t* n, int cap, int B, int C,
                           const int32_t* spatial_shape_host, float* dense, v3d_stream_t stream) {
  hipStream_t st = (hipStream_t)stream;
  if (!feat || !coords || !n || cap < 1 || B < 1 || C < 1 || !spatial_shape_host || !dense) return V3D_EINVAL;
  const int D = spatial_shape_host[0], H = spatial_shape_host[1], Wd = spatial_shape_host[2];
  if (D < 1 || H < 1 || Wd < 1) return V3D_EINVAL;
  V3D_CHECK_HIP(v3d_fill_async(dense, 0, (size_t)B * C * D * H * Wd * sizeof(float), st));
  const long long total = (long long)cap * C;
  const int blocks = (int)((total + V3D_BLOCK - 1) / V3D_BLOCK);
  hipLaunchKernelGGL(densify_kernel, dim3(blocks > 4096 ? 4096 : blocks), dim3(V3D_BLOCK), 0, st, feat,
                     (const int4*)coords, n, cap, C, D, H, Wd, dense);
  V3D_CHECK_LAUNCH();
  return V3D_OK;
}
