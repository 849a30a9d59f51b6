// brick.hip -- submanifold sparse convolution over SPATIALLY ORDERED rows: the neighbourhood of a 256-row pass lives in LDS.
//
// Reference path: spconv SubMConv3d as driven by vision3d/detector/sparse_cnn.py:15-30,151-175 (indice_conv: one gather + GEMM +
// scatter-add per kernel offset).  The packed kernels of spconv.hip gather the 16 input rows of a tile once PER OFFSET -- 27 gathers
// per output tile, 14 of them live at the Waymo-range stage 2 -- and are bound by the issue cost of those gathers (spconv.hip,
// spconv_fwd_rows_kouter: eight LDS-DMA pieces per wave and offset).  When the rows of a stage are numbered in a spatial order
// (a plan in brick order: second_plan.hip, v3d_backbone_set_row_order), the 27 x 256 neighbours of 256 consecutive rows are only
// ~1.2-1.7 x 256 DIFFERENT rows (tools/brick_union_analysis.py).  So:
//   * brick_plan_kernel (once per submanifold rulebook): per pass the sorted list of the distinct input rows it touches
//     (`ulist`, <= BRK_UMAX), every neighbour-table entry translated into a slot of that list (`lidx`, 16 bit), and per 16-row
//     tile the mask of offsets under which ANY of its rows has a neighbour (`tmask`);
//   * spconv_fwd_brick (per layer): a workgroup brings its pass's distinct rows into LDS ONCE (LDS-DMA, rows swizzled by slot),
//     then walks the offsets: W[k] L2 -> LDS (double buffered, as the offset-outer kernel), the MFMA fragments of a tile are
//     ds_read_b128s from the slots `lidx` names -- no global gather in the loop at all -- and a tile skips the offsets its mask
//     does not hold (reads and MFMAs).  Same products in the same order as spconv_fwd_rows_kouter: bit-identical rows.
//   A pass whose neighbourhood exceeds the LDS slots (never seen on the synthetic sweeps; possible for adversarial orders) runs a
//   plain direct-gather body instead: correct for ANY row order, fast for spatial ones.
#include <type_traits>

#include "sp_device.h"

#define BRK_ROWS 256   // output rows per pass: BRK_NW waves x BRK_T tiles
#define BRK_NW 4
#define BRK_T 4
#define BRK_UMAX 480   // LDS slots of a pass's neighbourhood (1.875 x BRK_ROWS); one more slot holds zeros (absent neighbours)
#define BRK_ABSENT 0xFFFFu
#ifndef BRK_DBG
#define BRK_DBG 0  // timing ablations (WRONG results): 1 no MFMAs, 2 no row-fragment reads, 4 no W-fragment reads, 8 no W stream / barrier, 16 no neighbourhood DMA
#endif
#define BRK_STRIDE(cap) (((cap) + 63) & ~63)  // row stride of the slot table (16-bit entries; rows of a wave's four tiles interleaved)

// ---------------------------------------------------------------------------------------------------- the per-rulebook planner
// One workgroup per pass.  LDS: a bitmap over the input row space (bit v = "row v is a neighbour of this pass") and its word-wise
// exclusive popcount prefix: slot of row v = prefix[v >> 5] + popc(bits below v) -- the distinct rows in ascending order.
template <int K>
__global__ __launch_bounds__(256) void brick_plan_kernel(const int* __restrict__ nbr, const int* __restrict__ n_in_ptr, int cap_in,
                                                         const int* __restrict__ n_out_ptr, int cap_out,
                                                         unsigned short* __restrict__ lidx, int* __restrict__ ulist,
                                                         int* __restrict__ ucnt, unsigned* __restrict__ tmask) {
  extern __shared__ unsigned brk_smem[];
  __shared__ int wsum[4];
  const int n = min(*n_out_ptr, cap_out), n_in = min(*n_in_ptr, cap_in);
  const int row0 = blockIdx.x * BRK_ROWS;
  if (row0 >= n) return;
  const int W = (n_in + 31) >> 5;
  unsigned* bm = brk_smem;
  int* pre = reinterpret_cast<int*>(brk_smem + W);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int w = tid; w < W; w += 256) bm[w] = 0u;
  __syncthreads();
  const int row = row0 + tid;
  const bool live = row < n;
  int v[K];
  unsigned tm = 0u;  // lanes 0..3 of a wave: the mask of tile 4 * wave + lane
#pragma unroll
  for (int k = 0; k < K; k++) {
    v[k] = live ? nbr[(size_t)k * cap_out + row] : -1;
    if (v[k] >= n_in) v[k] = -1;  // (a table entry beyond the live inputs cannot occur; never index the bitmap with it)
    if (v[k] >= 0) atomicOr(&bm[v[k] >> 5], 1u << (v[k] & 31));
    const unsigned long long b = __ballot(v[k] >= 0);
    if (lane < 4 && ((b >> (16 * lane)) & 0xFFFFull)) tm |= 1u << k;
  }
  if (lane < 4 && row0 + (wave * 4 + lane) * 16 < n) tmask[(row0 >> 4) + wave * 4 + lane] = tm;
  __syncthreads();
  // exclusive prefix of the words' popcounts: thread t owns words [t * wpt, (t + 1) * wpt)
  const int wpt = (W + 255) / 256;
  const int w_lo = min(tid * wpt, W), w_hi = min(w_lo + wpt, W);
  int local = 0;
  for (int w = w_lo; w < w_hi; w++) local += __popc(bm[w]);
  int incl = local;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int t = __shfl_up(incl, o);
    if (lane >= o) incl += t;
  }
  if (lane == 63) wsum[wave] = incl;
  __syncthreads();
  int base = incl - local;
  for (int w = 0; w < wave; w++) base += wsum[w];
  if (tid == 0) ucnt[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
  int* ul = ulist + (size_t)blockIdx.x * BRK_UMAX;
  for (int w = w_lo; w < w_hi; w++) {
    pre[w] = base;
    unsigned bits = bm[w];
    while (bits) {
      const int b = __ffs(bits) - 1;
      bits &= bits - 1;
      if (base < BRK_UMAX) ul[base] = w * 32 + b;
      base++;
    }
  }
  __syncthreads();
  if (row >= BRK_STRIDE(cap_out)) return;  // (rows beyond the live count read "no neighbour")
#pragma unroll
  for (int k = 0; k < K; k++) {
    unsigned s = BRK_ABSENT;
    if (v[k] >= 0) {
      const int w = v[k] >> 5;
      s = (unsigned)min(pre[w] + __popc(bm[w] & ((1u << (v[k] & 31)) - 1u)), 0xFFFE);
    }
    lidx[(size_t)k * BRK_STRIDE(cap_out) + (row & ~63) + 4 * (row & 15) + ((row >> 4) & 3)] = (unsigned short)s;  // (a wave's four tiles interleaved: 8 bytes per lane)
  }
}

extern "C" size_t v3d_sparse_brick_table_bytes(int cap, int K, size_t* lidx_bytes, size_t* ulist_bytes, size_t* ucnt_bytes,
                                               size_t* tmask_bytes) {
  if (cap < 1 || K < 1) return 0;
  const size_t passes = (size_t)v3d_ceil_div(cap, BRK_ROWS);
  const size_t a = v3d_align((size_t)K * BRK_STRIDE(cap) * 2), b = v3d_align(passes * BRK_UMAX * 4), c = v3d_align(passes * 4),
               d = v3d_align((size_t)v3d_ceil_div(cap, 16) * 4);
  if (lidx_bytes) *lidx_bytes = a;
  if (ulist_bytes) *ulist_bytes = b;
  if (ucnt_bytes) *ucnt_bytes = c;
  if (tmask_bytes) *tmask_bytes = d;
  return a + b + c + d;
}

int v3d_i_sparse_brick_plan(const int32_t* nbr, const int32_t* n_in, int cap_in, const int32_t* n_out, int cap_out, int K,
                            const V3dBrickTables& t, hipStream_t st) {
  if (!nbr || !n_in || !n_out || !t.lidx || !t.ulist || !t.ucnt || !t.tmask || cap_in < 1 || cap_out < 1) return V3D_EINVAL;
  if (K != 27) return V3D_EUNSUPPORTED;
  const size_t lds = (size_t)((cap_in + 31) / 32) * 8;
  if (lds > 150 * 1024) return V3D_EUNSUPPORTED;  // (600 k input rows)
  static V3dPerDeviceFlag raised;
  V3D_CHECK_HIP(v3d_set_max_lds(raised, (const void*)brick_plan_kernel<27>, 150 * 1024));
  hipLaunchKernelGGL(brick_plan_kernel<27>, dim3(v3d_ceil_div(cap_out, BRK_ROWS)), dim3(256), lds, st, nbr, n_in, cap_in, n_out, cap_out,
                     t.lidx, t.ulist, t.ucnt, t.tmask);
  V3D_CHECK_LAUNCH();
  return V3D_OK;
}

extern "C" int v3d_sparse_brick_plan(const int32_t* nbr, const int32_t* n_rows, int cap, int K, uint16_t* lidx, int32_t* ulist,
                                     int32_t* ucnt, uint32_t* tmask, v3d_stream_t stream) {
  return v3d_i_sparse_brick_plan(nbr, n_rows, cap, n_rows, cap, K, V3dBrickTables{lidx, ulist, ucnt, tmask}, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------------------------------- the convolution
typedef int i32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void asm_gld8(i32x2_t& d, const void* p) { asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(d) : "v"(p) : "memory"); }
// s_waitcnt vmcnt(N) with every register of the array tied (no use can be scheduled above the wait)
template <int N, int M> __device__ __forceinline__ void vm_wait_arr(f32x4 (&a)[M]) {
  asm volatile("s_waitcnt vmcnt(%0)" : : "n"(N) : "memory");
#pragma unroll
  for (int i = 0; i < M; i++) asm volatile("" : "+v"(a[i]) : : "memory");
}
template <int I, int N, typename F>
__device__ __forceinline__ void v3d_static_for_impl(F& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    v3d_static_for_impl<I + 1, N>(f);
  }
}
template <int N, typename F>
__device__ __forceinline__ void v3d_static_for(F& f) { v3d_static_for_impl<0, N>(f); }

// Rows are the producer's SPLIT rows: [hi: CIN x 16 bit | lo: CIN x 16 bit] = ROWB bytes = CH 16-byte chunks; the MFMA A fragment
// of lane (r, kg) for channel block ki is chunk ki * 4 + kg (hi) and CH / 2 + ki * 4 + kg (lo).  Slot s keeps chunk c at position
// (c + s) mod CH: the sixteen rows of a tile read the same chunk index from sixteen slots, and consecutive slots -- what spatially
// ordered neighbours mostly are -- then land on different banks.
//
// Workgroup = 4 waves (one per SIMD) x 4 tiles = one 256-row pass.  The offset loop is fully unrolled and software-pipelined by
// hand so that nothing a step needs is requested IN that step:
//   W[k + 2]  global -> registers, requested at the top of step k (two steps of flight time);
//   W[k + 1]  registers -> LDS at the top of step k, ONE barrier, then its fragments LDS -> the second fragment register set,
//             all in front of the MFMAs of step k, which run on the first set meanwhile;
//   A(k, t+1) the next tile's row fragments are read from the pass's LDS neighbourhood in front of tile t's MFMAs.
// (Round-6 measurements, profiles/r06_brick_ablation.txt: with the requests issued in the step that consumes them -- the structure
//  of spconv_fwd_rows_kouter -- the 8 waves of a workgroup run in lockstep between the per-step barriers: all read, then all multiply;
//  27 x (LDS burst + 96 MFMAs per SIMD + barrier) = 45 us at 56 k rows whatever else the loop contains.)
template <int CIN, int COUT, int PREC>
__global__ __launch_bounds__(256) void spconv_fwd_brick(const unsigned char* __restrict__ in_s, const unsigned short* __restrict__ wimg,
                                                        const int* __restrict__ nbr, const V3dBrickTables bt,
                                                        const int* __restrict__ n_ptr, int cap, const float* __restrict__ scale,
                                                        const float* __restrict__ shift, int relu, float* __restrict__ out,
                                                        const V3dActScale as, unsigned short* __restrict__ out_s) {
  static_assert(CIN % 32 == 0 && CIN <= 64 && COUT % 16 == 0 && COUT <= 64, "shape not covered by the brick kernel");
  constexpr int K = 27, T = BRK_T, NW = BRK_NW;
  constexpr int KI = CIN / 32, NB = COUT / 16;
  constexpr int NF = KI * NB * 2;     // 1 KB weight fragments per offset
  constexpr int WBYTES = NF * 1024;   // one W[k] image
  constexpr int WPT = WBYTES / (NW * 64 * 16);
  constexpr int ROWB = CIN * 4, CH = ROWB / 16, RPI = 64 / CH;  // bytes / chunks per row; rows per DMA instruction
  constexpr int NI = (BRK_UMAX + RPI * NW - 1) / (RPI * NW);   // DMA instructions per wave for a full neighbourhood
  static_assert(WPT >= 1 && WPT * NW * 64 * 16 == WBYTES, "weight image / workgroup shape");
  __shared__ __attribute__((aligned(256))) unsigned char halo[(BRK_UMAX + 1) * ROWB];
  __shared__ __attribute__((aligned(16))) unsigned char wbuf[2 * WBYTES];

  const int n = min(*n_ptr, cap);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, kg = lane >> 4;
  int pass = blockIdx.x;
  {  // XCD-contiguous pass order (workgroup b runs on XCD b % 8): neighbouring passes share most of their rows -> one L2
    const int npass = (n + BRK_ROWS - 1) / BRK_ROWS;
    if (pass >= npass) return;
    const int q = npass / 8, rmd = npass % 8, xcd = pass % 8, idx = pass / 8;
    pass = (xcd < rmd ? xcd * (q + 1) : rmd * (q + 1) + (xcd - rmd) * q) + idx;
  }
  const int row0 = pass * BRK_ROWS + wave * (T * 16);  // first row of this wave's T tiles
  const SpScales ss = sp_scales<PREC>(as, wimg, (size_t)K * NF * 512);
  const float s_next = (PREC == 1 && out_s) ? as.next[0] : 1.f;
  float vmax = 0.f;
  const int ucnt = __builtin_amdgcn_readfirstlane(bt.ucnt[pass]);

  f32x4 acc[T][NB];
#pragma unroll
  for (int t = 0; t < T; t++)
#pragma unroll
    for (int j = 0; j < NB; j++) acc[t][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  if (ucnt <= BRK_UMAX) {
    // ---- everything the pass needs from memory but the weights is requested up front: the slots of this wave's rows for ALL offsets
    //      (8 bytes per offset and lane: the planner interleaves the four tiles of a wave) and the pass's distinct rows (LDS-DMA).
    i32x2_t pk[K];  // 16-bit slots of row r of tiles 0..3
#pragma unroll
    for (int k = 0; k < K; k++)
      asm_gld8(pk[k], reinterpret_cast<const i32x2_t*>(bt.lidx + (size_t)k * BRK_STRIDE(cap)) + (row0 >> 2) + r);
    f32x4 wreg[2][WPT];
    auto issue_w = [&](int k, f32x4 (&w)[WPT]) {
      const f32x4* wp = reinterpret_cast<const f32x4*>(wimg) + (size_t)k * NF * 64 + tid;
#pragma unroll
      for (int i = 0; i < WPT; i++) asm_gld16(w[i], wp + i * NW * 64);
    };
    auto store_w = [&](int buf, const f32x4 (&w)[WPT]) {
      f32x4* dst = reinterpret_cast<f32x4*>(wbuf + buf * WBYTES) + tid;
#pragma unroll
      for (int i = 0; i < WPT; i++) dst[(size_t)i * NW * 64] = w[i];
    };
    issue_w(0, wreg[0]);
    issue_w(1, wreg[1]);
    {
      const int* ul = bt.ulist + (size_t)pass * BRK_UMAX;
      const int j = lane / CH, q = lane % CH;
      const unsigned halo0 = lds_addr_of(halo);
#pragma unroll
      for (int i0 = 0; i0 < NI; i0 += 8) {  // (eight row indices in flight per lane, then their eight DMA pieces)
        int src[8];
#pragma unroll
        for (int i = i0; i < i0 + 8 && i < NI; i++) src[i - i0] = ul[min((i * NW + wave) * RPI + j, BRK_UMAX - 1)];  // (beyond ucnt: stale, unused)
#pragma unroll
        for (int i = i0; i < i0 + 8 && i < NI; i++) {
          const int sb = (i * NW + wave) * RPI;  // wave-uniform
          if (sb < ucnt && !(BRK_DBG & 16)) {
            const int s = sb + j;
            const int c = (q - s) & (CH - 1);
            const unsigned char* g = s < ucnt ? in_s + (size_t)src[i - i0] * ROWB + c * 16 : reinterpret_cast<const unsigned char*>(spr_zero_row) + c * 16;
            asm_dma16(g, __builtin_amdgcn_readfirstlane(halo0 + sb * ROWB));
          }
        }
      }
      for (int i = tid; i < ROWB / 4; i += NW * 64) reinterpret_cast<unsigned*>(halo + BRK_UMAX * ROWB)[i] = 0u;  // the zero slot
    }
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(wreg[0][0]), "+v"(wreg[1][0]) : : "memory");  // (the neighbourhood, the slots, W[0], W[1])
#pragma unroll
    for (int i = 1; i < WPT; i++) asm volatile("" : "+v"(wreg[0][i]), "+v"(wreg[1][i]) : : "memory");
#pragma unroll
    for (int k = 0; k < K; k++) asm volatile("" : "+v"(pk[k]) : : "memory");
    store_w(0, wreg[0]);
    __syncthreads();

    u32x4_t breg[2][KI][2][NB];  // [set][ki][hi, lo][j] fragments of W[k], set = k & 1
    u32x4_t areg[2][KI][2];      // [set][ki][hi, lo] row fragments of one tile, set = (k * T + t) & 1
    // fragment f of W (order of first use: ki, then hi before lo, then j)
    auto read_b1 = [&](int buf, int f, u32x4_t (&bq)[KI][2][NB]) {
      const int ki = f / (2 * NB), hl = (f / NB) & 1, j = f % NB;
      bq[ki][hl][j] = *(reinterpret_cast<const u32x4_t*>(wbuf + buf * WBYTES) + lane + (size_t)((ki * NB + j) * 2 + hl) * 64);
    };
    int a_slot = 0;  // slot of the tile whose fragments are being read
    auto a_prepare = [&](const i32x2_t p, int t) {
      const int raw = ((t & 2 ? p.y : p.x) >> (16 * (t & 1))) & 0xFFFF;
      a_slot = raw == (int)BRK_ABSENT ? BRK_UMAX : raw;
    };
    // fragment f of the prepared tile (order of first use: ki, then lo before hi)
    auto read_a1 = [&](int f, u32x4_t (&aq)[KI][2]) {
      const int ki = f >> 1, hl = (f & 1) ^ 1;
      aq[ki][hl] = *reinterpret_cast<const u32x4_t*>(halo + a_slot * ROWB + (((hl ? CH / 2 : 0) + ki * 4 + kg + a_slot) & (CH - 1)) * 16);
    };
#pragma unroll
    for (int f = 0; f < NF; f++) read_b1(0, f, breg[0]);
    a_prepare(pk[0], 0);
#pragma unroll
    for (int f = 0; f < 2 * KI; f++) read_a1(f, areg[0]);

    // One offset = T tiles x G groups of NB MFMAs; behind every group ONE slot of other work, so that the wave never issues more
    // than a few non-matrix instructions in a row (one wave per SIMD: whatever it issues between two MFMAs beyond ~3 slots idles the
    // matrix pipe): W[k + 2] requests, W[k + 1] registers -> LDS, the barrier, W[k + 1] fragments -> registers, the next tile's rows.
    constexpr int G = KI * 3, S = T * G;           // groups per tile, slots per step
    constexpr int Q_ST = WPT, Q_BAR = 2 * WPT, Q_B = 2 * WPT + 1;  // first slot of: the LDS stores, the barrier, the fragment reads
    static_assert(Q_B < S, "slot schedule");
    constexpr int BPS = (NF + (S - Q_B) - 1) / (S - Q_B);  // W fragment reads per slot
    auto step = [&](auto kc) {
      constexpr int k = decltype(kc)::value;
      constexpr int cur = k & 1;
#pragma unroll
      for (int t = 0; t < T; t++) {
        const int aset = (k * T + t) & 1;
        const bool a_next = t + 1 < T || k + 1 < K;
        if (a_next) a_prepare(t + 1 < T ? pk[k] : pk[k + 1 < K ? k + 1 : k], t + 1 < T ? t + 1 : 0);
#pragma unroll
        for (int g = 0; g < G; g++) {
          const int ki = g / 3, term = g % 3, q = t * G + g;
#pragma unroll
          for (int j = 0; j < NB; j++) {  // smallest terms first: lo * Wh, hi * Wl, hi * Wh
#if BRK_DBG & 1
            asm volatile("" ::"v"(areg[aset][ki][term == 0 ? 1 : 0]), "v"(breg[cur][ki][term == 1 ? 1 : 0][j]));
#else
            acc[t][j] = sp_mfma<PREC>(areg[aset][ki][term == 0 ? 1 : 0], breg[cur][ki][term == 1 ? 1 : 0][j], acc[t][j]);
#endif
          }
          // ---- the slot behind this group
#if !(BRK_DBG & 8)
          if constexpr (k + 2 < K) {
            if (q < WPT) asm_gld16(wreg[cur][q], reinterpret_cast<const f32x4*>(wimg) + (size_t)(k + 2) * NF * 64 + tid + q * NW * 64);
          }
          if constexpr (k + 1 < K) {
            if (q == Q_ST) {  // W[k + 1]: everything but this step's own requests has landed
              if constexpr (k + 2 < K) vm_wait_arr<WPT>(wreg[cur ^ 1]);
              else vm_wait_arr<0>(wreg[cur ^ 1]);
            }
            if (q >= Q_ST && q < Q_ST + WPT)
              (reinterpret_cast<f32x4*>(wbuf + (cur ^ 1) * WBYTES) + tid)[(size_t)(q - Q_ST) * NW * 64] = wreg[cur ^ 1][q - Q_ST];
            if (q == Q_BAR) __syncthreads();
#endif
#if !(BRK_DBG & 4)
            if (q >= Q_B) {
#pragma unroll
              for (int f = (q - Q_B) * BPS; f < (q - Q_B + 1) * BPS && f < NF; f++) read_b1(cur ^ 1, f, breg[cur ^ 1]);
            }
#endif
#if !(BRK_DBG & 8)
          }
#endif
#if !(BRK_DBG & 2)
          if (a_next && g < 2 * KI) read_a1(g, areg[aset ^ 1]);
#endif
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    };
    v3d_static_for<K>(step);
  } else {
    // ---- a neighbourhood beyond the LDS slots: plain direct gathers (fragments straight from global memory), same products, same order
    for (int k = 0; k < K; k++) {
      const u32x4_t* bw = reinterpret_cast<const u32x4_t*>(wimg) + (size_t)k * NF * 64 + lane;
#pragma unroll
      for (int t = 0; t < T; t++) {
        const int row = row0 + t * 16 + r;
        const int src = row < n ? nbr[(size_t)k * cap + row] : -1;
        if (__ballot(src >= 0) == 0ull) continue;
#pragma unroll
        for (int ki = 0; ki < KI; ki++) {
          u32x4_t ah = u32x4_t{0u, 0u, 0u, 0u}, al = u32x4_t{0u, 0u, 0u, 0u};
          if (src >= 0) {
            const unsigned char* p = in_s + (size_t)src * ROWB + (ki * 4 + kg) * 16;
            ah = *reinterpret_cast<const u32x4_t*>(p);
            al = *reinterpret_cast<const u32x4_t*>(p + ROWB / 2);
          }
#pragma unroll
          for (int j = 0; j < NB; j++) acc[t][j] = sp_mfma<PREC>(al, bw[(size_t)((ki * NB + j) * 2) * 64], acc[t][j]);
#pragma unroll
          for (int j = 0; j < NB; j++) acc[t][j] = sp_mfma<PREC>(ah, bw[(size_t)((ki * NB + j) * 2 + 1) * 64], acc[t][j]);
#pragma unroll
          for (int j = 0; j < NB; j++) acc[t][j] = sp_mfma<PREC>(ah, bw[(size_t)((ki * NB + j) * 2) * 64], acc[t][j]);
        }
      }
    }
  }

  // epilogue: scale / shift / ReLU on the accumulators (D[row = kg*4 + rr][col = j*16 + r]), rows stored coalesced through LDS
  // (sp_device.h: sp_tile_store_*); the neighbourhood and weight buffers are dead by now
  __syncthreads();
  unsigned char* stage = halo + wave * SP_STAGE_BYTES(COUT);
  static_assert(BRK_NW * SP_STAGE_BYTES(COUT) <= (BRK_UMAX + 1) * ROWB, "staging blocks inside the neighbourhood buffer");
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
#if !(BRK_DBG & 32)
      if (out) sp_tile_store_f32<COUT, NB>(stage, out + (size_t)(row0 + t * 16) * COUT, v, nv, lane);
      if (out_s) sp_tile_store_split<PREC, COUT, NB>(stage, out_s + (size_t)(row0 + t * 16) * (2 * COUT), v, s_next, nv, lane);
#endif
    }
  }
  if constexpr (PREC == 1) sp_range_check(as, vmax, ss.limit);
}

template <int CIN, int COUT, int PREC>
static int launch_brick(const void* in_s, const void* wimg, const int* nbr, const V3dBrickTables& bt, const int* n_ptr, int cap,
                        const float* scale, const float* shift, int relu, float* out, const V3dActScale& as, void* out_s, hipStream_t st) {
  hipLaunchKernelGGL((spconv_fwd_brick<CIN, COUT, PREC>), dim3(v3d_ceil_div(cap, BRK_ROWS)), dim3(BRK_NW * 64), 0, st, (const unsigned char*)in_s,
                     (const unsigned short*)wimg, nbr, bt, n_ptr, cap, scale, shift, relu, out, as, (unsigned short*)out_s);
  V3D_CHECK_LAUNCH();
  return V3D_OK;
}

int v3d_i_sparse_conv_fwd_brick(const void* in_split, const void* weight_image, const int32_t* nbr, const V3dBrickTables& bt,
                                const int32_t* n_out, int cap, int K, int Cin, int Cout, const float* scale, const float* shift, int relu,
                                float* out, int prec, const V3dActScale* act, void* out_split, hipStream_t st) {
  if (!in_split || !weight_image || !nbr || !n_out || (!out && !out_split) || cap < 1 || !bt.lidx || !bt.ulist || !bt.ucnt || !bt.tmask)
    return V3D_EINVAL;
  if ((scale == nullptr) != (shift == nullptr)) return V3D_EINVAL;
  if (prec != V3D_PREC_BF16X3 && prec != V3D_PREC_F16S) return V3D_EINVAL;
  if (prec == V3D_PREC_F16S && (!act || !act->in || (out_split && !act->next))) return V3D_EINVAL;
  if (K != 27) return V3D_EUNSUPPORTED;
  const V3dActScale as = prec == V3D_PREC_F16S ? *act : V3dActScale{nullptr, nullptr, nullptr, nullptr};
#define V3D_TRY(ci, co)                                                                                                              \
  if (Cin == ci && Cout == co)                                                                                                       \
    return prec == V3D_PREC_F16S ? launch_brick<ci, co, 1>(in_split, weight_image, nbr, bt, n_out, cap, scale, shift, relu, out, as, out_split, st) \
                                 : launch_brick<ci, co, 0>(in_split, weight_image, nbr, bt, n_out, cap, scale, shift, relu, out, as, out_split, st);
  V3D_TRY(64, 64)
  V3D_TRY(32, 32)
#undef V3D_TRY
  return V3D_EUNSUPPORTED;
}

extern "C" int v3d_sparse_conv_fwd_brick(const void* in_split, const void* weight_image, const int32_t* nbr, const uint16_t* lidx,
                                         const int32_t* ulist, const int32_t* ucnt, const uint32_t* tmask, const int32_t* n_out,
                                         int cap, int K, int Cin, int Cout, const float* scale, const float* shift, int relu, float* out,
                                         int prec, const float* act_in, const float* act_next, int32_t* range_flag, void* out_split,
                                         v3d_stream_t stream) {
  const V3dActScale as{act_in, act_next, range_flag, nullptr};
  const V3dBrickTables bt{const_cast<uint16_t*>(lidx), const_cast<int32_t*>(ulist), const_cast<int32_t*>(ucnt), const_cast<uint32_t*>(tmask)};
  return v3d_i_sparse_conv_fwd_brick(in_split, weight_image, nbr, bt, n_out, cap, K, Cin, Cout, scale, shift, relu, out, prec, &as,
                                     out_split, (hipStream_t)stream);
}
