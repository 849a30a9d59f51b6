// dense_conv.hip -- dense 2-D convolutions of the BEV head (A8: RPN, A9: 1x1 heads) on the matrix cores.
//
// Reference: nn.Conv2d + BatchNorm2d + ReLU stacks (vision3d/detector/second.py:58-94) and the 1x1 heads
// (detector/proposal.py:19-22), executed by cuDNN in fp32.  This is the only genuinely dense, GEMM-shaped
// part of the model (63.4 GFLOP/frame), so it is the part that goes to MFMA (BASELINE.json north_star).
//
// Precision: every fp32 operand is split once into 16-bit pieces hi = rne(x), lo = rne(x - hi) and the product is evaluated as
// lo*hi + hi*lo + hi*hi with fp32 accumulation on the 16-bit matrix pipe.  Activations travel between layers ALREADY split (two
// 16-bit NHWC planes = the bytes of one fp32 tensor), weights are split and packed once per model, so the inner loop has no
// conversions.  Two arithmetics (template parameter PREC; spconv.hip "the split-precision product" has the measurements):
//   PREC 0 "bf16x3"  bf16 pieces, 2^-17 per product, scale-free.
//   PREC 1 "f16s"    f16 pieces of x * s under a power-of-two scale s per tensor (weights: per layer, chosen at pack time, inverse
//                    in the image's trailer; activations: static per layer, a device entry {s, 1/s, limit, ..} set by calibration --
//                    runtime.DenseHeadPlan.calibrate -- with headroom; an output beyond its entry's limit raises the frame's range
//                    flag): 2^-22 per product, i.e. the fp32 modules' results up to fp32 summation noise, at the same MFMA count.
//                    Scaling by powers of two is exact: results do not depend on the scales while nothing leaves the f16 range.
//
// Kernel: implicit GEMM, M = B*H*W pixels (flattened), N = Cout, K = taps * Cin.
//   workgroup  64 pixels x 128 couts, 4 waves as 2 (pixel halves) x 2 (cout halves); wave tile 32 x 64
//              = 2 x 4 MFMA fragments, 8 fp32 accumulators (32 VGPRs)
//   k-step     32 input channels of one tap: A (64 px x 32 ch x {hi,lo}) gathered with zero halo,
//              B (32 ch x 128 cout x {hi,lo}) copied linearly from the pre-packed weight image
//   LDS        B: fragment-major image (lane-linear ds_read_b128); A: pixel-major rows with an XOR swizzle that is
//              conflict-free for the coalesced writes and the fragment reads; double buffered, one barrier per
//              k-step; global loads run two k-steps ahead in two register sets.  (An 8-wave variant of the same
//              tile measured 70 us vs 57.6 us: the limiter is L1/L2 fill bandwidth -- weights are re-streamed per
//              64-pixel tile -- not latency.)
//   epilogue   accumulators -> LDS (fp32) -> bias + ReLU -> either the next layer's split bf16 NHWC planes
//              (16-byte stores) or fp32 NCHW for the consumer outside this file.
#include "v3d_internal.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef unsigned short bf16_t;  // storage type of a 16-bit piece at the C ABI (bf16 or f16 by the arithmetic)
// native vector type for the 16-byte staging registers: HIP's uint4 is a struct with a union inside and an
// array of them is NOT promoted to registers (it round-tripped through scratch every k-step: 142 us/conv)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <int PREC>
__device__ __forceinline__ f32x4 dc_mfma(const u32x4 a, const u32x4 b, const f32x4 c) {
  if constexpr (PREC == 0)
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

#define DC_BM 64
#define DC_BN 128
#define DC_KC 32  // input channels per k-step
#define DC_THREADS 256

__device__ __forceinline__ bf16_t f32_to_bf16_rne(float f) {
  unsigned u = __float_as_uint(f);
  if ((u & 0x7F800000u) == 0x7F800000u) return (bf16_t)(u >> 16);  // inf / nan: truncate
  u += 0x7FFFu + ((u >> 16) & 1u);
  return (bf16_t)(u >> 16);
}
__device__ __forceinline__ float bf16_to_f32(bf16_t h) { return __uint_as_float(((unsigned)h) << 16); }
// one value -> (hi, lo) pieces; PREC 1: of x * s
template <int PREC>
__device__ __forceinline__ void split_val(float x, float s, bf16_t& hi, bf16_t& lo) {
  if constexpr (PREC == 0) {
    hi = f32_to_bf16_rne(x);
    lo = f32_to_bf16_rne(x - bf16_to_f32(hi));
  } else {
    const float a = x * s;
    const _Float16 h = (_Float16)a;
    const _Float16 l = (_Float16)(a - (float)h);
    hi = __builtin_bit_cast(bf16_t, h);
    lo = __builtin_bit_cast(bf16_t, l);
  }
}
// trailer of a packed weight image: {max|w| bits, 1/s_w, s_w, precision}; bf16x3 images carry it unused
#define DC_WIMG_TRAILER 256
#define DC_F16S_WEIGHT_TARGET 13  // max|w| * s_w in [2^13, 2^14)
__host__ __device__ static inline float dc_pow2_scale(unsigned amax_bits, int target) {
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

// ------------------------------------------------------------------------------------------------
// weights (Cout, Cin, kh, kw) fp32 [+ per-cout scale folded in] -> packed split image
// image[s][plane][nf][kg][j][e]  with  s = tap*(Cin/32) + chunk, plane in {hi, lo}, nf = cout/16,
// kg = (cin%32)/8, j = cout%16, e = cin%8   -- exactly the byte order of the B tile in LDS.
// ------------------------------------------------------------------------------------------------
// max |w * scale[co]| into word 0 of the trailer (zeroed by the caller)
__global__ void dc_wmax_kernel(const float* __restrict__ w, const float* __restrict__ scale, int Cout, long long per_cout,
                               unsigned* __restrict__ trailer) {
  unsigned m = 0u;
  const long long n = (long long)Cout * per_cout;
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (long long)gridDim.x * blockDim.x) {
    float v = w[t];
    if (scale) v *= scale[t / per_cout];
    m = max(m, __float_as_uint(v) & 0x7FFFFFFFu);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
  if ((threadIdx.x & 63) == 0 && m) atomicMax(trailer, m);
}

template <int PREC>
__global__ void dc_pack_weights_kernel(const float* __restrict__ w, const float* __restrict__ scale, int Cout, int Cin,
                                       int ks, int CoutPad, bf16_t* __restrict__ img) {
  const int taps = ks * ks, chunks = Cin / DC_KC;
  const long long total = (long long)taps * chunks * (CoutPad / 16) * 4 * 16 * 8;
  float sw = 1.f;
  if constexpr (PREC == 1) {
    unsigned* trailer = reinterpret_cast<unsigned*>(img + total * 2);
    sw = dc_pow2_scale(trailer[0], DC_F16S_WEIGHT_TARGET);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      reinterpret_cast<float*>(trailer)[1] = 1.f / sw;
      reinterpret_cast<float*>(trailer)[2] = sw;
      trailer[3] = 1u;
    }
  }
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
    long long r = t;
    const int e = (int)(r % 8); r /= 8;
    const int j = (int)(r % 16); r /= 16;
    const int kg = (int)(r % 4); r /= 4;
    const int nf = (int)(r % (CoutPad / 16)); r /= (CoutPad / 16);
    const int chunk = (int)(r % chunks); r /= chunks;
    const int tap = (int)r;
    const int co = nf * 16 + j, ci = chunk * DC_KC + kg * 8 + e;
    float v = 0.f;
    if (co < Cout) {
      v = w[((size_t)co * Cin + ci) * taps + tap];
      if (scale) v *= scale[co];
    }
    bf16_t hi, lo;
    split_val<PREC>(v, sw, hi, lo);
    const size_t step = (size_t)tap * chunks + chunk;
    const size_t plane_elems = (size_t)(CoutPad / 16) * 4 * 16 * 8;
    const size_t off = ((size_t)(nf * 4 + kg) * 16 + j) * 8 + e;
    img[step * 2 * plane_elems + off] = hi;
    img[step * 2 * plane_elems + plane_elems + off] = lo;
  }
}

static size_t dc_image_payload_bytes(int Cin, int Cout, int ksize) {
  const int pad = (Cout + DC_BN - 1) / DC_BN * DC_BN;
  return (size_t)ksize * ksize * (Cin / DC_KC) * 2 * (size_t)pad * DC_KC * sizeof(bf16_t);
}
extern "C" size_t v3d_conv2d_weight_image_bytes(int Cin, int Cout, int ksize) {
  return dc_image_payload_bytes(Cin, Cout, ksize) + DC_WIMG_TRAILER;
}

extern "C" int v3d_conv2d_pack_weights(const float* weight, const float* scale, int Cout, int Cin, int ksize, int prec,
                                        void* image, v3d_stream_t stream) {
  if (!weight || !image || Cout < 1 || Cin < DC_KC || Cin % DC_KC || (ksize != 1 && ksize != 3)) return V3D_EINVAL;
  if (prec != V3D_PREC_BF16X3 && prec != V3D_PREC_F16S) return V3D_EINVAL;
  const int pad = (Cout + DC_BN - 1) / DC_BN * DC_BN;
  hipStream_t st = (hipStream_t)stream;
  if (prec == V3D_PREC_F16S) {
    unsigned* trailer = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(image) + dc_image_payload_bytes(Cin, Cout, ksize));
    V3D_CHECK_HIP(v3d_fill_async(trailer, 0, DC_WIMG_TRAILER, st));
    hipLaunchKernelGGL(dc_wmax_kernel, dim3(128), dim3(256), 0, st, weight, scale, Cout, (long long)Cin * ksize * ksize, trailer);
    hipLaunchKernelGGL(dc_pack_weights_kernel<1>, dim3(512), dim3(256), 0, st, weight, scale, Cout, Cin, ksize, pad, (bf16_t*)image);
  } else {
    hipLaunchKernelGGL(dc_pack_weights_kernel<0>, dim3(512), dim3(256), 0, st, weight, scale, Cout, Cin, ksize, pad, (bf16_t*)image);
  }
  V3D_CHECK_LAUNCH();
  return V3D_OK;
}

// ------------------------------------------------------------------------------------------------
// .dense() straight into the conv input format: zero planes + scatter of split values.
// out[(b*H + y)*W + x][c*D + z]  (the reference's (B, C*D, H, W) view, channels innermost)
// ------------------------------------------------------------------------------------------------
template <int PREC>
__global__ __launch_bounds__(V3D_BLOCK) void densify_split_kernel(const float* __restrict__ feat,
                                                                  const int4* __restrict__ coords,
                                                                  const int* __restrict__ n_ptr, int cap, int C, int D,
                                                                  int H, int Wd, bf16_t* __restrict__ hi,
                                                                  bf16_t* __restrict__ lo, unsigned* __restrict__ occ,
                                                                  int* __restrict__ written_pix, int* __restrict__ written_n,
                                                                  const float* __restrict__ entry, int* __restrict__ range_flag,
                                                                  unsigned* __restrict__ seen) {
  const int n = min(*n_ptr, cap);
  const long long total = (long long)n * C;
  const float s_out = PREC == 1 ? entry[0] : 1.f, limit = PREC == 1 ? entry[2] : 0.f;
  float vmax = 0.f;
  if (written_n && blockIdx.x == 0 && threadIdx.x == 0) *written_n = n;
  for (long long t = (long long)blockIdx.x * V3D_BLOCK + threadIdx.x; t < total; t += (long long)gridDim.x * V3D_BLOCK) {
    const int i = (int)(t / C), ch = (int)(t % C);
    const int4 c = coords[i];
    if (occ && ch == 0)  // the occupancy bitmap of the background-skipping head rides along (inverted bits, see bev_occupancy_kernel)
      atomicAnd(occ + ((size_t)c.x * H + c.z) * ((Wd + 31) >> 5) + (c.w >> 5), ~(1u << (c.w & 31)));
    // PERSISTENT planes (second_plan.hip): the pixels this frame writes, so that the next frame zeroes exactly those
    if (written_pix && ch == 0) written_pix[i] = (c.x * H + c.z) * Wd + c.w;
    const size_t o = (((size_t)c.x * H + c.z) * Wd + c.w) * ((size_t)C * D) + (size_t)ch * D + c.y;
    bf16_t h, l;
    const float v = feat[t];
    if constexpr (PREC == 1) vmax = fmaxf(vmax, fabsf(v));
    split_val<PREC>(v, s_out, h, l);
    hi[o] = h;
    lo[o] = l;
  }
  if constexpr (PREC == 1) {
    if (range_flag && vmax > limit) atomicMax(range_flag, V3D_FLAG_RANGE);
    if (seen) v3d_mark_seen(seen, vmax, limit * (1.f / (float)(1 << V3D_QUIET_BITS)));
  }
}

extern "C" int v3d_densify_nhwc_split(const float* feat, const int32_t* coords, const int32_t* n, int cap, int B, int C,
                                      const int32_t* spatial_shape_host, void* out_hi, void* out_lo,
                                      v3d_stream_t stream) {
  return v3d_i_densify_nhwc_split(feat, coords, n, cap, B, C, spatial_shape_host, out_hi, out_lo, nullptr, (hipStream_t)stream);
}

// Zero the pixels a list names in both planes (16-byte stores): the start-of-frame job of a plan's PERSISTENT BEV planes.
__global__ __launch_bounds__(V3D_BLOCK) void bev_clear_pixels_kernel(const int* __restrict__ pix, const int* __restrict__ n_ptr, int cap,
                                                                     int units /*16-byte units per pixel and plane*/,
                                                                     uint4* __restrict__ hi, uint4* __restrict__ lo) {
  const long long total = (long long)min(*n_ptr, cap) * units;
  for (long long t = (long long)blockIdx.x * V3D_BLOCK + threadIdx.x; t < total; t += (long long)gridDim.x * V3D_BLOCK) {
    const size_t o = (size_t)pix[t / units] * units + (size_t)(t % units);
    hi[o] = make_uint4(0u, 0u, 0u, 0u);
    lo[o] = make_uint4(0u, 0u, 0u, 0u);
  }
}

int v3d_i_bev_clear_pixels(const int32_t* pix, const int32_t* n, int cap, int channels, void* hi, void* lo, hipStream_t st) {
  if (!pix || !n || cap < 1 || channels < 8 || channels % 8 || !hi || !lo) return V3D_EINVAL;
  const int units = channels / 8;
  const long long total = (long long)cap * units;
  hipLaunchKernelGGL(bev_clear_pixels_kernel, dim3((int)std::min<long long>(v3d_ceil_div(total, V3D_BLOCK), 2048)), dim3(V3D_BLOCK), 0, st,
                     pix, n, cap, units, (uint4*)hi, (uint4*)lo);
  V3D_CHECK_LAUNCH();
  return V3D_OK;
}

// occ_inv (nullable): inverted occupancy bitmap, ALREADY filled with 0xFF by the caller; bits of the occupied pixels are cleared
// written_pix / written_n (nullable, together): the planes are PERSISTENT and already zero outside the pixels the caller has just
// cleared (v3d_i_bev_clear_pixels on the list the previous call left here): no fill, and this call's pixel list is left behind.
int v3d_i_densify_nhwc_split(const float* feat, const int32_t* coords, const int32_t* n, int cap, int B, int C,
                             const int32_t* spatial_shape_host, void* out_hi, void* out_lo, uint32_t* occ_inv, hipStream_t st,
                             int32_t* written_pix, int32_t* written_n, int prec, const float* act_entry, int32_t* range_flag,
                             unsigned* seen) {
  if (!feat || !coords || !n || cap < 1 || B < 1 || C < 1 || !spatial_shape_host || !out_hi || !out_lo) return V3D_EINVAL;
  if (prec == V3D_PREC_F16S ? !act_entry : prec != V3D_PREC_BF16X3) return V3D_EINVAL;
  if ((written_pix == nullptr) != (written_n == nullptr)) return V3D_EINVAL;
  const int D = spatial_shape_host[0], H = spatial_shape_host[1], Wd = spatial_shape_host[2];
  const size_t bytes = (size_t)B * H * Wd * C * D * sizeof(bf16_t);
  if (written_pix) {
    // (nothing to fill)
  } else if ((char*)out_lo == (char*)out_hi + bytes) {  // planes allocated back to back: one launch
    V3D_CHECK_HIP(v3d_fill_async(out_hi, 0, 2 * bytes, st));
  } else {
    V3D_CHECK_HIP(v3d_fill_async(out_hi, 0, bytes, st));
    V3D_CHECK_HIP(v3d_fill_async(out_lo, 0, bytes, st));
  }
  const long long total = (long long)cap * C;
  const int blocks = (int)((total + V3D_BLOCK - 1) / V3D_BLOCK);
  if (prec == V3D_PREC_F16S)
    hipLaunchKernelGGL(densify_split_kernel<1>, dim3(blocks > 4096 ? 4096 : blocks), dim3(V3D_BLOCK), 0, st, feat, (const int4*)coords, n,
                       cap, C, D, H, Wd, (bf16_t*)out_hi, (bf16_t*)out_lo, occ_inv, written_pix, written_n, act_entry, range_flag, seen);
  else
    hipLaunchKernelGGL(densify_split_kernel<0>, dim3(blocks > 4096 ? 4096 : blocks), dim3(V3D_BLOCK), 0, st, feat, (const int4*)coords, n,
                       cap, C, D, H, Wd, (bf16_t*)out_hi, (bf16_t*)out_lo, occ_inv, written_pix, written_n, nullptr, nullptr, nullptr);
  V3D_CHECK_LAUNCH();
  return V3D_OK;
}

// fp32 NCHW -> split NHWC planes (entry for callers that hold a torch-style tensor); f16s: pieces of x * entry[0]
template <int PREC>
__global__ void nchw_to_split_nhwc_kernel(const float* __restrict__ x, int B, int C, int HW, bf16_t* __restrict__ hi,
                                          bf16_t* __restrict__ lo, const float* __restrict__ entry) {
  const long long total = (long long)B * C * HW;
  const float s = PREC == 1 ? entry[0] : 1.f;
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(t % C);
    const long long bp = t / C;  // b*HW + p
    const int b = (int)(bp / HW), p = (int)(bp % HW);
    bf16_t h, l;
    split_val<PREC>(x[((size_t)b * C + c) * HW + p], s, h, l);
    hi[t] = h;
    lo[t] = l;
  }
}

// The same through an LDS tile (even C <= 128 per pass): 64 pixels of one image x all channels -- rows of 64 consecutive pixels read
// per channel (256-byte runs), packed channel pairs written per pixel (256-byte rows of each plane).  The plain kernel above reads
// with a stride of H * W floats between lanes: 0.9 TB/s on an 8-image map, 323 us of every fp32-class train step.
#define NS_PX 64
template <int PREC>
__global__ __launch_bounds__(256) void nchw_to_split_nhwc_tiled_kernel(const float* __restrict__ x, int B, int C, int HW, bf16_t* __restrict__ hi,
                                                                       bf16_t* __restrict__ lo, const float* __restrict__ entry) {
  __shared__ float tile[NS_PX][129];
  const float s = PREC == 1 ? entry[0] : 1.f;
  const int tpi = (HW + NS_PX - 1) / NS_PX, tid = threadIdx.x;
  for (long long idx = blockIdx.x; idx < (long long)B * tpi; idx += gridDim.x) {
    const int b = (int)(idx / tpi), p0 = (int)(idx % tpi) * NS_PX;
    for (int c0 = 0; c0 < C; c0 += 128) {
      const int cn = min(128, C - c0);
      __syncthreads();
      {
        const int p = tid & 63;
        if (p0 + p < HW)
          for (int c = tid >> 6; c < cn; c += 4) tile[p][c] = x[((size_t)b * C + c0 + c) * HW + p0 + p];
      }
      __syncthreads();
      const int cp = tid & 63;  // channels 2 cp, 2 cp + 1
      if (2 * cp < cn)
        for (int p = tid >> 6; p < NS_PX && p0 + p < HW; p += 4) {
          bf16_t h0, l0, h1, l1;
          split_val<PREC>(tile[p][2 * cp], s, h0, l0);
          split_val<PREC>(tile[p][2 * cp + 1], s, h1, l1);
          const size_t o = ((size_t)b * HW + p0 + p) * C + c0 + 2 * cp;
          *reinterpret_cast<unsigned*>(hi + o) = (unsigned)h0 | ((unsigned)h1 << 16);
          *reinterpret_cast<unsigned*>(lo + o) = (unsigned)l0 | ((unsigned)l1 << 16);
        }
    }
  }
}

// bf16 split NHWC planes -> fp32 NCHW (hi + lo): the way back (the gradient of the BEV map leaving the fp32-class dense train step)
__global__ __launch_bounds__(256) void split_nhwc_to_nchw_kernel(const bf16_t* __restrict__ hi, const bf16_t* __restrict__ lo, int B, int C, int HW,
                                                                 float* __restrict__ x) {
  __shared__ float tile[NS_PX][129];
  const int tpi = (HW + NS_PX - 1) / NS_PX, tid = threadIdx.x;
  for (long long idx = blockIdx.x; idx < (long long)B * tpi; idx += gridDim.x) {
    const int b = (int)(idx / tpi), p0 = (int)(idx % tpi) * NS_PX;
    for (int c0 = 0; c0 < C; c0 += 128) {
      const int cn = min(128, C - c0);
      __syncthreads();
      const int cp = tid & 63;
      if (2 * cp < cn)
        for (int p = tid >> 6; p < NS_PX && p0 + p < HW; p += 4) {
          const size_t o = ((size_t)b * HW + p0 + p) * C + c0 + 2 * cp;
          const unsigned h = *reinterpret_cast<const unsigned*>(hi + o), l = *reinterpret_cast<const unsigned*>(lo + o);
          tile[p][2 * cp] = __uint_as_float(h << 16) + __uint_as_float(l << 16);
          tile[p][2 * cp + 1] = __uint_as_float(h & 0xFFFF0000u) + __uint_as_float(l & 0xFFFF0000u);
        }
      __syncthreads();
      const int p = tid & 63;
      if (p0 + p < HW)
        for (int c = tid >> 6; c < cn; c += 4) x[((size_t)b * C + c0 + c) * HW + p0 + p] = tile[p][c];
    }
  }
}

extern "C" int v3d_split_nhwc_to_nchw(const void* x_hi, const void* x_lo, int B, int C, int H, int W, float* out, v3d_stream_t stream) {
  if (!x_hi || !x_lo || !out || B < 1 || C < 2 || (C & 1) || H < 1 || W < 1) return V3D_EINVAL;
  const long long tiles = (long long)B * ((H * W + NS_PX - 1) / NS_PX);
  hipLaunchKernelGGL(split_nhwc_to_nchw_kernel, dim3((unsigned)std::min<long long>(tiles, 8192)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)x_hi, (const bf16_t*)x_lo, B, C, H * W, out);
  V3D_CHECK_LAUNCH();
  return V3D_OK;
}

extern "C" int v3d_nchw_to_split_nhwc(const float* x, int B, int C, int H, int W, void* out_hi, void* out_lo, int prec,
                                       const float* act_entry, v3d_stream_t stream) {
  if (!x || !out_hi || !out_lo || B < 1 || C < 1 || H < 1 || W < 1) return V3D_EINVAL;
  if (prec == V3D_PREC_F16S ? !act_entry : prec != V3D_PREC_BF16X3) return V3D_EINVAL;
  if (!(C & 1)) {
    const long long tiles = (long long)B * ((H * W + NS_PX - 1) / NS_PX);
    const dim3 grid((unsigned)std::min<long long>(tiles, 8192));
    if (prec == V3D_PREC_F16S)
      hipLaunchKernelGGL(nchw_to_split_nhwc_tiled_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, x, B, C, H * W, (bf16_t*)out_hi,
                         (bf16_t*)out_lo, act_entry);
    else
      hipLaunchKernelGGL(nchw_to_split_nhwc_tiled_kernel<0>, grid, dim3(256), 0, (hipStream_t)stream, x, B, C, H * W, (bf16_t*)out_hi,
                         (bf16_t*)out_lo, nullptr);
    V3D_CHECK_LAUNCH();
    return V3D_OK;
  }
  if (prec == V3D_PREC_F16S)
    hipLaunchKernelGGL(nchw_to_split_nhwc_kernel<1>, dim3(2048), dim3(256), 0, (hipStream_t)stream, x, B, C, H * W, (bf16_t*)out_hi,
                       (bf16_t*)out_lo, act_entry);
  else
    hipLaunchKernelGGL(nchw_to_split_nhwc_kernel<0>, dim3(2048), dim3(256), 0, (hipStream_t)stream, x, B, C, H * W, (bf16_t*)out_hi,
                       (bf16_t*)out_lo, nullptr);
  V3D_CHECK_LAUNCH();
  return V3D_OK;
}

// ------------------------------------------------------------------------------------------------
// the convolution
// ------------------------------------------------------------------------------------------------
struct DcParams {
  int B, H, W, Cin, Cout, CoutPad, ks, relu;
  int M;          // B*H*W
  int cout_store; // channels actually written
  // background skipping (large-tile kernel, split-plane output only; see v3d_conv2d_nhwc_split: occ)
  const unsigned* occ;   // BEV occupancy, one bit per pixel, INVERTED (0 = occupied), rows of ceil(W / 32) words (nullptr:
                         // compute every tile)
  int reach;             // output pixels further than this (Chebyshev) from every occupied pixel equal the empty-map response
  const bf16_t* bg_hi;   // that response for ONE image, (H, W, Cout) split planes
  const bf16_t* bg_lo;
  unsigned* work;        // [2] tile counter + workgroups-done counter of the persistent grid (zero between launches)
  unsigned* reset_ptr;   // nullable.  NULL: the pair resets ITSELF (last workgroup out -- every workgroup then ends on an atomic round
  int reset_words;       // trip, ~1.5 us at the end of every launch).  Else: this launch zeroes reset_words words at reset_ptr when it
                         // starts -- the counters of OTHER call sites, whose launches lie behind it in stream order -- and leaves
                         // its own pair for a later launch to zero (runtime.DenseHeadPlan chains the layers that way)
  unsigned* tile_state;  // nullable, one word per tile of a PERSISTENT output buffer: nonzero = the tile holds computed values.
                         // The empty-map response of a layer does not depend on the frame, so a background tile whose state
                         // is 0 already holds it from an earlier frame and is not written at all.
  // f16s arithmetic (PREC 1; null / unused for bf16x3): device scale entries {s, 1/s, limit, ..} of the input planes and of the
  // output planes (null when only the fp32 NCHW tensor is written), the weight image's trailer {.., 1/s_w, ..} and the frame's
  // range flag (nullable)
  const float* in_entry;
  const float* out_entry;
  const float* w_inv;  // 1 / s_w: the caller's hot copy (v3d_conv2d_prec::w_inv) or the image's trailer (a cold line: ~1 us per launch)
  int* range_flag;
};

// what an epilogue applies: v = acc * undo + bias; [ReLU]; planes <- pieces of v * s_out; |v| > limit raises the range flag
struct DcScales {
  float undo, s_out, limit;
};
template <int PREC>
__device__ __forceinline__ DcScales dc_scales(const DcParams& p) {
  DcScales r{1.f, 1.f, 3.0e38f};
  if constexpr (PREC == 1) {
    r.undo = p.in_entry[1] * p.w_inv[0];
    if (p.out_entry) {
      r.s_out = p.out_entry[0];
      r.limit = p.out_entry[2];
    }
  }
  return r;
}
template <int PREC>
__device__ __forceinline__ float dc_finish(float acc, float bias, const DcScales& sc) {
  if constexpr (PREC == 1) return acc * sc.undo + bias;
  else return acc + bias;
}
__device__ __forceinline__ void dc_range_check(const DcParams& p, float vmax, float limit) {
  if (p.range_flag && vmax > limit) atomicMax(p.range_flag, V3D_FLAG_RANGE);
}

template <int KS, int PREC>
__global__ __launch_bounds__(DC_THREADS, 2) void conv2d_bf16x3_kernel(const bf16_t* __restrict__ x_hi,
                                                                      const bf16_t* __restrict__ x_lo,
                                                                      const bf16_t* __restrict__ w_img,
                                                                      const float* __restrict__ bias, const DcParams p,
                                                                      bf16_t* __restrict__ y_hi, bf16_t* __restrict__ y_lo,
                                                                      float* __restrict__ y_nchw) {
  // LDS: 2 x (A 8 KB + B 16 KB) for the loop; reused as a 64 x 128 fp32 tile (32 KB) by the epilogue
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * (8192 + 16384)];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;  // pixel half / cout half
  // XCD-aware tile order: the dispatcher deals consecutive workgroups round-robin to the 8 XCDs (private L2s);
  // remap so each XCD walks a CONTIGUOUS run of pixel tiles -- neighbouring tiles share two of their three
  // input rows (the 3x3 halo), which then hit in that XCD's L2 instead of being fetched by eight of them.
  const int nt = gridDim.x;
  int mtile = blockIdx.x;
  {
    const int q = nt / 8, rmd = nt % 8, xcd = mtile % 8, idx = mtile / 8;
    mtile = (xcd < rmd ? xcd * (q + 1) : rmd * (q + 1) + (xcd - rmd) * q) + idx;  // bijective for any nt
  }
  const int m0 = mtile * DC_BM;
  const int n0 = blockIdx.y * DC_BN;
  const int chunks = p.Cin / DC_KC;
  const int steps = KS * KS * chunks;

  // ---- this thread's A-gather role: pixel a_px of the tile, 16-byte part a_kg of the 64-byte channel slice
  const int a_px = tid >> 2, a_kg = tid & 3;
  const int am = m0 + a_px;
  const bool a_live = am < p.M;
  int a_b = 0, a_h = 0, a_w = 0;
  if (a_live) {
    a_b = am / (p.H * p.W);
    const int rem = am - a_b * p.H * p.W;
    a_h = rem / p.W;
    a_w = rem - a_h * p.W;
  }
  // A image in LDS: pixel-major rows of 64 B (4 slots of 16 B) with the slot XOR-swizzled by bit 3 of the row:
  //   slot(px, kg) = px*4 + (kg ^ ((px & 8) >> 2))
  // found by exhaustive search to be conflict-free for BOTH access patterns: the coalesced writer (8 consecutive
  // lanes = 2 pixels x 4 parts -> 8 distinct 16-byte slots mod 128 B) and the MFMA fragment reader (each
  // ds_read_b128 lane group of 16 -> 16 distinct slots mod 256 B).  A fragment-major image had 4-way write
  // conflicts (35 % of all LDS cycles, PMC SQ_LDS_BANK_CONFLICT 4.2 M -> 0.4 M).
  const int a_dst = (a_px * 4 + (a_kg ^ ((a_px & 8) >> 2))) * 16;

  // staging registers: TWO sets -- the loads of step s+2 are issued while step s multiplies; step s+1's operands
  // (issued one step earlier) are written to the other LDS buffer at the end of step s
  struct Stage { u32x4 a0, a1, b0, b1, b2, b3; };
  Stage R0, R1;
  int ld_tap = 0, ld_chunk = 0;  // (tap, chunk) of the next step to load: steps are loaded strictly in order, so two
                                 // counters replace a per-step integer division
  auto load_step = [&](int s, Stage& R) {
    const int tap = ld_tap, chunk = ld_chunk;
    if (++ld_chunk == chunks) {
      ld_chunk = 0;
      ++ld_tap;
    }
    const int dy = KS == 3 ? tap / 3 - 1 : 0, dx = KS == 3 ? tap % 3 - 1 : 0;
    const int hh = a_h + dy, ww = a_w + dx;
    const bool ok = a_live && hh >= 0 && hh < p.H && ww >= 0 && ww < p.W;
    if (ok) {
      const size_t off = (((size_t)a_b * p.H + hh) * p.W + ww) * p.Cin + chunk * DC_KC + a_kg * 8;
      R.a0 = *reinterpret_cast<const u32x4*>(x_hi + off);
      R.a1 = *reinterpret_cast<const u32x4*>(x_lo + off);
    } else {
      R.a0 = u32x4{0, 0, 0, 0};
      R.a1 = u32x4{0, 0, 0, 0};
    }
    // B: 16 KB of this (step, cout tile): [plane][nf 8][kg 4][j 16][8] -> plane stride = CoutPad/16*4*16*8 elems
    const size_t plane_elems = (size_t)(p.CoutPad / 16) * 4 * 16 * 8;
    const bf16_t* wb = w_img + (size_t)s * 2 * plane_elems + (size_t)(n0 / 16) * 4 * 16 * 8;
    const bf16_t* wlo = wb + plane_elems;
    R.b0 = *reinterpret_cast<const u32x4*>(wb + (size_t)tid * 8);
    R.b1 = *reinterpret_cast<const u32x4*>(wb + (size_t)(tid + 256) * 8);
    R.b2 = *reinterpret_cast<const u32x4*>(wlo + (size_t)tid * 8);
    R.b3 = *reinterpret_cast<const u32x4*>(wlo + (size_t)(tid + 256) * 8);
  };
  auto store_step = [&](int buf, const Stage& R) {
    unsigned char* A = smem + buf * (8192 + 16384);
    unsigned char* Bm = A + 8192;
    *reinterpret_cast<u32x4*>(A + a_dst) = R.a0;          // hi plane
    *reinterpret_cast<u32x4*>(A + 4096 + a_dst) = R.a1;   // lo plane
    *reinterpret_cast<u32x4*>(Bm + tid * 16) = R.b0;
    *reinterpret_cast<u32x4*>(Bm + (tid + 256) * 16) = R.b1;
    *reinterpret_cast<u32x4*>(Bm + (tid + 512) * 16) = R.b2;
    *reinterpret_cast<u32x4*>(Bm + (tid + 768) * 16) = R.b3;
  };

  f32x4 acc[2][4];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto multiply = [&](int buf) {
    const unsigned char* A = smem + buf * (8192 + 16384);
    const unsigned char* Bm = A + 8192;
    u32x4 ah[2], al[2], bh[4], bl[4];
#pragma unroll
    for (int i = 0; i < 2; i++) {
      const int px = (wr * 2 + i) * 16 + (lane & 15);  // fragment row of this lane
      const int off = (px * 4 + ((lane >> 4) ^ ((lane & 8) >> 2))) * 16;
      ah[i] = *reinterpret_cast<const u32x4*>(A + off);
      al[i] = *reinterpret_cast<const u32x4*>(A + 4096 + off);
    }
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int nf = wc * 4 + j;
      bh[j] = *reinterpret_cast<const u32x4*>(Bm + (nf * 64 + lane) * 16);
      bl[j] = *reinterpret_cast<const u32x4*>(Bm + 8192 + (nf * 64 + lane) * 16);
    }
    // term-major order: 8 independent accumulators between two uses of the same one
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
      for (int j = 0; j < 4; j++) acc[i][j] = dc_mfma<PREC>(al[i], bh[j], acc[i][j]);
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
      for (int j = 0; j < 4; j++) acc[i][j] = dc_mfma<PREC>(ah[i], bl[j], acc[i][j]);
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
      for (int j = 0; j < 4; j++) acc[i][j] = dc_mfma<PREC>(ah[i], bh[j], acc[i][j]);
  };

  // prologue: step 0 -> LDS buffer 0, step 1 in flight in R1
  load_step(0, R0);
  if (steps > 1) load_step(1, R1);
  store_step(0, R0);
  __syncthreads();
  // step s: issue loads(s+2) -> multiply(s) -> store(s+1) (its loads were issued during step s-1: two multiplies
  // of cover) -> barrier.  Unrolled by two so each register set / LDS buffer has a fixed role.
  for (int s = 0; s < steps; s += 2) {
    if (s + 2 < steps) load_step(s + 2, R0);
    multiply(0);
    if (s + 1 < steps) store_step(1, R1);
    __syncthreads();
    if (s + 1 < steps) {
      if (s + 3 < steps) load_step(s + 3, R1);
      multiply(1);
      if (s + 2 < steps) store_step(0, R0);
      __syncthreads();
    }
  }

  // ---- epilogue: accumulators -> LDS tile [64 px][128 + 4 pad] fp32
  float* tile = reinterpret_cast<float*>(smem);
  constexpr int TS = DC_BN + 4;
  static_assert(DC_BM * TS * 4 <= 2 * (8192 + 16384), "epilogue tile must fit the loop buffers");
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int row = wr * 32 + i * 16 + (lane >> 4) * 4 + r;  // D: row = (lane>>4)*4 + r, col = lane&15
        const int col = wc * 64 + j * 16 + (lane & 15);
        tile[row * TS + col] = acc[i][j][r];
      }
  __syncthreads();
  const DcScales sc = dc_scales<PREC>(p);
  float vmax = 0.f;
  if (y_hi) {  // split 16-bit NHWC planes: thread handles 8 consecutive couts of one pixel, 16-byte stores
    for (int q = tid; q < DC_BM * (DC_BN / 8); q += DC_THREADS) {
      const int row = q / (DC_BN / 8), c8 = q % (DC_BN / 8);
      const int m = m0 + row, co = n0 + c8 * 8;
      if (m >= p.M || co >= p.cout_store) continue;
      u32x4 vh, vl;
#pragma unroll
      for (int e2 = 0; e2 < 4; e2++) {  // two channels per 32-bit word
        bf16_t h0, l0, h1, l1;
        float v0 = dc_finish<PREC>(tile[row * TS + c8 * 8 + 2 * e2], bias ? bias[co + 2 * e2] : 0.f, sc);
        float v1 = dc_finish<PREC>(tile[row * TS + c8 * 8 + 2 * e2 + 1], bias ? bias[co + 2 * e2 + 1] : 0.f, sc);
        if (p.relu) {
          v0 = fmaxf(v0, 0.f);
          v1 = fmaxf(v1, 0.f);
        }
        if constexpr (PREC == 1) vmax = fmaxf(vmax, fmaxf(fabsf(v0), fabsf(v1)));
        split_val<PREC>(v0, sc.s_out, h0, l0);
        split_val<PREC>(v1, sc.s_out, h1, l1);
        vh[e2] = (unsigned)h0 | ((unsigned)h1 << 16);
        vl[e2] = (unsigned)l0 | ((unsigned)l1 << 16);
      }
      *reinterpret_cast<u32x4*>(y_hi + (size_t)m * p.cout_store + co) = vh;
      *reinterpret_cast<u32x4*>(y_lo + (size_t)m * p.cout_store + co) = vl;
    }
    if constexpr (PREC == 1) dc_range_check(p, vmax, sc.limit);
  }
  if (y_nchw) {  // fp32 (B, cout_store, H, W): thread handles 4 consecutive pixels of one cout
    const int HW = p.H * p.W;
    for (int q = tid; q < (DC_BM / 4) * DC_BN; q += DC_THREADS) {
      const int col = q / (DC_BM / 4), r4 = q % (DC_BM / 4);
      const int co = n0 + col;
      if (co >= p.cout_store) continue;
      const float bv = bias ? bias[co] : 0.f;
#pragma unroll
      for (int e = 0; e < 4; e++) {
        const int m = m0 + r4 * 4 + e;
        if (m >= p.M) continue;
        float v = dc_finish<PREC>(tile[(r4 * 4 + e) * TS + col], bv, sc);
        if (p.relu) v = fmaxf(v, 0.f);
        const int b = m / HW, pix = m - b * HW;
        y_nchw[((size_t)b * p.cout_store + co) * HW + pix] = v;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Large-tile variant (Cin >= 64, a multiple of 32): 144 pixels x 128 couts per workgroup, wave-specialised.
//
// Why: the 64-pixel kernel above is bound by LDS and L1-fill bandwidth, not by the matrix cores -- per k-step each
// wave issues 12 ds_read_b128 for 24 MFMAs and the block re-streams the whole 590 KB weight image for only 64
// pixels (550 blocks x 864 KB through the 64 B/clk TCPs, three scheduling rounds on 256 CUs).  Here
//   * 35 200 BEV pixels = 245 workgroups = ONE round on 256 CUs, weights streamed 245x instead of 550x;
//   * waves 0-3 are MATRIX waves, side by side along Cout: each owns ALL 144 pixels x 32 couts (9 x 2 fragments,
//     72 accumulator registers); A fragments come from LDS (requested three 6-MFMA steps ahead), its private B
//     fragments straight from the packed image in L2 (no LDS, no duplication between waves), each of the four
//     k-substep register sets refilled for the next stage as soon as its last MFMA has issued;
//   * waves 4-7 are LOADER waves: while the matrix waves multiply stage s they gather stage s+1 (all 128 input
//     channels of one tap, 72 KB, zero halo) into the other LDS buffer with global_load_lds_dwordx4 (LDS-DMA).  With one workgroup per
//     CU nothing else could overlap the gather latency and the 13-cycle ds_write_b128s with the MFMAs (measured
//     with the cycle-counter stamps below: 15 us fixed + 22 us multiply + 13 us exposed gather when one set of
//     4 waves did both);
//   * one barrier per stage (9 per 3x3 convolution at Cin = 128).
// ------------------------------------------------------------------------------------------------
#define DL_KC 128
#define DL_SS (DL_KC / 32)  // MFMA k-substeps per stage
#define DL_THREADS 512
#define DL_TS (DC_BN + 4)
// dynamic LDS of a tile of MT fragments: two stages (hi + lo planes of MT * 16 pixels x 128 channels) or the fp32 epilogue tile
static constexpr int dl_smem(int mt) {
  const int bm = mt * 16, stages = 2 * 2 * bm * DL_KC * 2, tile = bm * DL_TS * 4;
  return tile > stages ? tile : stages;
}

#ifndef DL_TIMELINE
#define DL_TIMELINE 0  // 1: cycle-counter stamps of one workgroup (tools/mb_dense.py prints them)
#endif
#if DL_TIMELINE
__device__ unsigned long long dl_tl[2][64];
extern "C" int v3d_debug_dense_timeline(unsigned long long* host_out) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(dl_tl), sizeof(dl_tl));
}
#define DL_STAMP(role, idx) do { if (blockIdx.x == 8 && lane == 0 && (wave == 0 || wave == 4)) dl_tl[role][idx] = __builtin_readcyclecounter(); } while (0)
#else
#define DL_STAMP(role, idx)
#endif

__device__ __attribute__((aligned(16))) const unsigned dl_zero16[4] = {0u, 0u, 0u, 0u};  // halo source of the LDS-DMA gather

typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
// (v0, v1) -> packed hi pair and lo pair (hi = RNE(v), lo = RNE(v - hi)) on the hardware converters; PREC 1: f16 pieces of v * s
template <int PREC>
__device__ __forceinline__ void split_pair(float v0, float v1, float s, unsigned& hi, unsigned& lo) {
  if constexpr (PREC == 0) {
    const bf16x2_t h = __builtin_convertvector(f32x2_t{v0, v1}, bf16x2_t);
    hi = __builtin_bit_cast(unsigned, h);
    const float r0 = v0 - __uint_as_float(hi << 16), r1 = v1 - __uint_as_float(hi & 0xFFFF0000u);
    lo = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{r0, r1}, bf16x2_t));
  } else {
    v3d_split_f16_pair(v0, v1, __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, s))), hi, lo);
  }
}

// one tile (MT 16-pixel fragments x 128 couts) by the whole workgroup; returns with every thread past its last LDS access
// except the epilogue reads (the caller separates tiles with a barrier)
template <int KS, int MT, int PREC>  // MT = 16-pixel fragments per tile: 9 (144 pixels), or 5 (80 pixels) with background skipping
__device__ __forceinline__ void dl_tile(unsigned char* smem_l, const bf16_t* __restrict__ x_hi, const bf16_t* __restrict__ x_lo,
                                        const bf16_t* __restrict__ w_img, const float* __restrict__ bias, const DcParams& p,
                                        bf16_t* __restrict__ y_hi, bf16_t* __restrict__ y_lo, float* __restrict__ y_nchw,
                                        const int mtile) {
  constexpr int BM = MT * 16, A_PLANE = BM * DL_KC * 2, STAGE = 2 * A_PLANE;  // pixels per tile; bytes of one plane / of hi + lo of a stage
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool loader = wave >= 4;
  const int m0 = mtile * BM;
  const int n0 = blockIdx.y * DC_BN;
  if (p.occ) {
    // Background tiles.  An output pixel whose receptive field (through ALL layers so far: `reach` pixels) holds no occupied
    // BEV pixel sees exactly the inputs it sees in an empty map, so its value is the empty map's response at that position,
    // bit for bit -- precomputed once per weight set by this same kernel (bg planes).  Every wave takes the decision from
    // the occupancy bitmap (wave-uniform, no barrier); a background tile is a 74 KB copy instead of 1 944 MFMAs per wave.
    // "is any occupied pixel within `reach` of any pixel of this tile": the tile is a run of consecutive pixels = one
    // segment per image row it crosses; a segment's neighbourhood is the rectangle (rows +- reach, columns +- reach),
    // tested against the bitmap one (row, word) pair per lane
    const int R = p.reach, wpr = (p.W + 31) >> 5, span = 2 * R + 1;
    const int mlast = min(m0 + BM, p.M) - 1;
    const int r0 = m0 / p.W, r1 = mlast / p.W;  // global rows b * H + y
    const int per_seg = span * wpr, total = (r1 - r0 + 1) * per_seg;
    bool near_any = false;
    for (int q = lane; q < total; q += 64) {
      const int seg = q / per_seg, rem = q - seg * per_seg, dy = rem / wpr - R, wd = rem % wpr;
      const int r = r0 + seg, rr = r + dy;
      if (rr < 0 || rr >= p.B * p.H || rr / p.H != r / p.H) continue;  // neighbours live in the same image
      const int xs = (r == r0 ? m0 - r0 * p.W : 0) - R, xe = (r == r1 ? mlast - r1 * p.W : p.W - 1) + R;
      const int lo = max(xs, wd * 32), hi = min(min(xe, p.W - 1), wd * 32 + 31);
      if (lo > hi) continue;
      const unsigned mask = (0xFFFFFFFFu >> (31 - (hi - wd * 32))) & (0xFFFFFFFFu << (lo - wd * 32));
      near_any |= (~p.occ[(size_t)rr * wpr + wd] & mask) != 0u;
    }
    if (!__builtin_amdgcn_readfirstlane(__ballot(near_any) != 0ull)) {
      if (p.tile_state) {
        if (p.tile_state[mtile] == 0u) return;  // the response is already in place (workgroup-uniform: one word)
        __syncthreads();                        // every thread has read the word before it is cleared
        if (tid == 0) p.tile_state[mtile] = 0u;
      }
      const int HW = p.H * p.W, parts = (min(p.cout_store, n0 + DC_BN) - n0) >> 3;  // 16-byte parts of this cout block
      for (int q = tid; q < BM * parts; q += DL_THREADS) {
        const int row = q / parts, part = q - row * parts;
        const int m = m0 + row;
        if (m >= p.M) break;
        const size_t src = (size_t)(m % HW) * p.cout_store + n0 + part * 8, dst = (size_t)m * p.cout_store + n0 + part * 8;
        *reinterpret_cast<u32x4*>(y_hi + dst) = *reinterpret_cast<const u32x4*>(p.bg_hi + src);
        *reinterpret_cast<u32x4*>(y_lo + dst) = *reinterpret_cast<const u32x4*>(p.bg_lo + src);
      }
      return;
    }
  }
  if (p.tile_state && tid == 0) p.tile_state[mtile] = 1u;
  const int chunks = (p.Cin + DL_KC - 1) / DL_KC;  // 128-channel stages per tap (the last one may be partial: zero-filled)
  const int stages = KS * KS * chunks;
  // LDS image of one plane: pixel rows of 256 B (16 parts of 16 B), the part slot XOR-ed with px & 15.  Every row
  // aliases the same 64 banks, so the XOR alone has to separate the 16 lanes of a ds_read_b128 group (16 pixels,
  // parts p and p^1: the pixel sets of the two parts never differ in bit 0 only) and the 8 lanes of a
  // ds_write_b128 group (8 parts of one pixel).
  auto a_slot = [](int px, int part) { return px * 256 + ((part ^ (px & 15)) << 4); };

  f32x4 acc[MT][2];
#pragma unroll
  for (int i = 0; i < MT; i++) {
    acc[i][0] = f32x4{0.f, 0.f, 0.f, 0.f};
    acc[i][1] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  if (loader) {
    // ---- gather role, LDS-DMA: one global_load_lds_dwordx4 moves 64 lanes x 16 B = four 256-byte pixel rows of one
    // plane straight into LDS (wave-uniform base + lane*16, no VGPR round trip, no ds_write, no select).  The LDS
    // image is swizzled, so the permutation goes on the SOURCE side: lane L of a row block feeds slot L%16 of pixel
    // L/16 with channel part (L%16) ^ (px & 15).  Halo / out-of-range lanes read a 16-byte block of zeros.
    const int lw = wave - 4;
    const int sub = lane >> 4, slot = lane & 15;
    int a_pix[MT], a_hw[MT];  // this lane's 9 pixels: px = (lw*9 + j)*4 + sub; flattened index (-1: beyond M), (h << 16 | w)
    {
      const int m = m0 + lw * MT * 4 + sub;
      const int b = m / (p.H * p.W), rem = m - b * p.H * p.W;
      int h = rem / p.W, w = rem - h * p.W;
#pragma unroll
      for (int j = 0; j < MT; j++) {
        const int mj = m + 4 * j;
        a_pix[j] = mj < p.M ? mj : -1;
        a_hw[j] = mj < p.M ? ((h << 16) | w) : 0;
        w += 4;
        if (w >= p.W) {
          w -= p.W;
          if (++h == p.H) h = 0;
        }
      }
    }
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    int tap = 0, chunk = 0;
    auto gather = [&](int buf) {
      const int dy = KS == 3 ? tap / 3 - 1 : 0, dx = KS == 3 ? tap % 3 - 1 : 0;
      unsigned char* A = smem_l + buf * STAGE + lw * MT * 1024;
#pragma unroll
      for (int j = 0; j < MT; j++) {
        const int px = (lw * MT + j) * 4 + sub;
        const int ch = chunk * DL_KC + ((slot ^ (px & 15)) << 3);
        const int hh = (a_hw[j] >> 16) + dy, ww = (a_hw[j] & 0xFFFF) + dx;
        const bool ok = a_pix[j] >= 0 && hh >= 0 && hh < p.H && ww >= 0 && ww < p.W && ch < p.Cin;
        const size_t off = (size_t)(a_pix[j] + dy * p.W + dx) * p.Cin + ch;
        const bf16_t* sh = ok ? x_hi + off : reinterpret_cast<const bf16_t*>(dl_zero16);
        const bf16_t* sl = ok ? x_lo + off : reinterpret_cast<const bf16_t*>(dl_zero16);
        __builtin_amdgcn_global_load_lds((gptr_t)sh, (lptr_t)(A + j * 1024), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gptr_t)sl, (lptr_t)(A + A_PLANE + j * 1024), 16, 0, 0);
      }
      if (++chunk == chunks) {
        chunk = 0;
        ++tap;
      }
    };
    DL_STAMP(1, 0);
    gather(0);
    DL_STAMP(1, 1);
    __syncthreads();
    DL_STAMP(1, 2);
    for (int s = 0; s < stages; s++) {
      if (s + 1 < stages) gather((s + 1) & 1);
      if (s < 8) DL_STAMP(1, 3 + 2 * s);
      __syncthreads();
      if (s < 8) DL_STAMP(1, 4 + 2 * s);
    }
    DL_STAMP(1, 30);
  } else {
    // ---- matrix role
    const size_t plane_elems = (size_t)(p.CoutPad / 16) * 4 * 16 * 8;
    u32x4 cb[DL_SS][4];  // B fragments per k-substep: [n-tile*2 + plane]
    int tap = 0, chunk = 0;  // stage whose B fragments are requested next
    auto load_b = [&](int ss) {
      const int steps32 = p.Cin / DC_KC;  // 32-channel steps per tap in the packed image
      const size_t step = (size_t)tap * steps32 + min(chunk * DL_SS + ss, steps32 - 1);  // clamped: its A rows are zero
      const bf16_t* wb = w_img + step * 2 * plane_elems + ((size_t)(n0 / 16 + wave * 2) * 64 + lane) * 8;
#pragma unroll
      for (int nt2 = 0; nt2 < 2; nt2++) {
        cb[ss][nt2 * 2 + 0] = *reinterpret_cast<const u32x4*>(wb + (size_t)nt2 * 512);
        cb[ss][nt2 * 2 + 1] = *reinterpret_cast<const u32x4*>(wb + plane_elems + (size_t)nt2 * 512);
      }
    };
    auto next_stage = [&]() {
      if (++chunk == chunks) {
        chunk = 0;
        ++tap;
      }
    };
    // 36 (k-substep, pixel-tile) pairs per stage, two per step.  A step issues its 12 MFMAs term-major over FOUR
    // accumulators, so an accumulator is re-used every 4th MFMA (64 clocks): with two accumulators alternating the
    // dependent-issue latency (~40 clocks) stalled every MFMA (measured 25 instead of 16 clocks per MFMA).  The A
    // fragments of step u+2 are requested before step u's MFMAs issue (ring of 3 slots).
    auto multiply = [&](int buf, bool more, bool stamp) {
      const unsigned char* A = smem_l + buf * STAGE;
      constexpr int NSTEP = DL_SS * MT / 2;
      u32x4 fh[3][2], fl[3][2];
      auto frag = [&](int t, u32x4& h, u32x4& l) {
        const int ss = t / MT, i = t % MT;
        const int off = a_slot(i * 16 + (lane & 15), ss * 4 + (lane >> 4));
        h = *reinterpret_cast<const u32x4*>(A + off);
        l = *reinterpret_cast<const u32x4*>(A + A_PLANE + off);
      };
      frag(0, fh[0][0], fl[0][0]);
      frag(1, fh[0][1], fl[0][1]);
      frag(2, fh[1][0], fl[1][0]);
      frag(3, fh[1][1], fl[1][1]);
#pragma unroll
      for (int u = 0; u < NSTEP; u++) {
        if (u + 2 < NSTEP) {
          frag(2 * u + 4, fh[(u + 2) % 3][0], fl[(u + 2) % 3][0]);
          frag(2 * u + 5, fh[(u + 2) % 3][1], fl[(u + 2) % 3][1]);
        }
        const int t0 = 2 * u, t1 = 2 * u + 1;
        const int s0 = t0 / MT, i0 = t0 % MT, s1 = t1 / MT, i1 = t1 % MT;
        const u32x4 b0h0 = cb[s0][0], b0l0 = cb[s0][1];
        const u32x4 b0h1 = cb[s0][2], b0l1 = cb[s0][3];
        const u32x4 b1h0 = cb[s1][0], b1l0 = cb[s1][1];
        const u32x4 b1h1 = cb[s1][2], b1l1 = cb[s1][3];
        const u32x4 a0h = fh[u % 3][0], a0l = fl[u % 3][0], a1h = fh[u % 3][1], a1l = fl[u % 3][1];
        acc[i0][0] = dc_mfma<PREC>(a0l, b0h0, acc[i0][0]);
        acc[i0][1] = dc_mfma<PREC>(a0l, b0h1, acc[i0][1]);
        acc[i1][0] = dc_mfma<PREC>(a1l, b1h0, acc[i1][0]);
        acc[i1][1] = dc_mfma<PREC>(a1l, b1h1, acc[i1][1]);
        __builtin_amdgcn_sched_barrier(0);
        acc[i0][0] = dc_mfma<PREC>(a0h, b0l0, acc[i0][0]);
        acc[i0][1] = dc_mfma<PREC>(a0h, b0l1, acc[i0][1]);
        acc[i1][0] = dc_mfma<PREC>(a1h, b1l0, acc[i1][0]);
        acc[i1][1] = dc_mfma<PREC>(a1h, b1l1, acc[i1][1]);
        __builtin_amdgcn_sched_barrier(0);
        acc[i0][0] = dc_mfma<PREC>(a0h, b0h0, acc[i0][0]);
        acc[i0][1] = dc_mfma<PREC>(a0h, b0h1, acc[i0][1]);
        acc[i1][0] = dc_mfma<PREC>(a1h, b1h0, acc[i1][0]);
        acc[i1][1] = dc_mfma<PREC>(a1h, b1h1, acc[i1][1]);
        // a substep's B registers are free once its last tile has issued: refill them for the next stage
        // (unconditionally -- the last stage re-reads its own fragments: a branch here makes the compiler's
        // s_waitcnt bookkeeping merge two histories and wait for vmcnt(0) in the middle of the stage, 1350 clocks)
        if (i0 == MT - 1) load_b(s0);
        if (i1 == MT - 1) load_b(s1);
        __builtin_amdgcn_sched_barrier(0);
#if DL_TIMELINE
        if (stamp && (u == 0 || u == 1 || u == 8 || u == 16 || u == 17)) DL_STAMP(0, 40 + u);
#endif
      }
    };
    DL_STAMP(0, 0);
#pragma unroll
    for (int ss = 0; ss < DL_SS; ss++) load_b(ss);
    if (stages > 1) next_stage();
    DL_STAMP(0, 1);
    __syncthreads();
    DL_STAMP(0, 2);
    for (int s = 0; s < stages; s++) {
      const bool more = s + 1 < stages;
      multiply(s & 1, more, s == 3);
      if (s + 2 < stages) next_stage();  // (tap, chunk) = stage s+2, clamped to the last one
      if (s < 8) DL_STAMP(0, 3 + 2 * s);
      __syncthreads();
      if (s < 8) DL_STAMP(0, 4 + 2 * s);
    }
    DL_STAMP(0, 30);
  }

  DL_STAMP(loader ? 1 : 0, 31);
  // ---- epilogue (all 8 waves): accumulators -> LDS tile [144 px][128 + 4] fp32 -> bias + ReLU -> outputs
  const DcScales sc = dc_scales<PREC>(p);
  float vmax = 0.f;
  float* tile = reinterpret_cast<float*>(smem_l);
  // this thread's output role: 8 consecutive couts (y_hi requires Cout % 8 == 0); the two bias vectors are requested
  // before the accumulator exchange so their latency hides behind it
  f32x4 eb0 = f32x4{0.f, 0.f, 0.f, 0.f}, eb1 = eb0;
  {
    const int co = n0 + (tid & 15) * 8;
    if (bias && y_hi && co < p.cout_store) {
      eb0 = *reinterpret_cast<const f32x4*>(bias + co);
      eb1 = *reinterpret_cast<const f32x4*>(bias + co + 4);
    }
  }
  if (!loader) {
#pragma unroll
    for (int i = 0; i < MT; i++)
#pragma unroll
      for (int j = 0; j < 2; j++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const int row = i * 16 + (lane >> 4) * 4 + r;
          const int col = wave * 32 + j * 16 + (lane & 15);
          tile[row * DL_TS + col] = acc[i][j][r];
        }
  }
  __syncthreads();
  DL_STAMP(loader ? 1 : 0, 32);
  if (y_hi) {  // thread = 8 consecutive couts (fixed) of pixel rows r, r+32, ...: two 16-byte LDS reads, two 16-byte stores
    const int c8 = tid & 15, co = n0 + c8 * 8;
    if (co < p.cout_store) {
#pragma unroll
      for (int k = 0; k < (BM + 31) / 32; k++) {
        const int row = (tid >> 4) + 32 * k;
        const int m = m0 + row;
        if (row < BM && m < p.M) {
          const f32x4 t0 = *reinterpret_cast<const f32x4*>(tile + row * DL_TS + c8 * 8);
          const f32x4 t1 = *reinterpret_cast<const f32x4*>(tile + row * DL_TS + c8 * 8 + 4);
          float v[8] = {dc_finish<PREC>(t0[0], eb0[0], sc), dc_finish<PREC>(t0[1], eb0[1], sc), dc_finish<PREC>(t0[2], eb0[2], sc),
                        dc_finish<PREC>(t0[3], eb0[3], sc), dc_finish<PREC>(t1[0], eb1[0], sc), dc_finish<PREC>(t1[1], eb1[1], sc),
                        dc_finish<PREC>(t1[2], eb1[2], sc), dc_finish<PREC>(t1[3], eb1[3], sc)};
          u32x4 vh, vl;
#pragma unroll
          for (int e2 = 0; e2 < 4; e2++) {
            float v0 = v[2 * e2], v1 = v[2 * e2 + 1];
            if (p.relu) {
              v0 = fmaxf(v0, 0.f);
              v1 = fmaxf(v1, 0.f);
            }
            if constexpr (PREC == 1) vmax = fmaxf(vmax, fmaxf(fabsf(v0), fabsf(v1)));
            unsigned h, l;
            split_pair<PREC>(v0, v1, sc.s_out, h, l);
            vh[e2] = h;
            vl[e2] = l;
          }
          *reinterpret_cast<u32x4*>(y_hi + (size_t)m * p.cout_store + co) = vh;
          *reinterpret_cast<u32x4*>(y_lo + (size_t)m * p.cout_store + co) = vl;
        }
      }
    }
    if constexpr (PREC == 1) dc_range_check(p, vmax, sc.limit);
  }
  if (y_nchw) {
    const int HW = p.H * p.W;
    for (int q = tid; q < (BM / 4) * DC_BN; q += DL_THREADS) {
      const int col = q / (BM / 4), r4 = q % (BM / 4);
      const int co = n0 + col;
      if (co >= p.cout_store) continue;
      const float bv = bias ? bias[co] : 0.f;
#pragma unroll
      for (int e = 0; e < 4; e++) {
        const int m = m0 + r4 * 4 + e;
        if (m >= p.M) continue;
        float v = dc_finish<PREC>(tile[(r4 * 4 + e) * DL_TS + col], bv, sc);
        if (p.relu) v = fmaxf(v, 0.f);
        const int b = m / HW, pix = m - b * HW;
        y_nchw[((size_t)b * p.cout_store + co) * HW + pix] = v;
      }
    }
  }
  DL_STAMP(loader ? 1 : 0, 33);
}

// ------------------------------------------------------------------------------------------------
// 2-D tiles for the 3x3 layers of the background-skipping path (Cin = 128, one cout block, split output).
// dl_tile gathers a tile's input once PER TAP: 9 stages x 40 KB = 360 LDS-DMA requests for 80 pixels, and a live tile lasts as
// long as that chain of gathers (~20 us for 7 us of MFMA) -- with the persistent grid a kernel lasts as long as ONE live tile.
// Here a tile is 5 rows x 16 columns and its (5 + 2) x (16 + 2) neighbourhood goes to LDS ONCE (64 requests): per plane two
// 64-channel halves of 126 pixel rows x 128 B, chunk p of pixel row hp at chunk p ^ (hp & 7) (every ds_read_b128 lane group
// covers the bank row once, for every tap alignment); the nine taps read their fragments from it at immediate offsets of eight
// per-lane bases, the weights come straight from L2 as before, and there is no barrier between the taps.
// FOUR waves (round 6; rounds 4-5: 4 matrix + 4 loader waves, 84 KB): every wave requests a quarter of the image (16 LDS-DMA
// instructions), then multiplies its 32 output channels -- the loaders' only job was those 16 requests, yet as waves of the
// same kernel they held 256 VGPRs each and, with the LDS padded past half a CU, the workgroup owned the CU: one live tile per
// CU and launch at 0.23 matrix-pipe occupancy, and no other frame's tile beside it.  At 256 threads x 256 VGPRs and 64 KB (the
// image; the epilogue tile reuses it) TWO workgroups share a CU -- with frames in flight the tiles of two frames' launches: one's
// occupancy test, image load and epilogue under the other's matrix phase.  Same bits (the summation order of an output is
// unchanged).  Same box: 3 750 -> 4 090 frames/s pipelined (bs = 1), one frame at a time 0.478 -> 0.483 ms (the epilogue's
// stores on half the threads); Waymo-range 1 148 -> 1 218, bs = 8 5 870 -> 6 580 frames/s.
// ------------------------------------------------------------------------------------------------
#define D2_TH 5
#define D2_TW 16
#define D2_HW (D2_TW + 2)
#define D2_HALO ((D2_TH + 2) * D2_HW)    // 126 pixel rows
#define D2_PIECES ((D2_HALO + 7) / 8)    // 16 requests of 8 pixel rows x 128 B per (plane, half)
#define D2_HALF (D2_PIECES * 1024)
#define D2_PLANE (2 * D2_HALF)
static_assert(2 * D2_PLANE <= dl_smem(D2_TH), "the neighbourhood image lives in the tile kernel's LDS request");
static_assert(D2_TH * 16 * DL_TS * 4 <= 2 * D2_PLANE, "the epilogue tile reuses the image's LDS");
#define D2_THREADS 256
// Returns true (workgroup-uniform) when the tile was LIVE and the workgroup's next tile has already been drawn into *s_next: the
// draw -- an atomic round trip of ~1.5 us -- is issued behind the matrix phase by one thread and travels while the
// epilogue stores, instead of standing between this tile's last store and the next tile.  (Asking at the START of a live tile hands
// tiles to workgroups that stay busy for 15 us: 21 -> 32 us; at this point the workgroup is one epilogue away from being free.)
template <int PREC>
__device__ __forceinline__ bool dl_tile2d(unsigned char* smem_l, const bf16_t* __restrict__ x_hi, const bf16_t* __restrict__ x_lo,
                                          const bf16_t* __restrict__ w_img, const float* __restrict__ bias, const DcParams& p,
                                          bf16_t* __restrict__ y_hi, bf16_t* __restrict__ y_lo, const int mtile, int* s_next) {
  constexpr int MT = D2_TH, BM = MT * 16, NT = D2_THREADS;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tx_n = (p.W + D2_TW - 1) / D2_TW, ty_n = (p.H + D2_TH - 1) / D2_TH;
  const int b = mtile / (ty_n * tx_n), rt = mtile - b * ty_n * tx_n;
  const int y0 = (rt / tx_n) * D2_TH, x0 = (rt % tx_n) * D2_TW;
  if (p.occ) {
    // background tile <=> no occupied pixel within `reach` of any pixel of the tile: the rectangle (rows y0 - R .. y0 + 4 + R,
    // columns x0 - R .. x0 + 15 + R, clipped to the image) against the bitmap, one (row, word) pair per lane
    const int R = p.reach, wpr = (p.W + 31) >> 5;
    const int ylo = max(y0 - R, 0), yhi = min(min(y0 + D2_TH - 1, p.H - 1) + R, p.H - 1);
    const int xlo = max(x0 - R, 0), xhi = min(min(x0 + D2_TW - 1, p.W - 1) + R, p.W - 1);
    const int w0 = xlo >> 5, nw = (xhi >> 5) - w0 + 1, total = (yhi - ylo + 1) * nw;
    const unsigned in_place = p.tile_state ? p.tile_state[mtile] : 1u;  // (requested with the occupancy words, not behind them)
    bool near_any = false;
    for (int q = lane; q < total; q += 64) {
      const int row = ylo + q / nw, wd = w0 + q % nw;
      const int lo = max(xlo, wd * 32), hi = min(xhi, wd * 32 + 31);
      const unsigned mask = (0xFFFFFFFFu >> (31 - (hi - wd * 32))) & (0xFFFFFFFFu << (lo - wd * 32));
      near_any |= (~p.occ[((size_t)b * p.H + row) * wpr + wd] & mask) != 0u;
    }
    if (!__builtin_amdgcn_readfirstlane(__ballot(near_any) != 0ull)) {
      if (p.tile_state) {
        if (in_place == 0u) return false;  // the response is already in place (workgroup-uniform: one word)
        __syncthreads();             // every thread has read the word before it is cleared
        if (tid == 0) p.tile_state[mtile] = 0u;
      }
      const int parts = p.cout_store >> 3;
      for (int q = tid; q < BM * parts; q += NT) {
        const int row = q / parts, part = q - row * parts;
        const int yy = y0 + (row >> 4), xx = x0 + (row & 15);
        if (yy >= p.H || xx >= p.W) continue;
        const size_t src = ((size_t)yy * p.W + xx) * p.cout_store + part * 8, dst = (((size_t)b * p.H + yy) * p.W + xx) * p.cout_store + part * 8;
        *reinterpret_cast<u32x4*>(y_hi + dst) = *reinterpret_cast<const u32x4*>(p.bg_hi + src);
        *reinterpret_cast<u32x4*>(y_lo + dst) = *reinterpret_cast<const u32x4*>(p.bg_lo + src);
      }
      return false;
    }
  }
  if (p.tile_state && tid == 0) p.tile_state[mtile] = 1u;

  f32x4 acc[MT][2];
#pragma unroll
  for (int i = 0; i < MT; i++) {
    acc[i][0] = f32x4{0.f, 0.f, 0.f, 0.f};
    acc[i][1] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  {
    // wave w brings (plane w & 1, half w >> 1) of the image: 16 requests of 8 pixel rows x 128 B, zero rows outside the image
    const int plane = wave & 1, half = wave >> 1;
    const int sub = lane >> 3, slot = lane & 7;
    const bf16_t* xp = plane ? x_lo : x_hi;
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    unsigned char* dstb = smem_l + plane * D2_PLANE + half * D2_HALF;
#pragma unroll
    for (int j = 0; j < D2_PIECES; j++) {
      const int hp = j * 8 + sub;
      const int hy = hp / D2_HW, hx = hp - hy * D2_HW;
      const int yy = y0 - 1 + hy, xx = x0 - 1 + hx;
      const bool ok = hp < D2_HALO && yy >= 0 && yy < p.H && xx >= 0 && xx < p.W;
      const int part = half * 8 + (slot ^ (hp & 7));
      const bf16_t* src = ok ? xp + (((size_t)b * p.H + yy) * p.W + xx) * p.Cin + part * 8 : reinterpret_cast<const bf16_t*>(dl_zero16);
      __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(dstb + j * 1024), 16, 0, 0);
    }
  }
  {
    // ---- matrix role: as dl_tile (4 waves side by side along Cout, B fragments straight from L2), A fragments from the image
    const size_t plane_elems = (size_t)(p.CoutPad / 16) * 4 * 16 * 8;
    u32x4 cb[DL_SS][4];
    int tapn = 0;  // tap whose B fragments are requested next
    auto load_b = [&](int ss) {
      const size_t step = (size_t)min(tapn, 8) * DL_SS + ss;
      const bf16_t* wb = w_img + step * 2 * plane_elems + ((size_t)(wave * 2) * 64 + lane) * 8;
#pragma unroll
      for (int nt2 = 0; nt2 < 2; nt2++) {
        cb[ss][nt2 * 2 + 0] = *reinterpret_cast<const u32x4*>(wb + (size_t)nt2 * 512);
        cb[ss][nt2 * 2 + 1] = *reinterpret_cast<const u32x4*>(wb + plane_elems + (size_t)nt2 * 512);
      }
    };
    typedef const __attribute__((address_space(3))) unsigned char* lds_t;
    const lds_t lds = (lds_t)smem_l;
    const int pxl = lane & 15, kg = lane >> 4;
    unsigned pw[8];  // pixel row hp = c + pxl (c: a constant of the unrolled code): byte 128 hp + 16 ((4 (ss & 1) + kg) ^ (hp & 7))
#pragma unroll
    for (int j = 0; j < 8; j++) pw[j] = (unsigned)(pxl * 128 + ((kg ^ ((pxl + j) & 7)) << 4));
    auto multiply = [&](auto tapc) {
      constexpr int TAP = decltype(tapc)::value;
      constexpr int dyc = TAP / 3, dxc = TAP % 3;  // halo offsets (the image starts one row / column before the tile)
      constexpr int NSTEP = DL_SS * MT / 2;
      u32x4 fh[3][2], fl[3][2];
      auto frag = [&](auto tc, u32x4& h, u32x4& l) {
        constexpr int t = decltype(tc)::value;
        constexpr int ss = t / MT, i = t % MT;
        constexpr int c = (i + dyc) * D2_HW + dxc;
        const unsigned a = (pw[c & 7] ^ (unsigned)((ss & 1) << 6)) + (unsigned)((ss >> 1) * D2_HALF + c * 128);
        h = *reinterpret_cast<const __attribute__((address_space(3))) u32x4*>(lds + a);
        l = *reinterpret_cast<const __attribute__((address_space(3))) u32x4*>(lds + a + D2_PLANE);
      };
#define D2_IC(v) std::integral_constant<int, (v)>{}
      frag(D2_IC(0), fh[0][0], fl[0][0]);
      frag(D2_IC(1), fh[0][1], fl[0][1]);
      frag(D2_IC(2), fh[1][0], fl[1][0]);
      frag(D2_IC(3), fh[1][1], fl[1][1]);
      auto step = [&](auto uc) {
        constexpr int u = decltype(uc)::value;
        if constexpr (u + 2 < NSTEP) {
          frag(D2_IC(2 * u + 4), fh[(u + 2) % 3][0], fl[(u + 2) % 3][0]);
          frag(D2_IC(2 * u + 5), fh[(u + 2) % 3][1], fl[(u + 2) % 3][1]);
        }
        constexpr int t0 = 2 * u, t1 = 2 * u + 1;
        constexpr int s0 = t0 / MT, i0 = t0 % MT, s1 = t1 / MT, i1 = t1 % MT;
        const u32x4 b0h0 = cb[s0][0], b0l0 = cb[s0][1];
        const u32x4 b0h1 = cb[s0][2], b0l1 = cb[s0][3];
        const u32x4 b1h0 = cb[s1][0], b1l0 = cb[s1][1];
        const u32x4 b1h1 = cb[s1][2], b1l1 = cb[s1][3];
        const u32x4 a0h = fh[u % 3][0], a0l = fl[u % 3][0], a1h = fh[u % 3][1], a1l = fl[u % 3][1];
        acc[i0][0] = dc_mfma<PREC>(a0l, b0h0, acc[i0][0]);
        acc[i0][1] = dc_mfma<PREC>(a0l, b0h1, acc[i0][1]);
        acc[i1][0] = dc_mfma<PREC>(a1l, b1h0, acc[i1][0]);
        acc[i1][1] = dc_mfma<PREC>(a1l, b1h1, acc[i1][1]);
        __builtin_amdgcn_sched_barrier(0);
        acc[i0][0] = dc_mfma<PREC>(a0h, b0l0, acc[i0][0]);
        acc[i0][1] = dc_mfma<PREC>(a0h, b0l1, acc[i0][1]);
        acc[i1][0] = dc_mfma<PREC>(a1h, b1l0, acc[i1][0]);
        acc[i1][1] = dc_mfma<PREC>(a1h, b1l1, acc[i1][1]);
        __builtin_amdgcn_sched_barrier(0);
        acc[i0][0] = dc_mfma<PREC>(a0h, b0h0, acc[i0][0]);
        acc[i0][1] = dc_mfma<PREC>(a0h, b0h1, acc[i0][1]);
        acc[i1][0] = dc_mfma<PREC>(a1h, b1h0, acc[i1][0]);
        acc[i1][1] = dc_mfma<PREC>(a1h, b1h1, acc[i1][1]);
        // a substep's B registers are free once its last tile has issued: refill them for the next tap (unconditionally:
        // behind the last tap a re-read nobody uses -- see dl_tile)
        if constexpr (i0 == MT - 1) load_b(s0);
        if constexpr (i1 == MT - 1) load_b(s1);
        __builtin_amdgcn_sched_barrier(0);
      };
      step(D2_IC(0)); step(D2_IC(1)); step(D2_IC(2)); step(D2_IC(3)); step(D2_IC(4));
      step(D2_IC(5)); step(D2_IC(6)); step(D2_IC(7)); step(D2_IC(8)); step(D2_IC(9));
      static_assert(NSTEP == 10, "the steps are written out");
    };
#pragma unroll
    for (int ss = 0; ss < DL_SS; ss++) load_b(ss);  // (behind the image requests: the barrier waits for both)
    tapn = 1;
    __syncthreads();  // image complete (the compiler waits for the requests in front of the barrier)
    multiply(D2_IC(0)); tapn = 2;
    multiply(D2_IC(1)); tapn = 3;
    multiply(D2_IC(2)); tapn = 4;
    multiply(D2_IC(3)); tapn = 5;
    multiply(D2_IC(4)); tapn = 6;
    multiply(D2_IC(5)); tapn = 7;
    multiply(D2_IC(6)); tapn = 8;
    multiply(D2_IC(7));
    multiply(D2_IC(8));
#undef D2_IC
  }
  __syncthreads();  // every fragment read is behind us: the image's LDS becomes the epilogue tile
  const bool draws = s_next != nullptr && p.work != nullptr;
  unsigned drawn = 0u;
  if (draws && tid == 0) drawn = gridDim.x + atomicAdd(p.work, 1u);  // the next tile, requested now (see the header)
  // ---- epilogue (all 8 waves): accumulators -> LDS tile [80 px][128 + 4] fp32 -> bias + ReLU -> split planes
  const DcScales sc = dc_scales<PREC>(p);
  float vmax = 0.f;
  float* tile = reinterpret_cast<float*>(smem_l);
  f32x4 eb0 = f32x4{0.f, 0.f, 0.f, 0.f}, eb1 = eb0;
  {
    const int co = (tid & 15) * 8;
    if (bias && co < p.cout_store) {
      eb0 = *reinterpret_cast<const f32x4*>(bias + co);
      eb1 = *reinterpret_cast<const f32x4*>(bias + co + 4);
    }
  }
#pragma unroll
  for (int i = 0; i < MT; i++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int r = 0; r < 4; r++) tile[(i * 16 + (lane >> 4) * 4 + r) * DL_TS + wave * 32 + j * 16 + (lane & 15)] = acc[i][j][r];
  __syncthreads();
  const int c8 = tid & 15, co = c8 * 8;
  if (co < p.cout_store) {
#pragma unroll
    for (int k = 0; k < (BM + NT / 16 - 1) / (NT / 16); k++) {
      const int row = (tid >> 4) + (NT / 16) * k;
      const int yy = y0 + (row >> 4), xx = x0 + (row & 15);
      if (row < BM && yy < p.H && xx < p.W) {
        const size_t m = ((size_t)b * p.H + yy) * p.W + xx;
        const f32x4 t0 = *reinterpret_cast<const f32x4*>(tile + row * DL_TS + c8 * 8);
        const f32x4 t1 = *reinterpret_cast<const f32x4*>(tile + row * DL_TS + c8 * 8 + 4);
        float v[8] = {dc_finish<PREC>(t0[0], eb0[0], sc), dc_finish<PREC>(t0[1], eb0[1], sc), dc_finish<PREC>(t0[2], eb0[2], sc),
                      dc_finish<PREC>(t0[3], eb0[3], sc), dc_finish<PREC>(t1[0], eb1[0], sc), dc_finish<PREC>(t1[1], eb1[1], sc),
                      dc_finish<PREC>(t1[2], eb1[2], sc), dc_finish<PREC>(t1[3], eb1[3], sc)};
        u32x4 vh, vl;
#pragma unroll
        for (int e2 = 0; e2 < 4; e2++) {
          float v0 = v[2 * e2], v1 = v[2 * e2 + 1];
          if (p.relu) {
            v0 = fmaxf(v0, 0.f);
            v1 = fmaxf(v1, 0.f);
          }
          if constexpr (PREC == 1) vmax = fmaxf(vmax, fmaxf(fabsf(v0), fabsf(v1)));
          unsigned h, l;
          split_pair<PREC>(v0, v1, sc.s_out, h, l);
          vh[e2] = h;
          vl[e2] = l;
        }
        *reinterpret_cast<u32x4*>(y_hi + m * p.cout_store + co) = vh;
        *reinterpret_cast<u32x4*>(y_lo + m * p.cout_store + co) = vl;
      }
    }
  }
  if constexpr (PREC == 1) dc_range_check(p, vmax, sc.limit);
  if (draws && tid == 0) *s_next = (int)drawn;
  return draws;
}

// persistent grid drawing 2-D tiles from the counter pair (see conv2d_bf16x3_large_kernel)
template <int PREC>
__global__ __launch_bounds__(D2_THREADS, 2) void conv2d_bf16x3_tile2d_kernel(const bf16_t* __restrict__ x_hi, const bf16_t* __restrict__ x_lo,
                                                                          const bf16_t* __restrict__ w_img, const float* __restrict__ bias,
                                                                          const DcParams p, bf16_t* __restrict__ y_hi,
                                                                          bf16_t* __restrict__ y_lo) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_l[];
  __shared__ int s_next;
  const int ntiles = p.B * ((p.H + D2_TH - 1) / D2_TH) * ((p.W + D2_TW - 1) / D2_TW);
  // the first tile of a workgroup is its own index (no round trip to the counter in front of the first tile); later draws
  // continue behind the grid
  // (requesting the NEXT draw before the current tile starts hides the counter's round trip but hands tiles to workgroups
  // that are busy with a live one: 21.0 -> 32.4 us on the real frame)
  if (p.reset_ptr && blockIdx.x == 0)
    for (int i = threadIdx.x; i < p.reset_words; i += D2_THREADS) p.reset_ptr[i] = 0u;
  for (int mtile = blockIdx.x; mtile < ntiles;) {  // workgroup-uniform
    const bool drew = dl_tile2d<PREC>(smem_l, x_hi, x_lo, w_img, bias, p, y_hi, y_lo, mtile, &s_next);
    if (!drew && threadIdx.x == 0) s_next = (int)(gridDim.x + atomicAdd(p.work, 1u));
    __syncthreads();  // everyone is done with this tile's LDS
    mtile = s_next;
    __syncthreads();  // ... and has read s_next
  }
  if (!p.reset_ptr && threadIdx.x == 0 && atomicAdd(p.work + 1, 1u) == gridDim.x - 1) {  // last workgroup out: every draw has happened
    p.work[0] = 0u;
    p.work[1] = 0u;
  }
}

// The kernel: one tile per workgroup in launch order (XCD-contiguous remap), or -- p.work set: background skipping -- a
// PERSISTENT grid of at most one workgroup per CU that draws tiles from a counter.  Why persistent: on a sparse map fewer than
// half of the tiles convolve (the others are a copy), a kernel lasts as long as the CU that drew the most live tiles, and the
// hardware dispatcher hands out workgroups in order, not to the first free CU (measured: 440 80-pixel tiles of which ~200 live
// took as long as 440 live ones).  Drawing from a counter, CUs that hit background tiles come back within a few microseconds
// and take the next tile: ~200 live tiles spread over 256 CUs, one each.  The counter pair resets itself (last workgroup out).
template <int KS, int MT, int PREC>
__global__ __launch_bounds__(DL_THREADS) void conv2d_bf16x3_large_kernel(const bf16_t* __restrict__ x_hi,
                                                                         const bf16_t* __restrict__ x_lo,
                                                                         const bf16_t* __restrict__ w_img,
                                                                         const float* __restrict__ bias, const DcParams p,
                                                                         bf16_t* __restrict__ y_hi, bf16_t* __restrict__ y_lo,
                                                                         float* __restrict__ y_nchw) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_l[];
  __shared__ int s_next;
  if (!p.work) {
    const int nt = gridDim.x;
    int mtile = blockIdx.x;
    {  // XCD-aware order (see the 64-pixel kernel)
      const int q = nt / 8, rmd = nt % 8, xcd = mtile % 8, idx = mtile / 8;
      mtile = (xcd < rmd ? xcd * (q + 1) : rmd * (q + 1) + (xcd - rmd) * q) + idx;
    }
    dl_tile<KS, MT, PREC>(smem_l, x_hi, x_lo, w_img, bias, p, y_hi, y_lo, y_nchw, mtile);
    return;
  }
  const int ntiles = (p.M + MT * 16 - 1) / (MT * 16);
  if (p.reset_ptr && blockIdx.x == 0)
    for (int i = threadIdx.x; i < p.reset_words; i += DL_THREADS) p.reset_ptr[i] = 0u;
  // (the first tile of a workgroup is its own index: no round trip to the counter in front of it)
  for (int mtile = blockIdx.x; mtile < ntiles;) {  // workgroup-uniform
    dl_tile<KS, MT, PREC>(smem_l, x_hi, x_lo, w_img, bias, p, y_hi, y_lo, y_nchw, mtile);
    if (threadIdx.x == 0) s_next = (int)(gridDim.x + atomicAdd(p.work, 1u));
    __syncthreads();  // everyone is done with this tile's LDS
    mtile = s_next;
    __syncthreads();  // ... and has read s_next
  }
  if (!p.reset_ptr && threadIdx.x == 0 && atomicAdd(p.work + 1, 1u) == gridDim.x - 1) {  // last workgroup out: every draw has happened
    p.work[0] = 0u;
    p.work[1] = 0u;
  }
}

// ------------------------------------------------------------------------------------------------
// 1x1 convolution with <= 16 output channels and an fp32 NCHW output only: the fused [cls | reg] head (256 -> 16 at
// 200 x 176).  The tile kernels above pad Cout to 128 and stage the input through LDS for a weight reuse that does not
// exist here; the layer is a pure stream of the input planes (36 MB) against 16 KB of weights.  One wave = 16 pixels:
// the whole packed weight column (n-tile 0 of the image, hi and lo) sits in registers, the A fragments come straight
// from global memory (a lane's 8 channels are one 16-byte load), 3 MFMAs per 32 channels into one accumulator per term,
// D[pixel = (lane >> 4) * 4 + r][cout = lane & 15] goes out as NCHW with bias (+ ReLU).  14 -> ~9 us.
// ------------------------------------------------------------------------------------------------
template <int STEPS, int PREC>  // Cin / 32: compile-time, so that the fragment arrays are plain registers (a run-time bound spilled them)
__global__ __launch_bounds__(256) void conv1x1_bf16x3_small_cout_kernel(const bf16_t* __restrict__ x_hi,
                                                                        const bf16_t* __restrict__ x_lo,
                                                                        const bf16_t* __restrict__ w_img,
                                                                        const float* __restrict__ bias, const DcParams p,
                                                                        float* __restrict__ y_nchw) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int m0 = (blockIdx.x * 4 + wave) * 16;
  if (m0 >= p.M) return;
  const size_t plane_elems = (size_t)(p.CoutPad / 16) * 4 * 16 * 8;
  const int px = min(m0 + (lane & 15), p.M - 1);  // rows beyond M are computed on a clamped pixel and not stored
  const bf16_t* xh = x_hi + (size_t)px * p.Cin + (lane >> 4) * 8;
  const bf16_t* xl = x_lo + (size_t)px * p.Cin + (lane >> 4) * 8;
  const bf16_t* wb = w_img + (size_t)lane * 8;
  u32x4 bh[STEPS], bl[STEPS], ah[STEPS], al[STEPS];
#pragma unroll
  for (int s = 0; s < STEPS; s++) {
      bh[s] = *reinterpret_cast<const u32x4*>(wb + (size_t)s * 2 * plane_elems);
      bl[s] = *reinterpret_cast<const u32x4*>(wb + (size_t)s * 2 * plane_elems + plane_elems);
      ah[s] = *reinterpret_cast<const u32x4*>(xh + s * DC_KC);
      al[s] = *reinterpret_cast<const u32x4*>(xl + s * DC_KC);
    }
  f32x4 c_lh = {0.f, 0.f, 0.f, 0.f}, c_hl = c_lh, c_hh = c_lh;
#pragma unroll
  for (int s = 0; s < STEPS; s++) {
      c_lh = dc_mfma<PREC>(al[s], bh[s], c_lh);
      c_hl = dc_mfma<PREC>(ah[s], bl[s], c_hl);
      c_hh = dc_mfma<PREC>(ah[s], bh[s], c_hh);
    }
  const int co = lane & 15;
  if (co >= p.cout_store) return;
  const DcScales sc = dc_scales<PREC>(p);
  const float bv = bias ? bias[co] : 0.f;
  const int HW = p.H * p.W;
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const int m = m0 + (lane >> 4) * 4 + r;
    if (m >= p.M) continue;
    float v = dc_finish<PREC>((c_lh[r] + c_hl[r]) + c_hh[r], bv, sc);  // small terms first
    if (p.relu) v = fmaxf(v, 0.f);
    const int b = m / HW, pix = m - b * HW;
    y_nchw[((size_t)b * p.cout_store + co) * HW + pix] = v;
  }
}

// ------------------------------------------------------------------------------------------------
// The RPN's last layer (1x1, 128 -> 128, folded BatchNorm + ReLU: detector/second.py:73-79) and the fused [cls | reg] head on top of
// it (1x1, 128 -> <= 16: detector/proposal.py:19-22) as ONE pass over the pixels.  Both are per-pixel products: as two launches they
// cost 15.3 + 7.2 us of the KITTI frame for ~1 us of matrix work (the tile kernel's bitmap test, draw, epilogue and a 36 MB round
// trip of the intermediate planes through HBM).  Here a wave owns 16 pixels: A fragments straight from the input planes (a lane's 8
// channels = one 16-byte load), the 64 KB packed image of the 1x1 layer once per workgroup in LDS, 96 MFMAs into eight
// accumulators (per accumulator the order of the tile kernels: k-steps ascending, lo*Wh, hi*Wl, hi*Wh within a step), then bias +
// ReLU + the hi / lo split on the hardware converter exactly as the tile kernel's epilogue stores them, a transpose of the wave's
// 16 x 128 tile through a private LDS slab into A fragments again, and the head's 12 MFMAs with its three term accumulators summed
// (lh + hl) + hh + bias as in conv1x1_bf16x3_small_cout_kernel.  Bit-identical to the two launches it replaces
// (tests/test_gpu_dense_conv.py); every pixel is computed (background included: same values as the skipping path, no tile state).
// Persistent grid: <= one workgroup of FH_WAVES waves per CU, waves stride over the 16-pixel tiles.
// ------------------------------------------------------------------------------------------------
#define FH_WAVES 9  // 256 workgroups x 9 waves = 2 304 >= the 2 200 sixteen-pixel tiles of the 200 x 176 KITTI map: one tile per wave, no second pass
#define FH_ROW 272                                  // bytes of one pixel row of a transpose slab (256 + 16: conflict-free 16-byte reads)
#define FH_SLAB (2 * 16 * FH_ROW)                   // hi + lo planes of one wave's 16 x 128 tile
#define FH_W1_BYTES (4 * 2 * 128 * 32 * 2)          // 4 k-steps x (hi, lo) x 128 couts x 32 cins, bf16
#define FH_SMEM (FH_W1_BYTES + FH_WAVES * FH_SLAB)
struct FhScales {  // f16s (PREC 1): entries of the input planes and of the intermediate tensor, the two images' 1 / s_w, the flag
  const float *in_entry, *mid_entry, *w1_inv, *w2_inv;
  int* range_flag;
};
template <int PREC>
__global__ __launch_bounds__(FH_WAVES * 64) void conv1x1_head_fused_kernel(const bf16_t* __restrict__ x_hi, const bf16_t* __restrict__ x_lo,
                                                                          const bf16_t* __restrict__ w1_img, const float* __restrict__ b1,
                                                                          int relu1, const bf16_t* __restrict__ w2_img,
                                                                          const float* __restrict__ b2, int relu2, int M, int HW,
                                                                          int cout2, int cout2_pad, float* __restrict__ y_nchw,
                                                                          const FhScales fs) {
  extern __shared__ __attribute__((aligned(16))) unsigned char fh_smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  {  // the 1x1 layer's packed image, as it lies in memory: [step][plane][n-tile][lane][8]
    const u32x4* src = reinterpret_cast<const u32x4*>(w1_img);
    u32x4* dst = reinterpret_cast<u32x4*>(fh_smem);
    for (int i = tid; i < FH_W1_BYTES / 16; i += FH_WAVES * 64) dst[i] = src[i];
  }
  // the head's packed weight column (n-tile 0, hi and lo of the four k-steps) lives in registers for all of the wave's tiles
  u32x4 h_bh[4], h_bl[4];
  {
    const size_t pe2 = (size_t)cout2_pad * 32;
#pragma unroll
    for (int s = 0; s < 4; s++) {
      h_bh[s] = *reinterpret_cast<const u32x4*>(w2_img + (size_t)s * 2 * pe2 + (size_t)lane * 8);
      h_bl[s] = *reinterpret_cast<const u32x4*>(w2_img + (size_t)s * 2 * pe2 + pe2 + (size_t)lane * 8);
    }
  }
  __syncthreads();
  DcScales sc1{1.f, 1.f, 3.0e38f}, sc2{1.f, 1.f, 3.0e38f};
  if constexpr (PREC == 1) {
    sc1.undo = fs.in_entry[1] * fs.w1_inv[0];
    sc1.s_out = fs.mid_entry[0];
    sc1.limit = fs.mid_entry[2];
    sc2.undo = fs.mid_entry[1] * fs.w2_inv[0];
  }
  float vmax = 0.f;
  const unsigned char* w1s = fh_smem;
  unsigned char* slab = fh_smem + FH_W1_BYTES + wave * FH_SLAB;
  const int px_l = lane & 15, kg = lane >> 4;
  const int ntiles = (M + 15) / 16;
  for (int t = blockIdx.x * FH_WAVES + wave; t < ntiles; t += gridDim.x * FH_WAVES) {
    const int m0 = t * 16;
    const int px = min(m0 + px_l, M - 1);  // rows beyond M are computed on a clamped pixel and not stored
    u32x4 ah[4], al[4];
#pragma unroll
    for (int s = 0; s < 4; s++) {
      ah[s] = *reinterpret_cast<const u32x4*>(x_hi + (size_t)px * 128 + s * 32 + kg * 8);
      al[s] = *reinterpret_cast<const u32x4*>(x_lo + (size_t)px * 128 + s * 32 + kg * 8);
    }
    f32x4 acc[8];
#pragma unroll
    for (int nf = 0; nf < 8; nf++) acc[nf] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 4; s++) {
      u32x4 bh[8], bl[8];
#pragma unroll
      for (int nf = 0; nf < 8; nf++) {
        bh[nf] = *reinterpret_cast<const u32x4*>(w1s + s * 16384 + (nf * 64 + lane) * 16);
        bl[nf] = *reinterpret_cast<const u32x4*>(w1s + s * 16384 + 8192 + (nf * 64 + lane) * 16);
      }
      const u32x4 a_h = ah[s], a_l = al[s];
#pragma unroll
      for (int nf = 0; nf < 8; nf++) acc[nf] = dc_mfma<PREC>(a_l, bh[nf], acc[nf]);
#pragma unroll
      for (int nf = 0; nf < 8; nf++) acc[nf] = dc_mfma<PREC>(a_h, bl[nf], acc[nf]);
#pragma unroll
      for (int nf = 0; nf < 8; nf++) acc[nf] = dc_mfma<PREC>(a_h, bh[nf], acc[nf]);
    }
    // bias + ReLU + split: D[pixel = kg * 4 + r][cout = nf * 16 + px_l] -> slab[plane][pixel][cout] (bf16)
#pragma unroll
    for (int nf = 0; nf < 8; nf++) {
      const float bv = b1 ? b1[nf * 16 + px_l] : 0.f;
#pragma unroll
      for (int r2 = 0; r2 < 2; r2++) {
        float v0 = dc_finish<PREC>(acc[nf][2 * r2], bv, sc1), v1 = dc_finish<PREC>(acc[nf][2 * r2 + 1], bv, sc1);
        if (relu1) {
          v0 = fmaxf(v0, 0.f);
          v1 = fmaxf(v1, 0.f);
        }
        if constexpr (PREC == 1) vmax = fmaxf(vmax, fmaxf(fabsf(v0), fabsf(v1)));
        unsigned h, l;
        split_pair<PREC>(v0, v1, sc1.s_out, h, l);
        const int p0 = kg * 4 + 2 * r2, off = (nf * 16 + px_l) * 2;
        *reinterpret_cast<unsigned short*>(slab + p0 * FH_ROW + off) = (unsigned short)(h & 0xFFFFu);
        *reinterpret_cast<unsigned short*>(slab + (p0 + 1) * FH_ROW + off) = (unsigned short)(h >> 16);
        *reinterpret_cast<unsigned short*>(slab + 16 * FH_ROW + p0 * FH_ROW + off) = (unsigned short)(l & 0xFFFFu);
        *reinterpret_cast<unsigned short*>(slab + 16 * FH_ROW + (p0 + 1) * FH_ROW + off) = (unsigned short)(l >> 16);
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the wave's own slab: its LDS writes have landed (one wave, in order)
    f32x4 c_lh = {0.f, 0.f, 0.f, 0.f}, c_hl = c_lh, c_hh = c_lh;
#pragma unroll
    for (int s = 0; s < 4; s++) {
      const u32x4 a2h = *reinterpret_cast<const u32x4*>(slab + px_l * FH_ROW + s * 64 + kg * 16);
      const u32x4 a2l = *reinterpret_cast<const u32x4*>(slab + 16 * FH_ROW + px_l * FH_ROW + s * 64 + kg * 16);
      c_lh = dc_mfma<PREC>(a2l, h_bh[s], c_lh);
      c_hl = dc_mfma<PREC>(a2h, h_bl[s], c_hl);
      c_hh = dc_mfma<PREC>(a2h, h_bh[s], c_hh);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the fragment reads are done before the next tile overwrites the slab
    const int co = px_l;
    if (co < cout2) {
      const float bv = b2 ? b2[co] : 0.f;
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int m = m0 + kg * 4 + r;
        if (m >= M) continue;
        float v = dc_finish<PREC>((c_lh[r] + c_hl[r]) + c_hh[r], bv, sc2);  // small terms first
        if (relu2) v = fmaxf(v, 0.f);
        const int b = m / HW, pix = m - b * HW;
        y_nchw[((size_t)b * cout2 + co) * HW + pix] = v;
      }
    }
  }
  if constexpr (PREC == 1)
    if (fs.range_flag && vmax > sc1.limit) atomicMax(fs.range_flag, V3D_FLAG_RANGE);
}

static int dc_check_prec(const v3d_conv2d_prec* pr, bool planes_out) {
  if (!pr || pr->prec == V3D_PREC_BF16X3) return V3D_OK;
  if (pr->prec != V3D_PREC_F16S || !pr->in_entry || (planes_out && !pr->out_entry)) return V3D_EINVAL;
  return V3D_OK;
}

extern "C" int v3d_conv2d_1x1_head_fused(const void* x_hi, const void* x_lo, const void* w1_image, const float* b1, int relu1,
                                          const void* w2_image, const float* b2, int relu2, int B, int H, int W, int Cmid, int Cout2,
                                          float* y_nchw, const v3d_conv2d_prec* pr, v3d_stream_t stream) {
  if (!x_hi || !x_lo || !w1_image || !w2_image || !y_nchw || B < 1 || H < 1 || W < 1) return V3D_EINVAL;
  if (Cmid != 128 || Cout2 < 1 || Cout2 > 16) return V3D_EUNSUPPORTED;  // 128 -> 128 -> <= 16: the SECOND RPN tail
  if (dc_check_prec(pr, true)) return V3D_EINVAL;  // (out_entry = the entry of the intermediate 128-channel tensor)
  const int n_cu = v3d_device_cu_count();
  if (n_cu < 1) return V3D_EINVAL;
  const int M = B * H * W, ntiles = (M + 15) / 16;
  const int grid = std::min(v3d_ceil_div(ntiles, FH_WAVES), n_cu);
  const int cout2_pad = (Cout2 + DC_BN - 1) / DC_BN * DC_BN;
  const bool f16s = pr && pr->prec == V3D_PREC_F16S;
  FhScales fs{nullptr, nullptr, nullptr, nullptr, nullptr};
  if (f16s) {
    fs.in_entry = pr->in_entry;
    fs.mid_entry = pr->out_entry;
    fs.w1_inv = pr->w_inv ? pr->w_inv : reinterpret_cast<const float*>(reinterpret_cast<const char*>(w1_image) + dc_image_payload_bytes(128, 128, 1)) + 1;
    fs.w2_inv = pr->w_inv2 ? pr->w_inv2 : reinterpret_cast<const float*>(reinterpret_cast<const char*>(w2_image) + dc_image_payload_bytes(128, Cout2, 1)) + 1;
    fs.range_flag = pr->range_flag;
  }
  auto kern = f16s ? conv1x1_head_fused_kernel<1> : conv1x1_head_fused_kernel<0>;
  V3D_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, FH_SMEM));
  hipLaunchKernelGGL(kern, dim3(grid), dim3(FH_WAVES * 64), FH_SMEM, (hipStream_t)stream, (const bf16_t*)x_hi,
                     (const bf16_t*)x_lo, (const bf16_t*)w1_image, b1, relu1, (const bf16_t*)w2_image, b2, relu2, M, H * W, Cout2,
                     cout2_pad, y_nchw, fs);
  V3D_CHECK_LAUNCH();
  return V3D_OK;
}

extern "C" int v3d_conv2d_bg_tiles(int B, int H, int W);
// BEV occupancy bitmap, INVERTED (bit cleared = occupied) so that the 0xFF fill that resets the rest of a plan's per-frame
// state also resets it: row (b, y) = ceil(W / 32) words, bit x % 32 of word x / 32.  From the site list of the last sparse
// stage (rows (b, z, y, x); z is folded into the channels).
__global__ __launch_bounds__(V3D_BLOCK) void bev_occupancy_kernel(const int4* __restrict__ coords, const int* __restrict__ n_ptr,
                                                                  int cap, int H, int W, unsigned* __restrict__ occ) {
  const int n = min(*n_ptr, cap), wpr = (W + 31) >> 5;
  for (int i = blockIdx.x * V3D_BLOCK + threadIdx.x; i < n; i += gridDim.x * V3D_BLOCK) {
    const int4 c = coords[i];
    atomicAnd(occ + ((size_t)c.x * H + c.z) * wpr + (c.w >> 5), ~(1u << (c.w & 31)));
  }
}

extern "C" size_t v3d_bev_occupancy_words(int B, int H, int W) { return (size_t)B * H * ((W + 31) >> 5); }

extern "C" int v3d_bev_occupancy_bits(const int32_t* coords, const int32_t* n, int cap, int B, int H, int W, uint32_t* occ,
                                      v3d_stream_t stream) {
  if (!coords || !n || !occ || cap < 1 || B < 1 || H < 1 || W < 1) return V3D_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  V3D_CHECK_HIP(v3d_fill_async(occ, 0xFF, v3d_bev_occupancy_words(B, H, W) * sizeof(uint32_t), st));
  hipLaunchKernelGGL(bev_occupancy_kernel, dim3(std::min(v3d_ceil_div(cap, V3D_BLOCK), 1024)), dim3(V3D_BLOCK), 0, st,
                     (const int4*)coords, n, cap, H, W, occ);
  V3D_CHECK_LAUNCH();
  return V3D_OK;
}

// tiles of the persistent background-skipping grid over a (B, H, W) map = words of a `tile_state` array
// (the larger of the two tilings the persistent kernels use: runs of 80 pixels, or 5 x 16 blocks)
extern "C" int v3d_conv2d_bg_tiles(int B, int H, int W) {
  const long long flat = ((long long)B * H * W + 5 * 16 - 1) / (5 * 16);
  const long long blocks = (long long)B * ((H + D2_TH - 1) / D2_TH) * ((W + D2_TW - 1) / D2_TW);
  return (int)(flat > blocks ? flat : blocks);
}

template <int PREC>
static int dc_launch(const void* x_hi, const void* x_lo, const void* weight_image, const float* bias, DcParams p, int ksize,
                     void* y_hi, void* y_lo, float* y_nchw, const uint32_t* occ, uint32_t* work, uint32_t* tile_state,
                     uint32_t* reset_ptr, int reset_words, hipStream_t st) {
  const int B = p.B, H = p.H, W = p.W, Cin = p.Cin, Cout = p.Cout;
  if (ksize == 1 && !y_hi && Cout <= 16 && (Cin == 128 || Cin == 256)) {  // the head: stream kernel
    const dim3 sgrid(v3d_ceil_div(v3d_ceil_div(p.M, 16), 4));
    if (Cin == 128)
      hipLaunchKernelGGL((conv1x1_bf16x3_small_cout_kernel<4, PREC>), sgrid, dim3(256), 0, st, (const bf16_t*)x_hi, (const bf16_t*)x_lo,
                         (const bf16_t*)weight_image, bias, p, y_nchw);
    else
      hipLaunchKernelGGL((conv1x1_bf16x3_small_cout_kernel<8, PREC>), sgrid, dim3(256), 0, st, (const bf16_t*)x_hi, (const bf16_t*)x_lo,
                         (const bf16_t*)weight_image, bias, p, y_nchw);
    V3D_CHECK_LAUNCH();
    if (reset_ptr) V3D_CHECK_HIP(v3d_fill_async(reset_ptr, 0, (size_t)reset_words * 4, st));  // (not a persistent path: see below)
    return V3D_OK;
  }
  const bool large_ok = Cin >= 64 && H < 32768 && W < 65536 && W >= 16;  // (Cin % 32 == 0 checked by the caller) else: the 64-pixel kernel
  if (large_ok && Cout > 32) {
    // With background skipping (and a work counter) the tiles are 80 pixels instead of 144 and the grid is persistent, one
    // workgroup per CU drawing tiles from the counter: see the kernel.  The 80-pixel kernel's LDS request is padded past half a
    // CU's LDS so that two of ITS workgroups never share a CU (the 2-D tile kernel below shares on purpose).
    const int n_cu = v3d_device_cu_count();
    if (n_cu < 1) return V3D_EINVAL;
    const bool persistent = p.occ && work && p.CoutPad == DC_BN;
    if (persistent && ksize == 3 && Cin == DL_KC) {  // 2-D tiles with an LDS-resident neighbourhood
      if (v3d_ablate('d')) return V3D_OK;
      p.work = work;
      p.tile_state = tile_state;
      p.reset_ptr = reset_ptr;
      p.reset_words = reset_words;
      const int tiles2 = B * v3d_ceil_div(H, D2_TH) * v3d_ceil_div(W, D2_TW);
      // Four waves per workgroup, 64 KB of LDS and <= 256 VGPRs: TWO workgroups fit a CU (see the kernel's header).  The grid stays one
      // workgroup per CU while the live tiles are fewer than the CUs (KITTI bs = 1: 146-201 of 440 tiles -- a second resident
      // workgroup would only pair live tiles on one CU; the co-resident is another FRAME's launch); from 4 tiles per CU on (a batch,
      // the Waymo-range map) two per CU: one tile's test / image load / epilogue runs under the other's matrix phase (bs = 8:
      // 1.84 -> 1.68 ms per batch, 6 420 -> 6 580 frames/s pipelined)
      const int smem2 = 2 * D2_PLANE;
      static V3dPerDeviceFlag attr2;
      V3D_CHECK_HIP(v3d_set_max_lds(attr2, (const void*)conv2d_bf16x3_tile2d_kernel<PREC>, smem2));
      const int per_cu = tiles2 >= 4 * n_cu ? 2 : 1;
      // (192 / 384 workgroups at bs = 1 measured again under the round-6 pipeline: 4 770 / 4 730 vs 4 757 frames/s, 192 costs 20 us of latency)
      hipLaunchKernelGGL(conv2d_bf16x3_tile2d_kernel<PREC>, dim3(std::min(tiles2, n_cu * per_cu)), dim3(D2_THREADS), smem2, st, (const bf16_t*)x_hi,
                         (const bf16_t*)x_lo, (const bf16_t*)weight_image, bias, p, (bf16_t*)y_hi, (bf16_t*)y_lo);
      V3D_CHECK_LAUNCH();
      return V3D_OK;
    }
    const int mt = persistent ? 5 : 9;  // (in launch order the small tiles only add rounds: 33 vs 27 us on a full map)
    const int tiles = v3d_ceil_div(p.M, mt * 16);
    p.work = persistent ? work : nullptr;
    p.tile_state = persistent ? tile_state : nullptr;
    p.reset_ptr = persistent ? reset_ptr : nullptr;
    p.reset_words = persistent ? reset_words : 0;
    if (tile_state && y_hi && !persistent)  // every pixel of the persistent buffer is about to be computed
      V3D_CHECK_HIP(v3d_fill_async(tile_state, 0x01, (size_t)v3d_conv2d_bg_tiles(B, H, W) * sizeof(uint32_t), st));
    // a launch that is NOT persistent cannot zero other call sites' counters from inside the kernel: do it in front (same stream
    // order as the in-kernel reset: before this layer's tiles, behind the previous launch)
    if (reset_ptr && !persistent) V3D_CHECK_HIP(v3d_fill_async(reset_ptr, 0, (size_t)reset_words * 4, st));
    dim3 lgrid(persistent ? std::min(tiles, n_cu) : tiles, p.CoutPad / DC_BN);
    auto kern = mt == 9 ? (ksize == 3 ? conv2d_bf16x3_large_kernel<3, 9, PREC> : conv2d_bf16x3_large_kernel<1, 9, PREC>)
                        : (ksize == 3 ? conv2d_bf16x3_large_kernel<3, 5, PREC> : conv2d_bf16x3_large_kernel<1, 5, PREC>);
    const int smem = std::max(dl_smem(mt), 84 * 1024);
    V3D_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
    hipLaunchKernelGGL(kern, lgrid, dim3(DL_THREADS), smem, st, (const bf16_t*)x_hi, (const bf16_t*)x_lo,
                       (const bf16_t*)weight_image, bias, p, (bf16_t*)y_hi, (bf16_t*)y_lo, y_nchw);
    V3D_CHECK_LAUNCH();
    return V3D_OK;
  }
  if (tile_state && y_hi) V3D_CHECK_HIP(v3d_fill_async(tile_state, 0x01, (size_t)v3d_conv2d_bg_tiles(B, H, W) * sizeof(uint32_t), st));
  if (reset_ptr) V3D_CHECK_HIP(v3d_fill_async(reset_ptr, 0, (size_t)reset_words * 4, st));
  dim3 grid(v3d_ceil_div(p.M, DC_BM), p.CoutPad / DC_BN);
  if (ksize == 3)
    hipLaunchKernelGGL((conv2d_bf16x3_kernel<3, PREC>), grid, dim3(DC_THREADS), 0, st, (const bf16_t*)x_hi, (const bf16_t*)x_lo,
                       (const bf16_t*)weight_image, bias, p, (bf16_t*)y_hi, (bf16_t*)y_lo, y_nchw);
  else
    hipLaunchKernelGGL((conv2d_bf16x3_kernel<1, PREC>), grid, dim3(DC_THREADS), 0, st, (const bf16_t*)x_hi, (const bf16_t*)x_lo,
                       (const bf16_t*)weight_image, bias, p, (bf16_t*)y_hi, (bf16_t*)y_lo, y_nchw);
  V3D_CHECK_LAUNCH();
  return V3D_OK;
}

// The convolution on split planes in either arithmetic (`pr` NULL or prec 0: bf16x3 -- the entry points above).  reset_ptr /
// reset_words: this launch zeroes those counter words of OTHER call sites before its tiles run -- inside the kernel on the persistent
// paths, by a fill in front of it otherwise, so a chain of layers may mix both kinds (a layer off the persistent paths used to drop
// the request silently: ADVICE round 4).
extern "C" int v3d_conv2d_nhwc_split(const void* x_hi, const void* x_lo, const void* weight_image, const float* bias,
                                     int relu, int B, int H, int W, int Cin, int Cout, int ksize, void* y_hi, void* y_lo,
                                     float* y_nchw, const uint32_t* occ, int reach, const void* bg_hi, const void* bg_lo,
                                     uint32_t* work, uint32_t* tile_state, uint32_t* reset_ptr, int reset_words,
                                     const v3d_conv2d_prec* pr, v3d_stream_t stream) {
  if (reset_ptr && (reset_words < 0 || reset_words > 4096 || !work)) return V3D_EINVAL;
  if (!x_hi || !x_lo || !weight_image || B < 1 || H < 1 || W < 1 || Cout < 1) return V3D_EINVAL;
  if (occ && (!bg_hi || !bg_lo || reach < 0 || reach > 64)) return V3D_EINVAL;
  if (Cin < DC_KC || Cin % DC_KC || (ksize != 1 && ksize != 3)) return V3D_EUNSUPPORTED;
  if ((y_hi == nullptr) != (y_lo == nullptr) || (!y_hi && !y_nchw)) return V3D_EINVAL;
  if (y_hi && (Cout % 8)) return V3D_EUNSUPPORTED;
  if (dc_check_prec(pr, y_hi != nullptr)) return V3D_EINVAL;
  DcParams p;
  p.B = B; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.ks = ksize; p.relu = relu;
  p.CoutPad = (Cout + DC_BN - 1) / DC_BN * DC_BN;
  p.M = B * H * W;
  p.cout_store = Cout;
  // background skipping applies to the large-tile kernel writing split planes only (an fp32 NCHW consumer gets every
  // pixel computed: the background planes hold the SPLIT value, not the fp32 one)
  const bool skip = occ && y_hi && !y_nchw;
  p.occ = skip ? occ : nullptr;
  p.reach = reach;
  p.bg_hi = (const bf16_t*)bg_hi;
  p.bg_lo = (const bf16_t*)bg_lo;
  p.work = nullptr;
  p.tile_state = nullptr;
  p.reset_ptr = nullptr;
  p.reset_words = 0;
  const bool f16s = pr && pr->prec == V3D_PREC_F16S;
  p.in_entry = f16s ? pr->in_entry : nullptr;
  p.out_entry = (f16s && y_hi) ? pr->out_entry : nullptr;
  p.range_flag = f16s ? pr->range_flag : nullptr;
  p.w_inv = (f16s && pr->w_inv) ? pr->w_inv
                                : reinterpret_cast<const float*>(reinterpret_cast<const char*>(weight_image) + dc_image_payload_bytes(Cin, Cout, ksize)) + 1;
  hipStream_t st = (hipStream_t)stream;
  if (f16s)
    return dc_launch<1>(x_hi, x_lo, weight_image, bias, p, ksize, y_hi, y_lo, y_nchw, occ, work, tile_state, reset_ptr, reset_words, st);
  return dc_launch<0>(x_hi, x_lo, weight_image, bias, p, ksize, y_hi, y_lo, y_nchw, occ, work, tile_state, reset_ptr, reset_words, st);
}
