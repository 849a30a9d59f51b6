// voxelize.hip -- device voxelizer fused with the VoxelFeatureExtractor mean (T1 + A5 + A6).
//
// Replaces spconv.utils.VoxelGenerator.generate as driven by vision3d/core/preprocess.py:17-33
// (host C++ loop over points + a 360 MB dense index grid per call) and detector/layers.py:10-17.
//
// The sequential reference semantics (voxel ids in first-touch order of the input points, the first
// max_pts points of a voxel kept in input order) are reproduced EXACTLY by a parallel formulation:
//   K1 insert : every point inserts its linear voxel key into a hash table; per slot an
//               atomicMin keeps the smallest point index (the "first toucher") and an atomicExch
//               threads the point onto the slot's linked list.
//   K2 count  : flag[i] = "point i is the first toucher of its voxel"; per-256-chunk popcounts
//               (wave64 ballots).
//   K3 scan   : exclusive scan of the chunk counts (+ per-frame voxel bases, for max_voxels).
//   K4 emit   : rank of a first toucher (chunk offset + ballot prefix) == voxel id of the
//               sequential loop.  The same thread walks its voxel's list, keeps the max_pts smallest
//               point indices (== first-come order), writes coords/occupancy/points and the mean.
//   (K2 + K3 share a launch; for ONE frame K2 + K3 + K4 are a single-pass chained scan in one launch.)
// No host synchronisation: the voxel count stays in device memory.
#include "v3d_internal.h"

#define VOX_MAX_FRAMES 64
#define VOX_MAX_PTS 8
// points per block of the count/emit kernels: ONE 256-thread round, so the per-voxel list walks of a
// whole frame run concurrently (a 2048-point chunk serialised 8 latency-bound rounds per block)
#define VOX_CHUNK 256

struct VoxParams {
  float vs[3], lo[3];
  int grid[3];
  int max_pts, max_voxels, C, B, n_points;
  int frame_off[VOX_MAX_FRAMES + 1];
};

__device__ __forceinline__ int frame_of(const VoxParams& p, int i) {
  int b = 0;
  while (b + 1 < p.B && i >= p.frame_off[b + 1]) b++;
  return b;
}

__global__ __launch_bounds__(V3D_BLOCK) void vox_insert_kernel(const float* __restrict__ pts, const VoxParams p,
                                                               const V3dHash h, unsigned* __restrict__ first,
                                                               int* __restrict__ head, int* __restrict__ pt_slot,
                                                               int* __restrict__ next) {
  for (int i = blockIdx.x * V3D_BLOCK + threadIdx.x; i < p.n_points; i += gridDim.x * V3D_BLOCK) {
    const float* q = pts + (size_t)i * p.C;
    int c[3];
    bool ok = true;
#pragma unroll
    for (int j = 0; j < 3; j++) {
      // true IEEE division in fp32: bit-exact voxel indices
      const float f = floorf((q[j] - p.lo[j]) / p.vs[j]);
      ok = ok && (f >= 0.0f) && (f < (float)p.grid[j]);
      c[j] = (int)f;
    }
    if (!ok) {
      pt_slot[i] = -1;
      continue;
    }
    const int b = frame_of(p, i);
    const v3d_key_t key = (((v3d_key_t)b * p.grid[2] + c[2]) * p.grid[1] + c[1]) * p.grid[0] + c[0];
    const int s = v3d_hash_insert(h, key);
    if (s < 0) {  // unreachable: the table holds 2x n_points slots
      pt_slot[i] = -1;
      continue;
    }
    atomicMin(&first[s], (unsigned)i);
    next[i] = atomicExch(&head[s], i);
    pt_slot[i] = s;
  }
}

__device__ __forceinline__ bool vox_is_first(const int* pt_slot, const unsigned* first, int i, int n) {
  if (i >= n) return false;
  const int s = pt_slot[i];
  return s >= 0 && first[s] == (unsigned)i;
}

// count + scan in ONE launch: every block counts the first touchers of its chunk and publishes the count; the
// highest-index block turns the counts into exclusive offsets (v3d_common.h) and derives the per-frame bases.
// chunk_counts must read -1 at launch.
// frame_base[b] = number of first-touch points before frame b's first point; out_base[b] = first output
// row of frame b after clipping every earlier frame to max_voxels; n_voxels = total rows.
__global__ __launch_bounds__(V3D_BLOCK) void vox_count_scan_kernel(const int* __restrict__ pt_slot,
                                                                   const unsigned* __restrict__ first, const VoxParams p,
                                                                   int* __restrict__ chunk_counts, int n_chunks,
                                                                   int* __restrict__ frame_base, int* __restrict__ out_base,
                                                                   int* __restrict__ n_voxels) {
  __shared__ int lds[8];
  __shared__ int fb_s[VOX_MAX_FRAMES + 1];
  const int base = blockIdx.x * VOX_CHUNK;
  int cnt = 0;
  for (int r = 0; r < VOX_CHUNK / V3D_BLOCK; r++) {
    int tot;
    v3d_block_rank(vox_is_first(pt_slot, first, base + r * V3D_BLOCK + threadIdx.x, p.n_points), tot, lds);
    cnt += tot;
  }
  if (threadIdx.x == 0) v3d_publish_count(chunk_counts + blockIdx.x, cnt);
  if (blockIdx.x != gridDim.x - 1) return;

  const int total = v3d_block_exclusive_scan_global(chunk_counts, n_chunks, lds);
  __syncthreads();
  // per-frame bases: wave w handles frames w, w+4, ...
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  for (int b = w; b <= p.B; b += V3D_BLOCK / V3D_WAVE) {
    const int pos = p.frame_off[b];
    int val;
    if (pos >= p.n_points) {
      val = total;
    } else {
      const int chunk = pos / VOX_CHUNK;
      int c = 0;
      for (int i = chunk * VOX_CHUNK + lane; i < pos; i += 64) c += vox_is_first(pt_slot, first, i, p.n_points);
      for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o);
      val = v3d_load_coherent(chunk_counts + chunk) + c;
    }
    if (lane == 0) {
      frame_base[b] = val;
      fb_s[b] = val;
    }
  }
  __syncthreads();
  if (tid == 0) {
    int acc = 0;
    for (int b = 0; b < p.B; b++) {
      out_base[b] = acc;
      acc += min(fb_s[b + 1] - fb_s[b], p.max_voxels);
    }
    out_base[p.B] = acc;
    *n_voxels = acc;
  }
}

// CHAINED (one frame, B == 1): count, scan and emit in ONE launch -- every block counts the first touchers of its chunk, PUBLISHES the
// count (chunk_offsets reads -1 at launch: part of the frame's 0xFF fill), adds up the counts of all its predecessors (spinning on
// slots that still read -1: workgroups are dispatched in index order and publish before they wait, so every wait is on a block that
// is running or done) and emits at prefix + rank; the highest-index block writes the voxel count.  The same single-pass chained
// scan as rb_scan_emit_kernel (rulebook.hip); replaces vox_count_scan_kernel + this kernel's unchained form, which stay for
// batches (B > 1: the per-frame bases need every earlier frame's total).
template <bool CHAINED>
__global__ __launch_bounds__(V3D_BLOCK) void vox_emit_kernel(const float* __restrict__ pts, const VoxParams p,
                                                             const int* __restrict__ pt_slot,
                                                             const unsigned* __restrict__ first,
                                                             const int* __restrict__ head, const int* __restrict__ next,
                                                             int* __restrict__ chunk_offsets,
                                                             const int* __restrict__ frame_base,
                                                             const int* __restrict__ out_base,
                                                             float* __restrict__ voxels, int* __restrict__ coords,
                                                             int* __restrict__ occupancy, float* __restrict__ mean,
                                                             const V3dHash site_hash, int* __restrict__ site_vals,
                                                             int site_d, int site_h, int site_w, int* __restrict__ n_voxels) {
  __shared__ int lds[4];
  static_assert(VOX_CHUNK == V3D_BLOCK, "one round per block");
  const int i = blockIdx.x * VOX_CHUNK + threadIdx.x;
  const bool flag = vox_is_first(pt_slot, first, i, p.n_points);
  int tot;
  int rank = v3d_block_rank(flag, tot, lds);
  if constexpr (CHAINED) {
    if (threadIdx.x == 0) v3d_publish_count(chunk_offsets + blockIdx.x, tot);
  }
  // ---- everything about the voxel that does not depend on its NUMBER, in front of the (chained) wait for the number: the cell,
  //      the walk of the voxel's point list (the max_pts smallest point indices, ascending = first-come order) and the points
  int c[3] = {0, 0, 0};
  int best[VOX_MAX_PTS];
#pragma unroll
  for (int k = 0; k < VOX_MAX_PTS; k++) best[k] = 0x7FFFFFFF;
  int cnt = 0;
  float4 pv[VOX_MAX_PTS];
  if (flag) {
    const float* q = pts + (size_t)i * p.C;  // voxel coordinates from the first toucher itself
#pragma unroll
    for (int j = 0; j < 3; j++) c[j] = (int)floorf((q[j] - p.lo[j]) / p.vs[j]);
    for (int j = head[pt_slot[i]], guard = 0; j >= 0 && guard < p.n_points; j = next[j], guard++) {
      cnt++;
      int x = j;
#pragma unroll
      for (int k = 0; k < VOX_MAX_PTS; k++) {  // sorted insert by compare-exchange
        const int lo_ = min(best[k], x);
        x = max(best[k], x);
        best[k] = lo_;
      }
    }
  }
  const int occ = min(cnt, p.max_pts);
  if (p.C == 4) {
    // four channels (x, y, z, intensity): the points as 16-byte loads, all in flight at once (the generic loop below issues
    // max_pts x C scalar loads); same sums in the same order
#pragma unroll
    for (int k = 0; k < VOX_MAX_PTS; k++)
      pv[k] = (flag && k < p.max_pts && k < occ) ? reinterpret_cast<const float4*>(pts)[best[k]] : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  if constexpr (CHAINED) {
    __shared__ int s_part[V3D_BLOCK / V3D_WAVE];
    asm volatile("" ::: "memory");  // (the loads above are issued in front of the spin)
    int part = 0;
    for (int cb = threadIdx.x; cb < (int)blockIdx.x; cb += V3D_BLOCK) part += v3d_wait_count(chunk_offsets + cb);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off);
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = part;
    __syncthreads();
    const int prefix = s_part[0] + s_part[1] + s_part[2] + s_part[3];
    rank += prefix;
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) *n_voxels = min(prefix + tot, p.max_voxels);
  } else {
    rank += chunk_offsets[blockIdx.x];
  }
  if (!flag) return;
  const int b = CHAINED ? 0 : frame_of(p, i);
  const int local = CHAINED ? rank : rank - frame_base[b];
  if (local >= p.max_voxels) return;  // `continue` variant of the max_voxels rule
  const int v = CHAINED ? local : out_base[b] + local;
  reinterpret_cast<int4*>(coords)[v] = make_int4(b, c[2], c[1], c[0]);
  if (site_vals) {  // the coordinate hash the first submanifold rulebook needs (saves the rb_hash_build launch)
    // key over the RULEBOOK's grid (D, H, W), which may be larger than the voxel grid (SECOND pads z by one)
    const v3d_key_t key = (((v3d_key_t)b * site_d + c[2]) * site_h + c[1]) * site_w + c[0];
    const int hs = v3d_site_insert(site_hash, key, (unsigned)v);  // (each voxel is inserted once, with its row)
    if (hs >= 0) site_vals[hs] = v;
  }
  occupancy[v] = occ;
  if (p.C == 4) {
    float4 sm = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int k = 0; k < VOX_MAX_PTS; k++) {
      if (k < p.max_pts) {
        if (voxels) reinterpret_cast<float4*>(voxels)[(size_t)v * p.max_pts + k] = pv[k];
        sm.x += pv[k].x; sm.y += pv[k].y; sm.z += pv[k].z; sm.w += pv[k].w;
      }
    }
    const float fo = (float)occ;
    if (mean) reinterpret_cast<float4*>(mean)[v] = make_float4(sm.x / fo, sm.y / fo, sm.z / fo, sm.w / fo);
    return;
  }
  for (int ch = 0; ch < p.C; ch++) {
    float sacc = 0.f;
#pragma unroll
    for (int k = 0; k < VOX_MAX_PTS; k++) {
      if (k < p.max_pts) {
        const float val = k < occ ? pts[(size_t)best[k] * p.C + ch] : 0.f;
        if (voxels) voxels[((size_t)v * p.max_pts + k) * p.C + ch] = val;
        sacc += val;
      }
    }
    if (mean) mean[(size_t)v * p.C + ch] = sacc / (float)occ;
  }
}

extern "C" size_t v3d_voxelize_workspace(int n_points) {
  const size_t n = (size_t)(n_points > 0 ? n_points : 1);
  const size_t cap = v3d_hash_capacity((long long)n);
  const size_t chunks = (n + VOX_CHUNK - 1) / VOX_CHUNK;
  return v3d_align(cap * 8) + 2 * v3d_align(cap * 4) + 2 * v3d_align(n * 4) + v3d_align(chunks * 4) +
         2 * v3d_align((VOX_MAX_FRAMES + 1) * 4) + 256;
}

extern "C" int v3d_voxelize(const float* points, int n_points, int C, const int32_t* frame_offsets_host, int B,
                            const float* voxel_size_host, const float* bounds_host, int max_pts, int max_voxels,
                            float* voxels, int32_t* coords, int32_t* occupancy, float* mean, int32_t* n_voxels,
                            void* workspace, size_t workspace_bytes, v3d_stream_t stream) {
  return v3d_i_voxelize(points, n_points, C, frame_offsets_host, B, voxel_size_host, bounds_host, max_pts, max_voxels,
                        voxels, coords, occupancy, mean, n_voxels, workspace, workspace_bytes, 1, nullptr, nullptr,
                        (hipStream_t)stream);
}

// clear_tables = 0: the caller has already filled the head of the workspace (hash keys | first | head) with
// 0xFF, e.g. as part of one arena-wide memset (second_plan.hip).
int v3d_i_voxelize(const float* points, int n_points, int C, const int32_t* frame_offsets_host, int B,
                   const float* voxel_size_host, const float* bounds_host, int max_pts, int max_voxels, float* voxels,
                   int32_t* coords, int32_t* occupancy, float* mean, int32_t* n_voxels, void* workspace,
                   size_t workspace_bytes, int clear_tables, const V3dRbHash* site_hash, const int32_t* site_shape,
                   hipStream_t st) {
  if (n_points < 0 || C < 3 || B < 1 || B > VOX_MAX_FRAMES || max_pts < 1 || max_pts > VOX_MAX_PTS || max_voxels < 1)
    return V3D_EINVAL;
  if (!frame_offsets_host || !voxel_size_host || !bounds_host || !coords || !occupancy || !n_voxels) return V3D_EINVAL;
  if (frame_offsets_host[0] != 0 || frame_offsets_host[B] != n_points) return V3D_EINVAL;
  if (n_points == 0) {
    V3D_CHECK_HIP(v3d_fill_async(n_voxels, 0, sizeof(int32_t), st));
    return V3D_OK;
  }
  if (!points || !workspace) return V3D_EINVAL;
  if (site_hash && (long long)B * max_voxels > V3D_SITE_MAX_ROWS && n_points > V3D_SITE_MAX_ROWS) return V3D_EUNSUPPORTED;  // rows ride in 24 bits
  VoxParams p;
  for (int j = 0; j < 3; j++) {
    p.vs[j] = voxel_size_host[j];
    p.lo[j] = bounds_host[j];
    p.grid[j] = (int)lroundf((bounds_host[3 + j] - bounds_host[j]) / voxel_size_host[j]);
    if (p.grid[j] < 1) return V3D_EINVAL;
  }
  p.max_pts = max_pts;
  p.max_voxels = max_voxels;
  p.C = C;
  p.B = B;
  p.n_points = n_points;
  for (int b = 0; b <= B; b++) {
    p.frame_off[b] = frame_offsets_host[b];
    if (b > 0 && p.frame_off[b] < p.frame_off[b - 1]) return V3D_EINVAL;
  }
  const unsigned cap = v3d_hash_capacity(n_points);
  const int chunks = v3d_ceil_div(n_points, VOX_CHUNK);
  V3dArena ar(workspace, workspace_bytes);
  // keys | first | head are contiguous: ONE memset(0xFF) resets all three (EMPTY / UINT_MAX / -1)
  v3d_key_t* keys = ar.take<v3d_key_t>(cap);
  unsigned* first = ar.take<unsigned>(cap);
  int* head = ar.take<int>(cap);
  int* pt_slot = ar.take<int>(n_points);
  int* next = ar.take<int>(n_points);
  int* chunk_counts = ar.take<int>(chunks);
  int* frame_base = ar.take<int>(VOX_MAX_FRAMES + 1);
  int* out_base = ar.take<int>(VOX_MAX_FRAMES + 1);
  if (!ar.ok()) return V3D_EWORKSPACE;
  if (clear_tables) {
    V3D_CHECK_HIP(v3d_fill_async(keys, 0xFF, (size_t)((char*)head - (char*)keys) + (size_t)cap * 4, st));
    V3D_CHECK_HIP(v3d_fill_async(chunk_counts, 0xFF, (size_t)chunks * 4, st));  // -1 = "count not published yet"
  }
  V3dHash h = v3d_make_hash(keys, cap);
  const int ins_blocks = min(v3d_ceil_div(n_points, V3D_BLOCK), 2048);
  hipLaunchKernelGGL(vox_insert_kernel, dim3(ins_blocks), dim3(V3D_BLOCK), 0, st, points, p, h, first, head, pt_slot,
                     next);
  V3dHash sh = v3d_make_hash(site_hash ? site_hash->keys : keys, site_hash ? site_hash->hcap : cap);
  int* site_vals = (site_hash && site_shape) ? site_hash->vals : (int*)nullptr;
  const int sd = site_shape ? site_shape[0] : 0, shh = site_shape ? site_shape[1] : 0, sw = site_shape ? site_shape[2] : 0;
  if (B == 1) {  // one frame: count + scan + emit as a single-pass chained scan, 2 launches per voxelization instead of 3
    hipLaunchKernelGGL(vox_emit_kernel<true>, dim3(chunks), dim3(V3D_BLOCK), 0, st, points, p, pt_slot, first, head, next,
                       chunk_counts, frame_base, out_base, voxels, coords, occupancy, mean, sh, site_vals, sd, shh, sw, n_voxels);
  } else {
    hipLaunchKernelGGL(vox_count_scan_kernel, dim3(chunks), dim3(V3D_BLOCK), 0, st, pt_slot, first, p, chunk_counts, chunks,
                       frame_base, out_base, n_voxels);
    hipLaunchKernelGGL(vox_emit_kernel<false>, dim3(chunks), dim3(V3D_BLOCK), 0, st, points, p, pt_slot, first, head, next,
                       chunk_counts, frame_base, out_base, voxels, coords, occupancy, mean, sh, site_vals, sd, shh, sw, n_voxels);
  }
  V3D_CHECK_LAUNCH();
  return V3D_OK;
}
