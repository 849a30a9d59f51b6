// rulebook.hip -- hash-indexed sparse-convolution rulebooks (T3, spconv ops.get_indice_pairs) on gfx950.
//
// Reached in the reference through spconv.SubMConv3d / spconv.SparseConv3d
// (vision3d/detector/sparse_cnn.py:15-30,153-175).  spconv builds per-offset (in,out) pair lists with a
// dense index grid; here the rulebook is an OUTPUT-STATIONARY neighbour table nbr[k][o] (k-major, so
// lanes = consecutive output rows read/write coalesced) built from an open-addressing hash of the
// active coordinates -- no dense grid, so the 0.55 G-cell Waymo-range volume costs the same as KITTI.
//
// Strided convolutions must also CREATE the output site list.  Sequential semantics ("number the
// outputs in the order a loop over (input i, offset k) first touches them") are reproduced in
// parallel: ticket t = i*K + k, per output-hash slot an atomicMin keeps the smallest ticket, and a
// wave64 ballot/popcount scan over "ticket t is its slot's minimum" yields exactly the sequential
// numbering.  Everything is sized by capacities; live counts stay in device memory.
#include "rb_device.h"

// ---------------------------------------------------------------------------------- hash of the inputs
__global__ __launch_bounds__(V3D_BLOCK) void rb_hash_build_kernel(const int4* __restrict__ coords,
                                                                  const int* __restrict__ n_ptr, int cap,
                                                                  const RbGeom g, const V3dHash h,
                                                                  int* __restrict__ vals) {
  const int n = min(*n_ptr, cap);
  for (int i = blockIdx.x * V3D_BLOCK + threadIdx.x; i < n; i += gridDim.x * V3D_BLOCK) {
    const int4 c = coords[i];
    if ((unsigned)c.x >= (unsigned)RB_MAX_BATCH) continue;  // no key (see rb_key)
    const int s = v3d_site_insert(h, rb_key(c.x, c.y, c.z, c.w, g.in_shape), (unsigned)i);
    if (s >= 0) vals[s] = i;
  }
}

__global__ __launch_bounds__(V3D_BLOCK) void rb_candidates_kernel(const int4* __restrict__ coords,
                                                                  const int* __restrict__ n_ptr, int cap_in,
                                                                  const RbGeom g, const V3dHash h,
                                                                  unsigned* __restrict__ first_ticket,
                                                                  int* __restrict__ cand_slot,
                                                                  int* __restrict__ overflow, int* __restrict__ overflow_any) {
  rb_candidates_body(coords, n_ptr, cap_in, g, h, first_ticket, cand_slot, overflow, overflow_any, blockIdx.x, gridDim.x);
}

// grid = rb_subm_blocks(cap, K) blocks (block -> (group of kernel offsets, 256 output rows)) + the blocks of a candidate job riding along
__global__ __launch_bounds__(V3D_BLOCK) void rb_subm_nbr_kernel(const int4* __restrict__ coords,
                                                                const int* __restrict__ n_ptr, int cap,
                                                                const RbGeom g, const V3dHash h,
                                                                const int* __restrict__ vals, int* __restrict__ nbr,
                                                                int subm_blocks, const RbCandJob job) {
  if ((int)blockIdx.x >= subm_blocks) {
    rb_candidates_body(job.coords, job.n_ptr, job.cap_in, job.g, job.h, job.first_ticket, job.cand_slot, job.overflow,
                       job.overflow_any, blockIdx.x - subm_blocks, job.blocks);
    return;
  }
  rb_subm_block(coords, min(*n_ptr, cap), cap, g, h, nbr, blockIdx.x);
}

// ---------------------------------------------------------------------------------- host side
static int fill_geom(RbGeom& g, const int32_t* shape, const int32_t* ks, const int32_t* stride, const int32_t* pad) {
  g.K = 1;
  for (int j = 0; j < 3; j++) {
    g.in_shape[j] = shape[j];
    g.ks[j] = ks[j];
    g.stride[j] = stride ? stride[j] : 1;
    g.pad[j] = pad ? pad[j] : ks[j] / 2;
    if (g.in_shape[j] < 1 || g.ks[j] < 1 || g.stride[j] < 1 || g.pad[j] < 0) return V3D_EINVAL;
    g.out_shape[j] = (g.in_shape[j] + 2 * g.pad[j] - g.ks[j]) / g.stride[j] + 1;
    if (g.out_shape[j] < 1) return V3D_EINVAL;
    g.K *= g.ks[j];
  }
  g.Kc = 1;
  for (int j = 0; j < 3; j++) {
    g.kc[j] = (g.ks[j] + g.stride[j] - 1) / g.stride[j];
    g.Kc *= g.kc[j];
  }
  if (g.K > 62) return V3D_EUNSUPPORTED;  // the kernel offset rides in 6 bits of a candidate word beside the slot (-1 = dead)
  // linear cell keys (batch index in front: up to 64 frames) must fit the 40 key bits of a site table's words (v3d_common.h)
  // -- input AND output grid, and the batch index itself is bounded where keys are made (RB_MAX_BATCH)
  if ((long long)g.in_shape[0] * g.in_shape[1] * g.in_shape[2] >= (1ll << 34)) return V3D_EUNSUPPORTED;
  if ((long long)g.out_shape[0] * g.out_shape[1] * g.out_shape[2] >= (1ll << 34)) return V3D_EUNSUPPORTED;
  return V3D_OK;
}

// ---- internal (C++) entry points shared by the C ABI wrappers below and the fused backbone plan
// (second_plan.hip).  A V3dRbHash is a coordinate hash of ONE active-site set: keys[hcap] + vals[hcap]
// (vals[slot] = row index).  The strided rulebook leaves exactly such a hash of its OUTPUT sites behind,
// which is the table the next stage's submanifold rulebook needs -- the plan reuses it.
int v3d_i_hash_build(const int32_t* coords, const int32_t* n, int cap, const int32_t* shape, V3dRbHash h, int clear,
                     hipStream_t st) {
  RbGeom g;
  const int32_t ones[3] = {1, 1, 1};
  int rc = fill_geom(g, shape, ones, nullptr, nullptr);
  if (rc) return rc;
  if (cap > V3D_SITE_MAX_ROWS) return V3D_EUNSUPPORTED;  // rows ride in 24 bits of the site table's key words
  if (clear) V3D_CHECK_HIP(v3d_fill_async(h.keys, 0xFF, (size_t)h.hcap * 8, st));
  V3dHash hh = v3d_make_hash(h.keys, h.hcap);
  hipLaunchKernelGGL(rb_hash_build_kernel, dim3(min(v3d_ceil_div(cap, V3D_BLOCK), 2048)), dim3(V3D_BLOCK), 0, st,
                     (const int4*)coords, n, cap, g, hh, h.vals);
  V3D_CHECK_LAUNCH();
  return V3D_OK;
}

// the candidate pass of a strided layer as a job for another launch (see RbCandJob); next == nullptr: an empty job
static int make_cand_job(const V3dRbCandNext* next, RbCandJob& job) {
  job = RbCandJob{};
  if (!next) return V3D_OK;
  int rc = fill_geom(job.g, next->shape, next->ksize, next->stride, next->padding);
  if (rc) return rc;
  const long long tickets = (long long)next->cap_in * job.g.Kc;
  if (tickets >= (1ll << 31) || next->out.hcap > (1u << RB_SLOT_BITS)) return V3D_EUNSUPPORTED;
  if ((char*)next->first_ticket != (char*)next->out.keys + (size_t)next->out.hcap * 8 ||
      (char*)next->out.vals != (char*)next->first_ticket + (size_t)next->out.hcap * 4)
    return V3D_EINVAL;
  job.coords = (const int4*)next->coords_in;
  job.n_ptr = next->n_in;
  job.cap_in = next->cap_in;
  job.h = v3d_make_hash(next->out.keys, next->out.hcap);
  job.first_ticket = next->first_ticket;
  job.cand_slot = next->cand_slot;
  job.overflow = next->overflow;
  job.overflow_any = next->overflow_any;
  job.blocks = min(v3d_ceil_div(tickets, V3D_BLOCK), 4096);
  return V3D_OK;
}

int v3d_i_subm_nbr(const int32_t* coords, const int32_t* n, int cap, const int32_t* shape, const int32_t* ksize,
                   V3dRbHash h, int32_t* nbr, hipStream_t st, const V3dRbCandNext* next) {
  RbGeom g;
  int rc = fill_geom(g, shape, ksize, nullptr, nullptr);
  if (rc) return rc;
  for (int j = 0; j < 3; j++)
    if (!(g.ks[j] & 1)) return V3D_EINVAL;  // submanifold needs odd kernels
  V3dHash hh = v3d_make_hash(h.keys, h.hcap);
  RbCandJob job;
  rc = make_cand_job(next, job);
  if (rc) return rc;
  const int subm_blocks = rb_subm_blocks(cap, g);
  hipLaunchKernelGGL(rb_subm_nbr_kernel, dim3(subm_blocks + job.blocks), dim3(V3D_BLOCK), 0, st,
                     (const int4*)coords, n, cap, g, hh, h.vals, nbr, subm_blocks, job);
  V3D_CHECK_LAUNCH();
  return V3D_OK;
}

// a rulebook step as a launch of its own (what a plan does with a step no sparse layer's launch carried)
__global__ __launch_bounds__(V3D_BLOCK) void rb_scan_emit_kernel(const RbScanJob r) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[RB_SCAN_LDS];
  rb_rider_run(r, blockIdx.x, lds);
}
__global__ __launch_bounds__(V3D_BLOCK) void rb_fill_nbr_kernel(const int* __restrict__ n_ptr, int cap_in, int K,
                                                                const int* __restrict__ cand_slot,
                                                                const int* __restrict__ vals, int cap_out,
                                                                int* __restrict__ nbr, int fill_blocks,
                                                                const int4* __restrict__ coords_out,
                                                                const int* __restrict__ n_out_ptr, const RbGeom sg,
                                                                const V3dHash sh, int* __restrict__ subm_nbr, int subm_blocks,
                                                                const RbCandJob job) {
  rb_fill_nbr_body(n_ptr, cap_in, K, cand_slot, vals, cap_out, nbr, fill_blocks, coords_out, n_out_ptr, sg, sh, subm_nbr, subm_blocks, job,
                   blockIdx.x);
}
int v3d_i_rb_step_launch(const RbStep& s, hipStream_t st) {
  if (s.kind == 1) {
    if (s.scan.blocks < 1) return V3D_OK;
    hipLaunchKernelGGL(rb_scan_emit_kernel, dim3(s.scan.blocks), dim3(V3D_BLOCK), 0, st, s.scan);
  } else {
    const RbFillJob& f = s.fill;
    if (f.blocks < 1) return V3D_OK;
    hipLaunchKernelGGL(rb_fill_nbr_kernel, dim3(f.blocks), dim3(V3D_BLOCK), 0, st, f.n_ptr, f.cap_in, f.Kc, f.cand_slot, f.vals, f.cap_out, f.nbr,
                       f.fill_blocks, f.coords_out, f.n_out, f.sg, f.sh, f.subm_nbr, f.subm_blocks, f.job);
  }
  V3D_CHECK_LAUNCH();
  return V3D_OK;
}

// The two steps of a strided layer's rulebook behind its candidate pass, as descriptors: `scan` (count + scan + emit: numbers the
// output sites, fills the output hash's rows, initialises the table's live columns) and `fill` (the table, the submanifold table of
// the output sites, the NEXT strided layer's candidate pass).  Arguments as v3d_i_sparse_rulebook; nothing is launched.
int v3d_i_sparse_rulebook_steps(const int32_t* coords_in, const int32_t* n_in, int cap_in, const int32_t* shape,
                                const int32_t* ksize, const int32_t* stride, const int32_t* padding, int32_t* coords_out,
                                int32_t* n_out, int cap_out, int32_t* nbr, int32_t* overflow, V3dRbHash out,
                                unsigned* first_ticket, int* cand_slot, int* chunk_counts, const int32_t* next_subm_ksize,
                                int32_t* next_subm_nbr, int32_t* overflow_any, const V3dRbCandNext* next, int init_columns,
                                RbStep* scan, RbStep* fill) {
  RbGeom g;
  int rc = fill_geom(g, shape, ksize, stride, padding);
  if (rc) return rc;
  const long long tickets = (long long)cap_in * g.Kc;
  if (tickets >= (1ll << 31) || cap_out > V3D_SITE_MAX_ROWS || out.hcap > (1u << RB_SLOT_BITS)) return V3D_EUNSUPPORTED;
  if ((char*)first_ticket != (char*)out.keys + (size_t)out.hcap * 8 || (char*)out.vals != (char*)first_ticket + (size_t)out.hcap * 4)
    return V3D_EINVAL;
  *scan = RbStep{};
  scan->kind = 1;
  RbScanJob& r = scan->scan;
  r.blocks = v3d_ceil_div(tickets, V3D_SCAN_CHUNK);
  r.cap_in = cap_in;
  r.cap_out = cap_out;
  r.coords = (const int4*)coords_in;
  r.n_ptr = n_in;
  r.g = g;
  r.cand_slot = cand_slot;
  r.first_ticket = first_ticket;
  r.chunk_counts = chunk_counts;
  r.coords_out = (int4*)coords_out;
  r.vals = out.vals;
  r.n_out = n_out;
  r.overflow = overflow;
  r.overflow_any = overflow_any;
  r.nbr_init = init_columns ? nbr : nullptr;
  r.out_hash = v3d_make_hash(out.keys, out.hcap);
  *fill = RbStep{};
  fill->kind = 2;
  RbFillJob& f = fill->fill;
  f.n_ptr = n_in;
  f.cap_in = cap_in;
  f.Kc = g.Kc;
  f.cand_slot = cand_slot;
  f.vals = out.vals;
  f.cap_out = cap_out;
  f.nbr = nbr;
  f.fill_blocks = min(v3d_ceil_div(tickets, V3D_BLOCK), 4096);
  f.coords_out = (const int4*)coords_out;
  f.n_out = n_out;
  f.sg = g;
  f.sh = r.out_hash;
  if (next_subm_ksize && next_subm_nbr) {
    rc = fill_geom(f.sg, g.out_shape, next_subm_ksize, nullptr, nullptr);
    if (rc) return rc;
    for (int j = 0; j < 3; j++)
      if (!(f.sg.ks[j] & 1)) return V3D_EINVAL;
    f.subm_blocks = rb_subm_blocks(cap_out, f.sg);
    f.subm_nbr = next_subm_nbr;
  }
  rc = make_cand_job(next, f.job);
  if (rc) return rc;
  f.blocks = f.fill_blocks + f.subm_blocks + f.job.blocks;
  return V3D_OK;
}

// scratch: first_ticket[out.hcap] (must directly follow out.keys and precede out.vals in memory so that ONE
// memset resets keys|first_ticket|vals), cand_slot[cap_in*K], chunk_counts[ceil(cap_in*K/2048)].
int v3d_i_sparse_rulebook(const int32_t* coords_in, const int32_t* n_in, int cap_in, const int32_t* shape,
                          const int32_t* ksize, const int32_t* stride, const int32_t* padding, int32_t* coords_out,
                          int32_t* n_out, int cap_out, int32_t* nbr, int32_t* overflow, V3dRbHash out,
                          unsigned* first_ticket, int* cand_slot, int* chunk_counts, int32_t* out_shape, int clear,
                          const int32_t* next_subm_ksize, int32_t* next_subm_nbr, hipStream_t st, int32_t* overflow_any,
                          int candidates_done, const V3dRbCandNext* next, int init_columns) {
  RbStep scan, fill;
  int rc = v3d_i_sparse_rulebook_steps(coords_in, n_in, cap_in, shape, ksize, stride, padding, coords_out, n_out, cap_out, nbr, overflow,
                                       out, first_ticket, cand_slot, chunk_counts, next_subm_ksize, next_subm_nbr, overflow_any, next,
                                       init_columns, &scan, &fill);
  if (rc) return rc;
  const RbGeom& g = scan.scan.g;
  if (out_shape)
    for (int j = 0; j < 3; j++) out_shape[j] = g.out_shape[j];
  if (clear) {  // overflow flag convention: <= 0 (0 or -1) = fine, 1 = a capacity was hit
    V3D_CHECK_HIP(v3d_fill_async(out.keys, 0xFF, (size_t)out.hcap * 16, st));
    V3D_CHECK_HIP(v3d_fill_async(nbr, 0xFF, (size_t)g.K * cap_out * 4, st));
    V3D_CHECK_HIP(v3d_fill_async(overflow, 0, 4, st));
    V3D_CHECK_HIP(v3d_fill_async(chunk_counts, 0xFF, (size_t)scan.scan.blocks * 4, st));  // -1 = "count not published yet"
  }
  if (!candidates_done)  // (else: the pass rode in an earlier launch of the caller's -- v3d_i_subm_nbr / the previous strided layer)
    hipLaunchKernelGGL(rb_candidates_kernel, dim3(fill.fill.fill_blocks), dim3(V3D_BLOCK), 0, st, (const int4*)coords_in, n_in, cap_in,
                       g, scan.scan.out_hash, first_ticket, cand_slot, overflow, overflow_any);
  rc = v3d_i_rb_step_launch(scan, st);
  if (rc) return rc;
  return v3d_i_rb_step_launch(fill, st);
}

extern "C" size_t v3d_rulebook_workspace(int cap_in, int cap_out, int K) {
  const size_t ci = (size_t)(cap_in > 0 ? cap_in : 1), co = (size_t)(cap_out > 0 ? cap_out : 1);
  const size_t hcap = v3d_hash_capacity((long long)(ci > co ? ci : co));
  const size_t tickets = ci * (size_t)(K > 0 ? K : 1);
  const size_t chunks = (tickets + V3D_SCAN_CHUNK - 1) / V3D_SCAN_CHUNK;
  return v3d_align(hcap * 8) + 2 * v3d_align(hcap * 4) + v3d_align(tickets * 4) + v3d_align(chunks * 4) + 256;
}

extern "C" int v3d_rulebook_subm(const int32_t* coords, const int32_t* n, int cap, const int32_t* spatial_shape_host,
                                 const int32_t* ksize_host, int32_t* nbr, void* workspace, size_t workspace_bytes,
                                 v3d_stream_t stream) {
  hipStream_t st = (hipStream_t)stream;
  if (!coords || !n || cap < 1 || !spatial_shape_host || !ksize_host || !nbr || !workspace) return V3D_EINVAL;
  const unsigned hcap = v3d_hash_capacity(cap);
  V3dArena ar(workspace, workspace_bytes);
  V3dRbHash h;
  h.keys = ar.take<v3d_key_t>(hcap);
  h.vals = ar.take<int>(hcap);
  h.hcap = hcap;
  if (!ar.ok()) return V3D_EWORKSPACE;
  int rc = v3d_i_hash_build(coords, n, cap, spatial_shape_host, h, 1, st);
  if (rc) return rc;
  return v3d_i_subm_nbr(coords, n, cap, spatial_shape_host, ksize_host, h, nbr, st, nullptr);
}

extern "C" int v3d_rulebook_sparse(const int32_t* coords_in, const int32_t* n_in, int cap_in,
                                   const int32_t* spatial_shape_host, const int32_t* ksize_host,
                                   const int32_t* stride_host, const int32_t* padding_host, int32_t* coords_out,
                                   int32_t* n_out, int cap_out, int32_t* nbr, int32_t* overflow, void* workspace,
                                   size_t workspace_bytes, v3d_stream_t stream) {
  hipStream_t st = (hipStream_t)stream;
  if (!coords_in || !n_in || cap_in < 1 || cap_out < 1 || !spatial_shape_host || !ksize_host || !stride_host ||
      !padding_host || !coords_out || !n_out || !nbr || !overflow || !workspace)
    return V3D_EINVAL;
  const unsigned hcap = v3d_hash_capacity(cap_in > cap_out ? cap_in : cap_out);
  long long K = (long long)ksize_host[0] * ksize_host[1] * ksize_host[2];
  if (K < 1) return V3D_EINVAL;
  const long long tickets = (long long)cap_in * K;
  const int chunks = v3d_ceil_div(tickets, V3D_SCAN_CHUNK);
  V3dArena ar(workspace, workspace_bytes);
  // keys | first_ticket | vals contiguous -> one memset(0xFF): EMPTY / UINT_MAX / -1 (hcap*4 is a multiple of 256)
  V3dRbHash h;
  h.keys = ar.take<v3d_key_t>(hcap);
  unsigned* first_ticket = ar.take<unsigned>(hcap);
  h.vals = ar.take<int>(hcap);
  h.hcap = hcap;
  int* cand_slot = ar.take<int>((size_t)tickets);
  int* chunk_counts = ar.take<int>(chunks);
  if (!ar.ok()) return V3D_EWORKSPACE;
  return v3d_i_sparse_rulebook(coords_in, n_in, cap_in, spatial_shape_host, ksize_host, stride_host, padding_host,
                               coords_out, n_out, cap_out, nbr, overflow, h, first_ticket, cand_slot, chunk_counts,
                               nullptr, 1, nullptr, nullptr, st, nullptr, 0, nullptr, 0);
}

// ---------------------------------------------------------------------------------- transposed table (backward)
// nbrT[k][i] = o  <=>  nbr[k][o] = i.  Every (input row, offset) feeds at most one output row, so the scatter has
// no collisions.  Used by the data-gradient of strided layers: dX[i] = sum_k dY[nbrT[k][i]] @ W[k]^T is the same
// output-stationary gather as the forward pass (submanifold tables are their own transpose with k reversed).
__global__ __launch_bounds__(V3D_BLOCK) void rb_transpose_kernel(const int* __restrict__ nbr, const int* __restrict__ n_out_ptr,
                                                                 int cap_out, int K, int cap_in, int* __restrict__ nbrT) {
  const int n = min(*n_out_ptr, cap_out);
  const long long total = (long long)K * n;
  for (long long t = (long long)blockIdx.x * V3D_BLOCK + threadIdx.x; t < total; t += (long long)gridDim.x * V3D_BLOCK) {
    const int k = (int)(t / n), o = (int)(t % n);
    const int i = nbr[(size_t)k * cap_out + o];
    if (i >= 0 && i < cap_in) nbrT[(size_t)k * cap_in + i] = o;
  }
}

extern "C" int v3d_rulebook_transpose(const int32_t* nbr, const int32_t* n_out, int cap_out, int K, int cap_in,
                                      int32_t* nbr_t, v3d_stream_t stream) {
  hipStream_t st = (hipStream_t)stream;
  if (!nbr || !n_out || !nbr_t || cap_out < 1 || cap_in < 1 || K < 1) return V3D_EINVAL;
  V3D_CHECK_HIP(v3d_fill_async(nbr_t, 0xFF, (size_t)K * cap_in * 4, st));
  const long long total = (long long)K * cap_out;
  hipLaunchKernelGGL(rb_transpose_kernel, dim3(min(v3d_ceil_div(total, V3D_BLOCK), 4096)), dim3(V3D_BLOCK), 0, st, nbr, n_out,
                     cap_out, K, cap_in, nbr_t);
  V3D_CHECK_LAUNCH();
  return V3D_OK;
}
