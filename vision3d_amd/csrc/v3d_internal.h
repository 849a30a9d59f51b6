// v3d_internal.h -- C++ entry points shared inside libvision3d_hip (not part of the C ABI).
#pragma once
#include "v3d_common.h"

struct V3dRbHash {
  v3d_key_t* keys;
  int* vals;
  unsigned hcap;  // power of two
};

// `clear` = 0 on any of these: the caller has pre-filled the tables with 0xFF itself (one arena-wide memset).
int v3d_i_voxelize(const float* points, int n_points, int C, const int32_t* frame_offsets_host, int B,
                   const float* voxel_size_host, const float* bounds_host, int max_pts, int max_voxels, float* voxels,
                   int32_t* coords, int32_t* occupancy, float* mean, int32_t* n_voxels, void* workspace,
                   size_t workspace_bytes, int clear_tables, const V3dRbHash* site_hash /*nullable: also insert every
                   emitted voxel into this (pre-cleared) coordinate hash*/,
                   const int32_t* site_shape /*(D, H, W) of the grid that hash is keyed on*/, hipStream_t st);
int v3d_i_hash_build(const int32_t* coords, const int32_t* n, int cap, const int32_t* shape, V3dRbHash h, int clear,
                     hipStream_t st);
// The candidate pass of a strided rulebook handed to an EARLIER launch (rulebook.hip RbCandJob): the next strided layer's
// inputs (= the sites the carrying launch's stage owns), geometry, output hash and scratch.
struct V3dRbCandNext {
  const int32_t* coords_in;
  const int32_t* n_in;
  int cap_in;
  const int32_t* shape;  // input grid of that layer
  const int32_t *ksize, *stride, *padding;
  V3dRbHash out;
  unsigned* first_ticket;
  int* cand_slot;
  int32_t* overflow;
  int32_t* overflow_any;
};
int v3d_i_subm_nbr(const int32_t* coords, const int32_t* n, int cap, const int32_t* shape, const int32_t* ksize,
                   V3dRbHash h, int32_t* nbr, hipStream_t st, const V3dRbCandNext* next = nullptr);
int v3d_i_sparse_rulebook(const int32_t* coords_in, const int32_t* n_in, int cap_in, const int32_t* shape,
                          const int32_t* ksize, const int32_t* stride, const int32_t* padding, int32_t* coords_out,
                          int32_t* n_out, int cap_out, int32_t* nbr, int32_t* overflow, V3dRbHash out,
                          unsigned* first_ticket, int* cand_slot, int* chunk_counts /*reads -1 at launch unless clear*/,
                          int32_t* out_shape, int clear,
                          const int32_t* next_subm_ksize /*nullable: also build the submanifold table of the OUTPUT sites*/,
                          int32_t* next_subm_nbr, hipStream_t st,
                          int32_t* overflow_any = nullptr /*nullable: a second flag raised together with *overflow (the plan's summary)*/,
                          int candidates_done = 0 /*the candidate pass already ran as `next` of an earlier launch*/,
                          const V3dRbCandNext* next = nullptr /*carry the NEXT strided layer's candidate pass in the last launch*/,
                          int init_columns = 0 /*the emit pass writes -1 into the table columns of the sites it creates: the table then
                          needs no -1 fill (clear = 0 callers that do not pre-fill it)*/);

// iou_nms.hip: mask + greedy reduction on boxes already sorted by (score desc, index asc) and prepped (BoxPrep rows)
int v3d_i_nms_sorted(const void* prep_sorted, const int* order, int N, float iou_threshold, int64_t* keep, int32_t* n_keep,
                     unsigned long long* mask, unsigned long long* remv, hipStream_t st);

int v3d_i_nms_mask_sorted(const void* prep_sorted, int N, float iou_threshold, unsigned long long* mask, hipStream_t st);

// .dense() riding in the epilogue of the LAST sparse layer (the 16-row kernel): besides its rows the layer writes them, split into
// bf16 hi / lo, into the plan's persistent BEV planes out[(b * H + y) * W + x][c * D + z], clears the pixel's bit in the inverted
// occupancy bitmap and lists the pixel -- what densify_split_kernel (dense_conv.hip) does in a launch of its own.
struct V3dDensifyOut {
  const int32_t* coords;  // (cap, 4) = (b, z, y, x) of the layer's OUTPUT rows
  int D, H, W;
  void *hi, *lo;          // hi == nullptr: off
  uint32_t* occ;          // nullable
  int32_t *pix, *pix_n;   // written pixel list of the persistent planes
};
// spconv.hip: v3d_sparse_conv_fwd_packed with an explicit estimate of the live row count (kernel choice only)
int v3d_i_sparse_conv_fwd_packed(const float* in, const void* weight_image, const int32_t* nbr, const int32_t* n_out,
                                 int cap_out, int K, int Cin, int Cout, const float* scale, const float* shift, int relu,
                                 float* out, int rows_hint, hipStream_t st, const V3dDensifyOut* densify = nullptr /*the 16-row
                                 kernel is used whatever the row count; V3D_EUNSUPPORTED if the shape has no packed kernel*/,
                                 int ring_tiles_min = 2 /*64 -> 64 ring kernel: at least this many 16-row tiles per workgroup
                                 (throughput mode of a plan: fewer, fatter workgroups = less CU-time per launch)*/);

// spconv.hip: several packed weight images in one launch (mode 0: W (K, Cin, Cout); 1 / 2: the transposed layer of a source
// (K, Cout, Cin), 2 with the offsets reversed)
#define V3D_PACK_JOBS_MAX 32
struct V3dPackJobs {
  const float* w[V3D_PACK_JOBS_MAX];
  void* img[V3D_PACK_JOBS_MAX];
  int K[V3D_PACK_JOBS_MAX], cin[V3D_PACK_JOBS_MAX], cout[V3D_PACK_JOBS_MAX], mode[V3D_PACK_JOBS_MAX];
};
int v3d_i_sparse_conv_pack_batch(const V3dPackJobs& jobs, int n, hipStream_t stream);

// sparse_bn.hip: the C-ABI BatchNorm entry points with the row count optionally in device memory (n_dev != nullptr:
// min(*n_dev, n) rows, n = the buffers' capacity).  Same chunking, bit-identical results either way.
int v3d_i_sparse_bn_relu_fwd(const float* x, int n, const int32_t* n_dev, int C, const float* gamma, const float* beta, float eps,
                             int relu, float* y, float* save_mean, float* save_invstd, float* var_unbiased, float* running_mean,
                             float* running_var, float momentum, int64_t* num_batches_tracked, void* workspace,
                             size_t workspace_bytes, hipStream_t st);
int v3d_i_sparse_bn_relu_bwd(const float* x, const float* dy, int n, const int32_t* n_dev, int C, const float* gamma,
                             const float* beta, const float* save_mean, const float* save_invstd, int relu, float* dx,
                             float* dgamma, float* dbeta, void* workspace, size_t workspace_bytes, hipStream_t st);

// dense_conv.hip: v3d_densify_nhwc_split that also clears the bits of the occupied pixels in an inverted BEV occupancy bitmap
// (pre-filled with 0xFF by the caller): the input of the background-skipping dense head.
int v3d_i_densify_nhwc_split(const float* feat, const int32_t* coords, const int32_t* n, int cap, int B, int C,
                             const int32_t* spatial_shape_host, void* out_hi, void* out_lo, uint32_t* occ_inv, hipStream_t st,
                             int32_t* written_pix = nullptr /*with written_n: PERSISTENT planes -- no fill, the written pixels are listed*/,
                             int32_t* written_n = nullptr);
// zero the listed pixels (channels bf16 values each) of both planes: start-of-frame job of persistent BEV planes
int v3d_i_bev_clear_pixels(const int32_t* pix, const int32_t* n, int cap, int channels, void* hi, void* lo, hipStream_t st);
