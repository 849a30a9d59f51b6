// v3d_internal.h -- C++ entry points shared inside libvision3d_hip (not part of the C ABI).
#pragma once
#include "v3d_common.h"
// Ablation builds only (bash tools/build_variant.sh abl "-DV3D_ABLATE" second_plan.hip dense_conv.hip proposal.hip): launches of a
// class named in $V3D_ABL are skipped -- c: sparse convolutions, r: the 64 -> 64 3x3x3 ones only, d: dense 3x3 tile launches,
// p: the proposal stage.  Results are garbage; the frame period with a class removed is its marginal cost (docs/rounds/round6.md).
#ifdef V3D_ABLATE
#include <cstdlib>
#include <cstring>
static inline bool v3d_ablate(char c) { const char* e = getenv("V3D_ABL"); return e && strchr(e, c); }
#else
static inline bool v3d_ablate(char) { return false; }
#endif

struct RbScanJob;  // rb_device.h
struct RbStep;
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

// The same two steps as descriptors (rb_device.h RbStep) instead of launches: a plan hands the scans to the sparse layers' launches,
// which run them as their first workgroups (the rulebook chain depends on coordinates only and runs ahead of the convolutions), and
// launches the others with v3d_i_rb_step_launch.
int v3d_i_sparse_rulebook_steps(const int32_t* coords_in, const int32_t* n_in, int cap_in, const int32_t* shape,
                                const int32_t* ksize, const int32_t* stride, const int32_t* padding, int32_t* coords_out,
                                int32_t* n_out, int cap_out, int32_t* nbr, int32_t* overflow, V3dRbHash out,
                                unsigned* first_ticket, int* cand_slot, int* chunk_counts, const int32_t* next_subm_ksize,
                                int32_t* next_subm_nbr, int32_t* overflow_any, const V3dRbCandNext* next, int init_columns,
                                RbStep* scan, RbStep* fill);
int v3d_i_rb_step_launch(const RbStep& s, hipStream_t st);

// iou_nms.hip: mask + greedy reduction on boxes already sorted by (score desc, index asc) and prepped (BoxPrep rows)
int v3d_i_nms_sorted(const void* prep_sorted, const int* order, int N, float iou_threshold, int64_t* keep, int32_t* n_keep,
                     unsigned long long* mask, unsigned long long* remv, hipStream_t st);

int v3d_i_nms_mask_sorted(const void* prep_sorted, int N, float iou_threshold, unsigned long long* mask, hipStream_t st);

// Arithmetic of the packed sparse kernels and of the dense head (spconv.hip, "the split-precision product"): bf16 pieces (2^-17 per
// product, scale-free) or f16 pieces under per-tensor power-of-two scales (2^-22: fp32-class at the same three MFMAs).
// (V3D_PREC_BF16X3 = 0 / V3D_PREC_F16S = 1: include/vision3d_hip.h)
// values of a frame's summary flag word (reset to -1 by the plan's per-frame 0xFF fill; raised with atomicMax)
#define V3D_FLAG_CAPACITY 1  // an active-site capacity was hit: rows were dropped
#define V3D_FLAG_RANGE 2     // f16s arithmetic: a tensor exceeded the range its scale was calibrated for
#define V3D_FLAG_QUIET 3     // f16s arithmetic: a tensor's largest magnitude of this frame lies 2^12 or more below the limit its scale was
                             // calibrated for (the small entries of such a tensor no longer keep 22 bits): recalibrate downward, run again
#define V3D_QUIET_BITS 12
// f16s: one entry per tensor in device memory = {s, 1/s, limit, 0}: the tensor is split as f16(x * s) by its consumer, and its
// producer raises V3D_FLAG_RANGE when |x| > limit (= 2^15 / s: a factor two inside f16's 65 504).  All null for bf16x3.
struct V3dActScale {
  const float* in;    // entry of the rows this launch gathers
  const float* next;  // nullable: entry of this launch's OUTPUT (checked in the epilogue; the scale of planes written there)
  int32_t* flag;      // nullable: where a range violation is recorded
  const float* w_inv; // nullable: 1 / s_w of the weight image in memory the caller keeps HOT (a plan's table: one line for all layers).
                      // NULL: read from the image's trailer -- a line nothing else touches, i.e. a cold miss of ~1 us at the top of
                      // every launch (measured: every packed layer +1 us against bf16x3 until the plan passed this)
  unsigned* seen;     // nullable (needs `next`): word of this launch's OUTPUT tensor, zero at the start of a frame, set to 1 by a wave whose
                      // outputs reach next[2] * 2^-V3D_QUIET_BITS (v3d_mark_seen) -- what the plan's quiet check reads
};

// .dense() riding in the epilogue of the LAST sparse layer (the 16-row kernel): besides its rows the layer writes them, split into
// 16-bit hi / lo pieces (of the launch's arithmetic; f16s: scaled by its `next` entry), into the plan's persistent BEV planes out[(b * H + y) * W + x][c * D + z], clears the pixel's bit in the inverted
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
                                 (throughput mode of a plan: fewer, fatter workgroups = less CU-time per launch)*/,
                                 int prec = V3D_PREC_BF16X3 /*the arithmetic the image was packed for*/,
                                 const V3dActScale* act = nullptr /*V3D_PREC_F16S: required*/,
                                 const void* in_split = nullptr /*the gathered rows ALREADY split into this arithmetic's pieces under
                                 act->in (row = [hi: Cin x 16 bit | lo: Cin x 16 bit]): what an earlier call wrote through out_split;
                                 the main loop then has no conversion work*/,
                                 void* out_split = nullptr /*besides `out`: the output rows split under act->next (f16s) for the next
                                 packed layer, (cap_out, 2 * Cout) 16-bit*/,
                                 const RbScanJob* rider = nullptr /*a rulebook scan the launch may carry as its first workgroups*/,
                                 bool* rider_taken = nullptr /*set when it did (the 16-row and the ring kernels)*/);

// brick.hip: tables of a submanifold rulebook over spatially ordered rows (a plan in brick order), per pass of 256 output rows:
// the distinct input rows it touches, the neighbour table translated into slots of that list, per-tile offset masks.
struct V3dBrickTables {
  uint16_t* lidx;   // (K, cap) slot of nbr[k][o] in its pass's list, 0xFFFF = no neighbour
  int32_t* ulist;   // (passes, 480) distinct input rows of a pass, ascending
  int32_t* ucnt;    // (passes) their number (beyond 480: the pass runs the direct-gather body)
  uint32_t* tmask;  // (ceil(cap / 16)) bit k = some row of the 16-row tile has a neighbour under offset k
};
int v3d_i_sparse_brick_plan(const int32_t* nbr, const int32_t* n_in, int cap_in, const int32_t* n_out, int cap_out, int K,
                            const V3dBrickTables& t, hipStream_t st);
int v3d_i_sparse_conv_fwd_brick(const void* in_split, const void* weight_image, const int32_t* nbr, const V3dBrickTables& bt,
                                const int32_t* n_out, int cap, int K, int Cin, int Cout, const float* scale, const float* shift, int relu,
                                float* out, int prec, const V3dActScale* act, void* out_split, hipStream_t st);

bool v3d_i_sparse_conv_packed_supported(int Cin, int Cout);

// spconv.hip: v3d_sparse_conv_fwd (exact fp32 kernels) whose output is additionally checked against the limit of the f16s scale
// entry of the tensor it produces (wave kernel only; both nullable)
int v3d_i_sparse_conv_fwd_exact(const float* in, const float* weight, const int32_t* nbr, const int32_t* n_out, int cap_out, int K,
                                int Cin, int Cout, const float* scale, const float* shift, int relu, float* out, int algo,
                                hipStream_t st, const float* next_entry, int32_t* range_flag, unsigned* seen = nullptr,
                                const RbScanJob* rider = nullptr, bool* rider_taken = nullptr /*as above (the wave kernel)*/);

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
                             int32_t* written_n = nullptr, int prec = V3D_PREC_BF16X3,
                             const float* act_entry = nullptr /*f16s: {s, 1/s, limit, ..} of the planes (device)*/,
                             int32_t* range_flag = nullptr /*f16s, nullable: raised to V3D_FLAG_RANGE by a value beyond the limit*/,
                             unsigned* seen = nullptr /*f16s, nullable: the planes' word of the downward range check (V3dActScale::seen)*/);
// zero the listed pixels (channels bf16 values each) of both planes: start-of-frame job of persistent BEV planes
int v3d_i_bev_clear_pixels(const int32_t* pix, const int32_t* n, int cap, int channels, void* hi, void* lo, hipStream_t st);
