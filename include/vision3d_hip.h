/*
 * vision3d_hip.h -- C ABI of libvision3d_hip.so, the MI355X (gfx950) implementation of the
 * vision3d point-cloud hot path.
 *
 * Conventions:
 *   - plain pointers + sizes, no torch types.  Every pointer is a DEVICE pointer unless its name ends
 *     in `_host`.  The caller owns every buffer, including workspaces (query the *_workspace() size
 *     first); the library never allocates or frees across this boundary (the v3d_second_plan object
 *     below owns a private arena for its own lifetime).
 *   - `stream` is a hipStream_t passed as void*; every call only ENQUEUES work on it -- no call
 *     synchronises the device, data-dependent sizes are returned through device-side counters.
 *   - return value: 0 = ok; < 0 = invalid argument (V3D_E*); > 0 = hipError_t from the runtime.
 *   - stateless and re-entrant: no global mutable state; concurrent calls on different streams are
 *     safe (reference ops are called from 6 DataLoader worker processes, train.py:18).
 *   - row-major contiguous inputs of exactly the stated dtype (reference: raw data_ptr use without
 *     .contiguous(), box_iou_rotated_cuda.cu:81-82).
 *
 * Each entry point names the reference interface it replaces (paths relative to the reference root).
 */
#ifndef VISION3D_HIP_H
#define VISION3D_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define V3D_OK 0
#define V3D_EINVAL (-1)     /* bad size / null pointer / unsupported shape */
#define V3D_EWORKSPACE (-2) /* workspace too small */
#define V3D_EUNSUPPORTED (-3)

typedef void* v3d_stream_t;

/* ---- version probes: vision3d/ops/csrc/vision.cpp:14-58 (get_cuda_version/get_compiler_version) */
const char* v3d_version(void);          /* "vision3d_hip x.y (gfx950)" */
int v3d_hip_runtime_version(void);      /* hipRuntimeGetVersion, replaces cuda_version.cu:6-8 */
const char* v3d_compiler_version(void); /* "clang M.m.p" */
const char* v3d_error_string(int code);

/* ---- A2: pairwise rotated BEV IoU.
 * Replaces detectron2::box_iou_rotated (ops/csrc/box_iou_rotated/box_iou_rotated.h:20-32,
 * box_iou_rotated_cuda.cu:65-121).  boxes (.,5) f32 = (xc, yc, w, h, angle_DEGREES); ious (M,N) f32.
 * Arithmetic follows the HOST branch of box_iou_rotated_utils.h (the CPU path is the parity target). */
int v3d_box_iou_rotated(const float* boxes1, int M, const float* boxes2, int N, float* ious, v3d_stream_t stream);

/* 3-D IoU of (x, y, z, w, l, h, yaw) boxes, z = centre: (M,7) x (N,7) -> (M,N).  BEV intersection area through the operator of
 * v3d_box_iou_rotated on columns (0,1,3,4,6) -- the yaw is read the way that operator reads it -- times the overlap of the z
 * extents, over the union of the volumes.  Replaces the stub vision3d/ops/iou_nms.py:12-13 (`box_iou_rotated_3d` raises
 * NotImplementedError upstream): defined by oracle/v3d_oracle.c:orc_box_iou_rotated_3d. */
int v3d_box_iou_rotated_3d(const float* boxes1, int M, const float* boxes2, int N, float* ious, v3d_stream_t stream);

/* ---- A3: greedy rotated NMS.
 * Replaces detectron2::nms_rotated (ops/csrc/nms_rotated/nms_rotated.h:22-36; CPU semantics
 * nms_rotated_cpu.cpp:7-59: suppress when IoU >= thr).  keep (N) i64 receives indices into the
 * ORIGINAL arrays in decreasing-score order (ties: lower index first); *n_keep (device i32) the count.
 * The reference's CUDA build suppresses on IoU > thr (nms_rotated_cuda.cu:62-63): pass nextafterf(thr, INFINITY) for that rule
 * (the two differ only at IoU == thr exactly; vision3d_amd.ops.nms_rotated(..., rule="cuda") does so).
 * The score sort, the IoU bitmask and the sequential reduction all run on the device. */
size_t v3d_nms_rotated_workspace(int N);
int v3d_nms_rotated(const float* boxes, const float* scores, int N, float iou_threshold, int64_t* keep,
                    int32_t* n_keep, void* workspace, size_t workspace_bytes, v3d_stream_t stream);

/* ---- Proposal stage of the BEV head, fused on the device.
 * Replaces ProposalLayer.inference after the 1x1 heads (detector/proposal.py:39-80: sigmoid, top-k per (frame, class),
 * core/box_encode.py:13-21 decode, ops/iou_nms.py:90-134 coordinate-offset batched rotated NMS, per-class score cut).
 * head_maps (B, n_cls*n_yaw*8, H, W) f32 = [class logits (c, yaw) | box deltas (c, dof, yaw)] channels (the
 * conv_cls | conv_reg outputs concatenated); anchors (n_cls, n_yaw, H, W, 7) f32; score_thresh_host (n_cls) on the
 * HOST.  Outputs are padded to N = B*n_cls*topk rows, sorted by decreasing score; *n_out (device i32) = rows valid.
 * Candidate order inside a group: (score descending, anchor index ascending).  No host synchronisation. */
size_t v3d_proposals_workspace(int B, int n_cls, int topk);
int v3d_proposals(const float* head_maps, const float* anchors, int B, int n_cls, int n_yaw, int H, int W, int topk,
                  const float* score_thresh_host, float iou_threshold, float* out_boxes, int64_t* out_batch_idx,
                  int64_t* out_class_idx, float* out_scores, int32_t* n_out, void* workspace, size_t workspace_bytes,
                  v3d_stream_t stream);
/* Same, and the last kernel also copies one device word into n_out[1] (n_out then has TWO words): the backbone plan's
 * capacity-overflow summary (v3d_backbone_overflow_flags()[n_layers]), so that the caller's single host read of the frame --
 * the proposal count -- brings the "this frame dropped rows" verdict with it. */
int v3d_proposals_flag(const float* head_maps, const float* anchors, int B, int n_cls, int n_yaw, int H, int W, int topk,
                       const float* score_thresh_host, float iou_threshold, float* out_boxes, int64_t* out_batch_idx,
                       int64_t* out_class_idx, float* out_scores, int32_t* n_out /*[2]*/, const int32_t* aux_flag,
                       void* workspace, size_t workspace_bytes, v3d_stream_t stream);
/* The first half alone: the decoded top-k candidates before NMS, (B, n_cls, topk) group-major, score descending inside a group
 * (ties: anchor index ascending): boxes (N, 7), scores (N), N = B * n_cls * topk.  What PV-RCNN's stage 2 refines
 * (vision3d_amd/detector/model.py stage1_proposals; the reference's ProposalLayer top-k + decode, detector/proposal.py:61-77).
 * Workspace: v3d_proposals_workspace. */
int v3d_proposals_topk(const float* head_maps, const float* anchors, int B, int n_cls, int n_yaw, int H, int W, int topk,
                       float* boxes, float* scores, void* workspace, size_t workspace_bytes, v3d_stream_t stream);
/* Stage-2 tail on candidates in that layout: refined = core/box_encode.py:13-21 decode of `deltas` against `proposals` (written to
 * `refined` (N, 7) when not NULL), score = sigmoid(conf), coordinate-offset batched rotated NMS per (frame, class) group
 * (ops/iou_nms.py:90-134), per-class score cut; outputs as v3d_proposals (padded to N rows, decreasing score, *n_out valid).
 * The reference's refinement.py:32-33 raises: the definition is the repository's. */
size_t v3d_refine_nms_workspace(int B, int n_cls, int topk);
int v3d_refine_nms(const float* deltas, const float* proposals, const float* conf, int B, int n_cls, int topk,
                   const float* score_thresh_host, float iou_threshold, float* refined, float* out_boxes, int64_t* out_batch_idx,
                   int64_t* out_class_idx, float* out_scores, int32_t* n_out, void* workspace, size_t workspace_bytes,
                   v3d_stream_t stream);

/* ---- Training-mode BatchNorm1d (+ ReLU) over sparse features (n, C), C a power of two in [4, 256].
 * Replaces nn.BatchNorm1d(eps, momentum) + nn.ReLU on SparseConvTensor.features (detector/sparse_cnn.py:15-30) in
 * training: forward = chunk statistics (two-pass per chunk) + Chan merge in double + fused normalise/affine/ReLU;
 * backward = chunk sums + merge + fused input gradient.  save_mean / save_invstd (C) feed the backward; var_unbiased (C)
 * is what the running_var update uses.  Deterministic. */
size_t v3d_sparse_bn_workspace(int n, int C);
int v3d_sparse_bn_relu_fwd(const float* x, int n, int C, const float* gamma, const float* beta, float eps, int relu, float* y,
                           float* save_mean, float* save_invstd, float* var_unbiased, float* running_mean /*nullable pair:*/,
                           float* running_var /*updated in place with `momentum`*/, float momentum,
                           int64_t* num_batches_tracked /*nullable: += 1*/, void* workspace, size_t workspace_bytes,
                           v3d_stream_t stream);
int v3d_sparse_bn_relu_bwd(const float* x, const float* dy, int n, int C, const float* gamma, const float* beta,
                           const float* save_mean, const float* save_invstd, int relu, float* dx, float* dgamma, float* dbeta,
                           void* workspace, size_t workspace_bytes, v3d_stream_t stream);

/* ---- A13: anchor <-> ground-truth target assignment, fused around the rotated-IoU core.
 * Replaces ProposalTargetAssigner.forward (core/proposal_targets.py:10-88) + Matcher (ops/matcher.py:55-130) +
 * box_encode.encode (core/box_encode.py:26-36) without materialising the (n_gt x anchors) IoU matrix.
 * gt_boxes (n_gt,7) f32, gt_class (n_gt) i64, anchors (n_cls, A, 7) f32; iou_thresh_host (n_cls, 2) = [lo, hi] on
 * the HOST: best IoU < lo -> label 0, [lo, hi) -> ignored, >= hi -> +1; allow_low_quality: every anchor attaining
 * some ground truth's best IoU (ties included) is positive.  Outputs (n_cls, A): G_cls i8 in {0,1}, M_cls u8
 * (0 = ignored), G_reg (.,7) f32 VoxelNet encoding at positives / 0, M_reg u8 = positives; matches (nullable) i64 =
 * index of the best ground truth (first maximal).  box_ignore is not an input: the reference never applies it. */
size_t v3d_assign_targets_workspace(int n_gt, int n_cls, int anchors_per_class);
int v3d_assign_targets(const float* gt_boxes, const int64_t* gt_class, int n_gt, const float* anchors, int n_cls,
                       int anchors_per_class, const float* iou_thresh_host, int allow_low_quality, int8_t* G_cls, uint8_t* M_cls,
                       float* G_reg, uint8_t* M_reg, int64_t* matches, void* workspace, size_t workspace_bytes,
                       v3d_stream_t stream);

/* ---- f2: GT-sampling + global augmentation of one training frame, fused (two launches with sampling, one without).
 * Replaces the chain vision3d/dataset/augmentation.py:31-48 -- SampleAugmentation :117-198 (paste at float64 positions :162-166,
 * collision filter IoU > 1e-2 :140-149, scene points under the pasted rectangles removed :195), FlipAugmentation :78-95,
 * ScaleAugmentation :98-114, RotateAugmentation :51-75 -- which the reference runs in numpy.  The random draws are the caller's
 * (made on the host with the reference's numpy calls in the reference's order): k samples {int32 box_row, pt_start, pt_len, cls;
 * float64 px, py} (32 bytes each, device memory) indexing the flat database tensors db_points (P, 4) / db_boxes (K, 7), flip,
 * the scale factor, cos / sin / value of the float32 rotation angle.  Arithmetic: float64 with sampling (numpy's promotion at the
 * paste), float32 without (k == 0: out_class_idx and work are not touched; any C >= 3).  With sampling C must be 4.
 * Outputs at capacity: out_points (N + sample_points, 4), out_boxes (n + k, 7), out_class_idx (n + k); the ragged sizes are the
 * first int32 words of `work`: {kept samples, kept sample points, kept scene points}, followed by keep[k] (0 / 1 per sample).
 * Result rows: kept scene points in order, then the kept samples' points in draw order; scene boxes, then kept samples' boxes. */
size_t v3d_augment_work_bytes(int N, int n, int k);
int v3d_augment_frame(const float* points, int N, int C, const float* boxes, const int64_t* class_idx, int n,
                      const float* db_points, const float* db_boxes, const void* samples, int k, int sample_points, int flip,
                      double factor, double cos_theta, double sin_theta, double theta, float* out_points, float* out_boxes,
                      int64_t* out_class_idx, void* work, size_t work_bytes, v3d_stream_t stream);

/* ---- A11: points in cuboids / rectangles.
 * Replaces core/geometry.py:27-65 (PointsInCuboids._get_mask when use_z != 0,
 * PointsNotInRectangles._get_mask otherwise).  points (N,C>=3) f32, boxes (n,7) f32
 * (x,y,z,w,l,h,yaw_rad); mask (N,n) u8. */
int v3d_points_in_boxes(const float* points, int N, int C, const float* boxes, int n, int use_z, uint8_t* mask,
                        v3d_stream_t stream);

/* ---- T1 (+A5, A6): voxelizer fused with the VoxelFeatureExtractor mean.
 * Replaces spconv.utils.VoxelGenerator.generate as driven by core/preprocess.py:17-33 (per-frame
 * voxelisation, batch index prefixed, frames concatenated) and detector/layers.py:10-17.
 * points (n_points,C) f32 = the frames concatenated; frame_offsets_host (B+1) i32 on the HOST.
 * voxel_size_host[3] (x,y,z), bounds_host[6] (x0,y0,z0,x1,y1,z1).
 * Outputs sized for B*max_voxels rows: voxels (.,max_pts,C) f32 zero padded [may be NULL],
 * coords (.,4) i32 = (b,z,y,x), occupancy (.) i32, mean (.,C) f32 [may be NULL]; *n_voxels (device i32).
 * Order: frame-major, first-touch order of the input points within a frame; the first max_pts points
 * of a voxel (in input order) are kept -- identical to the sequential reference loop. */
size_t v3d_voxelize_workspace(int n_points);
int v3d_voxelize(const float* points, int n_points, int C, const int32_t* frame_offsets_host, int B,
                 const float* voxel_size_host, const float* bounds_host, int max_pts, int max_voxels, float* voxels,
                 int32_t* coords, int32_t* occupancy, float* mean, int32_t* n_voxels, void* workspace,
                 size_t workspace_bytes, v3d_stream_t stream);

/* ---- T3: sparse-convolution rulebooks (spconv ops.get_indice_pairs, reached through
 * spconv.SubMConv3d / SparseConv3d at detector/sparse_cnn.py:15-30,153-175).
 * coords (cap,4) i32 (b,z,y,x); *n (device i32) active rows.  The rulebook is an output-stationary
 * neighbour table nbr (K, cap_out) i32, k-major: nbr[k*cap_out + o] = input row feeding output row o
 * through kernel offset k (k = (kz*ky_n + ky)*kx_n + kx), or -1.
 *   subm:   output sites == input sites (cap_out == cap).
 *   sparse: out sites = { (i + pad - k)/stride integral, in range }, numbered in first-touch order of
 *           the ticket sequence t = i*K + k; coords_out (cap_out,4), *n_out (device i32, clipped to
 *           cap_out; bit 0 of *overflow is set when clipped). */
size_t v3d_rulebook_workspace(int cap_in, int cap_out, int K);
int v3d_rulebook_subm(const int32_t* coords, const int32_t* n, int cap, const int32_t* spatial_shape_host,
                      const int32_t* ksize_host, int32_t* nbr, void* workspace, size_t workspace_bytes,
                      v3d_stream_t stream);
int v3d_rulebook_sparse(const int32_t* coords_in, const int32_t* n_in, int cap_in,
                        const int32_t* spatial_shape_host, const int32_t* ksize_host, const int32_t* stride_host,
                        const int32_t* padding_host, int32_t* coords_out, int32_t* n_out, int cap_out, int32_t* nbr,
                        int32_t* overflow, void* workspace, size_t workspace_bytes, v3d_stream_t stream);

/* ---- T3: sparse convolution forward (spconv ops.indice_conv) with optional fused per-channel affine
 * (BatchNorm1d in eval mode folded to scale/shift, sparse_cnn.py:18,27) and ReLU.
 * in (>=n_in,Cin) f32, weight (K,Cin,Cout) f32 [= spconv's (k0,k1,k2,Cin,Cout)], out (cap_out,Cout).
 * out[o,:] = act( (sum_k in[nbr[k,o],:] @ weight[k]) * scale + shift ).  Deterministic (no atomics).
 * algo: 0 = auto, 1 = scalar reference kernel, 3 = wave-autonomous exact-fp32 MFMA kernel (needs Cin%4==0, Cout%16==0 and a
 * compiled (Cin,Cout) instance). */
int v3d_sparse_conv_fwd(const float* in, const float* weight, const int32_t* nbr, const int32_t* n_out, int cap_out,
                        int K, int Cin, int Cout, const float* scale, const float* shift, int relu, float* out,
                        int algo, v3d_stream_t stream);

/* Split-precision variant (algo 4): operands are split into 16-bit hi + lo pieces (weights once per layer into a packed image,
 * activations in registers) and multiplied as lo*hi + hi*lo + hi*hi on the 16-bit matrix pipe with fp32 accumulation; one wave owns
 * 16 output rows with register accumulators.  Cout % 16 == 0.  Two arithmetics (`prec`):
 *   V3D_PREC_BF16X3 (0)  bf16 pieces, 2^-17 per product, scale-free (any fp32 magnitude).  What the un-suffixed entry points run.
 *   V3D_PREC_F16S   (1)  f16 pieces of x * s under a power-of-two scale s per tensor, 2^-22 per product: the result differs from
 *                        the reference's fp32 modules (detector/sparse_cnn.py:15-30 run spconv's fp32 GEMMs) by fp32's own
 *                        summation noise, at the same three MFMAs.  The weight scale is chosen by the pack call (device-side,
 *                        from max|W|); the activation scale is the caller's: `act_in` / `act_next` point at device entries
 *                        {s, 1/s, limit, 0} of the gathered tensor and (nullable) of the output -- an output magnitude beyond
 *                        `limit` = 2^15 / s raises *range_flag to 2 (atomicMax; nullable): the caller recalibrates and re-runs.
 *                        v3d_act_scale_from_rows fills an entry from the rows themselves (exact maximum: no flag needed). */
#define V3D_PREC_BF16X3 0
#define V3D_PREC_F16S 1
size_t v3d_sparse_conv_weight_image_bytes(int K, int Cin, int Cout);
/* image of `prec` (V3D_PREC_*) for the packed kernels: weight (K, Cin, Cout) fp32 -> split 16-bit fragments in MFMA order */
int v3d_sparse_conv_pack_weights(const float* weight, int K, int Cin, int Cout, int prec, void* image, v3d_stream_t stream);
/* entry[0..3] = {s, 1/s, 2^15 / s, max} for the n = min(*n_rows, cap) rows of `rows` (cap, C) [n_rows NULL: all cap rows]:
 * s = the power of two that puts the largest magnitude into [2^(13 - headroom_bits), 2^(14 - headroom_bits)).  No host
 * synchronisation.  headroom_bits in [0, 12].  scratch NULL: one workgroup.  scratch != NULL: a grid of workgroups -- two uint32
 * words in device memory that are ZERO when the launch starts (running maximum, arrival ticket); the last workgroup to arrive
 * writes the entry and zeroes them again, so one zeroed scratch serves every later call on the same stream. */
int v3d_act_scale_from_rows(const float* rows, const int32_t* n_rows, int cap, int C, int headroom_bits, float* entry,
                            uint32_t* scratch, v3d_stream_t stream);
/* rows_hint > 0: the caller's estimate of the LIVE row count (*n_out is device-side), used only to choose the kernel: 3x3x3 with
 * Cin, Cout in {32, 64}: LDS-ring kernel up to 16 384 rows, 64-row LDS-shared-weights kernel from 32 768, else the 16-row kernel;
 * 0 = unknown (ring / 16-row).  The 64 -> 64 ring kernel owns a CU per workgroup: it takes 2, 3 or 4 sixteen-row tiles per workgroup,
 * the smallest count that keeps rows_hint + 10 % inside one round of 256 workgroups (8 192 / 12 288 / 16 384 rows).
 * rows_hint < 0 FORCES a kernel (tests, benchmarks): -1 = 16-row, -5 = 64-row, -6 / -7 = offset-outer (staged / register gathers),
 * -10 = LDS ring, -16 = its register-gather form, -12 / -13 / -14 = the 64 -> 64 ring with 2 / 3 / 4 tiles per workgroup.
 * There is no process-global switch.
 * prec = the arithmetic the image was packed for; V3D_PREC_F16S needs act_in (and act_next where split rows are written); act_in /
 * act_next / range_flag are ignored (may be NULL) for V3D_PREC_BF16X3.
 * in_split / out_split (both nullable): rows ALREADY split into the arithmetic's 16-bit pieces -- a row = [hi: C x 16 bit |
 * lo: C x 16 bit], the bytes of the fp32 row; f16s: pieces of x * s of the tensor's scale entry.  in_split (then `in` may be NULL):
 * the layer gathers these and its main loop converts nothing; out_split (Cout % 8 == 0; f16s needs act_next; `out` may then be NULL):
 * the output rows once more in that form, for the next layer's in_split.  A chain of layers this way computes the same bits as
 * on fp32 rows (v3d_sparse_rows_split makes split rows from fp32 rows). */
int v3d_sparse_conv_fwd_packed(const float* in, const void* weight_image, const int32_t* nbr, const int32_t* n_out,
                               int cap_out, int K, int Cin, int Cout, const float* scale, const float* shift, int relu,
                               float* out, int rows_hint, int prec, const float* act_in, const float* act_next,
                               int32_t* range_flag, const void* in_split, void* out_split, v3d_stream_t stream);
int v3d_sparse_rows_split(const float* rows, const int32_t* n_rows, int cap, int C, int prec, const float* act_entry,
                          void* out_split, v3d_stream_t stream);
/* ---- T3 over SPATIALLY ORDERED rows (csrc/brick.hip; same interface the reference reaches through spconv.SubMConv3d,
 * detector/sparse_cnn.py:15-30).  When the rows of a stage are numbered in a spatial (brick / Morton) order, the 27 x 256
 * neighbours of 256 consecutive rows are ~1.2-1.7 x 256 distinct rows: v3d_sparse_brick_plan derives, once per submanifold
 * table `nbr` (K = 27), per 256-row pass the list of those rows (`ulist`, 480 per pass; `ucnt` their number), every table entry
 * as a slot of its pass's list (`lidx`, 0xFFFF = no neighbour) and per 16-row tile the mask of offsets any of its rows has
 * (`tmask`); v3d_sparse_conv_fwd_brick (Cin -> Cout = 64 -> 64, 32 -> 32; split rows in, rows and / or split rows out) keeps a pass's
 * rows in LDS and walks the offsets without any global gather.  Correct for ANY row order (a pass with more than 480 distinct
 * neighbours gathers directly); same bits as v3d_sparse_conv_fwd_packed2 with rows_hint = -6.
 * v3d_sparse_brick_table_bytes: byte sizes of the four tables for a capacity (each a multiple of 256), returns their sum. */
size_t v3d_sparse_brick_table_bytes(int cap, int K, size_t* lidx_bytes, size_t* ulist_bytes, size_t* ucnt_bytes,
                                    size_t* tmask_bytes);
int v3d_sparse_brick_plan(const int32_t* nbr, const int32_t* n_rows, int cap, int K, uint16_t* lidx, int32_t* ulist,
                          int32_t* ucnt, uint32_t* tmask, v3d_stream_t stream);
int v3d_sparse_conv_fwd_brick(const void* in_split, const void* weight_image, const int32_t* nbr, const uint16_t* lidx,
                              const int32_t* ulist, const int32_t* ucnt, const uint32_t* tmask, const int32_t* n_out,
                              int cap, int K, int Cin, int Cout, const float* scale, const float* shift, int relu, float* out,
                              int prec, const float* act_in, const float* act_next, int32_t* range_flag, void* out_split,
                              v3d_stream_t stream);

/* ---- T3 backward (spconv indice_conv backward; the reference trains through it at train.py:65).
 * Data gradient: dX[i] = sum_k dY[nbrT[k][i]] @ W[k]^T -- the forward entry points above on the TRANSPOSED
 * rulebook (v3d_rulebook_transpose; a submanifold table is its own transpose with the offsets reversed) and the
 * transposed weights.  Weight gradient: dW[k] = sum_{pairs} X[i]^T dY[o], exact-fp32 MFMA, deterministic. */
int v3d_rulebook_transpose(const int32_t* nbr, const int32_t* n_out, int cap_out, int K, int cap_in, int32_t* nbr_t,
                           v3d_stream_t stream);
size_t v3d_sparse_conv_bwd_weight_workspace(int K, int Cin, int Cout);
int v3d_sparse_conv_bwd_weight(const float* X, const float* dY, const int32_t* nbr, const int32_t* n_out, int cap_out,
                               int K, int Cin, int Cout, float* dW, void* workspace, size_t workspace_bytes,
                               v3d_stream_t stream);

/* ---- T2: SparseConvTensor.dense() (detector/sparse_cnn.py:128-133): zero-fill + scatter.
 * feat (cap,C), coords (cap,4), *n rows -> dense (B,C,D,H,W) f32. */
int v3d_densify(const float* feat, const int32_t* coords, const int32_t* n, int cap, int B, int C,
                const int32_t* spatial_shape_host, float* dense, v3d_stream_t stream);

/* ---- T4/T5: pointnet2 ops (pointnet2_utils.furthest_point_sample / gather_operation / ball_query /
 * grouping_operation; call sites detector/model.py:46-66, detector/roi_grid_pool.py:64-72). */
size_t v3d_fps_workspace(int B, int N);
int v3d_furthest_point_sample(const float* xyz, int B, int N, int K, int32_t* idx, void* workspace,
                              size_t workspace_bytes, v3d_stream_t stream);
int v3d_gather_points(const float* feat, const int32_t* idx, int B, int C, int N, int K, float* out,
                      v3d_stream_t stream);
/* idx_b == NULL: one radius (radius_b / nsample_b ignored).  idx_b != NULL: two radii around the same queries in ONE scan of the
 * database (what PointnetSAModuleMSG asks for per feature source): per (query, radius) the same result as the one-radius call. */
int v3d_ball_query(const float* xyz, const float* new_xyz, int B, int N, int M, float radius_a, int nsample_a, int32_t* idx_a,
                   float radius_b, int nsample_b, int32_t* idx_b, v3d_stream_t stream);
/* The same operation through a cell grid of the database: the points are binned into (x, y) cells no smaller than the larger radius
 * (one launch), every query looks at the 3 x 3 cells around its own and reads "the first nsample hits in index order" off a
 * per-wave LDS bitmap (second launch).  Same arguments, same results as v3d_ball_query for every input (the hit test is the scan
 * kernel's expression on the same operands); ~6x faster at PV-RCNN's sizes whatever the order of the cloud.  `workspace`:
 * v3d_ball_query_grid_workspace(B, N) bytes, 16-byte aligned, scratch. */
size_t v3d_ball_query_grid_workspace(int B, int N);
int v3d_ball_query_grid(const float* xyz, const float* new_xyz, int B, int N, int M, float radius_a, int nsample_a, int32_t* idx_a,
                        float radius_b, int nsample_b, int32_t* idx_b, void* workspace, size_t workspace_bytes, v3d_stream_t stream);
/* The two halves on their own.  build: the grids of up to 8 databases in ONE launch (a workgroup per database and frame; host arrays
 * of n_db entries: xyz[i] (B, N[i], 3), radius_max[i] = the largest radius that will be asked of grid i, workspace[i] of
 * v3d_ball_query_grid_workspace(B, N[i]) bytes) -- the six databases of a PV-RCNN frame (raw points, four voxel levels, keypoints:
 * detector/model.py:58-66, roi_grid_pool.py:64-72) cost one launch instead of six.  query: any radii <= radius_max of the build,
 * as often as wanted; same results as v3d_ball_query.  V3D_EUNSUPPORTED: N too large for the query's LDS bitmaps (use the scan). */
int v3d_ball_query_grid_build(int n_db, const float* const* xyz, const int32_t* N, const float* radius_max, void* const* workspace,
                              const size_t* workspace_bytes, int B, v3d_stream_t stream);
int v3d_ball_query_grid_query(const float* new_xyz, int B, int N, int M, float radius_a, int nsample_a, int32_t* idx_a,
                              float radius_b, int nsample_b, int32_t* idx_b, const void* workspace, size_t workspace_bytes,
                              v3d_stream_t stream);
/* query_many: up to 8 queries around the SAME new_xyz (B, M, 3), each in its own database / grid with its own radii, in one launch
 * (host arrays of n_jobs entries; idx_b[i] NULL: one radius for job i) -- the five set-abstraction modules of a PV-RCNN frame all
 * ask around the frame's keypoints (detector/model.py:58-66). */
int v3d_ball_query_grid_query_many(int n_jobs, const float* new_xyz, int B, int M, const int32_t* N, const float* radius_a,
                                   const int32_t* nsample_a, int32_t* const* idx_a, const float* radius_b, const int32_t* nsample_b,
                                   int32_t* const* idx_b, const void* const* workspace, const size_t* workspace_bytes,
                                   v3d_stream_t stream);
int v3d_group_points(const float* feat, const int32_t* idx, int B, int C, int N, int M, int nsample, float* out,
                     v3d_stream_t stream);
/* Bilinear lookup of BEV features at keypoints: F.grid_sample(feature_map, grid, bilinear, zeros, align_corners=True) for a
 * (B, 1, K, 2) grid, as BEVFeatureGatherer.forward calls it (detector/layers.py:29-47).  feature_map (B, C, H, W) f32, grid
 * (B, K, 2) f32 = (x, y) in [-1, 1], out (B, C, K).  Same taps, weights and summation order as torch's kernel. */
int v3d_bev_bilinear(const float* feature_map, const float* grid, int B, int C, int H, int W, int K, float* out,
                     v3d_stream_t stream);
/* BEVFeatureGatherer.forward in one launch (detector/layers.py:29-47): grid coordinates from the keypoints by the module's own fp32
 * statements (offset / pixel = its pixel_offset and base_pixel_size * STRIDES[-1], x and y), clamp, normalise, flip, then the
 * bilinear lookup above.  keypoint_xyz (B, K, 3); out is POINT-major: out[(b * K + k) * ldo + c], ldo >= C.  Same values as
 * v3d_bev_bilinear on the module's torch-computed grid, bit for bit. */
int v3d_bev_gather_keypoints(const float* feature_map, const float* keypoint_xyz, int B, int C, int H, int W, int K, float offset_x,
                             float offset_y, float pixel_x, float pixel_y, float* out, int ldo, v3d_stream_t stream);

/* SparseCNNBase.to_global in one launch (detector/sparse_cnn.py:91-105): out (n, 3) = (x, y, z) = float(indices[:, (3, 2, 1)]) *
 * scale + offset, one conversion, one multiply and one add per coordinate; indices (n, 4) i32 = (b, z, y, x), 16-byte aligned;
 * scale = base_voxel_size * stride as the caller rounded it in fp32. */
int v3d_voxel_centers(const int32_t* indices, int n, float scale_x, float scale_y, float scale_z, float offset_x, float offset_y,
                      float offset_z, float* out, v3d_stream_t stream);
/* RoiGridPool.sample_gridpoints in one launch (detector/roi_grid_pool.py:52-62): points[b, n, j] = centre + Rz(yaw) (size * (sample - 0.5))
 * with the module's own fp32 statements, one IEEE operation each; boxes (B*n, 7) = x, y, z, w, l, h, yaw, samples (B*n, m, 3) in
 * [0, 1), cos_yaw / sin_yaw (B*n) from the caller (torch's cos / sin: the same values as the op-by-op path) or both NULL (cosf / sinf
 * of the device library inside the launch: equal to torch's on this stack, which the GPU tests check), out (B*n, m, 3). */
int v3d_roi_grid_points(const float* boxes, const float* samples, const float* cos_yaw, const float* sin_yaw, int n_boxes, int m,
                        float* out, v3d_stream_t stream);

/* ---- T5: one layer of a set-abstraction shared MLP on gathered rows, exact fp32 on the matrix cores (csrc/sa_mlp.hip).
 * Replaces, per scale of pointnet2_modules.PointnetSAModuleMSG (call sites detector/model.py:58-66, detector/roi_grid_pool.py:
 * 64-72), grouping_operation + SharedMLP (Conv2d 1x1, no bias + BatchNorm2d + ReLU per layer) + max over the samples, without the
 * grouped (B, C+3, M, ns) tensor.  Rows are (b, m, s) in that order, rows = B*M*ns.
 *   first layer  (idx != NULL): feat (B, N, Kf) POINT-major, xyz (B, N, 3), new_xyz (B, M, 3), idx (B, M, ns) i32 into N;
 *                row = [xyz[i] - new_xyz[m], 0 | feat[i, :]], W (4 + Kf, Nout) row-major (row 3 multiplies the zero);
 *   later layers (idx == NULL, xyz == new_xyz == NULL): feat (rows, Kf) = the previous layer's output, N must equal M*ns, W (Kf, Nout).
 * out[row, :] = act(row @ W + bias); pool != 0: out (B*M, Nout) = max over the ns rows of each (b, m) (ns = 16 or 32).
 * Kf % 4 == 0, Nout in {16, 32, 64, 96, 128, 192, 256}; BatchNorm(eval) is folded into W / bias by the caller.
 * ldo = row stride of `out` in floats (0: Nout), n_store = columns stored (0: Nout): a scale's pooled rows land directly in their
 * column block of the (B*M, C_total) keypoint feature matrix -- no torch.cat of the scales / sources (model.py:72-74). */
int v3d_sa_mlp_layer(const float* feat, const float* xyz, const float* new_xyz, const int32_t* idx, int B, int N, int M, int ns,
                     int Kf, const float* W, const float* bias, int Nout, int relu, int pool, float* out, int ldo, int n_store,
                     v3d_stream_t stream);
/* The first TWO layers of a scale in one launch.  The first layer is linear in [xyz[i] - new_xyz[m], 0 | feat[i]] and its feature
 * part depends on the gathered point alone: the caller computes P (B, N, K1) = feat @ W1[4:] once per DATABASE point
 * (v3d_linear_rows; N rows instead of M * ns), and this kernel rebuilds relu(P[i] + rel . W1[0:3] + b1) as the operand rows of the
 * second layer W (K1, Nout) -- 37x less matrix work at RoI-grid pooling (2 048 keypoints, 76 800 grouped rows, roi_grid_pool.py:64-72),
 * and the (rows, K1) intermediate is never written.  wx (3, K1) = W1[0:3], b1 (K1); K1 % 4 == 0, K1 <= 256; ldp = row stride of P in
 * floats (0: K1; the two scales of a module share one product with concatenated weights); the rest as above. */
int v3d_sa_mlp_pair(const float* P, const float* xyz, const float* new_xyz, const int32_t* idx, int B, int N, int M, int ns, int K1,
                    int ldp, const float* wx, const float* b1, const float* W, const float* bias, int Nout, int relu, int pool,
                    float* out, int ldo, int n_store, v3d_stream_t stream);
/* v3d_sa_mlp_pair for the two scales of a module in ONE launch (same K1 and Nout; per scale: its column block of P, neighbour list,
 * sample count, wx / b1, W / bias and the column block of `out` it writes): fills the chip where two launches left it half empty. */
int v3d_sa_mlp_pair2(const float* P_a, const float* P_b, const float* xyz, const float* new_xyz, const int32_t* idx_a, const int32_t* idx_b,
                     int B, int N, int M, int ns_a, int ns_b, int K1, int ldp, const float* wx_a, const float* b1_a, const float* wx_b,
                     const float* b1_b, const float* W_a, const float* bias_a, const float* W_b, const float* bias_b, int Nout, int relu,
                     int pool, float* out_a, float* out_b, int ldo, int n_store, v3d_stream_t stream);
/* The MLP tail of PV-RCNN on a hundred rows: out[r, n] = act(sum_k A[r * lda + k] * W[k * Nout + n] + bias[n]) for r < R,
 * n < n_store (0: Nout), out row stride ldo (0: Nout).  Replaces nn.Linear (+ bias, + ReLU) of detector/layers.py:53-73 as used by
 * the RoI-grid reduction (roi_grid_pool.py:64-72: 3 072 -> 256 -> 256) and the refinement head (refinement.py:47-50: 256 -> 128 -> 8).
 * W is the Linear's weight TRANSPOSED (K, Nout), Nout padded to a multiple of 16 by the caller, K % 4 == 0, A 16-byte aligned.
 * Exact fp32 products on the matrix cores, fixed summation order (K split over 8 waves, partial tiles added in wave order). */
int v3d_linear_rows(const float* A, int lda, int R, int K, const float* W, const float* bias, int Nout, int relu, float* out, int ldo,
                    int n_store, v3d_stream_t stream);
/* up to 8 such products in one launch (host arrays of n_jobs entries; bias / relu / ldo / n_store arrays may be NULL = none / 0). */
int v3d_linear_rows_many(int n_jobs, const float* const* A, const int32_t* lda, const int32_t* R, const int32_t* K, const float* const* W,
                         const float* const* bias, const int32_t* Nout, const int32_t* relu, float* const* out, const int32_t* ldo,
                         const int32_t* n_store, v3d_stream_t stream);

/* ---- Fused sparse-backbone plan: voxelizer -> [rulebooks + sparse conv layers] -> .dense().
 * The native form of the sparse half of Second.feature_extract (detector/second.py:20-24,41-46 over
 * detector/sparse_cnn.py:151-175): created once per model, every forward only ENQUEUES kernels on
 * `stream` -- no host synchronisation, counts stay in device memory, one coordinate hash per stage is
 * shared by the strided rulebook that creates the stage and the submanifold rulebook that follows.
 * The plan owns its arena (hipMalloc at create, hipFree at destroy) and private copies of the layer
 * parameters (set_layer copies device -> device).  Not thread-safe per plan; use one plan per stream. */
typedef struct v3d_backbone v3d_backbone;
typedef struct {
  int32_t subm;                            /* 1 = SubMConv3d, 0 = SparseConv3d */
  int32_t cin, cout;
  int32_t ksize[3], stride[3], padding[3]; /* z, y, x */
  int32_t key;                             /* >= 0: submanifold layers with equal key share a rulebook (indice_key) */
  int32_t relu;                            /* fused ReLU */
} v3d_layer_desc;
typedef struct {
  float voxel_size[3];                     /* x, y, z */
  float bounds[6];                         /* x0,y0,z0,x1,y1,z1 */
  int32_t max_pts, max_voxels;             /* per voxel / per frame */
  int32_t point_channels;                  /* C of the input points == Cin of layer 0 */
  int32_t grid_shape[3];                   /* D,H,W of the CNN input grid (sparse_cnn.py:40-45: z + 1) */
  int32_t max_batch, max_points;           /* capacities: frames per forward, total points per forward */
  int32_t n_layers;
  float growth;                            /* active-site capacity of later stages = growth * voxel capacity (<=0: 2.0) */
  int32_t conv_algo;                       /* 0 = default; 3 = fp32-MFMA wave kernel; 4 = bf16x3 row-owner kernel */
} v3d_backbone_config;
int v3d_backbone_create(const v3d_backbone_config* cfg, const v3d_layer_desc* layers, v3d_backbone** out);
void v3d_backbone_destroy(v3d_backbone* plan);
size_t v3d_backbone_arena_bytes(const v3d_backbone* plan);
int v3d_backbone_num_layers(const v3d_backbone* plan);
/* weight (K,Cin,Cout) f32; scale/shift (Cout) or both NULL (no affine). */
int v3d_backbone_set_layer(v3d_backbone* plan, int layer, const float* weight, const float* scale, const float* shift,
                           v3d_stream_t stream);
/* points (n_points,C) f32 = frames concatenated.  Outputs, any may be NULL: dense_nchw (B, Cout_last*D, H, W) f32 and / or the
 * split planes dense_hi / dense_lo ((B, H, W, Cout_last*D) 16-bit pieces of the plan's arithmetic: the dense head's input). */
int v3d_backbone_forward(v3d_backbone* plan, const float* points, int n_points, const int32_t* frame_offsets_host,
                         int B, float* dense_nchw, void* dense_hi, void* dense_lo, v3d_stream_t stream);
/* Device-resident results of the last forward: layer = -1 -> voxelizer output (mean features, coords);
 * layer >= 0 -> that layer's output rows.  *n_rows_dev is a device int32; cap = row capacity. */
int v3d_backbone_layer_output(v3d_backbone* plan, int layer, float** features, int32_t** coords,
                              int32_t** n_rows_dev, int* cap, int* channels, int32_t* shape_host);
/* Kernel-choice tuning from observed sparsity: reads the live row counts of the last forward (blocking, a few bytes)
 * and uses them as size hints for the following forwards (capacities are upper bounds).  Not capturable. */
int v3d_backbone_tune(v3d_backbone* plan);
int32_t* v3d_backbone_occupancy(v3d_backbone* plan);       /* (cap0) i32, voxel occupancies of the last forward */
int32_t* v3d_backbone_overflow_flags(v3d_backbone* plan);  /* (n_layers+1) i32 device flags of the last forward: [l] > 0 = layer l
                                                              hit its active-site capacity (rows were dropped), [n_layers] > 0 = any */

/* ---- A8/A9: dense 2-D convolutions of the BEV head on the matrix cores (csrc/dense_conv.hip).
 * Replaces nn.Conv2d + BatchNorm2d(eval) + ReLU (detector/second.py:58-94) and the 1x1 heads
 * (detector/proposal.py:19-22).  "bf16 x 3" split precision: x = hi + lo (two bf16), products evaluated as
 * hi*hi + hi*lo + lo*hi with fp32 accumulation => fp32-class accuracy.  Activations are exchanged as two
 * bf16 NHWC planes (B,H,W,C) "hi" and "lo"; weights are packed once with v3d_conv2d_pack_weights.
 * Cin % 32 == 0, ksize in {1,3} (stride 1, "same" zero padding).  Outputs: split NHWC planes (Cout % 8 == 0)
 * and/or fp32 NCHW (B,Cout,H,W). */
size_t v3d_conv2d_weight_image_bytes(int Cin, int Cout, int ksize);
/* (v3d_conv2d_pack_weights: weight (Cout,Cin,k,k) f32; optional per-output-channel scale (folded BatchNorm) multiplied in;
 *  v3d_conv2d_nhwc_split: the convolution -- both declared below, with the arithmetic's scale entries) */
/* Background skipping for the BEV head.  The BEV map of a sparse scene is mostly empty: an output pixel of layer L whose
 * receptive field through layers 1..L (`reach` pixels: +1 per 3x3 layer) contains no occupied BEV pixel sees exactly the inputs it
 * sees in an EMPTY map, so its value is the empty map's response at that position -- bit for bit, borders included.  The caller
 * computes that response once per weight set (the same convolutions on an all-zero map, occ = NULL) and passes it as bg_hi / bg_lo
 * ((H, W, Cout) split planes of ONE image); tiles whose pixels are all further than `reach` (Chebyshev) from every occupied pixel
 * become a copy of that response instead of a convolution.  Results are identical
 * to the call with occ = NULL.  Applies to the large-tile kernel with split-plane output;
 * any other configuration computes every pixel.
 * occ: BEV occupancy, one bit per pixel, INVERTED (bit cleared = occupied; a 0xFF fill = empty map), rows (b, y) of ceil(W / 32)
 * words.  A plan keeps one for its last forward (v3d_backbone_bev_occupancy: cleared by the per-frame 0xFF fill, set by the
 * densify kernel -- no extra launch); v3d_bev_occupancy_bits builds one from a site list. */
size_t v3d_bev_occupancy_words(int B, int H, int W);
int v3d_bev_occupancy_bits(const int32_t* coords /*(cap,4) b,z,y,x*/, const int32_t* n, int cap, int B, int H, int W,
                           uint32_t* occ, v3d_stream_t stream);
uint32_t* v3d_backbone_bev_occupancy(v3d_backbone* plan);
/* The plan's PERSISTENT split BEV planes ((max_batch, H, W, C_out * D) bf16 each, hi then lo, adjacent).  Passing exactly these two
 * pointers as dense_hi / dense_lo to v3d_backbone_forward / _forward_voxels / _forward_reuse makes the plan keep them zero outside
 * the occupied pixels itself: each frame clears the few thousand pixels the previous one wrote (in the launch of its per-frame fill)
 * instead of filling 2 x 9 MB per KITTI frame -- what .dense() (detector/sparse_cnn.py:128-133: zeros + scatter) costs.  Valid until the
 * next forward into them; nobody else may write them. */
int v3d_backbone_bev_planes(v3d_backbone* plan, void** hi, void** lo);
/* Kernel choice of the following forwards: on != 0 = THROUGHPUT mode, for plans whose frames run beside other frames on the same GPU
 * (one plan per frame in flight): the LDS-filling 64 -> 64 sparse kernel takes four 16-row tiles per workgroup whatever the row count --
 * fewer, fatter workgroups: ~39 % less CU-time per launch, ~20 % longer launches (results are bit-identical either way) --, and a
 * packed layer that is followed by a packed layer writes its output rows ONLY in the split form the next layer gathers
 * (v3d_backbone_set_presplit), not as fp32 rows: v3d_backbone_layer_output's feature views of such layers are then stale.
 * Default off: the shortest launch (one frame at a time), every layer's fp32 rows written. */
int v3d_backbone_set_throughput_mode(v3d_backbone* plan, int on);
/* Arithmetic of the plan's INFERENCE entry points (v3d_backbone_forward* ; the training plan is bf16x3): V3D_PREC_BF16X3 (default of
 * a new plan) or V3D_PREC_F16S.  Set it BEFORE v3d_backbone_set_layer: the weight images are packed per arithmetic.
 * f16s: v3d_backbone_act_scales() = (n_layers + 1) x {s, 1/s, limit, max} in device memory, entry l = the rows layer l gathers,
 * entry n_layers = the BEV map (the scale of the split planes).  New plans hold {1, 1, 2^15, 0}: correct for magnitudes below 2^15,
 * full precision once calibrated: v3d_backbone_set_calibrating(plan, 1) -> one forward (every layer on the exact-fp32 kernel) ->
 * v3d_backbone_calibrate(plan, headroom_bits, stream) (entries from that frame's maxima; enqueued, no host sync) ->
 * v3d_backbone_set_calibrating(plan, 0).  A later frame whose tensors exceed an entry's limit (2^(headroom_bits + 1) x the
 * calibration frame's maximum) raises v3d_backbone_overflow_flags()[n_layers] to 2: recalibrate on it and run it again. */
int v3d_backbone_set_precision(v3d_backbone* plan, int prec);
int v3d_backbone_precision(const v3d_backbone* plan);
/* on != 0 (default): a packed layer also writes its output rows split into the arithmetic's 16-bit pieces and the next packed layer
 * gathers those (no conversion work in its main loop); 0: every layer splits the fp32 rows it gathers (A/B).  Same results. */
int v3d_backbone_set_presplit(v3d_backbone* plan, int on);
float* v3d_backbone_act_scales(v3d_backbone* plan);
int v3d_backbone_set_calibrating(v3d_backbone* plan, int on);
int v3d_backbone_calibrate(v3d_backbone* plan, int headroom_bits, v3d_stream_t stream);
/* v3d_conv2d_nhwc_split arguments of the background-skipping path (all nullable / zero = every tile convolved):
 *   occ, reach, bg_hi, bg_lo  inverted BEV occupancy bitmap (v3d_backbone_bev_occupancy), the layer's reach in pixels and its
 *                             response to the empty map (planes): tiles with no occupied pixel within reach ARE that response;
 *   work        two zeroed words owned by the caller for THIS call site (one pair per layer and stream in flight).  With it the
 *               skipping kernel runs as a persistent grid that draws tiles from work[0]; the pair resets itself when the kernel ends;
 *   tile_state  with `work`, when y_hi / y_lo are PERSISTENT buffers of this call site (the same pair every frame):
 *               v3d_conv2d_bg_tiles words, nonzero = the tile holds computed values (initialise to nonzero).  A layer's empty-map
 *               response does not depend on the frame, so a background tile whose word is 0 already holds it and is not written
 *               at all; the kernel keeps the words up to date.  Reset them to nonzero when the weights (bg_hi / bg_lo) change;
 *   reset_ptr, reset_words    the counter pair reset by ANOTHER launch: reset_ptr != NULL makes this launch zero reset_words words
 *               at reset_ptr when it starts (the pairs of call sites whose launches lie behind it in stream order) and leave its
 *               own `work` pair as it ends -- some later launch of the caller's zeroes it (saves the atomic round trip every
 *               workgroup of a self-resetting launch ends on); `work` must read zero at launch;
 *   pr          NULL = bf16x3 images and planes; else the arithmetic and its scale entries (v3d_conv2d_prec below). */
int v3d_conv2d_bg_tiles(int B, int H, int W);
/* ---- the same dense head in fp32-class arithmetic (V3D_PREC_F16S; see the sparse section above for the two arithmetics).
 * The reference's RPN / heads are fp32 nn.Conv2d modules (detector/second.py:58-79, detector/proposal.py:19-22): f16s reproduces
 * them up to fp32 summation noise at the cost of bf16x3.  Planes then hold f16 pieces of x * s: every plane pair has a device entry
 * {s, 1/s, limit = 2^15 / s, ..} (set by the caller's calibration: vision3d_amd.runtime.DenseHeadPlan.calibrate; the BEV planes of
 * a plan use v3d_backbone_act_scales()[n_layers]), the weight image carries its own scale (v3d_conv2d_pack_weights), and an output
 * magnitude beyond its entry's limit raises *range_flag to 2 (atomicMax, nullable).  fp32 NCHW outputs are unscaled.
 * For v3d_conv2d_1x1_head_fused `out_entry` is the entry of the intermediate 128-channel tensor. */
typedef struct v3d_conv2d_prec {
  int32_t prec;           /* V3D_PREC_*: the arithmetic the weight image(s) were packed for */
  const float* in_entry;  /* f16s: device entry of the input planes */
  const float* out_entry; /* f16s: device entry of the output planes (NULL when only fp32 NCHW is written) */
  int32_t* range_flag;    /* f16s, nullable */
  const float* w_inv;     /* f16s, nullable: device copy of the weight image's 1 / s_w (float 1 of its 256-byte trailer) in memory the
                             caller keeps hot, e.g. beside its scale entries; NULL: read from the trailer (a cold line per launch) */
  const float* w_inv2;    /* same for the second image of v3d_conv2d_1x1_head_fused */
} v3d_conv2d_prec;
int v3d_conv2d_pack_weights(const float* weight, const float* scale, int Cout, int Cin, int ksize, int prec, void* image,
                            v3d_stream_t stream);
int v3d_conv2d_nhwc_split(const void* x_hi, const void* x_lo, const void* weight_image, const float* bias, int relu, int B,
                          int H, int W, int Cin, int Cout, int ksize, void* y_hi, void* y_lo, float* y_nchw,
                          const uint32_t* occ, int reach, const void* bg_hi, const void* bg_lo, uint32_t* work,
                          uint32_t* tile_state, uint32_t* reset_ptr, int reset_words, const v3d_conv2d_prec* pr,
                          v3d_stream_t stream);
/* The tail of the SECOND dense head in ONE launch: RPN up-conv 1x1 128 -> 128 (+ folded BatchNorm bias + ReLU, detector/second.py:73-79)
 * followed by the fused [cls | reg] 1x1 head 128 -> Cout2 <= 16 (detector/proposal.py:19-22), split planes (B, H, W, 128) in, fp32
 * NCHW (B, Cout2, H, W) out; w1_image / w2_image from v3d_conv2d_pack_weights(.., ksize 1).  Bit-identical to v3d_conv2d_nhwc_split
 * applied twice; the intermediate planes never exist.  V3D_EUNSUPPORTED for other widths.  pr NULL = bf16x3. */
int v3d_conv2d_1x1_head_fused(const void* x_hi, const void* x_lo, const void* w1_image, const float* b1, int relu1,
                              const void* w2_image, const float* b2, int relu2, int B, int H, int W, int Cmid, int Cout2,
                              float* y_nchw, const v3d_conv2d_prec* pr, v3d_stream_t stream);
/* fp32 (B, C, H, W) -> split planes (B, H, W, C) of `prec` (f16s: pieces of x * act_entry[0]; act_entry NULL for bf16x3) */
int v3d_nchw_to_split_nhwc(const float* x, int B, int C, int H, int W, void* out_hi, void* out_lo, int prec,
                           const float* act_entry, v3d_stream_t stream);
/* .dense() of the last sparse stage straight into that input format: planes (B,H,W,C*D), channel = c*D + z. */
int v3d_densify_nhwc_split(const float* feat, const int32_t* coords, const int32_t* n, int cap, int B, int C,
                           const int32_t* spatial_shape_host, void* out_hi, void* out_lo, v3d_stream_t stream);
/* ... and back: bf16 split planes (B,H,W,C) -> fp32 (B,C,H,W) = hi + lo (C even).  The gradient of the BEV map leaving
 * v3d_dense_train_backward_split for the sparse training plan, which takes fp32 NCHW. */
int v3d_split_nhwc_to_nchw(const void* x_hi, const void* x_lo, int B, int C, int H, int W, float* out, v3d_stream_t stream);
/* The convolutions (+ .dense()) of the frame the plan forwarded LAST, on the site lists and neighbour tables that call left
 * behind: no voxelizer, no rulebook build -- the "rulebooks prebuilt" timing variant.  Same outputs. */
int v3d_backbone_forward_reuse(v3d_backbone* plan, int B, float* dense_nchw, void* dense_hi, void* dense_lo, v3d_stream_t stream);

/* The same plan fed with EXISTING voxels -- voxel_mean (M, C) f32 and coords (M, 4) i32 (b, z, y, x), M <= the plan's voxel
 * capacity, rows frame-sorted as core/preprocess.py:26-33 produces them -- instead of raw points: what Second.forward(item) /
 * Second.inference(item) (detector/second.py:26-35, called by train.py:63 and inference.py:38) hold.  Two device copies replace
 * the voxelizer; everything after is v3d_backbone_forward. */
int v3d_backbone_forward_voxels(v3d_backbone* plan, const float* voxel_mean, const int32_t* coords, int n_voxels, int B,
                                float* dense_nchw, void* dense_hi, void* dense_lo, v3d_stream_t stream);

/* ---- ProposalLoss (detector/proposal.py:100-141) and its gradient with respect to the FUSED head maps (B, n_cls * n_yaw * 8, H, W:
 * class channel cls * n_yaw + yaw, box channel n_cls * n_yaw + (cls * 7 + d) * n_yaw + yaw -- ProposalLayer.reshape_cls / reshape_reg)
 * in one pass: sigmoid focal loss (ops/focal_loss.py; alpha < 0 disables the class weight) over M_cls + smooth-L1 over M_reg (yaw
 * term / pi, counted three times as upstream broadcasts it), both divided by max(#M_reg, 1).  losses[3] = cls_loss, reg_loss,
 * normalizer; dmaps = d(cls_loss)/d(maps) in the class channels, d(reg_loss)/d(maps) in the box channels; _scale multiplies the two
 * channel groups with the upstream gradients (device scalars).  fp32, bit-repeatable. */
size_t v3d_proposal_loss_workspace(void);
int v3d_proposal_loss_fwd_bwd(const float* maps, const int8_t* g_cls, const uint8_t* m_cls, const float* g_reg, const uint8_t* m_reg,
                              int B, int n_cls, int n_yaw, int H, int W, float alpha, float gamma, float* losses, float* dmaps,
                              void* workspace, size_t workspace_bytes, v3d_stream_t stream);
int v3d_proposal_loss_scale(float* dmaps, int B, int n_cls, int n_yaw, int H, int W, const float* g_cls, const float* g_reg,
                            v3d_stream_t stream);

/* ---- Training plan: the sparse half of a train step (train.py:63-67 through detector/second.py:41-46 and
 * detector/sparse_cnn.py:15-30,151-175) as ONE call forwards and ONE call backwards, no host synchronisation.
 * Every layer must be conv + BatchNorm1d (training mode: batch statistics) [+ ReLU] with a power-of-two Cout in [4, 256].
 * Parameters are read from the caller's device pointers on every call (they change every optimiser step); the running
 * statistics are updated in place exactly as nn.BatchNorm1d would (momentum, unbiased variance, num_batches_tracked += 1).
 * forward:  voxel_mean (n_voxels, C) f32, coords (n_voxels, 4) i32 [b, z, y, x] (the Preprocessor's item) -> dense_out
 *           (B, Cout, D, H, W) f32 or dense_nhwc_bf16; activations needed by the backward stay in the plan's training arena (allocated by the
 *           first call: v3d_backbone_train_arena_bytes).
 * backward: grad_dense (B, Cout, D, H, W) f32 -> grad_weight (K, Cin, Cout), grad_gamma (Cout), grad_beta (Cout) of every
 *           layer.  Must follow a train_forward of the same plan; the layer array is the same (host) array of device pointers.
 * Deterministic: no atomics in any reduction. */
typedef struct {
  const float* weight;          /* (K, Cin, Cout) f32 */
  const float* gamma;           /* BatchNorm weight (Cout) */
  const float* beta;            /* BatchNorm bias (Cout) */
  float* running_mean;          /* nullable pair */
  float* running_var;
  int64_t* num_batches_tracked; /* nullable */
  float eps, momentum;
  float* grad_weight;           /* backward outputs (ignored by the forward) */
  float* grad_gamma;
  float* grad_beta;
} v3d_train_layer;
int v3d_backbone_train_forward(v3d_backbone* plan, const float* voxel_mean, const int32_t* coords, int n_voxels, int B,
                               const v3d_train_layer* layers_host, float* dense_out /*exactly one of the two outputs*/,
                               void* dense_nhwc_bf16 /*(B, H, W, Cout * D) bf16, channel = c * D + z: the BEV map as the bf16
                               autocast RPN consumes it (torch channels_last), rounded to nearest even*/,
                               v3d_stream_t stream);
int v3d_backbone_train_backward(v3d_backbone* plan, const float* grad_dense /*exactly one of the two gradients*/,
                                const void* grad_nhwc_bf16, int B, const v3d_train_layer* layers_host, v3d_stream_t stream);
size_t v3d_backbone_train_arena_bytes(const v3d_backbone* plan);
/* Coordinate-only pass + v3d_backbone_tune: builds the rulebooks for these voxels (no convolution), waits for `stream` and takes
 * the kernel-choice hints from the row counts.  Lets the FIRST training step run the kernels of all later steps (a training
 * forward cannot be repeated after tuning: it updates the running statistics).  Blocking; never during stream capture. */
int v3d_backbone_tune_from_voxels(v3d_backbone* plan, const int32_t* coords, int n_voxels, int B, v3d_stream_t stream);


/* ---- Dense TRAIN path of the BEV head (csrc/dense_train.hip): the RPN's Conv2d(128 -> 128, 3x3 pad 1 | 1x1, bias-free) +
 * BatchNorm2d (batch statistics) + ReLU layers and the fused 1x1 [cls | reg] head, forward and backward, bf16 storage / fp32
 * accumulation (the arithmetic of the reference's stack under torch autocast).  Replaces what train.py:58-66 runs through
 * cuDNN for detector/second.py:58-94 and detector/proposal.py:19-22.  All activations: bf16 NHWC (B, H, W, 128) = torch
 * channels_last bfloat16; every reduction is two-level in a fixed order (bit-repeatable).  128 channels are fixed. */
size_t v3d_dense_train_weight_image_bytes(int ksize);
/* weight (128, 128, k, k) fp32 -> bf16 fragment image; transpose = 0: forward, 1: data gradient (channels swapped, taps flipped) */
int v3d_dense_train_pack_weights(const float* weight, int ksize, int transpose, void* image, v3d_stream_t stream);
int v3d_dense_train_conv_tiles(int B, int H, int W); /* tiles of one convolution = rows of its `stats` output */
/* y = conv(x) (no bias, stride 1, pad k/2); stats (nullable): (tiles, 2, 128) fp32 per-tile channel sums / sums of squares of y */
int v3d_dense_train_conv(const void* x, const void* image, int B, int H, int W, int ksize, void* y, float* stats,
                         v3d_stream_t stream);
/* per-tile sums -> mean / invstd (biased variance, eps); running statistics (nullable pair; momentum, unbiased variance) and
 * num_batches_tracked (nullable) updated in place -- nn.BatchNorm2d's training-mode bookkeeping */
int v3d_dense_train_bn_finalize(const float* stats, int tiles, long long count, float eps, float momentum, float* mean,
                                float* invstd, float* running_mean, float* running_var, int64_t* num_batches_tracked,
                                v3d_stream_t stream);
/* y = relu?((x - mean) * invstd * gamma + beta), M = B*H*W pixels */
int v3d_dense_train_bn_relu_apply(const void* x, long long M, const float* mean, const float* invstd, const float* gamma,
                                  const float* beta, int relu, void* y, v3d_stream_t stream);
size_t v3d_dense_train_bn_bwd_workspace(void);
/* x = the layer's raw convolution output, dy = gradient w.r.t. the post-ReLU output -> dx (may alias dy), dgamma, dbeta */
int v3d_dense_train_bn_relu_bwd(const void* x, const void* dy, long long M, const float* mean, const float* invstd,
                                const float* gamma, const float* beta, int relu, void* dx, float* dgamma, float* dbeta,
                                void* workspace, size_t workspace_bytes, v3d_stream_t stream);
/* dW (128, 128, k, k) fp32 = sum over pixels of dy (x) x shifted by the tap, straight from the NHWC tensors (x: the layer's bf16
 * input, dy: the bf16 gradient of its raw output).  Bit-repeatable (fixed-order two-level reduction). */
size_t v3d_dense_train_wgrad_workspace(int ksize);
int v3d_dense_train_wgrad(const void* x, const void* dy, int B, int H, int W, int ksize, float* dw, void* workspace,
                          size_t workspace_bytes, v3d_stream_t stream);
/* fused 1x1 head, O in {8, 16, 24, 32, 48, 64} outputs with bias: maps fp32 (B, O, H, W) */
size_t v3d_dense_train_head_workspace(int O);
int v3d_dense_train_head_fwd(const void* feat, int B, int H, int W, const float* weight, const float* bias, int O, float* maps,
                             v3d_stream_t stream);
int v3d_dense_train_head_bwd(const void* feat, const float* dmaps, int B, int H, int W, const float* weight, int O, void* dfeat,
                             float* dweight, float* dbias, void* workspace, size_t workspace_bytes, v3d_stream_t stream);
/* The whole dense half of a train step, one call per direction (n_layers x [conv + batch-statistics BatchNorm + ReLU] + head).
 * arena: a device buffer of v3d_dense_train_arena_bytes owned by the caller, cleared once by v3d_dense_train_arena_init; it
 * carries the forward's activations and statistics to the backward of the SAME step.  bev / dbev: bf16 NHWC (B, H, W, 128). */
typedef struct {
  const float* weight;               /* (128, 128, k, k) */
  const float* gamma;                /* BatchNorm2d weight / bias */
  const float* beta;
  float* running_mean;               /* nullable pair; updated by the forward */
  float* running_var;
  int64_t* num_batches_tracked;      /* nullable */
  float eps, momentum;
  int32_t ksize;                     /* 1 or 3 */
  float* grad_weight;                /* outputs of the backward (unused by the forward) */
  float* grad_gamma;
  float* grad_beta;
} v3d_dense_train_layer;
size_t v3d_dense_train_arena_bytes(int B, int H, int W, int n_layers, int O);
int v3d_dense_train_arena_init(void* arena, int B, int H, int W, int n_layers, int O, v3d_stream_t stream);
int v3d_dense_train_forward(const void* bev, int B, int H, int W, const v3d_dense_train_layer* layers, int n_layers,
                            const float* head_weight, const float* head_bias, int O, void* arena, float* maps, v3d_stream_t stream);
int v3d_dense_train_backward(const void* bev, const float* dmaps, int B, int H, int W, const v3d_dense_train_layer* layers,
                             int n_layers, const float* head_weight, int O, void* arena, float* dhead_weight, float* dhead_bias,
                             void* dbev, v3d_stream_t stream);
/* The same step in fp32-class arithmetic ("bf16x3"): what the reference's fp32 train.py:58-66 (no autocast) needs from the dense half.
 * Every tensor is a split pair of bf16 NHWC planes (value = hi + lo: 16 significant bits), every product three MFMA terms with fp32
 * accumulation (2^-17 per product; scale-free, so gradients need no calibration): convolutions and data gradients on
 * v3d_conv2d_nhwc_split, the weight gradient as three passes of the bf16 kernel over (hi, hi), (hi, lo), (lo, hi), statistics /
 * normalisation / head gradients in fp32 on hi + lo.  Same layer descriptors, same reductions (fixed order, bit-repeatable); O <= 16.
 * bev_hi / bev_lo, dbev_hi / dbev_lo: (B, H, W, 128) bf16 planes; the backward reads the SAME bev planes and arena as the forward. */
size_t v3d_dense_train_arena_bytes_split(int B, int H, int W, int n_layers, int O);
int v3d_dense_train_forward_split(const void* bev_hi, const void* bev_lo, int B, int H, int W, const v3d_dense_train_layer* layers,
                                  int n_layers, const float* head_weight, const float* head_bias, int O, void* arena, float* maps,
                                  v3d_stream_t stream);
int v3d_dense_train_backward_split(const void* bev_hi, const void* bev_lo, const float* dmaps, int B, int H, int W,
                                   const v3d_dense_train_layer* layers, int n_layers, const float* head_weight, int O, void* arena,
                                   float* dhead_weight, float* dhead_bias, void* dbev_hi, void* dbev_lo, v3d_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* VISION3D_HIP_H */
