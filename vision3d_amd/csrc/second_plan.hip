// second_plan.hip -- fused sparse-backbone plan: voxelizer -> [rulebooks + sparse conv layers] -> .dense().
//
// This is the native runtime for the sparse half of Second.feature_extract
// (vision3d/detector/second.py:20-24,41-46 + detector/sparse_cnn.py:151-175).  The reference crosses
// Python <-> C++ <-> CUDA ~1000 times per frame here (one gather/GEMM/scatter triple per kernel offset);
// the eager C-ABI path of this repository still pays ~70 us of interpreter time per op.  A plan is
// created once per model: it owns an arena sized from capacities, private copies of the folded layer
// parameters, and on every forward only ENQUEUES kernels on the caller's stream:
//   * no host synchronisation anywhere -- voxel / active-site counts live in device memory and every
//     kernel takes them by pointer with a capacity-sized grid;
//   * one coordinate hash per stage: the strided rulebook leaves the hash of its OUTPUT sites behind and
//     the next stage's submanifold rulebook reuses it (spconv rebuilds a dense index grid per call);
//   * submanifold layers that share an indice_key share one neighbour table.
#include <new>
#include <vector>

#include "v3d_internal.h"

struct PlanLayer {
  v3d_layer_desc d;
  int K;
  int stage_in, stage_out;  // coordinate sets
  int rulebook;             // index into nbr tables
  bool builds_rulebook;
  float *weight, *scale, *shift, *out;
  void* wimg;  // split + packed weights for the bf16x3 kernel
  int has_affine;
  int* chunk_counts;  // strided layers: published per-chunk counts -> offsets (inside the per-frame 0xFF region)
  int rows_hint;      // expected live output rows (kernel choice): capacity-free estimate, refined by v3d_backbone_tune
};

struct PlanStage {
  int cap;
  int shape[3];
  int32_t* coords;
  int32_t* n_dev;
  V3dRbHash hash;
  unsigned* first_ticket;
  bool hash_ready_by_sparse;  // hash filled by the strided rulebook that created this stage
};

struct v3d_backbone {
  v3d_backbone_config cfg;
  std::vector<PlanLayer> layers;
  std::vector<PlanStage> stages;
  std::vector<int32_t*> nbr;       // per rulebook
  std::vector<int> nbr_cap;        // stride of each table
  void* arena = nullptr;
  size_t arena_bytes = 0;
  // voxelizer
  void* vox_ws = nullptr;
  size_t vox_ws_bytes = 0;
  int32_t* occupancy = nullptr;
  float* mean = nullptr;
  // strided-rulebook scratch
  int* cand_slot = nullptr;
  int32_t* overflow = nullptr;  // one flag per layer (<= 0 fine, 1 = capacity hit)
  char* ff_begin = nullptr;     // arena region reset to 0xFF by one memset per forward
  size_t ff_bytes = 0;
  int out_channels = 0;
};

static int conv_fan(const v3d_layer_desc& d) {
  int fan = 1;
  for (int j = 0; j < 3; j++) fan *= (d.ksize[j] + d.stride[j] - 1) / d.stride[j];
  return fan;
}

extern "C" int v3d_backbone_create(const v3d_backbone_config* cfg, const v3d_layer_desc* descs, v3d_backbone** out) {
  if (!cfg || !descs || !out || cfg->n_layers < 1 || cfg->max_batch < 1 || cfg->max_points < 1) return V3D_EINVAL;
  v3d_backbone* p = new (std::nothrow) v3d_backbone();
  if (!p) return V3D_EINVAL;
  p->cfg = *cfg;
  const float growth = cfg->growth > 0.f ? cfg->growth : 2.0f;
  long long cap0 = (long long)cfg->max_batch * cfg->max_voxels;
  if (cap0 > cfg->max_points) cap0 = cfg->max_points;
  if (cap0 < 1) cap0 = 1;

  // ---- pass 1: geometry, capacities, rulebook sharing
  PlanStage s0{};
  s0.cap = (int)cap0;
  for (int j = 0; j < 3; j++) s0.shape[j] = cfg->grid_shape[j];
  p->stages.push_back(s0);
  std::vector<int> key_to_rb;  // (stage, key) -> rulebook
  std::vector<int> key_stage, key_id;
  int cur = 0, cin = cfg->point_channels;
  long long max_tickets = 1;
  for (int l = 0; l < cfg->n_layers; l++) {
    PlanLayer L{};
    L.d = descs[l];
    if (L.d.cin != cin) { delete p; return V3D_EINVAL; }
    L.K = L.d.ksize[0] * L.d.ksize[1] * L.d.ksize[2];
    if (L.K < 1 || L.K > 64) { delete p; return V3D_EUNSUPPORTED; }
    L.stage_in = cur;
    if (L.d.subm) {
      L.stage_out = cur;
      int found = -1;
      if (L.d.key >= 0)
        for (size_t i = 0; i < key_id.size(); i++)
          if (key_id[i] == L.d.key && key_stage[i] == cur) found = key_to_rb[i];
      if (found >= 0) {
        L.rulebook = found;
        L.builds_rulebook = false;
      } else {
        L.rulebook = (int)p->nbr_cap.size();
        L.builds_rulebook = true;
        p->nbr_cap.push_back(p->stages[cur].cap);
        if (L.d.key >= 0) { key_id.push_back(L.d.key); key_stage.push_back(cur); key_to_rb.push_back(L.rulebook); }
      }
    } else {
      PlanStage ns{};
      long long cells = cfg->max_batch;
      for (int j = 0; j < 3; j++) {
        ns.shape[j] = (p->stages[cur].shape[j] + 2 * L.d.padding[j] - L.d.ksize[j]) / L.d.stride[j] + 1;
        if (ns.shape[j] < 1) { delete p; return V3D_EINVAL; }
        cells *= ns.shape[j];
      }
      long long cap = (long long)p->stages[cur].cap * conv_fan(L.d);
      if (cap > cells) cap = cells;
      const long long lim = (long long)(cap0 * (double)growth);
      if (cap > lim) cap = lim;
      if (cap < 1) cap = 1;
      ns.cap = (int)cap;
      ns.hash_ready_by_sparse = true;
      const long long tickets = (long long)p->stages[cur].cap * L.K;
      if (tickets > max_tickets) max_tickets = tickets;
      p->stages.push_back(ns);
      cur = (int)p->stages.size() - 1;
      L.stage_out = cur;
      L.rulebook = (int)p->nbr_cap.size();
      L.builds_rulebook = true;
      p->nbr_cap.push_back(ns.cap);
    }
    L.rows_hint = 0;  // unknown until tuned: the 16-row kernel (right for KITTI-size frames)
    cin = L.d.cout;
    p->layers.push_back(L);
  }
  p->out_channels = cin;

  // ---- pass 2: size and carve the arena (two passes over the same carving code)
  auto carve = [&](V3dArena& ar) {
    // ---- everything that must read 0xFF at the start of a forward is contiguous: ONE memset per frame
    p->ff_begin = ar.base + ar.off;
    p->vox_ws_bytes = v3d_voxelize_workspace(cfg->max_points);
    p->vox_ws = ar.take<char>(p->vox_ws_bytes);
    for (auto& st : p->stages) {
      st.hash.hcap = v3d_hash_capacity(st.cap);
      st.hash.keys = ar.take<v3d_key_t>(st.hash.hcap);
      st.first_ticket = ar.take<unsigned>(st.hash.hcap);
      st.hash.vals = ar.take<int>(st.hash.hcap);
    }
    p->nbr.resize(p->nbr_cap.size());
    std::vector<int> rbK(p->nbr_cap.size(), 1), rbSparse(p->nbr_cap.size(), 0);
    for (auto& L : p->layers) {
      rbK[L.rulebook] = L.K;
      if (!L.d.subm) rbSparse[L.rulebook] = 1;
    }
    for (size_t i = 0; i < p->nbr.size(); i++)
      if (rbSparse[i]) p->nbr[i] = ar.take<int32_t>((size_t)rbK[i] * p->nbr_cap[i]);  // strided tables start as -1
    p->overflow = ar.take<int32_t>(p->layers.size() + 1);
    for (auto& L : p->layers)  // strided layers: per-layer count slots, -1 = "not published" at the start of a frame
      if (!L.d.subm) L.chunk_counts = ar.take<int>((size_t)((long long)p->stages[L.stage_in].cap * L.K / V3D_SCAN_CHUNK + 2));
    p->ff_bytes = (size_t)((ar.base + ar.off) - p->ff_begin);
    // ---- the rest needs no per-frame initialisation
    for (size_t i = 0; i < p->nbr.size(); i++)
      if (!rbSparse[i]) p->nbr[i] = ar.take<int32_t>((size_t)rbK[i] * p->nbr_cap[i]);
    p->occupancy = ar.take<int32_t>(p->stages[0].cap);
    p->mean = ar.take<float>((size_t)p->stages[0].cap * cfg->point_channels);
    for (auto& st : p->stages) {
      st.coords = ar.take<int32_t>((size_t)st.cap * 4);
      st.n_dev = ar.take<int32_t>(1);
    }
    p->cand_slot = ar.take<int>((size_t)max_tickets);
    for (auto& L : p->layers) {
      L.weight = ar.take<float>((size_t)L.K * L.d.cin * L.d.cout);
      L.wimg = ar.take<char>(v3d_sparse_conv_weight_image_bytes(L.K, L.d.cin, L.d.cout));
      L.scale = ar.take<float>(L.d.cout);
      L.shift = ar.take<float>(L.d.cout);
      L.out = ar.take<float>((size_t)p->stages[L.stage_out].cap * L.d.cout);
    }
  };
  {
    V3dArena probe((void*)256, (size_t)1 << 60);
    carve(probe);
    p->arena_bytes = probe.off + 4096;
  }
  hipError_t e = hipMalloc(&p->arena, p->arena_bytes);
  if (e != hipSuccess) { delete p; return (int)e; }
  V3dArena ar(p->arena, p->arena_bytes);
  carve(ar);
  if (!ar.ok()) { (void)hipFree(p->arena); delete p; return V3D_EWORKSPACE; }
  e = hipMemset(p->ff_begin, 0xFF, p->ff_bytes);
  if (e != hipSuccess) { (void)hipFree(p->arena); delete p; return (int)e; }
  *out = p;
  return V3D_OK;
}

extern "C" void v3d_backbone_destroy(v3d_backbone* p) {
  if (!p) return;
  if (p->arena) (void)hipFree(p->arena);
  delete p;
}

extern "C" size_t v3d_backbone_arena_bytes(const v3d_backbone* p) { return p ? p->arena_bytes : 0; }

extern "C" int v3d_backbone_set_layer(v3d_backbone* p, int layer, const float* weight, const float* scale,
                                      const float* shift, v3d_stream_t stream) {
  if (!p || layer < 0 || layer >= (int)p->layers.size() || !weight) return V3D_EINVAL;
  if ((scale == nullptr) != (shift == nullptr)) return V3D_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  PlanLayer& L = p->layers[layer];
  V3D_CHECK_HIP(hipMemcpyAsync(L.weight, weight, (size_t)L.K * L.d.cin * L.d.cout * 4, hipMemcpyDeviceToDevice, st));
  L.has_affine = scale != nullptr;
  if (L.d.cout % 16 == 0) {
    int rc = v3d_sparse_conv_pack_weights(L.weight, L.K, L.d.cin, L.d.cout, L.wimg, stream);
    if (rc) return rc;
  }
  if (scale) {
    V3D_CHECK_HIP(hipMemcpyAsync(L.scale, scale, (size_t)L.d.cout * 4, hipMemcpyDeviceToDevice, st));
    V3D_CHECK_HIP(hipMemcpyAsync(L.shift, shift, (size_t)L.d.cout * 4, hipMemcpyDeviceToDevice, st));
  }
  return V3D_OK;
}

extern "C" int v3d_backbone_forward(v3d_backbone* p, const float* points, int n_points,
                                    const int32_t* frame_offsets_host, int B, float* dense_out, v3d_stream_t stream) {
  return v3d_backbone_forward2(p, points, n_points, frame_offsets_host, B, dense_out, nullptr, nullptr, stream);
}

static int plan_run_layers(v3d_backbone* p, int B, bool hash0_done, float* dense_out, void* dense_hi, void* dense_lo,
                           hipStream_t st);

extern "C" int v3d_backbone_forward2(v3d_backbone* p, const float* points, int n_points,
                                     const int32_t* frame_offsets_host, int B, float* dense_out, void* dense_hi,
                                     void* dense_lo, v3d_stream_t stream) {
  if (!p || !frame_offsets_host || B < 1 || B > p->cfg.max_batch || n_points < 0 || n_points > p->cfg.max_points)
    return V3D_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const v3d_backbone_config& c = p->cfg;
  PlanStage& s0 = p->stages[0];
  V3D_CHECK_HIP(v3d_fill_async(p->ff_begin, 0xFF, p->ff_bytes, st));  // all hash tables, strided nbr tables, flags
  // the voxelizer also fills stage 0's coordinate hash (what rb_hash_build would do in a launch of its own)
  const bool vox_hash = !p->layers.empty() && p->layers[0].d.subm && p->layers[0].builds_rulebook;
  int rc = v3d_i_voxelize(points, n_points, c.point_channels, frame_offsets_host, B, c.voxel_size, c.bounds, c.max_pts,
                          c.max_voxels, nullptr, s0.coords, p->occupancy, p->mean, s0.n_dev, p->vox_ws, p->vox_ws_bytes, 0,
                          vox_hash ? &s0.hash : nullptr, s0.shape, st);
  if (rc) return rc;
  return plan_run_layers(p, B, vox_hash, dense_out, dense_hi, dense_lo, st);
}

// Same plan fed with voxels that already exist (the `item` of the reference's Preprocessor: voxel_mean + coordinates,
// core/preprocess.py:26-33 + detector/layers.py:10-17): two device copies into the plan's stage-0 arrays instead of the
// voxelizer, then the identical layer sequence.  Lets Second.forward(item) / Second.inference(item) -- the entry points
// train.py:63 and inference.py:38 call -- run the native path without voxelizing twice.
extern "C" int v3d_backbone_forward_voxels(v3d_backbone* p, const float* voxel_mean, const int32_t* coords, int n_voxels, int B,
                                           float* dense_out, void* dense_hi, void* dense_lo, v3d_stream_t stream) {
  if (!p || !voxel_mean || !coords || B < 1 || B > p->cfg.max_batch || n_voxels < 0 || n_voxels > p->stages[0].cap)
    return V3D_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  PlanStage& s0 = p->stages[0];
  V3D_CHECK_HIP(v3d_fill_async(p->ff_begin, 0xFF, p->ff_bytes, st));
  if (n_voxels > 0) {
    V3D_CHECK_HIP(hipMemcpyAsync(s0.coords, coords, (size_t)n_voxels * 4 * sizeof(int32_t), hipMemcpyDeviceToDevice, st));
    V3D_CHECK_HIP(hipMemcpyAsync(p->mean, voxel_mean, (size_t)n_voxels * p->cfg.point_channels * sizeof(float),
                                 hipMemcpyDeviceToDevice, st));
  }
  V3D_CHECK_HIP(hipMemsetD32Async((hipDeviceptr_t)s0.n_dev, n_voxels, 1, st));
  return plan_run_layers(p, B, false, dense_out, dense_hi, dense_lo, st);
}

static int plan_run_layers(v3d_backbone* p, int B, bool hash0_done, float* dense_out, void* dense_hi, void* dense_lo,
                           hipStream_t st) {
  const v3d_backbone_config& c = p->cfg;
  int rc = V3D_OK;
  std::vector<char> rb_done(p->layers.size(), 0);
  const float* feat = p->mean;
  for (size_t l = 0; l < p->layers.size(); l++) {
    PlanLayer& L = p->layers[l];
    PlanStage& si = p->stages[L.stage_in];
    PlanStage& so = p->stages[L.stage_out];
    if (L.builds_rulebook && !rb_done[l]) {
      if (L.d.subm) {
        if (!si.hash_ready_by_sparse && !(L.stage_in == 0 && hash0_done)) {
          rc = v3d_i_hash_build(si.coords, si.n_dev, si.cap, si.shape, si.hash, 0, st);
          if (rc) return rc;
          if (L.stage_in == 0) hash0_done = true;
        }
        rc = v3d_i_subm_nbr(si.coords, si.n_dev, si.cap, si.shape, L.d.ksize, si.hash, p->nbr[L.rulebook], st);
      } else {
        // the next layer's submanifold table (over THIS layer's output sites) rides in the same last launch
        const bool fuse = l + 1 < p->layers.size() && p->layers[l + 1].d.subm && p->layers[l + 1].builds_rulebook &&
                          p->layers[l + 1].stage_in == L.stage_out;
        rc = v3d_i_sparse_rulebook(si.coords, si.n_dev, si.cap, si.shape, L.d.ksize, L.d.stride, L.d.padding, so.coords,
                                   so.n_dev, so.cap, p->nbr[L.rulebook], p->overflow + l, so.hash, so.first_ticket,
                                   p->cand_slot, L.chunk_counts, nullptr, 0,
                                   fuse ? p->layers[l + 1].d.ksize : nullptr,
                                   fuse ? p->nbr[p->layers[l + 1].rulebook] : nullptr, st,
                                   p->overflow + p->layers.size() /*summary flag: any layer*/);
        if (fuse) rb_done[l + 1] = 1;
      }
      if (rc) return rc;
    }
    rc = V3D_EUNSUPPORTED;
    // default (0): bf16x3 row-owner kernel where the reduction dim fills an MFMA (Cin >= 16), fp32 wave kernel else
    if (c.conv_algo == 4 || (c.conv_algo == 0 && L.d.cin >= 16))
      rc = v3d_i_sparse_conv_fwd_packed(feat, L.wimg, p->nbr[L.rulebook], so.n_dev, so.cap, L.K, L.d.cin, L.d.cout,
                                        L.has_affine ? L.scale : nullptr, L.has_affine ? L.shift : nullptr, L.d.relu, L.out,
                                        L.rows_hint, st);
    if (rc == V3D_EUNSUPPORTED)
      rc = v3d_sparse_conv_fwd(feat, L.weight, p->nbr[L.rulebook], so.n_dev, so.cap, L.K, L.d.cin, L.d.cout,
                               L.has_affine ? L.scale : nullptr, L.has_affine ? L.shift : nullptr, L.d.relu, L.out,
                               (c.conv_algo == 4 || c.conv_algo == 0) ? 3 : c.conv_algo, st);
    if (rc) return rc;
    feat = L.out;
  }
  if (dense_out) {
    PlanStage& sl = p->stages.back();
    rc = v3d_densify(feat, sl.coords, sl.n_dev, sl.cap, B, p->out_channels, sl.shape, dense_out, st);
    if (rc) return rc;
  }
  if (dense_hi || dense_lo) {
    PlanStage& sl = p->stages.back();
    rc = v3d_densify_nhwc_split(feat, sl.coords, sl.n_dev, sl.cap, B, p->out_channels, sl.shape, dense_hi, dense_lo, st);
    if (rc) return rc;
  }
  return V3D_OK;
}

// Device-resident views of the last forward: what = 0 voxel mean (cap0, C), 1 occupancy (cap0);
// for layers use v3d_backbone_layer_output.
extern "C" int v3d_backbone_layer_output(v3d_backbone* p, int layer, float** features, int32_t** coords,
                                         int32_t** n_rows_dev, int* cap, int* channels, int32_t* shape_host) {
  if (!p || layer < -1 || layer >= (int)p->layers.size()) return V3D_EINVAL;
  if (layer < 0) {  // the voxelizer output feeding layer 0
    if (features) *features = p->mean;
    if (coords) *coords = p->stages[0].coords;
    if (n_rows_dev) *n_rows_dev = p->stages[0].n_dev;
    if (cap) *cap = p->stages[0].cap;
    if (channels) *channels = p->cfg.point_channels;
    if (shape_host)
      for (int j = 0; j < 3; j++) shape_host[j] = p->stages[0].shape[j];
    return V3D_OK;
  }
  PlanLayer& L = p->layers[layer];
  PlanStage& so = p->stages[L.stage_out];
  if (features) *features = L.out;
  if (coords) *coords = so.coords;
  if (n_rows_dev) *n_rows_dev = so.n_dev;
  if (cap) *cap = so.cap;
  if (channels) *channels = L.d.cout;
  if (shape_host)
    for (int j = 0; j < 3; j++) shape_host[j] = so.shape[j];
  return V3D_OK;
}

// Reads the live row counts of the LAST forward (one blocking 4-byte copy per stage) and stores them as the kernel-
// choice hints of the following forwards: capacities are upper bounds (up to 30x the live count in later stages), and
// the two sparse kernels cross over at ~32 k live rows.  Call after a representative forward, outside stream capture.
extern "C" int v3d_backbone_tune(v3d_backbone* p) {
  if (!p) return V3D_EINVAL;
  std::vector<int> n_stage(p->stages.size(), 0);
  for (size_t s = 0; s < p->stages.size(); s++) {
    hipError_t e = hipMemcpy(&n_stage[s], p->stages[s].n_dev, sizeof(int), hipMemcpyDeviceToHost);
    if (e != hipSuccess) return (int)e;
  }
  for (auto& L : p->layers) L.rows_hint = n_stage[L.stage_out];
  return V3D_OK;
}

extern "C" int32_t* v3d_backbone_occupancy(v3d_backbone* p) { return p ? p->occupancy : nullptr; }
// flags[l] = layer l hit its active-site capacity; flags[n_layers] = any of them (one word for the caller's per-frame read)
extern "C" int32_t* v3d_backbone_overflow_flags(v3d_backbone* p) { return p ? p->overflow : nullptr; }
extern "C" int v3d_backbone_num_layers(const v3d_backbone* p) { return p ? (int)p->layers.size() : 0; }
