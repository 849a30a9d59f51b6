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
#include <algorithm>
#include <new>
#include <vector>

#include "v3d_internal.h"
#include "rb_device.h"

struct PlanLayer {
  v3d_layer_desc d;
  int K;
  int stage_in, stage_out;  // coordinate sets
  int rulebook;             // index into nbr tables
  bool builds_rulebook;
  float *weight, *scale, *shift, *out;
  void* out_s;  // the output rows once more, split into the arithmetic's 16-bit pieces under the next layer's scale entry: what the
                // next packed layer gathers (its main loop then converts nothing); nullptr for the last layer
  void* wimg;  // split + packed weights for the bf16x3 kernel
  int has_affine;
  int* chunk_counts;  // strided layers: published per-chunk counts -> offsets (inside the per-frame 0xFF region)
  int rows_hint;      // expected live output rows (kernel choice): capacity-free estimate, refined by v3d_backbone_tune
  int rows_hint_in;   // same for the input stage (the data-gradient pass of the training plan gathers over input rows)
  int cand_buf;       // strided rulebook builders: which of the plan's two ticket scratch buffers (alternating)
};

struct PlanTrain;

struct PlanStage {
  int cap;
  int hash_items;  // what the coordinate hash is sized for: the capacity BEFORE it was padded to an odd multiple of 64 rows (the
                   // few padding rows would otherwise double the table -- and the bytes the per-frame 0xFF fill clears)
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
  int* cand_slot[2] = {nullptr, nullptr};  // two: a strided layer's candidate pass may run while the previous layer's table is filled
  int32_t* overflow = nullptr;  // one flag per layer (<= 0 fine, 1 = capacity hit)
  char* ff_begin = nullptr;     // arena region reset to 0xFF by one memset per forward
  size_t ff_bytes = 0;
  int out_channels = 0;
  uint32_t* bev_occ = nullptr;  // (max_batch * H, ceil(W / 32)) inverted occupancy bits of the last stage (in the 0xFF region)
  // PERSISTENT split BEV planes (v3d_backbone_bev_planes): zero everywhere but the pixels of the last frame written into them, which
  // that frame's densify kernel listed in bev_pix -- the next frame clears exactly those (a few thousand 256-byte rows instead of
  // an 18 MB fill per KITTI frame) before it writes its own.
  void *bev_hi = nullptr, *bev_lo = nullptr;
  int32_t *bev_pix = nullptr, *bev_pix_n = nullptr;
  int ring_tiles_min = 2;  // v3d_backbone_set_throughput_mode: 4
#ifdef V3D_EXP_NO_RIDERS  // (A/B builds: tools/build_variant.sh)
  bool rb_riders = false;
#else
  bool rb_riders = true;   // inference forwards: rulebook steps ride in the sparse layers' launches (PlanRbQueue)
#endif
  // Arithmetic of the packed layers in the INFERENCE entry points (the training plan always runs bf16x3): v3d_backbone_set_precision.
  // f16s reads one scale entry {s, 1/s, limit, max} per tensor from act_tab: entry l = the rows layer l gathers, entry n_layers =
  // the BEV map (scale of the split planes the last layer / the densify kernel writes).  Entries start as {1, 1, 2^15}: valid
  // for magnitudes below 2^15, full precision once v3d_backbone_calibrate has set them from a frame.
  int prec = V3D_PREC_BF16X3;
  bool calibrating = false;  // the next forwards run every layer on the exact-fp32 kernel (no scales involved): calibration pass
  bool presplit = true;      // packed layers also write their rows split for the next packed layer (v3d_backbone_set_presplit: A/B)
  bool fp32_rows = true;     // ... beside the fp32 rows (v3d_backbone_layer_output); false in throughput mode: split rows only
  float* act_tab = nullptr;
  float* w_inv_tab = nullptr;  // [n_layers] 1 / s_w of the layers' f16s images, beside act_tab: what the kernels read instead of the
                               // images' trailers (cold lines)
  unsigned* frame_max = nullptr;  // [n_layers + 1] f16s: "the tensor entry l describes held a value >= limit * 2^-12 in THIS frame" (0 / 1), zeroed
                                  // by the frame's first launch and set by the producing waves: what plan_quiet_check_kernel reads
  const int32_t** entry_rows = nullptr;  // [n_layers + 1] device: the live-row count of the stage each entry's tensor lives on
  struct PlanTrain* train = nullptr;  // training buffers, allocated by the first v3d_backbone_train_forward
  // (Measured and removed in round 4: the rulebook chain on a second stream with one event per finished rulebook -- inside a
  // captured graph the fork / join costs more than the overlap returns, 3 209 -> 2 526 frames/s pipelined; docs/rounds/design_rounds_1-4.md 5c.4.)
};

static int conv_fan(const v3d_layer_desc& d) {
  int fan = 1;
  for (int j = 0; j < 3; j++) fan *= (d.ksize[j] + d.stride[j] - 1) / d.stride[j];
  return fan;
}

extern "C" int v3d_backbone_create(const v3d_backbone_config* cfg, const v3d_layer_desc* descs, v3d_backbone** out) {
  if (!cfg || !descs || !out || cfg->n_layers < 1 || cfg->max_batch < 1 || cfg->max_points < 1) return V3D_EINVAL;
  if (cfg->max_batch > 64) return V3D_EUNSUPPORTED;  // the batch index rides in 6 bits of a site key (rulebook.hip RB_MAX_BATCH)
  v3d_backbone* p = new (std::nothrow) v3d_backbone();
  if (!p) return V3D_EINVAL;
  p->cfg = *cfg;
  const float growth = cfg->growth > 0.f ? cfg->growth : 2.0f;
  long long cap0 = (long long)cfg->max_batch * cfg->max_voxels;
  if (cap0 > cfg->max_points) cap0 = cfg->max_points;
  if (cap0 < 1) cap0 = 1;

  // ---- pass 1: geometry, capacities, rulebook sharing
  PlanStage s0{};
  s0.hash_items = (int)cap0;
  cap0 = (cap0 + 63) / 64 * 64;  // (an odd multiple of 64 rows: see the strided stages below)
  if ((cap0 / 64) % 2 == 0) cap0 += 64;
  s0.cap = (int)cap0;
  for (int j = 0; j < 3; j++) s0.shape[j] = cfg->grid_shape[j];
  p->stages.push_back(s0);
  std::vector<int> key_to_rb;  // (stage, key) -> rulebook
  std::vector<int> key_stage, key_id;
  int cur = 0, cin = cfg->point_channels, n_strided = 0;
  long long max_tickets = 1;
  for (int l = 0; l < cfg->n_layers; l++) {
    PlanLayer L{};
    L.d = descs[l];
    if (L.d.cin != cin) { delete p; return V3D_EINVAL; }
    L.K = L.d.ksize[0] * L.d.ksize[1] * L.d.ksize[2];
    if (L.K < 1 || L.K > 62) { delete p; return V3D_EUNSUPPORTED; }
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
      // The capacity is the row stride of the stage's neighbour tables (nbr[k][o], k-major): a power of two would put the K rows
      // a tile reads at the same offset of every 128 KB -- one memory channel, one cache set.  Make it an odd multiple of 64 rows.
      ns.hash_items = (int)cap;
      cap = (cap + 63) / 64 * 64;
      if ((cap / 64) % 2 == 0) cap += 64;
      ns.cap = (int)cap;
      ns.hash_ready_by_sparse = true;
      const long long tickets = (long long)p->stages[cur].cap * conv_fan(L.d);  // tickets per input row: rulebook.hip rb_ticket
      if (tickets > max_tickets) max_tickets = tickets;
      p->stages.push_back(ns);
      cur = (int)p->stages.size() - 1;
      L.stage_out = cur;
      L.rulebook = (int)p->nbr_cap.size();
      L.builds_rulebook = true;
      L.cand_buf = n_strided++ & 1;
      p->nbr_cap.push_back(ns.cap);
    }
    L.rows_hint = L.rows_hint_in = 0;  // unknown until tuned: the 16-row kernel (right for KITTI-size frames)
    cin = L.d.cout;
    p->layers.push_back(L);
  }
  p->out_channels = cin;

  for (const auto& st : p->stages)  // a site's row index rides in 24 bits of its table word (v3d_common.h)
    if (st.cap > V3D_SITE_MAX_ROWS) { delete p; return V3D_EUNSUPPORTED; }
  // ---- pass 2: size and carve the arena (two passes over the same carving code)
  auto carve = [&](V3dArena& ar) {
    // ---- everything that must read 0xFF at the start of a forward is contiguous: ONE memset per frame
    p->ff_begin = ar.base + ar.off;
    p->vox_ws_bytes = v3d_voxelize_workspace(cfg->max_points);
    p->vox_ws = ar.take<char>(p->vox_ws_bytes);
    for (auto& st : p->stages) {
      st.hash.hcap = v3d_hash_capacity(st.hash_items);  // load factor <= (cap / hash_items) / 2: 0.501 at worst
      st.hash.keys = ar.take<v3d_key_t>(st.hash.hcap);
      st.first_ticket = ar.take<unsigned>(st.hash.hcap);
      st.hash.vals = ar.take<int>(st.hash.hcap);
    }
    p->nbr.resize(p->nbr_cap.size());
    std::vector<int> rbK(p->nbr_cap.size(), 1);
    for (auto& L : p->layers) rbK[L.rulebook] = L.K;
    // (the strided tables are NOT in this region: the emit pass of their rulebook initialises the columns of the sites it creates,
    // 1.5 MB of stores instead of 14 MB of fill per KITTI frame)
    p->overflow = ar.take<int32_t>(p->layers.size() + 1);
    {  // inverted BEV occupancy bitmap of the last stage (0xFF = nothing occupied): input of the background-skipping head
      const PlanStage& sl = p->stages.back();
      p->bev_occ = ar.take<uint32_t>(v3d_bev_occupancy_words(cfg->max_batch, sl.shape[1], sl.shape[2]));
    }
    for (auto& L : p->layers)  // strided layers: per-layer count slots, -1 = "not published" at the start of a frame
      if (!L.d.subm) L.chunk_counts = ar.take<int>((size_t)v3d_ceil_div((long long)p->stages[L.stage_in].cap * conv_fan(L.d), V3D_SCAN_CHUNK) + 2);
    p->ff_bytes = (size_t)((ar.base + ar.off) - p->ff_begin);
    // ---- the rest needs no per-frame initialisation
    for (size_t i = 0; i < p->nbr.size(); i++) p->nbr[i] = ar.take<int32_t>((size_t)rbK[i] * p->nbr_cap[i]);
    p->occupancy = ar.take<int32_t>(p->stages[0].cap);
    p->mean = ar.take<float>((size_t)p->stages[0].cap * cfg->point_channels);
    for (auto& st : p->stages) {
      st.coords = ar.take<int32_t>((size_t)st.cap * 4);
      st.n_dev = ar.take<int32_t>(1);
    }
    {
      const PlanStage& sl = p->stages.back();
      const size_t plane = (size_t)cfg->max_batch * sl.shape[0] * sl.shape[1] * sl.shape[2] * p->out_channels * 2;  // bf16
      p->bev_hi = ar.take<char>(plane);
      p->bev_lo = ar.take<char>(plane);
      p->bev_pix = ar.take<int32_t>(sl.cap);
      p->bev_pix_n = ar.take<int32_t>(1);
    }
    p->act_tab = ar.take<float>(4 * (p->layers.size() + 1) + p->layers.size());
    p->w_inv_tab = p->act_tab + 4 * (p->layers.size() + 1);
    p->frame_max = ar.take<unsigned>(p->layers.size() + 1);
    p->entry_rows = ar.take<const int32_t*>(p->layers.size() + 1);
    p->cand_slot[0] = ar.take<int>((size_t)max_tickets);
    p->cand_slot[1] = ar.take<int>((size_t)max_tickets);
    for (auto& L : p->layers) {
      L.weight = ar.take<float>((size_t)L.K * L.d.cin * L.d.cout);
      L.wimg = ar.take<char>(v3d_sparse_conv_weight_image_bytes(L.K, L.d.cin, L.d.cout));
      L.scale = ar.take<float>(L.d.cout);
      L.shift = ar.take<float>(L.d.cout);
      L.out = ar.take<float>((size_t)p->stages[L.stage_out].cap * L.d.cout);
      L.out_s = (&L != &p->layers.back() && L.d.cout % 8 == 0) ? (void*)ar.take<float>((size_t)p->stages[L.stage_out].cap * L.d.cout) : nullptr;
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
  if (e == hipSuccess) e = hipMemset(p->bev_hi, 0, (size_t)((char*)p->bev_lo - (char*)p->bev_hi) * 2);  // the planes are adjacent
  if (e == hipSuccess) e = hipMemset(p->bev_pix_n, 0, sizeof(int32_t));
  if (e == hipSuccess) e = hipMemset(p->frame_max, 0, (p->layers.size() + 1) * sizeof(unsigned));
  if (e == hipSuccess) {
    std::vector<const int32_t*> rows(p->layers.size() + 1);
    rows[0] = p->stages[0].n_dev;
    for (size_t l = 0; l < p->layers.size(); l++) rows[l + 1] = p->stages[p->layers[l].stage_out].n_dev;
    e = hipMemcpy(p->entry_rows, rows.data(), rows.size() * sizeof(const int32_t*), hipMemcpyHostToDevice);
  }
  if (e == hipSuccess) {
    std::vector<float> tab;
    for (size_t i = 0; i <= p->layers.size(); i++) tab.insert(tab.end(), {1.f, 1.f, 32768.f, 0.f});
    for (size_t i = 0; i < p->layers.size(); i++) tab.push_back(1.f);
    e = hipMemcpy(p->act_tab, tab.data(), tab.size() * sizeof(float), hipMemcpyHostToDevice);
  }
  if (e != hipSuccess) { (void)hipFree(p->arena); delete p; return (int)e; }
  *out = p;
  return V3D_OK;
}

static void plan_train_free(v3d_backbone* p);

extern "C" void v3d_backbone_destroy(v3d_backbone* p) {
  if (!p) return;
  plan_train_free(p);
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
    int rc = v3d_sparse_conv_pack_weights(L.weight, L.K, L.d.cin, L.d.cout, p->prec, L.wimg, stream);
    if (rc) return rc;
    if (p->prec == V3D_PREC_F16S) {  // the image's 1 / s_w (written by the pack kernel) into the plan's hot table
      const char* trailer = (const char*)L.wimg + v3d_sparse_conv_weight_image_bytes(L.K, L.d.cin, L.d.cout) - 256;
      V3D_CHECK_HIP(hipMemcpyAsync(p->w_inv_tab + layer, trailer + 4, sizeof(float), hipMemcpyDeviceToDevice, st));
    }
  }
  if (scale) {
    V3D_CHECK_HIP(hipMemcpyAsync(L.scale, scale, (size_t)L.d.cout * 4, hipMemcpyDeviceToDevice, st));
    V3D_CHECK_HIP(hipMemcpyAsync(L.shift, shift, (size_t)L.d.cout * 4, hipMemcpyDeviceToDevice, st));
  }
  return V3D_OK;
}

static int plan_run_layers(v3d_backbone* p, int B, bool hash0_done, float* dense_out, void* dense_hi, void* dense_lo,
                           hipStream_t st, bool reuse_rulebooks = false);

// Start of a frame: the 0xFF fill of the plan's per-frame region and -- when the caller asked for the plan's OWN persistent planes
// -- the zeroing of the pixels the previous frame wrote, in ONE launch (the clear rides in the fill's launch, a dozen launches
// ahead of the densify kernel that needs it).
__global__ __launch_bounds__(256) void plan_frame_start_kernel(uint4* __restrict__ ff, size_t nvec, int fill_blocks,
                                                               const int* __restrict__ pix, const int* __restrict__ n_ptr, int cap,
                                                               int units, uint4* __restrict__ hi, uint4* __restrict__ lo,
                                                               unsigned* __restrict__ frame_max, int n_frame_max) {
  if (blockIdx.x == 0 && (int)threadIdx.x < n_frame_max) frame_max[threadIdx.x] = 0u;  // (f16s: the tensors' running maxima of this frame)
  if ((int)blockIdx.x < fill_blocks) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (size_t)fill_blocks * 256)
      ff[i] = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
    return;
  }
  const long long total = (long long)min(*n_ptr, cap) * units;
  const int nb = gridDim.x - fill_blocks;
  for (long long t = (long long)(blockIdx.x - fill_blocks) * 256 + threadIdx.x; t < total; t += (long long)nb * 256) {
    const size_t o = (size_t)pix[t / units] * units + (size_t)(t % units);
    hi[o] = make_uint4(0u, 0u, 0u, 0u);
    lo[o] = make_uint4(0u, 0u, 0u, 0u);
  }
}

static int plan_own_planes(v3d_backbone* p, void* dense_hi, void* dense_lo, bool& own) {
  own = false;
  if (!dense_hi && !dense_lo) return V3D_OK;
  if ((dense_hi == p->bev_hi) != (dense_lo == p->bev_lo)) return V3D_EINVAL;  // both planes of the plan, or neither
  own = dense_hi == p->bev_hi;
  return V3D_OK;
}

static int plan_frame_start(v3d_backbone* p, void* dense_hi, void* dense_lo, hipStream_t st) {
  bool own;
  int rc = plan_own_planes(p, dense_hi, dense_lo, own);
  if (rc) return rc;
  if (!own || (p->ff_bytes & 15) || ((uintptr_t)p->ff_begin & 15) || (p->out_channels * p->stages.back().shape[0]) % 8) {
    V3D_CHECK_HIP(v3d_fill_async(p->ff_begin, 0xFF, p->ff_bytes, st));  // all hash tables, count slots, flags, the occupancy bitmap
    V3D_CHECK_HIP(v3d_fill_async(p->frame_max, 0, (p->layers.size() + 1) * sizeof(unsigned), st));
    if (!own) return V3D_OK;
    const PlanStage& sl = p->stages.back();
    return v3d_i_bev_clear_pixels(p->bev_pix, p->bev_pix_n, sl.cap, p->out_channels * sl.shape[0], p->bev_hi, p->bev_lo, st);
  }
  const PlanStage& sl = p->stages.back();
  const int units = p->out_channels * sl.shape[0] / 8;
  const size_t nvec = p->ff_bytes / 16;
  const int fill_blocks = (int)std::min<size_t>((nvec + 255) / 256, 4096);
  const int clear_blocks = (int)std::min<long long>(v3d_ceil_div((long long)sl.cap * units, 256), 1024);
  hipLaunchKernelGGL(plan_frame_start_kernel, dim3(fill_blocks + clear_blocks), dim3(256), 0, st, (uint4*)p->ff_begin, nvec, fill_blocks,
                     p->bev_pix, p->bev_pix_n, sl.cap, units, (uint4*)p->bev_hi, (uint4*)p->bev_lo, p->frame_max, (int)p->layers.size() + 1);
  V3D_CHECK_LAUNCH();
  return V3D_OK;
}

// f16s, the DOWNWARD range check.  An output beyond its consumer's limit is caught where it is produced (V3D_FLAG_RANGE); a tensor
// that has become much SMALLER than the frame its scale entry was calibrated on is only known once the whole tensor exists: every
// producing wave that sees a value of at least limit * 2^-V3D_QUIET_BITS sets the tensor's word (v3d_mark_seen), and this one-wave
// launch behind the last layer raises the frame's summary word to V3D_FLAG_QUIET when a tensor WITH rows, whose calibration frame
// was not all zeros, set nothing (read in the frame's one host synchronisation: recalibrate on this frame, run it again -- as
// for V3D_FLAG_RANGE).  Entry 0 (the voxel means, consumed by the exact-fp32 input layer) is not judged.
__global__ __launch_bounds__(64) void plan_quiet_check_kernel(const float* __restrict__ act_tab, const unsigned* __restrict__ seen,
                                                              const int32_t* const* __restrict__ entry_rows, int n_entries,
                                                              int* __restrict__ flag) {
  bool quiet = false;
  for (int l = 1 + (int)threadIdx.x; l < n_entries; l += 64)
    quiet |= seen[l] == 0u && *entry_rows[l] > 0 && act_tab[4 * l + 3] != 0.f;
  if (__ballot(quiet) != 0ull && threadIdx.x == 0) atomicMax(flag, V3D_FLAG_QUIET);
}

// (timing variant without a per-frame fill: only the planes' clear)
static int plan_clear_own_planes(v3d_backbone* p, void* dense_hi, void* dense_lo, hipStream_t st) {
  bool own;
  int rc = plan_own_planes(p, dense_hi, dense_lo, own);
  if (rc || !own) return rc;
  const PlanStage& sl = p->stages.back();
  return v3d_i_bev_clear_pixels(p->bev_pix, p->bev_pix_n, sl.cap, p->out_channels * sl.shape[0], p->bev_hi, p->bev_lo, st);
}

extern "C" int v3d_backbone_set_throughput_mode(v3d_backbone* p, int on) {
  if (!p) return V3D_EINVAL;
  p->ring_tiles_min = on ? 4 : 2;
  p->fp32_rows = !on;  // (v3d_backbone_layer_output: the feature rows of layers followed by a packed layer are then not written)
  return V3D_OK;
}

extern "C" int v3d_backbone_bev_planes(v3d_backbone* p, void** hi, void** lo) {
  if (!p || !hi || !lo) return V3D_EINVAL;
  *hi = p->bev_hi;
  *lo = p->bev_lo;
  return V3D_OK;
}

// Arithmetic of the inference entry points.  The weight images are packed per precision: call before v3d_backbone_set_layer
// (runtime.BackbonePlan re-uploads the layers when it changes).
extern "C" int v3d_backbone_set_precision(v3d_backbone* p, int prec) {
  if (!p || (prec != V3D_PREC_BF16X3 && prec != V3D_PREC_F16S)) return V3D_EINVAL;
  p->prec = prec;
  return V3D_OK;
}
extern "C" int v3d_backbone_precision(const v3d_backbone* p) { return p ? p->prec : V3D_EINVAL; }

// on (default): every packed layer of the inference entry points also writes its output rows split into the arithmetic's 16-bit
// pieces (under the next layer's scale entry) and the next packed layer gathers those: no conversion work in its main loop.
// off: every layer splits the fp32 rows it gathers in registers (rounds 1-4; A/B measurements).  Same results bit for bit.
extern "C" int v3d_backbone_set_presplit(v3d_backbone* p, int on) {
  if (!p) return V3D_EINVAL;
  p->presplit = on != 0;
  return V3D_OK;
}

// f16s: (n_layers + 1) x {s, 1/s, limit, max} in device memory, entry l = input rows of layer l, entry n_layers = the BEV map
extern "C" float* v3d_backbone_act_scales(v3d_backbone* p) { return p ? p->act_tab : nullptr; }

// on: the following forwards run every layer on the exact-fp32 kernel (whatever the magnitudes: nothing is scaled) -- the pass
// v3d_backbone_calibrate reads.  Their split planes are written with the scale entry as it stands.
extern "C" int v3d_backbone_set_calibrating(v3d_backbone* p, int on) {
  if (!p) return V3D_EINVAL;
  p->calibrating = on != 0;
  return V3D_OK;
}

// Scale entries from the LAST forward's tensors (enqueued on `stream`, no host synchronisation): per tensor the power of two that
// puts its largest magnitude `headroom_bits` binades below the top of the scaled range, i.e. later frames may exceed this frame's
// maxima by 2^(headroom_bits + 1) before the range flag is raised.  Costs nothing in precision up to ~7 bits (spconv.hip).
extern "C" int v3d_backbone_calibrate(v3d_backbone* p, int headroom_bits, v3d_stream_t stream) {
  if (!p || headroom_bits < 0 || headroom_bits > 12) return V3D_EINVAL;
  for (size_t l = 0; l <= p->layers.size(); l++) {
    const float* rows = l == 0 ? p->mean : p->layers[l - 1].out;
    const int C = l == 0 ? p->cfg.point_channels : p->layers[l - 1].d.cout;
    const PlanStage& sg = p->stages[l == 0 ? 0 : p->layers[l - 1].stage_out];
    int rc = v3d_act_scale_from_rows(rows, sg.n_dev, sg.cap, C, headroom_bits, p->act_tab + 4 * l, nullptr, stream);
    if (rc) return rc;
  }
  return V3D_OK;
}

extern "C" int v3d_backbone_forward(v3d_backbone* p, const float* points, int n_points,
                                     const int32_t* frame_offsets_host, int B, float* dense_out, void* dense_hi,
                                     void* dense_lo, v3d_stream_t stream) {
  if (!p || !frame_offsets_host || B < 1 || B > p->cfg.max_batch || n_points < 0 || n_points > p->cfg.max_points)
    return V3D_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const v3d_backbone_config& c = p->cfg;
  PlanStage& s0 = p->stages[0];
  {
    const int rcc = plan_frame_start(p, dense_hi, dense_lo, st);
    if (rcc) return rcc;
  }
  // the voxelizer also fills stage 0's coordinate hash (what rb_hash_build would do in a launch of its own)
  const bool vox_hash = !p->layers.empty() && p->layers[0].d.subm && p->layers[0].builds_rulebook;
  int rc = v3d_i_voxelize(points, n_points, c.point_channels, frame_offsets_host, B, c.voxel_size, c.bounds, c.max_pts,
                          c.max_voxels, nullptr, s0.coords, p->occupancy, p->mean, s0.n_dev, p->vox_ws, p->vox_ws_bytes, 0,
                          vox_hash ? &s0.hash : nullptr, s0.shape, st);
  if (rc) return rc;
  return plan_run_layers(p, B, vox_hash, dense_out, dense_hi, dense_lo, st);
}

// The convolutions of the LAST forwarded frame once more, on the site lists and neighbour tables that forward left in the plan
// (no voxelizer, no rulebook build): the "rulebooks prebuilt" timing variant of SURVEY.md section 8(d).  Same outputs.
extern "C" int v3d_backbone_forward_reuse(v3d_backbone* p, int B, float* dense_out, void* dense_hi, void* dense_lo,
                                          v3d_stream_t stream) {
  if (!p || B < 1 || B > p->cfg.max_batch) return V3D_EINVAL;
  {
    const int rcc = plan_clear_own_planes(p, dense_hi, dense_lo, (hipStream_t)stream);
    if (rcc) return rcc;
  }
  return plan_run_layers(p, B, true, dense_out, dense_hi, dense_lo, (hipStream_t)stream, true);
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
  {
    const int rcc = plan_frame_start(p, dense_hi, dense_lo, st);
    if (rcc) return rcc;
  }
  if (n_voxels > 0) {
    V3D_CHECK_HIP(hipMemcpyAsync(s0.coords, coords, (size_t)n_voxels * 4 * sizeof(int32_t), hipMemcpyDeviceToDevice, st));
    V3D_CHECK_HIP(hipMemcpyAsync(p->mean, voxel_mean, (size_t)n_voxels * p->cfg.point_channels * sizeof(float),
                                 hipMemcpyDeviceToDevice, st));
  }
  V3D_CHECK_HIP(hipMemsetD32Async((hipDeviceptr_t)s0.n_dev, n_voxels, 1, st));
  return plan_run_layers(p, B, false, dense_out, dense_hi, dense_lo, st);
}

// rulebook of layer l (and, riding in the strided builder's last launch, the submanifold table of the layer after it)
// `cand_done` (inference chain, nullable): the candidate pass of a strided layer rides in the launch that produces its input
// sites' last table -- stage 0's submanifold table or the previous strided layer's fill (rulebook.hip RbCandJob) -- and is
// marked done here.
static bool plan_next_strided(v3d_backbone* p, size_t l, int stage, std::vector<char>* cand_done, V3dRbCandNext& nx) {
  if (!cand_done) return false;
  for (size_t m = l + 1; m < p->layers.size(); m++) {
    PlanLayer& M = p->layers[m];
    if (!M.builds_rulebook || M.d.subm || M.stage_in != stage) continue;
    if ((*cand_done)[m]) return false;
    PlanStage &mi = p->stages[M.stage_in], &mo = p->stages[M.stage_out];
    nx.coords_in = mi.coords; nx.n_in = mi.n_dev; nx.cap_in = mi.cap; nx.shape = mi.shape;
    nx.ksize = M.d.ksize; nx.stride = M.d.stride; nx.padding = M.d.padding;
    nx.out = mo.hash; nx.first_ticket = mo.first_ticket; nx.cand_slot = p->cand_slot[M.cand_buf];
    nx.overflow = p->overflow + m; nx.overflow_any = p->overflow + p->layers.size();
    (*cand_done)[m] = 1;
    return true;
  }
  return false;
}

// The rulebook chain of a frame as a QUEUE of steps (rb_device.h RbStep) instead of launches: it depends on coordinates only, so
// the inference forward builds the whole queue first and lets a sparse layer's launch carry the next pending SCAN as its first
// workgroups -- the chain runs ahead of the convolutions and the scans (4 of a KITTI frame's 38 launches, ~9 us of dependent round
// trips each) disappear under them.  rb_step[rulebook] = the step whose completion the table needs (-1: launched directly).  A fill,
// a scan no launch carried in time, and whatever must precede a direct launch are launched on their own -- the order of the
// steps never changes.
struct PlanRbQueue {
  std::vector<RbStep> steps;
  std::vector<int> rb_step;
  size_t launched = 0;
};
static int plan_rb_flush(PlanRbQueue& q, int upto /*last step index that must have been launched; -1: none*/, hipStream_t st) {
  while ((int)q.launched <= upto && q.launched < q.steps.size()) {
    const int rc = v3d_i_rb_step_launch(q.steps[q.launched++], st);
    if (rc) return rc;
  }
  return V3D_OK;
}

static int plan_layer_rulebook(v3d_backbone* p, size_t l, std::vector<char>& rb_done, bool& hash0_done, hipStream_t st,
                               std::vector<char>* cand_done = nullptr, PlanRbQueue* defer = nullptr) {
  PlanLayer& L = p->layers[l];
  if (!L.builds_rulebook || rb_done[l]) return V3D_OK;
  PlanStage& si = p->stages[L.stage_in];
  PlanStage& so = p->stages[L.stage_out];
  int rc;
  V3dRbCandNext nx;
  if (L.d.subm) {
    if (defer) {  // a direct launch: everything queued so far comes first
      rc = plan_rb_flush(*defer, (int)defer->steps.size() - 1, st);
      if (rc) return rc;
    }
    if (!si.hash_ready_by_sparse && !(L.stage_in == 0 && hash0_done)) {
      rc = v3d_i_hash_build(si.coords, si.n_dev, si.cap, si.shape, si.hash, 0, st);
      if (rc) return rc;
      if (L.stage_in == 0) hash0_done = true;
    }
    const bool carry = plan_next_strided(p, l, L.stage_in, cand_done, nx);
    return v3d_i_subm_nbr(si.coords, si.n_dev, si.cap, si.shape, L.d.ksize, si.hash, p->nbr[L.rulebook], st, carry ? &nx : nullptr);
  }
  const bool fuse = l + 1 < p->layers.size() && p->layers[l + 1].d.subm && p->layers[l + 1].builds_rulebook &&
                    p->layers[l + 1].stage_in == L.stage_out;
  const int mine_done = cand_done && (*cand_done)[l];
  const bool carry = plan_next_strided(p, l, L.stage_out, cand_done, nx);
  if (defer && mine_done) {
    RbStep scan, fill;
    rc = v3d_i_sparse_rulebook_steps(si.coords, si.n_dev, si.cap, si.shape, L.d.ksize, L.d.stride, L.d.padding, so.coords, so.n_dev, so.cap,
                                     p->nbr[L.rulebook], p->overflow + l, so.hash, so.first_ticket, p->cand_slot[L.cand_buf], L.chunk_counts,
                                     fuse ? p->layers[l + 1].d.ksize : nullptr, fuse ? p->nbr[p->layers[l + 1].rulebook] : nullptr,
                                     p->overflow + p->layers.size(), carry ? &nx : nullptr, 1, &scan, &fill);
    if (rc) return rc;
    defer->steps.push_back(scan);
    defer->steps.push_back(fill);
    defer->rb_step[L.rulebook] = (int)defer->steps.size() - 1;
    if (fuse) {
      rb_done[l + 1] = 1;
      defer->rb_step[p->layers[l + 1].rulebook] = (int)defer->steps.size() - 1;
    }
    return V3D_OK;
  }
  if (defer) {  // (its candidate pass is a launch of its own: direct, behind everything queued)
    rc = plan_rb_flush(*defer, (int)defer->steps.size() - 1, st);
    if (rc) return rc;
  }
  rc = v3d_i_sparse_rulebook(si.coords, si.n_dev, si.cap, si.shape, L.d.ksize, L.d.stride, L.d.padding, so.coords, so.n_dev,
                             so.cap, p->nbr[L.rulebook], p->overflow + l, so.hash, so.first_ticket, p->cand_slot[L.cand_buf],
                             L.chunk_counts, nullptr, 0, fuse ? p->layers[l + 1].d.ksize : nullptr,
                             fuse ? p->nbr[p->layers[l + 1].rulebook] : nullptr, st,
                             p->overflow + p->layers.size() /*summary flag: any layer*/, mine_done, carry ? &nx : nullptr,
                             1 /*the emit pass initialises the table's live columns*/);
  if (fuse) rb_done[l + 1] = 1;
  return rc;
}

// out = act((sum_k feat[nbr[k]] @ W[k]) * scale + shift) of layer l: the packed split-precision kernels where the reduction dim
// fills an MFMA (Cin >= 16), the exact-fp32 wave kernel else.  `inference`: the plan's arithmetic (f16s reads the layer's scale
// entries and checks its output against the next one); the training plan passes false (bf16x3 images, no scales).
static int plan_layer_conv(v3d_backbone* p, PlanLayer& L, const float* feat, const void* wimg, const float* weight,
                           const float* scale, const float* shift, int relu, float* out, hipStream_t st,
                           const V3dDensifyOut* densify = nullptr, bool* densified = nullptr, bool inference = false,
                           const void** feat_split = nullptr /*in: the input rows' split copy or null; out: this layer's or null*/,
                           const RbScanJob* rider = nullptr, bool* rider_taken = nullptr /*a rulebook scan the launch may carry*/) {
  const v3d_backbone_config& c = p->cfg;
  PlanStage& so = p->stages[L.stage_out];
  int rc = V3D_EUNSUPPORTED;
  if (densified) *densified = false;
  if (rider_taken) *rider_taken = false;
  if (v3d_ablate('c') || (v3d_ablate('r') && L.d.cin == 64 && L.d.cout == 64 && L.K == 27)) return V3D_OK;
  const bool exact_pass = inference && p->calibrating;
  if (!exact_pass && (c.conv_algo == 4 || (c.conv_algo == 0 && L.d.cin >= 16))) {
    const int prec = inference ? p->prec : V3D_PREC_BF16X3;
    const size_t l = (size_t)(&L - p->layers.data());
    const V3dActScale as{p->act_tab + 4 * l, p->act_tab + 4 * (l + 1), p->overflow + p->layers.size(), p->w_inv_tab + l,
#ifdef V3D_EXP_NO_SEEN
                         nullptr};
#else
                         inference ? p->frame_max + l + 1 : nullptr};
#endif
    const void* in_s = feat_split ? *feat_split : nullptr;
    // (f16s only: its split is conversion instructions that issue slowly -- profiles/r05_f16s_split_ab.txt --; with bf16 pieces the
    //  second copy of the rows costs the pipelined mode more than the plain split it saves: profiles/r05_presplit_ab.txt)
    //  In THROUGHPUT mode the fp32 rows of such a layer are not written at all -- nobody reads them: the next layer gathers the
    //  split copy, captured graphs expose no intermediate rows -- so the split copy replaces the fp32 stores instead of adding to
    //  them, and pays in both arithmetics.
    //  Both only when the NEXT layer really runs a packed kernel (a channel pair outside the packed table falls back to the exact-fp32
    //  kernel, which gathers fp32 rows: they must then exist, and a split copy would be written for nobody).
    const bool next_packed = l + 1 < p->layers.size() && p->layers[l + 1].d.cin >= 16 && (c.conv_algo == 4 || c.conv_algo == 0) &&
                             v3d_i_sparse_conv_packed_supported(p->layers[l + 1].d.cin, p->layers[l + 1].d.cout);
    void* out_s = (feat_split && p->presplit && next_packed && (prec == V3D_PREC_F16S || !p->fp32_rows)) ? L.out_s : nullptr;
    if (out_s && !p->fp32_rows && !densify && v3d_i_sparse_conv_packed_supported(L.d.cin, L.d.cout)) out = nullptr;
    rc = v3d_i_sparse_conv_fwd_packed(feat, wimg, p->nbr[L.rulebook], so.n_dev, so.cap, L.K, L.d.cin, L.d.cout, scale, shift,
                                      relu, out, L.rows_hint, st, densify, p->ring_tiles_min, prec, &as, in_s, out_s, rider, rider_taken);
    if (rc == V3D_OK && densify && densified) *densified = true;
    if (feat_split) *feat_split = rc == V3D_OK ? out_s : nullptr;
  } else if (feat_split) {
    *feat_split = nullptr;  // an exact-fp32 layer writes fp32 rows only
  }
  if (rc == V3D_EUNSUPPORTED) {
    // (an f16s plan outside its calibration pass: the exact layer's output feeds a scaled layer -- checked against that entry)
    const bool check = inference && p->prec == V3D_PREC_F16S && !p->calibrating;
    const size_t l = (size_t)(&L - p->layers.data());
    rc = v3d_i_sparse_conv_fwd_exact(feat, weight, p->nbr[L.rulebook], so.n_dev, so.cap, L.K, L.d.cin, L.d.cout, scale, shift, relu,
                                     out, exact_pass ? 0 : ((c.conv_algo == 4 || c.conv_algo == 0) ? 3 : c.conv_algo), st,
                                     check ? p->act_tab + 4 * (l + 1) : nullptr, check ? p->overflow + p->layers.size() : nullptr,
                                     check ? p->frame_max + l + 1 : nullptr, rider, rider_taken);
  }
  return rc;
}

static int plan_run_layers(v3d_backbone* p, int B, bool hash0_done, float* dense_out, void* dense_hi, void* dense_lo,
                           hipStream_t st, bool reuse_rulebooks) {
  int rc = V3D_OK;
  std::vector<char> rb_done(p->layers.size(), 0);
  const float* feat = p->mean;
  const void* feat_split = nullptr;  // the split copy of `feat` (written by the previous packed layer), or null
  // the candidate pass of a strided layer rides in the launch that produces its input sites' last table (rulebook.hip RbCandJob)
  std::vector<char> cand_done(p->layers.size(), 0);
  bool densified = false;
  // the whole rulebook chain as a queue of steps (see PlanRbQueue); what must be a launch of its own is launched here
  PlanRbQueue rbq;
  const bool riders = !reuse_rulebooks && p->rb_riders;
  if (riders) {
    rbq.rb_step.assign(p->nbr.size(), -1);
    for (size_t l = 0; l < p->layers.size(); l++) {
      rc = plan_layer_rulebook(p, l, rb_done, hash0_done, st, &cand_done, &rbq);
      if (rc) return rc;
    }
  }
  for (size_t l = 0; l < p->layers.size(); l++) {
    PlanLayer& L = p->layers[l];
    if (!reuse_rulebooks && !riders) {
      rc = plan_layer_rulebook(p, l, rb_done, hash0_done, st, &cand_done);
      if (rc) return rc;
    }
    const RbScanJob* rider = nullptr;
    bool rider_taken = false;
    if (riders) {  // this layer's table first, and a pending fill (a launch of its own); then the next scan rides in this layer's launch
      rc = plan_rb_flush(rbq, rbq.rb_step[L.rulebook], st);
      if (rc) return rc;
      if (rbq.launched < rbq.steps.size() && rbq.steps[rbq.launched].kind == 2) {
        rc = plan_rb_flush(rbq, (int)rbq.launched, st);
        if (rc) return rc;
      }
      if (rbq.launched < rbq.steps.size() && rbq.steps[rbq.launched].kind == 1) rider = &rbq.steps[rbq.launched].scan;
    }
    // the LAST layer writes the plan's own BEV planes from its epilogue (.dense() without a launch of its own) where it runs the
    // 16-row kernel anyway: not 3x3x3 (the SECOND backbone ends in a (3, 1, 1) layer) and below the 64-row kernel's row counts
    V3dDensifyOut dn{};
    const bool want_dn = l + 1 == p->layers.size() && !dense_out && dense_hi == p->bev_hi && dense_lo == p->bev_lo && dense_hi &&
                         L.K != 27 && L.rows_hint < 32768 && L.d.cin >= 16 && L.d.cout % 16 == 0;
    if (want_dn) {
      const PlanStage& sl = p->stages.back();
      dn = V3dDensifyOut{sl.coords, sl.shape[0], sl.shape[1], sl.shape[2], p->bev_hi, p->bev_lo, p->bev_occ, p->bev_pix, p->bev_pix_n};
    }
    rc = plan_layer_conv(p, L, feat, L.wimg, L.weight, L.has_affine ? L.scale : nullptr, L.has_affine ? L.shift : nullptr,
                         L.d.relu, L.out, st, want_dn ? &dn : nullptr, &densified, true, &feat_split, rider, &rider_taken);
    if (rc) return rc;
    if (rider_taken) rbq.launched++;
    feat = L.out;
  }
  if (riders) {
    rc = plan_rb_flush(rbq, (int)rbq.steps.size() - 1, st);
    if (rc) return rc;
  }
  if (dense_out) {
    PlanStage& sl = p->stages.back();
    rc = v3d_densify(feat, sl.coords, sl.n_dev, sl.cap, B, p->out_channels, sl.shape, dense_out, st);
    if (rc) return rc;
  }
  if ((dense_hi || dense_lo) && !densified) {
    PlanStage& sl = p->stages.back();
    const bool own = dense_hi == p->bev_hi && dense_lo == p->bev_lo;  // (plan_clear_own_planes ran at the start of this frame)
    rc = v3d_i_densify_nhwc_split(feat, sl.coords, sl.n_dev, sl.cap, B, p->out_channels, sl.shape, dense_hi, dense_lo, p->bev_occ,
                                  st, own ? p->bev_pix : nullptr, own ? p->bev_pix_n : nullptr, p->prec,
                                  p->act_tab + 4 * p->layers.size(), p->overflow + p->layers.size(),
                                  p->prec == V3D_PREC_F16S && !p->calibrating ? p->frame_max + p->layers.size() : nullptr);
    if (rc) return rc;
  }
#ifndef V3D_EXP_NO_SEEN
  if (p->prec == V3D_PREC_F16S && !p->calibrating) {
    hipLaunchKernelGGL(plan_quiet_check_kernel, dim3(1), dim3(64), 0, st, p->act_tab, p->frame_max, p->entry_rows, (int)p->layers.size() + 1,
                       p->overflow + (int)p->layers.size());
    V3D_CHECK_LAUNCH();
  }
#endif
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
  for (auto& L : p->layers) {
    L.rows_hint = n_stage[L.stage_out];
    L.rows_hint_in = n_stage[L.stage_in];
  }
  return V3D_OK;
}

// Coordinate-only pass: builds every rulebook for the given voxel coordinates (no features, no convolutions), waits for the
// stream and takes the kernel-choice hints from the row counts -- v3d_backbone_tune without a forward.  The training plan
// calls it before its FIRST step, so that step already runs the kernels every later step will use (a training forward
// cannot simply be repeated after tuning: it updates the BatchNorm running statistics).
extern "C" int v3d_backbone_tune_from_voxels(v3d_backbone* p, const int32_t* coords, int n_voxels, int B, v3d_stream_t stream) {
  if (!p || !coords || B < 1 || B > p->cfg.max_batch || n_voxels < 1 || n_voxels > p->stages[0].cap) return V3D_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  PlanStage& s0 = p->stages[0];
  V3D_CHECK_HIP(v3d_fill_async(p->ff_begin, 0xFF, p->ff_bytes, st));
  V3D_CHECK_HIP(hipMemcpyAsync(s0.coords, coords, (size_t)n_voxels * 4 * sizeof(int32_t), hipMemcpyDeviceToDevice, st));
  V3D_CHECK_HIP(hipMemsetD32Async((hipDeviceptr_t)s0.n_dev, n_voxels, 1, st));
  std::vector<char> rb_done(p->layers.size(), 0);
  bool hash0_done = false;
  for (size_t l = 0; l < p->layers.size(); l++) {
    int rc = plan_layer_rulebook(p, l, rb_done, hash0_done, st);
    if (rc) return rc;
  }
  V3D_CHECK_HIP(hipStreamSynchronize(st));
  return v3d_backbone_tune(p);
}

// Inverted BEV occupancy bitmap of the LAST forward that produced split planes (v3d_backbone_forward / _forward_voxels): device
// pointer into the plan's arena, (B * H, ceil(W / 32)) words; what v3d_conv2d_nhwc_split takes as `occ`.
extern "C" uint32_t* v3d_backbone_bev_occupancy(v3d_backbone* p) { return p ? p->bev_occ : nullptr; }

extern "C" int32_t* v3d_backbone_occupancy(v3d_backbone* p) { return p ? p->occupancy : nullptr; }
// flags[l] = layer l hit its active-site capacity; flags[n_layers] = any of them (one word for the caller's per-frame read)
extern "C" int32_t* v3d_backbone_overflow_flags(v3d_backbone* p) { return p ? p->overflow : nullptr; }
extern "C" int v3d_backbone_num_layers(const v3d_backbone* p) { return p ? (int)p->layers.size() : 0; }

// =================================================================================================
// Training plan: the same stage / rulebook machinery driven forwards AND backwards by one C call each way.
//
// Replaces, for the sparse half of a train step (train.py:63-67 through detector/second.py:41-46 and
// detector/sparse_cnn.py:15-30,151-175), the ~400 Python-level operator calls of the module-by-module autograd path
// (rulebook builders, SparseConvFunction, SparseBatchNormReLUFunction, .dense() and their backward twins):
//   forward   per layer: pack the CURRENT weights -> conv (no affine) -> batch-statistics BatchNorm (+ ReLU) with the
//             running statistics updated in the merge kernel; the pre-BN rows and the saved mean / invstd stay in the
//             training arena for the backward;
//   backward  d(BEV) gathered back onto the last stage's rows, then per layer in reverse: BatchNorm(+ReLU) backward ->
//             weight gradient (exact-fp32 MFMA reduction) -> data gradient = the forward kernel on the transposed
//             rulebook (a submanifold table is its own transpose with the offsets reversed) and transposed weights.
// Row counts never leave the device (BatchNorm included: v3d_i_sparse_bn_*), so a whole train step can be captured in a
// HIP graph.  Parameters are read from / gradients written to the caller's device pointers (v3d_train_layer).
// =================================================================================================
struct PlanTrainLayer {
  float* conv_out = nullptr;  // (cap_out, cout) pre-BatchNorm rows
  float* stats = nullptr;     // save_mean | save_invstd | unbiased variance, 3 x cout
  void* wimg_t = nullptr;     // packed image of the transposed weights (Cin >= 16 layers)
  bool wimg_t_ready = false;  // packed by this step's forward call
  int32_t* nbr_t = nullptr;   // strided layers: (K, cap_in) transposed rulebook
};

struct PlanTrain {
  void* arena = nullptr;
  size_t arena_bytes = 0;
  std::vector<PlanTrainLayer> tl;
  float* g[2] = {nullptr, nullptr};  // gradient wrt a layer's output / input rows (ping-pong)
  float* g_conv = nullptr;           // gradient wrt the pre-BatchNorm rows
  float* w_t = nullptr;              // fp32 transposed weights of the layer in flight
  void* dw_ws = nullptr;
  size_t dw_ws_bytes = 0;
  void* bn_ws = nullptr;
  size_t bn_ws_bytes = 0;
  bool forward_done = false;
};

static void plan_train_free(v3d_backbone* p) {
  if (!p->train) return;
  if (p->train->arena) (void)hipFree(p->train->arena);
  delete p->train;
  p->train = nullptr;
}

static int plan_train_alloc(v3d_backbone* p) {
  if (p->train) return V3D_OK;
  for (auto& L : p->layers) {
    const int c = L.d.cout;
    if (c < 4 || c > V3D_BLOCK || (c & (c - 1))) return V3D_EUNSUPPORTED;  // sparse_bn.hip: power-of-two channel counts
  }
  PlanTrain* t = new (std::nothrow) PlanTrain();
  if (!t) return V3D_EINVAL;
  t->tl.resize(p->layers.size());
  auto carve = [&](V3dArena& ar) {
    size_t max_rows = 1, max_w = 1, max_dw = 1;
    int max_c = 4;
    for (size_t l = 0; l < p->layers.size(); l++) {
      PlanLayer& L = p->layers[l];
      PlanTrainLayer& T = t->tl[l];
      const PlanStage &si = p->stages[L.stage_in], &so = p->stages[L.stage_out];
      T.conv_out = ar.take<float>((size_t)so.cap * L.d.cout);
      T.stats = ar.take<float>((size_t)3 * L.d.cout);
      if (L.d.cin >= 16 && L.d.cin % 16 == 0)
        T.wimg_t = ar.take<char>(v3d_sparse_conv_weight_image_bytes(L.K, L.d.cout, L.d.cin));
      if (!L.d.subm && l > 0) T.nbr_t = ar.take<int32_t>((size_t)L.K * si.cap);
      max_rows = std::max(max_rows, std::max((size_t)so.cap * L.d.cout, (size_t)si.cap * L.d.cin));
      max_w = std::max(max_w, (size_t)L.K * L.d.cin * L.d.cout);
      max_dw = std::max(max_dw, v3d_sparse_conv_bwd_weight_workspace(L.K, L.d.cin, L.d.cout));
      max_c = std::max(max_c, (int)L.d.cout);
    }
    t->g[0] = ar.take<float>(max_rows);
    t->g[1] = ar.take<float>(max_rows);
    t->g_conv = ar.take<float>(max_rows);
    t->w_t = ar.take<float>(max_w);
    t->dw_ws_bytes = max_dw;
    t->dw_ws = ar.take<char>(max_dw);
    t->bn_ws_bytes = v3d_sparse_bn_workspace(1, max_c);
    t->bn_ws = ar.take<char>(t->bn_ws_bytes);
  };
  {
    V3dArena probe((void*)256, (size_t)1 << 60);
    carve(probe);
    t->arena_bytes = probe.off + 4096;
  }
  hipError_t e = hipMalloc(&t->arena, t->arena_bytes);
  if (e != hipSuccess) { delete t; return (int)e; }
  V3dArena ar(t->arena, t->arena_bytes);
  carve(ar);
  if (!ar.ok()) { (void)hipFree(t->arena); delete t; return V3D_EWORKSPACE; }
  p->train = t;
  return V3D_OK;
}

extern "C" size_t v3d_backbone_train_arena_bytes(const v3d_backbone* p) { return (p && p->train) ? p->train->arena_bytes : 0; }

// Wt[k'][co][ci] = W[k][ci][co], k' = K-1-k when `flip` (submanifold layers: offset k of the transposed table is offset
// K-1-k of the forward table)
__global__ __launch_bounds__(V3D_BLOCK) void plan_transpose_weights_kernel(const float* __restrict__ w, int K, int cin, int cout,
                                                                           int flip, float* __restrict__ wt) {
  const int total = K * cin * cout;
  for (int t = blockIdx.x * V3D_BLOCK + threadIdx.x; t < total; t += gridDim.x * V3D_BLOCK) {
    const int k = t / (cin * cout), rem = t - k * cin * cout, co = rem / cin, ci = rem - co * cin;  // t indexes wt
    const int ks = flip ? K - 1 - k : k;
    wt[t] = w[((size_t)ks * cin + ci) * cout + co];
  }
}

// rows[i, c] = grad_dense[b, c, z, y, x]   ((B, C, D, H, W): the backward of densify_kernel)
__global__ __launch_bounds__(V3D_BLOCK) void plan_densify_bwd_kernel(const float* __restrict__ gdense, const int4* __restrict__ coords,
                                                                     const int* __restrict__ n_ptr, int cap, int C, int D, int H,
                                                                     int Wd, float* __restrict__ rows) {
  const int n = min(*n_ptr, cap);
  const long long total = (long long)n * C;
  const size_t vol = (size_t)D * H * Wd;
  for (long long t = (long long)blockIdx.x * V3D_BLOCK + threadIdx.x; t < total; t += (long long)gridDim.x * V3D_BLOCK) {
    const int i = (int)(t / C), ch = (int)(t % C);
    const int4 c = coords[i];
    rows[t] = gdense[((size_t)c.x * C + ch) * vol + ((size_t)c.y * H + c.z) * Wd + c.w];
  }
}

// BEV map straight in the layout / type the autocast RPN consumes: (B, H, W, C * D) bf16 (torch: a (B, C * D, H, W) tensor in
// channels_last), channel = c * D + z as volume.dense().flatten(1, 2) orders them; RNE rounding like torch's .to(bfloat16).
// Saves the fp32 NCHW write, the channels_last copy and the cast of a 144 MB tensor per step (and their backward twins).
__global__ __launch_bounds__(V3D_BLOCK) void plan_densify_nhwc_bf16_kernel(const float* __restrict__ feat, const int4* __restrict__ coords,
                                                                           const int* __restrict__ n_ptr, int cap, int C, int D, int H,
                                                                           int Wd, __bf16* __restrict__ out) {
  const int n = min(*n_ptr, cap);
  const long long total = (long long)n * C;
  for (long long t = (long long)blockIdx.x * V3D_BLOCK + threadIdx.x; t < total; t += (long long)gridDim.x * V3D_BLOCK) {
    const int i = (int)(t / C), ch = (int)(t % C);
    const int4 c = coords[i];
    out[(((size_t)c.x * H + c.z) * Wd + c.w) * ((size_t)C * D) + (size_t)ch * D + c.y] = (__bf16)feat[t];
  }
}

__global__ __launch_bounds__(V3D_BLOCK) void plan_densify_nhwc_bf16_bwd_kernel(const __bf16* __restrict__ g, const int4* __restrict__ coords,
                                                                               const int* __restrict__ n_ptr, int cap, int C, int D,
                                                                               int H, int Wd, float* __restrict__ rows) {
  const int n = min(*n_ptr, cap);
  const long long total = (long long)n * C;
  for (long long t = (long long)blockIdx.x * V3D_BLOCK + threadIdx.x; t < total; t += (long long)gridDim.x * V3D_BLOCK) {
    const int i = (int)(t / C), ch = (int)(t % C);
    const int4 c = coords[i];
    rows[t] = (float)g[(((size_t)c.x * H + c.z) * Wd + c.w) * ((size_t)C * D) + (size_t)ch * D + c.y];
  }
}

extern "C" int v3d_backbone_train_forward(v3d_backbone* p, const float* voxel_mean, const int32_t* coords, int n_voxels, int B,
                                          const v3d_train_layer* io, float* dense_out, void* dense_nhwc_bf16,
                                          v3d_stream_t stream) {
  if (!p || !voxel_mean || !coords || !io || (dense_out == nullptr) == (dense_nhwc_bf16 == nullptr) || B < 1 || B > p->cfg.max_batch || n_voxels < 1 ||
      n_voxels > p->stages[0].cap)
    return V3D_EINVAL;
  for (size_t l = 0; l < p->layers.size(); l++)
    if (!io[l].weight || !io[l].gamma || !io[l].beta || (io[l].running_mean == nullptr) != (io[l].running_var == nullptr))
      return V3D_EINVAL;
  if (p->prec != V3D_PREC_BF16X3) return V3D_EINVAL;  // the step re-packs the layers' images as bf16 pieces: a training plan is a bf16x3 plan
  int rc = plan_train_alloc(p);
  if (rc) return rc;
  PlanTrain* t = p->train;
  hipStream_t st = (hipStream_t)stream;
  PlanStage& s0 = p->stages[0];
  V3D_CHECK_HIP(v3d_fill_async(p->ff_begin, 0xFF, p->ff_bytes, st));
  V3D_CHECK_HIP(hipMemcpyAsync(s0.coords, coords, (size_t)n_voxels * 4 * sizeof(int32_t), hipMemcpyDeviceToDevice, st));
  V3D_CHECK_HIP(hipMemcpyAsync(p->mean, voxel_mean, (size_t)n_voxels * p->cfg.point_channels * sizeof(float),
                               hipMemcpyDeviceToDevice, st));
  V3D_CHECK_HIP(hipMemsetD32Async((hipDeviceptr_t)s0.n_dev, n_voxels, 1, st));
  std::vector<char> rb_done(p->layers.size(), 0);
  bool hash0_done = false;
  const float* feat = p->mean;
  {
    // every packed weight image of the step in one launch: the forward image of each layer and the image of its transposed
    // weights (the data gradient of v3d_backbone_train_backward; the weights do not change between the two calls of a step)
    V3dPackJobs jobs;
    int nj = 0;
    for (size_t l = 0; l < p->layers.size(); l++) {
      PlanLayer& L = p->layers[l];
      PlanTrainLayer& T = t->tl[l];
      if (L.d.cin >= 16 && L.d.cout % 16 == 0 && nj < V3D_PACK_JOBS_MAX) {
        jobs.w[nj] = io[l].weight; jobs.img[nj] = L.wimg; jobs.K[nj] = L.K; jobs.cin[nj] = L.d.cin; jobs.cout[nj] = L.d.cout; jobs.mode[nj] = 0;
        nj++;
      }
      T.wimg_t_ready = false;
      if (l > 0 && T.wimg_t && L.d.cout >= 16 && nj < V3D_PACK_JOBS_MAX) {  // transposed layer: Cin' = cout, Cout' = cin
        jobs.w[nj] = io[l].weight; jobs.img[nj] = T.wimg_t; jobs.K[nj] = L.K; jobs.cin[nj] = L.d.cout; jobs.cout[nj] = L.d.cin;
        jobs.mode[nj] = L.d.subm ? 2 : 1;
        nj++;
        T.wimg_t_ready = true;
      }
    }
    if (nj) {
      rc = v3d_i_sparse_conv_pack_batch(jobs, nj, st);
      if (rc) return rc;
    }
  }
  for (size_t l = 0; l < p->layers.size(); l++) {
    PlanLayer& L = p->layers[l];
    PlanTrainLayer& T = t->tl[l];
    PlanStage& so = p->stages[L.stage_out];
    rc = plan_layer_rulebook(p, l, rb_done, hash0_done, st);
    if (rc) return rc;
    rc = plan_layer_conv(p, L, feat, L.wimg, io[l].weight, nullptr, nullptr, 0, T.conv_out, st);
    if (rc) return rc;
    rc = v3d_i_sparse_bn_relu_fwd(T.conv_out, so.cap, so.n_dev, L.d.cout, io[l].gamma, io[l].beta, io[l].eps, L.d.relu, L.out,
                                  T.stats, T.stats + L.d.cout, T.stats + 2 * L.d.cout, io[l].running_mean, io[l].running_var,
                                  io[l].momentum, io[l].num_batches_tracked, t->bn_ws, t->bn_ws_bytes, st);
    if (rc) return rc;
    feat = L.out;
  }
  PlanStage& sl = p->stages.back();
  if (dense_out) {
    rc = v3d_densify(feat, sl.coords, sl.n_dev, sl.cap, B, p->out_channels, sl.shape, dense_out, st);
    if (rc) return rc;
  } else {
    const size_t bytes = (size_t)B * sl.shape[0] * sl.shape[1] * sl.shape[2] * p->out_channels * 2;
    V3D_CHECK_HIP(v3d_fill_async(dense_nhwc_bf16, 0, bytes, st));
    const long long total = (long long)sl.cap * p->out_channels;
    hipLaunchKernelGGL(plan_densify_nhwc_bf16_kernel, dim3((int)std::min<long long>(v3d_ceil_div(total, V3D_BLOCK), 4096)),
                       dim3(V3D_BLOCK), 0, st, feat, (const int4*)sl.coords, sl.n_dev, sl.cap, p->out_channels, sl.shape[0],
                       sl.shape[1], sl.shape[2], (__bf16*)dense_nhwc_bf16);
    V3D_CHECK_LAUNCH();
  }
  t->forward_done = true;
  return V3D_OK;
}

extern "C" int v3d_backbone_train_backward(v3d_backbone* p, const float* grad_dense, const void* grad_nhwc_bf16, int B,
                                           const v3d_train_layer* io, v3d_stream_t stream) {
  if (!p || (grad_dense == nullptr) == (grad_nhwc_bf16 == nullptr) || !io || B < 1 || B > p->cfg.max_batch) return V3D_EINVAL;
  if (!p->train || !p->train->forward_done) return V3D_EINVAL;  // backward of what?
  for (size_t l = 0; l < p->layers.size(); l++)
    if (!io[l].weight || !io[l].gamma || !io[l].beta || !io[l].grad_weight || !io[l].grad_gamma || !io[l].grad_beta)
      return V3D_EINVAL;
  PlanTrain* t = p->train;
  hipStream_t st = (hipStream_t)stream;
  int rc;
  {
    PlanStage& sl = p->stages.back();
    const long long total = (long long)sl.cap * p->out_channels;
    const dim3 grid((int)std::min<long long>(v3d_ceil_div(total, V3D_BLOCK), 4096));
    if (grad_dense)
      hipLaunchKernelGGL(plan_densify_bwd_kernel, grid, dim3(V3D_BLOCK), 0, st, grad_dense, (const int4*)sl.coords, sl.n_dev,
                         sl.cap, p->out_channels, sl.shape[0], sl.shape[1], sl.shape[2], t->g[0]);
    else
      hipLaunchKernelGGL(plan_densify_nhwc_bf16_bwd_kernel, grid, dim3(V3D_BLOCK), 0, st, (const __bf16*)grad_nhwc_bf16,
                         (const int4*)sl.coords, sl.n_dev, sl.cap, p->out_channels, sl.shape[0], sl.shape[1], sl.shape[2],
                         t->g[0]);
  }
  int cur = 0;
  for (int l = (int)p->layers.size() - 1; l >= 0; l--) {
    PlanLayer& L = p->layers[l];
    PlanTrainLayer& T = t->tl[l];
    PlanStage &si = p->stages[L.stage_in], &so = p->stages[L.stage_out];
    const float* x_in = l > 0 ? p->layers[l - 1].out : p->mean;
    rc = v3d_i_sparse_bn_relu_bwd(T.conv_out, t->g[cur], so.cap, so.n_dev, L.d.cout, io[l].gamma, io[l].beta, T.stats,
                                  T.stats + L.d.cout, L.d.relu, t->g_conv, io[l].grad_gamma, io[l].grad_beta, t->bn_ws,
                                  t->bn_ws_bytes, st);
    if (rc) return rc;
    rc = v3d_sparse_conv_bwd_weight(x_in, t->g_conv, p->nbr[L.rulebook], so.n_dev, so.cap, L.K, L.d.cin, L.d.cout,
                                    io[l].grad_weight, t->dw_ws, t->dw_ws_bytes, stream);
    if (rc) return rc;
    if (l == 0) break;  // the voxel features need no gradient
    const bool packed_t = T.wimg_t && T.wimg_t_ready && L.d.cout >= 16;
    if (!packed_t) {  // fp32 transposed weights for the wave kernel below
      const int total = L.K * L.d.cin * L.d.cout;
      hipLaunchKernelGGL(plan_transpose_weights_kernel, dim3(std::min(v3d_ceil_div(total, V3D_BLOCK), 1024)), dim3(V3D_BLOCK), 0,
                         st, io[l].weight, L.K, L.d.cin, L.d.cout, L.d.subm ? 1 : 0, t->w_t);
    }
    const int32_t* nbr_t = p->nbr[L.rulebook];
    if (!L.d.subm) {
      rc = v3d_rulebook_transpose(p->nbr[L.rulebook], so.n_dev, so.cap, L.K, si.cap, T.nbr_t, stream);
      if (rc) return rc;
      nbr_t = T.nbr_t;
    }
    rc = V3D_EUNSUPPORTED;
    if (packed_t) {  // transposed layer: Cin' = cout, Cout' = cin
      rc = v3d_i_sparse_conv_fwd_packed(t->g_conv, T.wimg_t, nbr_t, si.n_dev, si.cap, L.K, L.d.cout, L.d.cin, nullptr, nullptr,
                                        0, t->g[cur ^ 1], L.rows_hint_in, st);
    }
    if (rc == V3D_EUNSUPPORTED) {
      if (packed_t) {  // (the packed kernel declined the shape: the fp32 transposed weights were skipped above)
        const int total = L.K * L.d.cin * L.d.cout;
        hipLaunchKernelGGL(plan_transpose_weights_kernel, dim3(std::min(v3d_ceil_div(total, V3D_BLOCK), 1024)), dim3(V3D_BLOCK), 0,
                           st, io[l].weight, L.K, L.d.cin, L.d.cout, L.d.subm ? 1 : 0, t->w_t);
      }
      rc = v3d_sparse_conv_fwd(t->g_conv, t->w_t, nbr_t, si.n_dev, si.cap, L.K, L.d.cout, L.d.cin, nullptr, nullptr, 0,
                               t->g[cur ^ 1], 0, stream);
    }
    if (rc) return rc;
    cur ^= 1;
  }
  V3D_CHECK_LAUNCH();
  return V3D_OK;
}
