// rb_device.h -- device side of the hash-indexed rulebooks (rulebook.hip), shared with the kernels that carry rulebook steps as
// riders (spconv.hip): geometry, the submanifold look-ups, tickets and the candidate pass, the scan + emit and fill bodies.
#pragma once
#include "v3d_internal.h"

struct RbGeom {
  int in_shape[3];   // D, H, W of the input grid
  int out_shape[3];  // D, H, W of the output grid
  int ks[3], stride[3], pad[3];
  int K;
  int kc[3], Kc;     // strided layers: ceil(ks / stride) per axis and their product -- tickets per input row (see rb_ticket)
};

// A site table word is key << 24 | row (v3d_common.h): keys own 40 bits.  Grids are limited to 2^34 cells (fill_geom), which leaves
// 6 bits for the batch index: a coordinate row with b outside [0, 64) has no key -- it is never inserted (its neighbours read -1;
// the strided builder raises its overflow flag), instead of aliasing another site's key.
#define RB_MAX_BATCH 64
__device__ __forceinline__ v3d_key_t rb_key(int b, int z, int y, int x, const int* shape) {
  return (((v3d_key_t)b * shape[0] + z) * shape[1] + y) * shape[2] + x;
}

// ---------------------------------------------------------------------------------- submanifold table
// nbr[k][o] of offset k for output row o (the kernels that call this are further down: they can carry a candidate job),
// KPT consecutive offsets per thread.  One look-up per thread leaves a wave with a single dependent chain
// (coordinate row -> table word -> store) and the launch latency-bound: ~45 G look-ups/s at Waymo range, where 2.8 - 4 M look-ups
// per table cost 60 - 85 us.  Here a thread reads its coordinate row ONCE and issues the first probes of its KPT offsets as
// straight-line, unconditional loads (an offset with nothing to look up -- centre tap, outside the grid, past K -- probes slot 0 and
// ignores the word), so KPT table reads are in flight per lane before the first is consumed; only a probe that lands on ANOTHER
// site's word walks on (linear probing, load factor < 0.5 of the capacity: rare).  Stores stay coalesced along o for every offset.
template <int KPT>
__device__ __forceinline__ void rb_subm_entries(const int4* __restrict__ coords, int n, int cap, const RbGeom& g, const V3dHash& h,
                                                int* __restrict__ nbr, int kgroup, int o) {
  if (o >= n) return;
  const int4 c = coords[o];
  const bool keyed = (unsigned)c.x < (unsigned)RB_MAX_BATCH;
  v3d_key_t key[KPT], w[KPT];
  unsigned s[KPT];
  bool look[KPT];
#pragma unroll
  for (int j = 0; j < KPT; j++) {
    const int k = kgroup * KPT + j;
    const int kx = k % g.ks[2], ky = (k / g.ks[2]) % g.ks[1], kz = k / (g.ks[2] * g.ks[1]);
    const int z = c.y + kz - g.ks[0] / 2, y = c.z + ky - g.ks[1] / 2, x = c.w + kx - g.ks[2] / 2;
    look[j] = keyed && k < g.K && 2 * k + 1 != g.K && z >= 0 && z < g.in_shape[0] && y >= 0 && y < g.in_shape[1] && x >= 0 &&
              x < g.in_shape[2];
    key[j] = look[j] ? rb_key(c.x, z, y, x, g.in_shape) : 0;
    s[j] = look[j] ? v3d_hash_start(key[j], h) : 0u;
    w[j] = h.keys[s[j]];
  }
#pragma unroll
  for (int j = 0; j < KPT; j++) {
    const int k = kgroup * KPT + j;
    int v = -1;
    if (2 * k + 1 == g.K) {
      v = o;
    } else if (look[j]) {
      const v3d_key_t word = w[j];
      if ((word >> V3D_SITE_ROW_BITS) == key[j]) {
        const unsigned row = (unsigned)word & V3D_SITE_NO_ROW;
        if (row != V3D_SITE_NO_ROW) v = (int)row;
      } else if (word != V3D_EMPTY_KEY) {
        v = v3d_site_find_row_from(h, key[j], (s[j] + 1) & h.mask);  // another site's word: walk on
      }
    }
    if (k < g.K) nbr[(size_t)k * cap + o] = v;
  }
}
// The 3x3x3 form of the same: offsets decoded at compile time (KPT = 9: one kz plane per thread, kgroup = kz; KPT = 27: the whole
// kernel), keys as the centre's key plus a per-offset delta ((dz * H + dy) * W + dx: one 64-bit add instead of three multiply-adds).
template <int KPT>
__device__ __forceinline__ void rb_subm_entries_333(const int4* __restrict__ coords, int n, int cap, const RbGeom& g, const V3dHash& h,
                                                    int* __restrict__ nbr, int kgroup, int o) {
  static_assert(KPT == 9 || KPT == 27, "one kz plane or the whole kernel");
  if (o >= n) return;
  const int4 c = coords[o];
  const bool keyed = (unsigned)c.x < (unsigned)RB_MAX_BATCH;
  const int D = g.in_shape[0], H = g.in_shape[1], W = g.in_shape[2];
  const v3d_key_t key0 = rb_key(c.x, c.y, c.z, c.w, g.in_shape);
  v3d_key_t key[KPT], w[KPT];
  unsigned s[KPT];
  bool look[KPT];
#pragma unroll
  for (int j = 0; j < KPT; j++) {
    const int dz = (KPT == 27 ? j / 9 : kgroup) - 1, dy = (j / 3) % 3 - 1, dx = j % 3 - 1;
    const int z = c.y + dz, y = c.z + dy, x = c.w + dx;
    look[j] = keyed && (dz | dy | dx) != 0 && z >= 0 && z < D && y >= 0 && y < H && x >= 0 && x < W;
    const long long delta = ((long long)dz * H + dy) * W + dx;
    key[j] = look[j] ? key0 + (v3d_key_t)delta : 0;
    s[j] = look[j] ? v3d_hash_start(key[j], h) : 0u;
    w[j] = h.keys[s[j]];
  }
#pragma unroll
  for (int j = 0; j < KPT; j++) {
    const int k = kgroup * KPT + j;
    const int dz = (KPT == 27 ? j / 9 : kgroup) - 1, dy = (j / 3) % 3 - 1, dx = j % 3 - 1;
    int v = -1;
    if ((dz | dy | dx) == 0) {
      v = o;  // centre tap: the site itself
    } else if (look[j]) {
      const v3d_key_t word = w[j];
      if ((word >> V3D_SITE_ROW_BITS) == key[j]) {
        const unsigned row = (unsigned)word & V3D_SITE_NO_ROW;
        if (row != V3D_SITE_NO_ROW) v = (int)row;
      } else if (word != V3D_EMPTY_KEY) {
        v = v3d_site_find_row_from(h, key[j], (s[j] + 1) & h.mask);  // another site's word: walk on
      }
    }
    nbr[(size_t)k * cap + o] = v;
  }
}
// Offsets per thread: 9 (one kz plane of a 3x3x3 kernel) for the long site lists, 1 for the short ones.  Measured in the frame
// (profiles/r05_rb_subm_kpt.txt): at Waymo range (104 k - 180 k rows) 9 per thread takes the four tables from 62 / 85 / 50 / 25 us to
// 45 / 64 / 39 / 16 us; on a KITTI frame (16 k - 30 k rows, a few hundred workgroups) the same form is SLOWER (9.8 -> 22.6 us, 11.2 ->
// 15.7 us): too few threads are left and each walks nine address computations in a row.  The switch is on the capacity, which
// host and device both know.
#ifndef RB_SUBM_KPT
#define RB_SUBM_KPT 9
#endif
#define RB_SUBM_KPT_MIN_ROWS 40960
#ifndef RB_SUBM_KPT_SMALL
#define RB_SUBM_KPT_SMALL 1  // offsets per thread below RB_SUBM_KPT_MIN_ROWS (3x3x3: 1, or 3 -- measured on the KITTI frame in round 6:
                             // 4 787 vs 4 760 frames/s pipelined, 453 vs 449 us one frame at a time: not adopted)
#endif
__host__ __device__ inline int rb_subm_kpt(int cap, const RbGeom& g) {
  const bool k333 = g.K == 27 && g.ks[0] == 3 && g.ks[1] == 3 && g.ks[2] == 3;
  return (k333 && cap >= RB_SUBM_KPT_MIN_ROWS) ? RB_SUBM_KPT : (k333 ? RB_SUBM_KPT_SMALL : 1);
}
__host__ __device__ inline int rb_subm_blocks(int cap, const RbGeom& g) {
  const int kpt = rb_subm_kpt(cap, g);
  return ((cap + V3D_BLOCK - 1) / V3D_BLOCK) * ((g.K + kpt - 1) / kpt);
}
__device__ __forceinline__ void rb_subm_block(const int4* __restrict__ coords, int n, int cap, const RbGeom& g, const V3dHash& h,
                                              int* __restrict__ nbr, int idx) {
  const int nbx = (cap + V3D_BLOCK - 1) / V3D_BLOCK, o = (idx % nbx) * V3D_BLOCK + threadIdx.x;
  const int kpt = rb_subm_kpt(cap, g);
  if (kpt == RB_SUBM_KPT)
    rb_subm_entries_333<RB_SUBM_KPT>(coords, n, cap, g, h, nbr, idx / nbx, o);
  else if (RB_SUBM_KPT_SMALL > 1 && kpt == RB_SUBM_KPT_SMALL)
    rb_subm_entries<RB_SUBM_KPT_SMALL>(coords, n, cap, g, h, nbr, idx / nbx, o);
  else
    rb_subm_entries<1>(coords, n, cap, g, h, nbr, idx / nbx, o);
}

// ---------------------------------------------------------------------------------- strided conv
// TICKETS.  The sequential rule numbers the outputs in the order a loop over (input row i, kernel offset k) first touches them.
// Of the K offsets of an input only those on the output lattice can touch anything: along an axis with stride s an input at v =
// c + pad reaches the offsets k = v % s, v % s + s, ... (< ks) -- ceil(ks / s) of them, 2 x 2 x 2 = 8 of 27 for a 3x3x3 stride-2
// layer.  A ticket is t = i * Kc + j with j = (jz, jy, jx) enumerating those offsets in increasing k: the same relative order as
// i * K + k over the offsets that can hit, so "smallest ticket per output slot" numbers the outputs exactly as before -- on a
// ticket space 3.4x smaller (candidate pass, flag scan and table fill all walk it).
// rb_ticket: the kernel offset k and the output cell of ticket slot j of input c; false if that slot is empty / outside.
__device__ __forceinline__ bool rb_ticket(const int4 c, int j, const RbGeom& g, int& k, int& oz, int& oy, int& ox) {
  const int jx = j % g.kc[2], jy = (j / g.kc[2]) % g.kc[1], jz = j / (g.kc[2] * g.kc[1]);
  const int vz = c.y + g.pad[0], vy = c.z + g.pad[1], vx = c.w + g.pad[2];
  const int kz = vz % g.stride[0] + jz * g.stride[0], ky = vy % g.stride[1] + jy * g.stride[1], kx = vx % g.stride[2] + jx * g.stride[2];
  if (kz >= g.ks[0] || ky >= g.ks[1] || kx >= g.ks[2] || kz > vz || ky > vy || kx > vx) return false;
  oz = (vz - kz) / g.stride[0];
  oy = (vy - ky) / g.stride[1];
  ox = (vx - kx) / g.stride[2];
  k = (kz * g.ks[1] + ky) * g.ks[2] + kx;
  return oz < g.out_shape[0] && oy < g.out_shape[1] && ox < g.out_shape[2];
}
// a candidate word: the output slot and the kernel offset of a live ticket, -1 for a dead one (slot < 2^26, k < 63)
#define RB_SLOT_BITS 26
#define RB_SLOT_MASK ((1 << RB_SLOT_BITS) - 1)

// candidate output coordinate of (input coord c, offset k); false if not on the output lattice
__device__ __forceinline__ bool rb_candidate(const int4 c, int k, const RbGeom& g, int& oz, int& oy, int& ox) {
  const int kx = k % g.ks[2], ky = (k / g.ks[2]) % g.ks[1], kz = k / (g.ks[2] * g.ks[1]);
  const int vz = c.y + g.pad[0] - kz, vy = c.z + g.pad[1] - ky, vx = c.w + g.pad[2] - kx;
  if (vz < 0 || vy < 0 || vx < 0) return false;
  if (vz % g.stride[0] || vy % g.stride[1] || vx % g.stride[2]) return false;
  oz = vz / g.stride[0];
  oy = vy / g.stride[1];
  ox = vx / g.stride[2];
  return oz < g.out_shape[0] && oy < g.out_shape[1] && ox < g.out_shape[2];
}

// The candidate pass of a strided layer as a job that can ride in another launch (blocks == 0: none): it only needs the input
// site list, which exists as soon as the PREVIOUS stage's sites are numbered -- long before the layer itself runs.  The plan
// lets it ride in the launch that fills the previous strided layer's table (or builds stage 0's submanifold table): one launch
// less per strided layer on a chain of dependent ~7 us launches.
struct RbCandJob {
  const int4* coords;
  const int* n_ptr;
  int cap_in;
  RbGeom g;
  V3dHash h;
  unsigned* first_ticket;
  int* cand_slot;
  int* overflow;
  int* overflow_any;
  int blocks;
};

__device__ __forceinline__ void rb_candidates_body(const int4* __restrict__ coords, const int* __restrict__ n_ptr, int cap_in,
                                                   const RbGeom& g, const V3dHash& h, unsigned* __restrict__ first_ticket,
                                                   int* __restrict__ cand_slot, int* __restrict__ overflow,
                                                   int* __restrict__ overflow_any, int block, int nblocks) {
  const long long nt = (long long)min(*n_ptr, cap_in) * g.Kc;
  for (long long t = (long long)block * V3D_BLOCK + threadIdx.x; t < nt; t += (long long)nblocks * V3D_BLOCK) {
    const int i = (int)(t / g.Kc), j = (int)(t % g.Kc);
    const int4 c = coords[i];
    int k, oz, oy, ox, s = -1;
    if (rb_ticket(c, j, g, k, oz, oy, ox)) {
      s = (unsigned)c.x < (unsigned)RB_MAX_BATCH ? v3d_site_insert(h, rb_key(c.x, oz, oy, ox, g.out_shape), V3D_SITE_NO_ROW) : -1;  // numbered by the emit pass
      if (s >= 0) {
        atomicMin(&first_ticket[s], (unsigned)t);
        s |= k << RB_SLOT_BITS;
      } else {
        atomicExch(overflow, 1);
        if (overflow_any) atomicExch(overflow_any, 1);
      }
    }
    cand_slot[t] = s;
  }
}

__device__ __forceinline__ bool rb_is_first(const int* cand_slot, const unsigned* first_ticket, long long t,
                                            long long nt) {
  if (t >= nt) return false;
  const int s = cand_slot[t];
  return s != -1 && first_ticket[s & RB_SLOT_MASK] == (unsigned)t;
}

#define RB_PER_THREAD (V3D_SCAN_CHUNK / V3D_BLOCK)  // 8 consecutive tickets per thread
static_assert(RB_PER_THREAD == 8, "rb_first_flags loads two int4");
// bit r set <=> ticket t0 + r is the first toucher of its output slot.  t0 is a multiple of 8 -> 32-byte aligned loads.
__device__ __forceinline__ unsigned rb_first_flags(const int* __restrict__ cand_slot, const unsigned* __restrict__ first_ticket,
                                                   long long t0, long long nt) {
  if (t0 >= nt) return 0u;
  const int4 a = *reinterpret_cast<const int4*>(cand_slot + t0);  // cand_slot is sized to whole chunks
  const int4 b = *reinterpret_cast<const int4*>(cand_slot + t0 + 4);
  const int s[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
  unsigned flags = 0u;
#pragma unroll
  for (int r = 0; r < 8; r++)
    if (t0 + r < nt && s[r] != -1 && first_ticket[s[r] & RB_SLOT_MASK] == (unsigned)(t0 + r)) flags |= 1u << r;
  return flags;
}

// count + scan + emit in ONE launch: a single-pass scan over 2 048-ticket chunks through published counts.
// Every block counts its first-toucher tickets and PUBLISHES the count at once (agent-scope atomic store into a slot that reads
// -1 at launch: part of the frame's 0xFF fill, no fence needed), then adds up the counts of ALL its live predecessors -- every
// thread its share, spinning on slots that still read -1 -- and emits its output rows at prefix + local rank.  Workgroups are
// dispatched in index order and publish before they wait for anything, so every wait is on a block that is already running
// or done; the counts appear within ~2 us of the launch, so the prefix is ONE round of coherent loads (a wave-0 look-back over
// inclusive prefixes took up to three dependent rounds for the late blocks: 12 us per launch instead of ~8).  The
// highest-index block adds up the total (clipped to cap_out).  Replaces a count+scan launch and an emit launch.
//   chunk_counts[n_chunks], all -1 at launch.
// (a body: the launch of its own below, or the first blocks of ANOTHER launch -- RbRider; `b` of `nblocks` chunk blocks, the first
// V3D_BLOCK threads of the workgroup, RB_SCAN_LDS bytes of its LDS)
#define RB_SCAN_LDS (32 + 2 * V3D_SCAN_CHUNK)
__device__ __forceinline__ void rb_scan_emit_body(const int4* __restrict__ coords, const int* __restrict__ n_ptr, int cap_in, const RbGeom& g,
                                                  const int* __restrict__ cand_slot, const unsigned* __restrict__ first_ticket,
                                                  int* __restrict__ chunk_counts, int cap_out, int4* __restrict__ coords_out,
                                                  int* __restrict__ vals, int* __restrict__ n_out, int* __restrict__ overflow,
                                                  int* __restrict__ overflow_any, int* __restrict__ nbr_init, const V3dHash& out_hash,
                                                  const int b, const int nblocks, unsigned char* __restrict__ lds_raw) {
  int* lds = reinterpret_cast<int*>(lds_raw);
  int* s_part = lds + 4;
  unsigned short* list = reinterpret_cast<unsigned short*>(lds_raw + 32);
  const long long nt = (long long)min(*n_ptr, cap_in) * g.Kc;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const long long base = (long long)b * V3D_SCAN_CHUNK;
  const bool live = base < nt, last = b == nblocks - 1;
  if (!live && !last) return;  // nothing to count, and nobody looks at a dead block's slots
  const long long t0 = base + (long long)tid * RB_PER_THREAD;
  const unsigned flags = live ? rb_first_flags(cand_slot, first_ticket, t0, nt) : 0u;
  const int mine = __popc(flags);
  int incl = mine;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int t = __shfl_up(incl, off);
    if (lane >= off) incl += t;
  }
  if (lane == 63) lds[w] = incl;
  __syncthreads();
  const int cnt = lds[0] + lds[1] + lds[2] + lds[3];
  if (tid == 0 && live) v3d_publish_count(chunk_counts + b, cnt);
  // Emit runs one output per thread: the chunk's first-toucher tickets are compacted into an LDS list at their local ranks (a
  // thread that walked its own up to 8 flagged tickets one after the other put 8 dependent look-up chains in a row -- in the
  // compact ticket space most of a thread's 8 tickets are live), and thread q emits local rank q.  Its loads do not depend on the
  // chunk's global prefix, so the first round's are ISSUED here, in front of the wait for the predecessors' counts.
  {
    int lr = incl - mine;
    for (int i = 0; i < w; i++) lr += lds[i];
    unsigned f = flags;
    while (f) {
      const int r = __ffs(f) - 1;
      f &= f - 1;
      list[lr++] = (unsigned short)(tid * RB_PER_THREAD + r);
    }
  }
  __syncthreads();
  int word0 = -1;
  int4 c0 = make_int4(0, 0, 0, 0);
  if (tid < cnt) {
    const long long t = base + list[tid];
    word0 = cand_slot[t];
    c0 = coords[(int)(t / g.Kc)];
  }
  asm volatile("" ::: "memory");  // (the loads above stay in front of the spin below)
  int part = 0;
  {
    const int n_live = (int)((nt + V3D_SCAN_CHUNK - 1) / V3D_SCAN_CHUNK);
    for (int i = tid; i < min(b, n_live); i += V3D_BLOCK) part += v3d_wait_count(chunk_counts + i);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off);
    if (lane == 0) s_part[w] = part;
  }
  __syncthreads();
  const int prefix = s_part[0] + s_part[1] + s_part[2] + s_part[3];
  if (last && tid == 0) {
    const int total = prefix + cnt;
    if (total > cap_out) {
      atomicExch(overflow, 1);
      if (overflow_any) atomicExch(overflow_any, 1);
    }
    *n_out = min(total, cap_out);
  }
  if (!live) return;
  if (nbr_init) {
    // the columns of the sites this block creates (ranks prefix .. prefix + cnt - 1) start as "no input" -- the fill kernel behind
    // this launch writes the live entries -- so the table needs no per-frame -1 fill over its whole K x capacity extent.  All
    // threads share the work, lanes along the ranks (the emitting threads alone would do 27 stores per site, one after the other).
    const int c0r = min(prefix, cap_out), c1r = min(prefix + cnt, cap_out), w_ = c1r - c0r;
    for (int kk = tid >> 5; kk < g.K; kk += V3D_BLOCK / 32)  // 8 table rows at a time, 32 lanes along the ranks: no division
      for (int r = tid & 31; r < w_; r += 32) nbr_init[(size_t)kk * cap_out + c0r + r] = -1;
  }
  for (int q = tid; q < cnt; q += V3D_BLOCK) {
    const int rank = prefix + q;
    if (rank >= cap_out) break;
    int word = word0;
    int4 c = c0;
    if (q != tid) {
      const long long t = base + list[q];
      word = cand_slot[t];
      c = coords[(int)(t / g.Kc)];
    }
    const int s = word & RB_SLOT_MASK;
    int oz, oy, ox;
    rb_candidate(c, (int)((unsigned)word >> RB_SLOT_BITS), g, oz, oy, ox);
    coords_out[rank] = make_int4(c.x, oz, oy, ox);
    vals[s] = rank;
    v3d_site_set_row(out_hash, s, rb_key(c.x, oz, oy, ox, g.out_shape), (unsigned)rank);  // look-ups read the row with the key
  }
}

// nbr[k][out] = i for every live ticket (nbr pre-filled with -1).  The same launch can carry the submanifold table
// of the OUTPUT sites (the next layer's rulebook): both only need what rb_emit left behind (coords_out, vals), so
// blocks [0, fill_blocks) fill and the remaining (row block, offset) pairs look up neighbours -- one launch saved
// per stage.
// (a body, block `bid` of fill_blocks + subm_blocks + job.blocks: see rb_scan_emit_body)
__device__ __forceinline__ void rb_fill_nbr_body(const int* __restrict__ n_ptr, int cap_in, int K, const int* __restrict__ cand_slot,
                                                 const int* __restrict__ vals, int cap_out, int* __restrict__ nbr, int fill_blocks,
                                                 const int4* __restrict__ coords_out, const int* __restrict__ n_out_ptr, const RbGeom& sg,
                                                 const V3dHash& sh, int* __restrict__ subm_nbr, int subm_blocks, const RbCandJob& job,
                                                 const int bid) {
  if (bid < fill_blocks) {
    const long long nt = (long long)min(*n_ptr, cap_in) * K;  // K: tickets per input row here (RbGeom::Kc)
    for (long long t = (long long)bid * V3D_BLOCK + threadIdx.x; t < nt; t += (long long)fill_blocks * V3D_BLOCK) {
      const int word = cand_slot[t];
      if (word == -1) continue;
      const int o = vals[word & RB_SLOT_MASK];
      if (o < 0) continue;  // clipped by cap_out
      nbr[(size_t)((unsigned)word >> RB_SLOT_BITS) * cap_out + o] = (int)(t / K);
    }
    return;
  }
  if (bid >= fill_blocks + subm_blocks) {  // the NEXT strided layer's candidate pass (its inputs = these output sites)
    rb_candidates_body(job.coords, job.n_ptr, job.cap_in, job.g, job.h, job.first_ticket, job.cand_slot, job.overflow,
                       job.overflow_any, bid - fill_blocks - subm_blocks, job.blocks);
    return;
  }
  // ---- submanifold table of the output sites
  const int idx = bid - fill_blocks;
  rb_subm_block(coords_out, min(*n_out_ptr, cap_out), cap_out, sg, sh, subm_nbr, idx);
}


// ---------------------------------------------------------------------------------- riders
// The scan + emit step of a strided layer's rulebook as a JOB for the first `blocks` workgroups of ANOTHER launch.  The rulebooks
// depend on coordinates only, so the chain of a frame can run AHEAD of the convolutions: a sparse layer's launch carries the next
// pending scan, and the step's ~9 us of dependent round trips (flags -> counts -> predecessors' counts -> emit: a launch of its own on
// a chain of 38) disappear under the layer.  The rider blocks come FIRST in the grid: they are dispatched first, the chained scan's
// "every block I wait for is already running" holds among them, and they run beside the host's own workgroups.  A rider block uses
// the first V3D_BLOCK threads of its workgroup (the others leave at once: a barrier counts live waves only) and RB_SCAN_LDS bytes of
// the host kernel's LDS.  (The FILL step -- the table, the output sites' submanifold table, the next layer's candidate pass: thousands
// of one-look-up workgroups -- stays a launch of its own: carried the same way it ADDED its duration to the host's, 8.3 -> 19.2 us,
// and its look-up code took the small hosts from 40-58 to 73 VGPRs.)
struct RbScanJob {
  int blocks;  // workgroups at the front of the grid = 2 048-ticket chunks of the capacity (0: no job)
  int cap_in, cap_out;
  const int4* coords;
  const int* n_ptr;
  RbGeom g;
  const int* cand_slot;
  const unsigned* first_ticket;
  int* chunk_counts;
  int4* coords_out;
  int* vals;
  int* n_out;
  int* overflow;
  int* overflow_any;
  int* nbr_init;  // nullable: the table whose live columns the emit pass initialises
  V3dHash out_hash;
};
// the fill step (always a launch): the arguments of rb_fill_nbr_kernel
struct RbFillJob {
  int blocks;
  const int* n_ptr;
  int cap_in, Kc;
  const int* cand_slot;
  const int* vals;
  int cap_out;
  int* nbr;
  int fill_blocks, subm_blocks;
  const int4* coords_out;
  const int* n_out;
  RbGeom sg;
  V3dHash sh;
  int* subm_nbr;
  RbCandJob job;
};
struct RbStep {
  int kind;  // 1: scan, 2: fill
  RbScanJob scan;
  RbFillJob fill;
};

__device__ __forceinline__ void rb_rider_run(const RbScanJob& r, const int bid, unsigned char* __restrict__ lds) {
  rb_scan_emit_body(r.coords, r.n_ptr, r.cap_in, r.g, r.cand_slot, r.first_ticket, r.chunk_counts, r.cap_out, r.coords_out, r.vals, r.n_out,
                    r.overflow, r.overflow_any, r.nbr_init, r.out_hash, bid, r.blocks, lds);
}
