// PersSampler on gfx950: ray / octree intersection, perspective-warped ray marching, edge samples and
// occupancy bookkeeping.  Semantics follow PtsSampler/PersSampler.cu:21-680 of the reference (cited per
// kernel); the structure does not: two-phase count / device-side scan / fill with ray-ordered segments
// and no host read-back, wave64 blocks, LDS-resident DFS stacks.
#include "rows_dev.h"

#include <stdlib.h>


// ---------------------------------------------------------------------------------------------------
// Slab test, PersSampler.cu:21-51.
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ void f2n_slab(const float* o, const float* d, const float* c, float side, float& near_,
                                         float& far_) {
  // Straight-line form of the reference's three-way branch per axis: both quotients are always formed and the result is
  // selected (the same IEEE operations on the selected path, so the same bits).  The rays of a wave differ in the signs
  // of d, so the branchy form executed the positive AND the negative arm of every axis -- twelve divisions per test
  // instead of six -- behind exec-mask juggling that a lone wave pays ~4 cycles per instruction for.
  float lo[3], hi[3];
  float hf = side * .5f;
#pragma unroll
  for (int i = 0; i < 3; i++) {
    const bool par = d[i] < 1e-6f && d[i] > -1e-6f, pos = d[i] > 0;  // ray constants: hoisted out of the DFS loop
    const float cm = c[i] - hf, cp = c[i] + hf;
    const float qm = (cm - o[i]) / d[i], qp = (cp - o[i]) / d[i];
    const bool inside = o[i] > cm && o[i] < cp;
    lo[i] = par ? (inside ? -1e6f : 1e6f) : (pos ? qm : qp);
    hi[i] = par ? (inside ? 1e6f : -1e6f) : (pos ? qp : qm);
  }
  near_ = fmaxf(near_, fmaxf(lo[0], fmaxf(lo[1], lo[2])));
  far_ = fminf(far_, fminf(hi[0], fminf(hi[1], hi[2])));
}

// ---------------------------------------------------------------------------------------------------
// Front-to-back DFS over the octree (PersSampler.cu:53-152), 8 lanes per ray (one per child slot of the node being
// expanded).  A node visit loads the 8 child indices with one coalesced 32-byte read, slab-tests all 8 children in parallel and
// turns the results into masks with one wave ballot; leaf hits in front of the first interior hit are emitted at
// once (rank = popcount of the mask below), the first interior hit is descended into, the remaining positions are
// parked on an LDS stack as a bit mask.  Same slab test, same child order, same cap as
// of the reference's one-thread-per-ray loop (and therefore the same output, bit for bit), but ~5x fewer dependent
// memory round trips per ray and 8x more waves in flight to hide them (measured 0.63 -> 0.09 ms for 8192 rays).
// MODE 0: count only.  MODE 1: fill a compact, ray-ordered list (segments from f2n_segment_scan).
// MODE 2: single pass into fixed-stride per-ray segments [ray*max_hits, ray*max_hits + cnt): no count pass, no scan.
// MODE 3: REPAIR of a MODE-2 result that was computed SPECULATIVELY, i.e. while (or before) f2n_oct_update_stats_ex of epoch
//         >= spec_epoch was killing leaves (trans_idx -> -1: the only thing a stat update changes in the tree, and it is
//         monotone).  A ray's list is still the list of the updated tree unless it holds a node that died since: every leaf it
//         accepted is then still alive, every leaf it skipped as dead still dead, and nothing else in the walk depends on
//         trans_idx.  The 8 lanes of a ray scan its list for such nodes (died_at[node] >= spec_epoch); only flagged rays (and
//         rays at the max_hits cap, whose truncation point may move) walk the tree again, overwriting their slots.  Whole
//         launch exits at once when no leaf died at all (*death_epoch < spec_epoch: the common case).
// ---------------------------------------------------------------------------------------------------
#define F2N_COOP_RAYS_PER_BLOCK 32  // 256 threads
// Parked siblings per ray.  A ray crosses at most 4 of a node's 8 children (three mid-planes), so behind the child that is
// descended into at most 3 are parked per level of the current path: 72 entries cover paths of 24 levels, which is the
// deepest tree the reference's own 48-int stack (2 ints per level, PersSampler.cu:7,70) can walk without overrunning it.
#define F2N_COOP_STACK 72
// (repair_from values of the tail repair, F2N_REPAIR_NONE / F2N_REPAIR_FULL: include/f2n_abi.h)
// LDSREC (MODE 2 only): the child records of every INTERIOR node -- the only ones a walk ever expands -- are copied into LDS
// when the block starts (interior_nodes[r] = node index of the r-th interior node, rank_of[node] = its r; both change only
// when the tree is rebuilt) and the walk then never leaves the CU: `cur` and the parked interior entries hold RANKS instead of
// node indices (an interior node is never emitted, so its index is not needed).  Why: a walk is a chain of ~30-40 dependent
// record reads per ray with one wave per SIMD -- 0.06 ms for 8192 rays alone, but 0.36 ms underneath the hash gather of the
// step it is prefetched under, where every one of those reads queues behind the gather's L2 traffic
// (profiles/r03_fresh_timeline.txt).  The bulk copy is bandwidth-bound (256 B per interior node and block, all loads
// independent) and barely notices.  Same records, same order of tests: the output is bit-identical to the global-memory walk.
template <int MODE, bool LDSREC = false>
__global__ __launch_bounds__(256) void oct_intersect_coop_kernel(
    int n_rays, int max_hits, const uint8_t* __restrict__ search_order, const float* __restrict__ rays_o,
    const float* __restrict__ rays_d, float g_near, float g_far, const F2nTreeNode* __restrict__ nodes,
    const int32_t* __restrict__ oct_start_end, int32_t* __restrict__ hit_counts, int32_t* __restrict__ oct_idx,
    float* __restrict__ oct_near_far, int32_t* __restrict__ se_out, int32_t* __restrict__ total,
    int32_t* __restrict__ oct_trans, const F2nChildInfo* __restrict__ child_blocks, const int32_t* __restrict__ died_at,
    int spec_epoch, const int32_t* __restrict__ death_epoch, int32_t* __restrict__ repair_flags, int32_t* __restrict__ n_repaired,
    const int32_t* __restrict__ interior_nodes = nullptr, const int32_t* __restrict__ rank_of = nullptr, int n_interior = 0,
    const int32_t* __restrict__ n_flagged = nullptr) {
  static_assert(!LDSREC || MODE == 2, "the LDS-resident walk exists for the single-pass variant");
  if (MODE == 3) {
    if (*death_epoch < spec_epoch) return;  // no leaf died since the speculative walk: every list stands (grid-uniform)
    // flagged variant (behind f2n_oct_list_repair): only the rays it marked F2N_REPAIR_FULL in repair_flags are walked again
    if (n_flagged != nullptr && *n_flagged == 0) return;
  }
  extern __shared__ float4_t s_rec[];  // LDSREC: [n_interior][8 slots][2] = the F2nChildInfo records, pad = rank of an interior child
  if (LDSREC) {
    // four records per thread and round: node index, record, rank of an interior child are three DEPENDENT reads -- issued
    // for all four before any is waited for, so a round costs three round trips (each several microseconds underneath the
    // gather), not twelve
    const int n_rec = n_interior * 8;
    for (int base = 0; base < n_rec; base += 4 * 256) {
      int u[4], rk[4];
      float4_t cs[4], meta[4];
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const int i = base + q * 256 + (int) threadIdx.x;
        u[q] = interior_nodes[min(i, n_rec - 1) >> 3];
      }
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const int i = min(base + q * 256 + (int) threadIdx.x, n_rec - 1);
        const float4_t* rec = (const float4_t*) (child_blocks + (size_t) u[q] * 8 + (i & 7));
        cs[q] = rec[0];
        meta[q] = rec[1];
      }
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const int ch = __float_as_int(meta[q][0]);
        rk[q] = (ch >= 0 && __float_as_int(meta[q][2]) != 0) ? rank_of[ch] : 0;
      }
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const int i = base + q * 256 + (int) threadIdx.x;
        if (i < n_rec) {
          meta[q][3] = __int_as_float(rk[q]);
          s_rec[2 * i] = cs[q];
          s_rec[2 * i + 1] = meta[q];
        }
      }
    }
    __syncthreads();
  }
  // Work stack of a ray: every hit sibling behind the first interior hit of an expanded node is parked here, nearest on
  // top -- interior nodes as (index >= 0), valid leaves as (~index, near, far, trans) to be emitted when popped.  A node
  // is therefore expanded exactly once (parking only a "remaining siblings" mask meant re-reading and re-testing all
  // eight children of a node once per interior child: ~3x the memory round trips in a deep, converged tree).
  __shared__ int s_node[F2N_COOP_STACK][F2N_COOP_RAYS_PER_BLOCK];
  __shared__ int s_tr[F2N_COOP_STACK][F2N_COOP_RAYS_PER_BLOCK];
  __shared__ float s_near[F2N_COOP_STACK][F2N_COOP_RAYS_PER_BLOCK];
  __shared__ float s_far[F2N_COOP_STACK][F2N_COOP_RAYS_PER_BLOCK];
  const int tid = threadIdx.x, k = tid & 7, grp = tid >> 3;
  const int shift = ((tid & 63) >> 3) * 8;  // position of this group's 8 bits inside a wave ballot
  const int ray_raw = blockIdx.x * F2N_COOP_RAYS_PER_BLOCK + grp;
  const bool in_range = ray_raw < n_rays;
  const int ray = in_range ? ray_raw : 0;
  const float o[3] = {rays_o[3 * ray], rays_o[3 * ray + 1], rays_o[3 * ray + 2]};
  const float d[3] = {rays_d[3 * ray], rays_d[3 * ray + 1], rays_d[3 * ray + 2]};
  int limit = in_range ? max_hits : 0, base = 0;
  if (MODE == 1) {
    base = oct_start_end[2 * ray];
    limit = in_range ? oct_start_end[2 * ray + 1] - base : 0;
  }
  if (MODE >= 2) base = ray * max_hits;
  int old_cnt = 0;
  bool redo = false;
  if (MODE == 3 && n_flagged != nullptr) {
    if (in_range) old_cnt = se_out[2 * ray + 1] - se_out[2 * ray];
    redo = in_range && repair_flags[ray] == F2N_REPAIR_FULL;
    if (!redo) limit = 0;
  } else if (MODE == 3) {
    bool dead_hit = false;
    if (in_range) {
      old_cnt = se_out[2 * ray + 1] - se_out[2 * ray];
      for (int e = k; e < old_cnt; e += 8) dead_hit |= died_at[oct_idx[base + e]] >= spec_epoch;
    }
    redo = in_range && ((((__ballot(dead_hit) >> shift) & 0xffull) != 0ull) || old_cnt >= max_hits);
    if (!redo) limit = 0;
  }
  const int octant = (int(d[0] > 0.f) << 2) | (int(d[1] > 0.f) << 1) | int(d[2] > 0.f);
  const int my_slot = search_order[octant * 8 + k];  // the child slot this lane tests at every node

  int cnt = 0, sp = -1;
  int cur = -1;        // node to expand next (-1: pop from the stack)
  bool active = limit > 0;
  if (active) {        // the root is tested on its own box first (:93-95); a childless root is itself the only leaf
    float near_ = g_near, far_ = g_far;
    f2n_slab(o, d, nodes[0].center, nodes[0].side_len, near_, far_);
    bool any_child = false;
#pragma unroll
    for (int c = 0; c < 8; c++) any_child |= nodes[0].childs[c] >= 0;
    if (!(near_ < far_)) {
      active = false;
    } else if (!any_child) {
      if (nodes[0].trans_idx >= 0) {
        if (MODE != 0 && k == 0) {
          oct_idx[base] = 0;
          oct_near_far[2 * base] = near_;
          oct_near_far[2 * base + 1] = far_;
          if (oct_trans != nullptr) oct_trans[base] = nodes[0].trans_idx;
        }
        cnt = 1;
      }
      active = false;
    } else {
      cur = 0;
    }
  }
  while (__any(active)) {
    // ---- pick the node to expand; parked leaves are emitted on the way ----
    if (active && cur < 0) {
      if (sp < 0) {
        active = false;
      } else {
        const int e = s_node[sp][grp];
        if (e >= 0) {
          cur = e;
        } else {  // a parked leaf: it is the nearest thing left
          if (MODE != 0 && k == 0) {
            oct_idx[base + cnt] = ~e;
            oct_near_far[2 * (base + cnt)] = s_near[sp][grp];
            oct_near_far[2 * (base + cnt) + 1] = s_far[sp][grp];
            if (oct_trans != nullptr) oct_trans[base + cnt] = s_tr[sp][grp];
          }
          cnt++;
          if (cnt >= limit) active = false;
        }
        sp--;
      }
    }
    // ---- test this lane's child of `cur` ----
    bool hit = false, interior = false, valid_leaf = false;
    int child = -1, child_trans = -1;
    int child_rank = -1;  // LDSREC: what stands for an interior child on the stack and in `cur`
    float near_ = g_near, far_ = g_far;
    const bool expanding = active && cur >= 0;
    if (expanding) {
      if (LDSREC || child_blocks != nullptr) {  // one 32-byte record per child slot: no dependent second read
        float4_t cs, meta;
        if (LDSREC) {
          cs = s_rec[2 * (cur * 8 + my_slot)];
          meta = s_rec[2 * (cur * 8 + my_slot) + 1];
        } else {
          const float4_t* rec = (const float4_t*) (child_blocks + (size_t) cur * 8 + my_slot);
          cs = rec[0];
          meta = rec[1];
        }
        child = __float_as_int(meta[0]);
        if (child >= 0) {
          const float cc[3] = {cs[0], cs[1], cs[2]};
          f2n_slab(o, d, cc, cs[3], near_, far_);
          hit = near_ < far_;
          if (hit) {
            child_trans = __float_as_int(meta[1]);
            interior = __float_as_int(meta[2]) != 0;
            valid_leaf = !interior && child_trans >= 0;
            if (LDSREC) child_rank = __float_as_int(meta[3]);
          }
        }
      } else {
        child = nodes[cur].childs[my_slot];
        if (child >= 0) {
          const F2nTreeNode* nd = nodes + child;
          f2n_slab(o, d, nd->center, nd->side_len, near_, far_);
          hit = near_ < far_;
          if (hit) {
#pragma unroll
            for (int c = 0; c < 8; c++) interior |= nd->childs[c] >= 0;
            child_trans = nd->trans_idx;
            valid_leaf = !interior && child_trans >= 0;
          }
        }
      }
    }
    const unsigned long long b_int = __ballot(hit && interior);
    const unsigned long long b_leaf = __ballot(hit && valid_leaf);
    if (expanding) {
      const int m_int = (int) ((b_int >> shift) & 0xffull);
      const int m_leaf = (int) ((b_leaf >> shift) & 0xffull);
      const int k_int = m_int ? __ffs(m_int) - 1 : 8;          // first interior hit (order position)
      const int front = m_leaf & ((1 << k_int) - 1);            // leaves in front of it: emitted now
      const int rank = __popc(front & ((1 << k) - 1));
      if (((front >> k) & 1) && cnt + rank < limit) {
        if (MODE != 0) {
          oct_idx[base + cnt + rank] = child;
          oct_near_far[2 * (base + cnt + rank)] = near_;
          oct_near_far[2 * (base + cnt + rank) + 1] = far_;
          if (oct_trans != nullptr) oct_trans[base + cnt + rank] = child_trans;  // saves the march a dependent node read
        }
      }
      cnt = min(limit, cnt + __popc(front));
      if (cnt >= limit) {
        active = false;
      } else if (k_int < 8) {
        // park every hit behind the first interior child, farthest first so that the nearest ends on top
        int rest = (m_int | m_leaf) & ~((2 << k_int) - 1);
        int n_rest = __popc(rest);
        while (sp + n_rest >= F2N_COOP_STACK) {  // unreachable for trees of <= 24 levels (see F2N_COOP_STACK); memory safety only
          rest &= ~(1 << (31 - __clz(rest)));
          n_rest--;
        }
        if ((rest >> k) & 1) {
          const int pos = sp + 1 + __popc(rest >> (k + 1));
          s_node[pos][grp] = interior ? (LDSREC ? child_rank : child) : ~child;
          s_near[pos][grp] = near_;
          s_far[pos][grp] = far_;
          s_tr[pos][grp] = child_trans;
        }
        sp += n_rest;
        cur = __shfl(LDSREC ? child_rank : child, (tid & 56) + k_int);  // the interior child's node index / rank (lane k_int)
      } else {
        cur = -1;
      }
    }
  }
  if (in_range && k == 0) {
    if (MODE == 0) hit_counts[ray] = cnt;
    if (MODE == 2 || (MODE == 3 && redo)) {
      se_out[2 * ray] = base;
      se_out[2 * ray + 1] = base + cnt;
    }
    if (MODE == 3) {
      if (n_flagged == nullptr) repair_flags[ray] = redo ? 1 : 0;
      else if (redo) repair_flags[ray] = 0;  // (repair_from semantics: march this ray again from its origin)
    }
  }
  if (MODE >= 2) {  // wave-level hit total, one atomic per wave (repair: the change of the total)
    int s = (in_range && k == 0) ? (MODE == 3 ? (redo ? cnt - old_cnt : 0) : cnt) : 0;
    int r = (MODE == 3 && in_range && k == 0 && redo) ? 1 : 0;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      s += __shfl_xor(s, off);
      if (MODE == 3) r += __shfl_xor(r, off);
    }
    if ((tid & 63) == 0 && s != 0) atomicAdd(total, s);
    if (MODE == 3 && (tid & 63) == 0 && r != 0 && n_repaired != nullptr) atomicAdd(n_repaired, r);
  }
}

// ---------------------------------------------------------------------------------------------------
// Ray-ordered segment allocation.  One 1024-thread workgroup walks the count array in chunks of
// 1024 x ITEMS with a running carry; wave-level inclusive scans use DPP-free __shfl_up on 64 lanes.
// ---------------------------------------------------------------------------------------------------
#define F2N_SCAN_THREADS 1024
#define F2N_SCAN_ITEMS 4
__device__ __forceinline__ void f2n_segment_scan_block(int n, const int32_t* __restrict__ counts, int32_t* __restrict__ start_end,
                                                       int32_t* __restrict__ total, int32_t* __restrict__ mirror,
                                                       const int32_t* __restrict__ also, int n_also) {
  __shared__ int s_wave[F2N_SCAN_THREADS / F2N_WAVE];
  __shared__ int s_carry;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) s_carry = 0;
  __syncthreads();
  for (int chunk = 0; chunk < n; chunk += F2N_SCAN_THREADS * F2N_SCAN_ITEMS) {
    const int i0 = chunk + tid * F2N_SCAN_ITEMS;
    int v[F2N_SCAN_ITEMS];
    int local = 0;
#pragma unroll
    for (int k = 0; k < F2N_SCAN_ITEMS; k++) {
      v[k] = (i0 + k < n) ? counts[i0 + k] : 0;
      local += v[k];
    }
    int incl = local;  // inclusive scan of per-thread sums across the wave
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      int up = __shfl_up(incl, off);
      if (lane >= off) incl += up;
    }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    int wave_off = 0;
    for (int w = 0; w < wave; w++) wave_off += s_wave[w];
    int run = s_carry + wave_off + incl - local;
#pragma unroll
    for (int k = 0; k < F2N_SCAN_ITEMS; k++) {
      if (i0 + k < n) {
        start_end[2 * (i0 + k)] = run;
        run += v[k];
        start_end[2 * (i0 + k) + 1] = run;
      }
    }
    __syncthreads();
    if (tid == F2N_SCAN_THREADS - 1) s_carry = run;  // last thread holds the chunk's inclusive total
    __syncthreads();
  }
  if (tid == 0) {
    total[0] = s_carry;
    // the host's copy of the count(s), written where the count is produced: `mirror` is mapped host memory (a separate
    // device-to-host copy launch cost one ~5 us dependent boundary on the queue of every scan)
    if (mirror != nullptr) {
      for (int k = 0; k < n_also; k++) mirror[k] = also[k];
      mirror[n_also] = s_carry;
    }
  }
}

__global__ __launch_bounds__(F2N_SCAN_THREADS) void segment_scan_kernel(int n, const int32_t* __restrict__ counts,
                                                                        int32_t* __restrict__ start_end,
                                                                        int32_t* __restrict__ total, int32_t* __restrict__ mirror,
                                                                        const int32_t* __restrict__ also, int n_also) {
  f2n_segment_scan_block(n, counts, start_end, total, mirror, also, n_also);
}

// ---------------------------------------------------------------------------------------------------
// Repair of speculatively walked leaf lists WITHOUT a second walk (speculative sampling, round 4).  A stat update changes
// one thing in the tree -- trans_idx -> -1 of the leaves that die -- and the walk (PersSampler.cu:53-152) reads trans_idx only
// to decide whether a leaf it has reached is listed: the list a fresh walk would produce is the old list minus its dead
// entries, in the same order with the same near / far, PROVIDED the old list was not cut at max_hits (removing entries would
// then make room for leaves the old walk never listed: those rays are flagged F2N_REPAIR_FULL and walked again).
// One 16-lane row per ray: entries are tested 16 at a time (died_at[node] >= spec_epoch), survivors behind the first dead
// entry slide forward in place.  repair_from[ray] = position of the first removed entry if the march got that far
// (reached[ray], see ray_march_kernel<2, true>) -- the march then resumes from the state it recorded there -- else -1.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void oct_list_repair_kernel(int n_rays, int max_hits, int32_t* __restrict__ oct_start_end,
                                                              int32_t* oct_idx, float* oct_near_far, int32_t* oct_trans,
                                                              int32_t* __restrict__ total, const int32_t* __restrict__ died_at,
                                                              int spec_epoch, const int32_t* __restrict__ death_epoch,
                                                              const int32_t* __restrict__ reached, int32_t* __restrict__ repair_from,
                                                              int32_t* __restrict__ n_repaired, int32_t* __restrict__ n_full) {
  if (*death_epoch < spec_epoch) return;  // no leaf died since the speculative walk (grid-uniform; repair_from is not read then)
  const int c = threadIdx.x & 15;
  const int shift = (threadIdx.x & 48);  // position of this row's 16 bits inside a wave ballot
  const int ray = blockIdx.x * 16 + (threadIdx.x >> 4);
  if (ray >= n_rays) return;  // whole rows leave together
  const size_t base = (size_t) ray * max_hits;
  const int cnt = oct_start_end[2 * ray + 1] - oct_start_end[2 * ray];
  int w = 0, p_min = -1;
  const bool capped = cnt >= max_hits;
  for (int e0 = 0; e0 < cnt; e0 += 16) {
    const int e = e0 + c;
    const bool in = e < cnt;
    const int node = in ? oct_idx[base + e] : 0;
    const bool dead = in && died_at[node] >= spec_epoch;
    const bool keep = in && !dead;
    const unsigned m_dead = (unsigned) ((__ballot(dead) >> shift) & 0xffffull);
    const unsigned m_keep = (unsigned) ((__ballot(keep) >> shift) & 0xffffull);
    if (p_min < 0 && m_dead != 0u) p_min = e0 + __ffs(m_dead) - 1;
    if (capped) {
      if (p_min >= 0) break;  // (row-uniform) a cut list that lost an entry: walked again
      continue;
    }
    if (p_min >= 0 && keep) {
      const int dst = w + __popc(m_keep & ((1u << c) - 1u));
      if (dst != e) {  // every lane of the row has loaded before any stores: dst <= e, and slots >= e0 are read in this round
        const float nr = oct_near_far[2 * (base + e)], fr = oct_near_far[2 * (base + e) + 1];
        const int tr = oct_trans != nullptr ? oct_trans[base + e] : 0;
        oct_idx[base + dst] = node;
        oct_near_far[2 * (base + dst)] = nr;
        oct_near_far[2 * (base + dst) + 1] = fr;
        if (oct_trans != nullptr) oct_trans[base + dst] = tr;
      }
    }
    w += __popc(m_keep);
  }
  if (c != 0) return;
  int from = F2N_REPAIR_NONE;
  if (p_min >= 0) {
    if (capped) {
      from = F2N_REPAIR_FULL;
      atomicAdd(n_full, 1);
    } else {
      oct_start_end[2 * ray + 1] = oct_start_end[2 * ray] + w;
      atomicAdd(total, w - cnt);
      if (p_min <= reached[ray]) from = p_min;
    }
    if (from != F2N_REPAIR_NONE && n_repaired != nullptr) atomicAdd(n_repaired, 1);
  }
  repair_from[ray] = from;
}

// ---------------------------------------------------------------------------------------------------
// Perspective-warped ray marching (PersSampler.cu:189-314), sixteen lanes per ray.
//
// A march step is a strictly sequential chain (the step length comes from the warp Jacobian at the current point):
// a one-ray-per-lane kernel runs 8192 rays as 128 lonely waves, each issuing ~1200 dependent-ish instructions (36
// IEEE divisions) per step.  The Jacobian is a 12-term sum over the leaf's 12 camera projections in Eigen's tree
// ((e0+(e1+e2)) + (e3+(e4+e5))) + ((e6+(e7+e8)) + (e9+(e10+e11))), which maps onto DPP exchanges bit for bit.
// History: 1 lane/ray 0.73 ms, 4 lanes/ray (3 projections each) 0.32 ms, one pass instead of count+fill 0.18 ms,
// 16 lanes/ray 0.17 ms (fresh scene, 8192 rays x 97 samples); converged scene (13 k rays, 5..250 samples): 0.47 -> 0.37 ms.
//
// MODE 0: count only.  MODE 1: fill ray-ordered compact arrays (segments from f2n_segment_scan).  MODE 2: ONE pass into
// fixed-stride per-ray slots [ray * 1024 + k] (pts, dt, t, anchors as (trans, node) pairs; dirs are not written) plus the
// per-ray counts -- f2n_pack_samples then copies the filled prefix of every slot into the compact arrays, which costs a
// few tens of MB of streaming instead of a second march.
// ---------------------------------------------------------------------------------------------------
// Sum of one value per projection over the 12 projections of a 16-lane row, in Eigen's order (see ray_march16_kernel).
__device__ __forceinline__ float f2n_row12_sum(float e) {
  // lane 1 of each quad: e1 + e2
  const float t = e + __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(e), 0xE8, 0xF, 0xF, true));   // quad_perm [0,2,2,3]
  // every lane of the quad: e0 + (e1 + e2)
  const float t1 = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(t), 0x55, 0xF, 0xF, true));      // quad_perm [1,1,1,1]
  const float p = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(e), 0x00, 0xF, 0xF, true)) + t1;  // quad_perm [0,0,0,0]
  // groups (0,1) and (2,3): quads (0,3) and (1,2) are row_mirror images of each other
  const float u = p + __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(p), 0x140, 0xF, 0xF, true));
  // the two halves of the row
  return u + __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(u), 0x141, 0xF, 0xF, true));
}

// Sixteen lanes per ray: one projection per lane (12 of the 16 lanes of a DPP row; lane 3 of every quad idles).  The
// kernel's duration is the LONGEST ray's step count times the latency of one step (a converged scene has rays of
// 250+ samples next to rays of 5), and that latency is a dependent chain of VALU instructions -- so the per-lane
// instruction count is what matters, not the lane count.  Quad q of a row holds Eigen group G(q) = {0, 2, 3, 1}[q] of
// the 12-term tree ((e0+(e1+e2)) + (e3+(e4+e5))) + ((e6+(e7+e8)) + (e9+(e10+e11))): the triple sum is two quad_perm
// steps, row_mirror pairs quads (0,3) = groups (0,1) and (1,2) = groups (2,3), row_half_mirror pairs the two halves --
// every lane ends with the full sum, added in exactly the reference's order (fp add commutes, association is kept).
//
// TAIL (MODE 2 only; speculative sampling, see f2n_oct_list_repair): the walk also RECORDS, for every leaf-list entry it
// crosses into (or past), the state it had at the start of that iteration -- (t, samples emitted, list position, first-point
// flag), 8 bytes into leaf_state[ray * max_hits + entry] -- and the last list position it looked at (reached[ray]).  An
// iteration reads list entries only through that crossing loop, so when entry p is later REMOVED from the list (its leaf died)
// every iteration before the one that first reached p is untouched: a repair resumes from leaf_state[p] on the compacted list
// instead of from the ray's origin (repair_from[ray] = p > 0; 0 = from the origin, < 0 = nothing to do) and produces, bit for
// bit, what a fresh march over the new list produces.
template <int MODE, bool TAIL>
__device__ __forceinline__ void f2n_march_ray(
    const int ray, const int resume, float sample_l, int scale_by_dis, const float* __restrict__ rays_o, const float* __restrict__ rays_d,
    const float* __restrict__ noise_all, const int32_t* __restrict__ oct_start_end, const int32_t* __restrict__ oct_idx_all,
    const float* __restrict__ near_far_all, const F2nTreeNode* __restrict__ nodes, const F2nTransInfo* __restrict__ transes,
    const int32_t* __restrict__ pts_start_end, int32_t* __restrict__ pts_counts, float* __restrict__ pts,
    float* __restrict__ dirs, float* __restrict__ dts, float* __restrict__ ts, int32_t* __restrict__ anchors,
    float* __restrict__ first_oct_dis, const int32_t* __restrict__ oct_trans_all, uint2* __restrict__ leaf_state,
    int32_t* __restrict__ reached) {
  // (`ray` is uniform over the 16-lane row; resume: TAIL repair -- the list entry whose recorded state the walk resumes from,
  // 0 = from the ray's origin)
  const int lane16 = threadIdx.x & 15, kq = lane16 & 3, quad = lane16 >> 2;
  const int grp = quad == 0 ? 0 : quad == 1 ? 2 : quad == 2 ? 3 : 1;  // Eigen group of this quad (see above)
  const int proj = 3 * grp + min(kq, 2);                               // lane 3 of a quad shadows projection 3g+2 (unused)
  const int j = lane16;                                                // emission role
  const int oct_s = oct_start_end[2 * ray], n_oct = oct_start_end[2 * ray + 1] - oct_s;
  constexpr bool FILL = MODE != 0;
  int max_n = F2N_MAX_SAMPLE_PER_RAY;
  size_t base = 0;
  if (MODE == 1) {
    base = (size_t) pts_start_end[2 * ray];
    max_n = pts_start_end[2 * ray + 1] - (int) base;
  }
  if (MODE == 2) base = (size_t) ray * F2N_MAX_SAMPLE_PER_RAY;
  if (FILL && j == 0) first_oct_dis[ray] = n_oct > 0 ? near_far_all[2 * oct_s] : 1e9f;  // :226-231
  // Output roles of the 16 lanes of a row: lanes 0-2 pts.xyz, 3 t, 4 dt, 5-6 anchors (trans, node); in MODE 1 also lane 7
  // anchors[2] = 0 and lanes 8-10 dirs.xyz.  Every writer lane keeps ONE running pointer (4-byte units).
  uint32_t* out_ptr = nullptr;
  int out_stride = 0;
  uint32_t out_const = 0u;
  if (FILL) {
    if (j < 3) { if (pts != nullptr) { out_ptr = (uint32_t*) pts + 3 * base + j; out_stride = 3; } }
    else if (j == 3) { out_ptr = (uint32_t*) ts + base; out_stride = 1; }
    else if (j == 4) { out_ptr = (uint32_t*) dts + base; out_stride = 1; }
    else if (MODE == 2 && j < 7) { out_ptr = (uint32_t*) anchors + 2 * base + (j - 5); out_stride = 2; }
    else if (MODE == 1 && j < 8) { out_ptr = (uint32_t*) anchors + 3 * base + (j - 5); out_stride = 3; }
    else if (MODE == 1 && j < 11) {
      out_ptr = (uint32_t*) dirs + 3 * base + (j - 8);
      out_stride = 3;
      out_const = __float_as_uint(rays_d[3 * ray + (j - 8)]);
    }
  }
  // role masks: the emitted word is an AND/OR blend of the candidates (a `j == k ? ... : ...` ladder compiles to a
  // divergent branch tree of ~50 issue slots per step, and a lone wave pays ~4 cycles for every slot)
  const uint32_t m_w0 = j == 0 ? ~0u : 0u, m_w1 = j == 1 ? ~0u : 0u, m_w2 = j == 2 ? ~0u : 0u, m_t = j == 3 ? ~0u : 0u;
  const uint32_t m_dt = j == 4 ? ~0u : 0u, m_tr = j == 5 ? ~0u : 0u, m_oct = j == 6 ? ~0u : 0u;
  const uint32_t out_fixed = j >= 7 ? out_const : 0u;
  int n = 0, oct_ptr_final = 0;
  if (n_oct > 0 && max_n > 0) {
    const float o[3] = {rays_o[3 * ray], rays_o[3 * ray + 1], rays_o[3 * ray + 2]};
    const float d[3] = {rays_d[3 * ray], rays_d[3 * ray + 1], rays_d[3 * ray + 2]};
    const float* noise = noise_all + ray;  // shared, overlapping window (:203)
    const int32_t* oct_idx = oct_idx_all + oct_s;
    const float* near_far = near_far_all + 2 * oct_s;
    int oct_ptr = 0;
    bool first = true;
    // A window of the next four leaf entries (node, transform, near, far) lives in registers: a leaf crossing is a
    // register select, and the window is refilled with independent loads every fourth crossing.  In a converged scene a
    // ray crosses ~100 leaves; one dependent L2 round trip (entry -> node -> transform) per crossing was the march's
    // critical path there.
    const int32_t* oct_trans = oct_trans_all != nullptr ? oct_trans_all + oct_s : nullptr;
    int w_idx[4], w_tr[4], win_base = 0;
    float w_near[4], w_far[4];
    auto fill_window = [&](int b) {
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const int e = min(b + q, n_oct - 1);
        w_idx[q] = oct_idx[e];
        w_near[q] = near_far[2 * e];
        w_far[q] = near_far[2 * e + 1];
        w_tr[q] = oct_trans != nullptr ? oct_trans[e] : -2;
      }
    };
    float resume_t = 0.f;
    if (TAIL && resume > 0) {  // the state recorded when the walk first reached the (since removed) entry `resume`
      const uint2 rec = leaf_state[(size_t) oct_s + resume];
      resume_t = __uint_as_float(rec.x);
      n = (int) (rec.y & 0x7ffu);
      oct_ptr = (int) ((rec.y >> 11) & 0x7ffu);
      first = ((rec.y >> 22) & 1u) != 0u;
      win_base = oct_ptr;
      if (out_ptr != nullptr) out_ptr += (size_t) n * out_stride;
    }
    fill_window(win_base);
    int cur_oct = w_idx[0];
    int tidx = w_tr[0] != -2 ? w_tr[0] : nodes[cur_oct].trans_idx, cached_tidx = -1;
    float cur_t = (TAIL && resume > 0) ? resume_t : w_near[0], cur_far = w_far[0];
    float xyz[3] = {o[0] + d[0] * cur_t, o[1] + d[1] * cur_t, o[2] + d[2] * cur_t};
    float m[8], wg[3], radius_clip = 1.f;  // this lane's projection of the current TransInfo
    // noise[n] sits on the step's critical path (it scales the step length): four values at a time, the next four
    // already in flight.  The buffer has 1024 + n_rays + 10 floats, so ray + n + 7 stays inside it.
    float nz[4], nz_next[4];
    int nz_base = TAIL ? (n & ~3) : 0;
#pragma unroll
    for (int q = 0; q < 4; q++) {
      nz[q] = noise[nz_base + q];
      nz_next[q] = noise[nz_base + 4 + q];
    }
    while (n < max_n && oct_ptr < n_oct) {
      // (TAIL) the state this iteration starts from: what a later repair resumes with
      const uint32_t st_word = TAIL ? ((uint32_t) n | ((uint32_t) oct_ptr << 11) | (first ? (1u << 22) : 0u)) : 0u;
      if (n - nz_base >= 4) {
        nz_base += 4;
#pragma unroll
        for (int q = 0; q < 4; q++) {
          nz[q] = nz_next[q];
          nz_next[q] = noise[nz_base + 4 + q];
        }
      }
      if (tidx != cached_tidx) {
        const float* T = (const float*) (transes + tidx);
        const float4_t a4 = *(const float4_t*) (T + 8 * proj), b4 = *(const float4_t*) (T + 8 * proj + 4);  // w2xz[proj]
#pragma unroll
        for (int q = 0; q < 4; q++) {
          m[q] = a4[q];
          m[4 + q] = b4[q];
        }
#pragma unroll
        for (int r = 0; r < 3; r++) wg[r] = T[96 + 12 * r + proj];
        const float4_t cd = *(const float4_t*) (T + 132);  // center, dis_summary
        const float radius = f2n_norm3(o[0] - cd[0], o[1] - cd[1], o[2] - cd[2]) / cd[3];
        radius_clip = fmaxf(radius, 1.f);
        cached_tidx = tidx;
      }
      // this lane's projection and its contribution to the Jacobian (:171-187)
      const float px = f2n_sum4(m[0] * xyz[0], m[1] * xyz[1], m[2] * xyz[2], m[3] * 1.f);
      const float pz = f2n_sum4(m[4] * xyz[0], m[5] * xyz[1], m[6] * xyz[2], m[7] * 1.f);
      const float d0 = 1 / pz;
      const float d1 = -px / (pz * pz);
      float tj[3];
#pragma unroll
      for (int c = 0; c < 3; c++) tj[c] = d0 * m[c] + d1 * m[4 + c];
      float jac[3][3];
#pragma unroll
      for (int r = 0; r < 3; r++)
#pragma unroll
        for (int c = 0; c < 3; c++) jac[r][c] = f2n_row12_sum(wg[r] * tj[c]);
      float pj[3];
#pragma unroll
      for (int r = 0; r < 3; r++) pj[r] = f2n_sum3(jac[r][0] * d[0], jac[r][1] * d[1], jac[r][2] * d[2]);
      const float pj_norm = f2n_norm3(pj[0], pj[1], pj[2]) + 1e-6f;
      const int nq = n - nz_base;
      const float step_warp = sample_l * (nq == 0 ? nz[0] : nq == 1 ? nz[1] : nq == 2 ? nz[2] : nz[3]);
      float step = step_warp / pj_norm;
      if (scale_by_dis) step *= radius_clip;
      float march = step;
      if (!first) {  // the first point of a ray is never emitted (:274-289)
        if (FILL) {
          float w[3] = {0.f, 0.f, 0.f};  // the warped point (:155-169) shares the projection with the Jacobian
          if (MODE == 1 || pts != nullptr) {  // (MODE 2 without a pts buffer: f2n_pack_samples computes it, in parallel)
            const float v = px / pz;
#pragma unroll
            for (int r = 0; r < 3; r++) w[r] = f2n_row12_sum(wg[r] * v);
          }
          // one 4-byte store per writer lane (roles fixed before the loop) instead of per-array branches
          uint32_t val = out_fixed | (__float_as_uint(cur_t) & m_t) | (__float_as_uint(step * pj_norm) & m_dt) |
                         ((uint32_t) tidx & m_tr) | ((uint32_t) cur_oct & m_oct);
          if (MODE == 1 || pts != nullptr)
            val |= (__float_as_uint(w[0]) & m_w0) | (__float_as_uint(w[1]) & m_w1) | (__float_as_uint(w[2]) & m_w2);
          if (out_ptr != nullptr) {
            *out_ptr = val;
            out_ptr += out_stride;
          }
        }
        n++;
      }
      bool crossed = false;
      int new_tr = -2;
      while (cur_t + march > cur_far) {  // leaf crossing (:291-301)
        oct_ptr++;
        if (oct_ptr >= n_oct) break;
        if (TAIL && leaf_state != nullptr && j == 7) leaf_state[(size_t) oct_s + oct_ptr] = make_uint2(__float_as_uint(cur_t), st_word);
        if (oct_ptr - win_base >= 4) {
          win_base = oct_ptr;
          fill_window(win_base);
        }
        const int q = oct_ptr - win_base;
        cur_oct = q == 0 ? w_idx[0] : q == 1 ? w_idx[1] : q == 2 ? w_idx[2] : w_idx[3];
        new_tr = q == 0 ? w_tr[0] : q == 1 ? w_tr[1] : q == 2 ? w_tr[2] : w_tr[3];
        const float cur_near = q == 0 ? w_near[0] : q == 1 ? w_near[1] : q == 2 ? w_near[2] : w_near[3];
        cur_far = q == 0 ? w_far[0] : q == 1 ? w_far[1] : q == 2 ? w_far[2] : w_far[3];
        crossed = true;
        const int ex = (int) ceilf(fmaxf((cur_near - cur_t) / step, 1.f));
        march = step * (float) ex;
      }
      if (crossed) tidx = new_tr != -2 ? new_tr : nodes[cur_oct].trans_idx;
      cur_t += march;
#pragma unroll
      for (int c = 0; c < 3; c++) xyz[c] = o[c] + d[c] * cur_t;
      first = false;
    }
    oct_ptr_final = oct_ptr;
  }
  if (MODE != 1 && j == 0) pts_counts[ray] = n;
  if (TAIL && reached != nullptr && j == 0) reached[ray] = n_oct > 0 ? oct_ptr_final : 0;
}

template <int MODE, bool TAIL = false>
__global__ __launch_bounds__(64) void ray_march_kernel(
    int n_rays, float sample_l, int scale_by_dis, const float* __restrict__ rays_o, const float* __restrict__ rays_d,
    const float* __restrict__ noise_all, const int32_t* __restrict__ oct_start_end, const int32_t* __restrict__ oct_idx_all,
    const float* __restrict__ near_far_all, const F2nTreeNode* __restrict__ nodes, const F2nTransInfo* __restrict__ transes,
    const int32_t* __restrict__ pts_start_end, int32_t* __restrict__ pts_counts, float* __restrict__ pts,
    float* __restrict__ dirs, float* __restrict__ dts, float* __restrict__ ts, int32_t* __restrict__ anchors,
    float* __restrict__ first_oct_dis, const int32_t* __restrict__ oct_trans_all, const int32_t* __restrict__ repair_flags,
    const int32_t* __restrict__ death_epoch, int spec_epoch, uint2* __restrict__ leaf_state = nullptr,
    int32_t* __restrict__ reached = nullptr) {
  static_assert(!TAIL || MODE == 2, "resumable walks exist for the single-pass variant");
  const int ray = blockIdx.x * 4 + (threadIdx.x >> 4);
  int resume = 0;
  if (MODE == 2 && repair_flags != nullptr) {  // repair of a speculative march: only the rays whose leaf list was redone
    if (*death_epoch < spec_epoch) return;
    if (ray >= n_rays) return;
    if (TAIL) {
      resume = repair_flags[ray];  // (= repair_from of f2n_oct_list_repair)
      if (resume < 0) return;
    } else if (repair_flags[ray] == 0) {
      return;
    }
  }
  if (ray >= n_rays) return;  // whole rows leave together
  f2n_march_ray<MODE, TAIL>(ray, resume, sample_l, scale_by_dis, rays_o, rays_d, noise_all, oct_start_end, oct_idx_all, near_far_all,
                            nodes, transes, pts_start_end, pts_counts, pts, dirs, dts, ts, anchors, first_oct_dis, oct_trans_all,
                            leaf_state, reached);
}

// The same march on a SMALL, persistent grid (speculative batches of the two-deep pipeline, round 4): `gridDim.x` one-wave
// blocks take groups of four rays off a counter -- in the order of `order` (rays sorted by leaf count, longest first: the four
// rows of a wave then hold rays of similar length, and the longest chains start first) -- until the batch is done.  Why: the
// ordinary launch puts ~3.4 march waves of 96 registers on every SIMD of the chip for the first ~100 us of a march, and the
// occupancy-bound kernels of the main queue that run meanwhile (field_bwd: 23 us alone, 95 us underneath the march;
// shade_bwd, field_shade_fwd, compositing) lose most of their resident waves to them.  A batch that is sampled two steps
// ahead of its use has ~1.5 ms to finish: a few hundred waves do it, one per CU or two.  Same per-ray code, same bits.
template <bool TAIL>
__global__ __launch_bounds__(1024) void ray_march_persistent_kernel(
    int n_rays, float sample_l, int scale_by_dis, const float* __restrict__ rays_o, const float* __restrict__ rays_d,
    const float* __restrict__ noise_all, const int32_t* __restrict__ oct_start_end, const int32_t* __restrict__ oct_idx_all,
    const float* __restrict__ near_far_all, const F2nTreeNode* __restrict__ nodes, const F2nTransInfo* __restrict__ transes,
    int32_t* __restrict__ pts_counts, float* __restrict__ pts, float* __restrict__ dts, float* __restrict__ ts,
    int32_t* __restrict__ anchors, float* __restrict__ first_oct_dis, const int32_t* __restrict__ oct_trans_all,
    uint2* __restrict__ leaf_state, int32_t* __restrict__ reached, const int32_t* __restrict__ order, int32_t* __restrict__ counter) {
  const int n_groups = (n_rays + 3) >> 2;
  for (;;) {
    int g = 0;
    if ((threadIdx.x & 63) == 0) g = atomicAdd(counter, 1);
    g = __builtin_amdgcn_readfirstlane(g);
    if (g >= n_groups) break;
    const int slot = g * 4 + ((threadIdx.x & 63) >> 4);  // (every wave of the workgroup on its own)
    if (slot < n_rays) {
      const int ray = order != nullptr ? order[slot] : slot;
      f2n_march_ray<2, TAIL>(ray, 0, sample_l, scale_by_dis, rays_o, rays_d, noise_all, oct_start_end, oct_idx_all, near_far_all, nodes,
                             transes, nullptr, pts_counts, pts, nullptr, dts, ts, anchors, first_oct_dis, oct_trans_all, leaf_state,
                             reached);
    }
  }
}

// Rays ordered by leaf count, longest list first (counting sort, one block; the order among rays of equal count is whatever the
// LDS counters hand out: it only decides which wave marches a ray, not what the march produces).
__global__ __launch_bounds__(1024) void sort_rays_by_hits_kernel(int n_rays, int max_hits, const int32_t* __restrict__ oct_start_end,
                                                                 int32_t* __restrict__ order, int32_t* __restrict__ counter) {
  __shared__ int s_bin[2048 + 2];  // bin b = rays with max_hits - count == b (so that ascending bins are descending counts)
  __shared__ int s_wave[16];
  const int tid = threadIdx.x;
  for (int b = tid; b < 2048 + 2; b += 1024) s_bin[b] = 0;
  if (tid == 0 && counter != nullptr) *counter = 0;  // (the persistent march's group counter)
  __syncthreads();
  for (int r = tid; r < n_rays; r += 1024) {
    const int k = min(max(oct_start_end[2 * r + 1] - oct_start_end[2 * r], 0), max_hits);
    atomicAdd(&s_bin[max_hits - k], 1);
  }
  __syncthreads();
  {  // exclusive prefix over the 2050 bins: two bins per thread, wave scans, wave carries
    const int a = s_bin[2 * tid], b = s_bin[2 * tid + 1];
    int incl = a + b;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int up = __shfl_up(incl, off);
      if ((tid & 63) >= off) incl += up;
    }
    if ((tid & 63) == 63) s_wave[tid >> 6] = incl;
    __syncthreads();
    int carry = 0;
    for (int w = 0; w < (tid >> 6); w++) carry += s_wave[w];
    const int excl = carry + incl - (a + b);
    s_bin[2 * tid] = excl;
    s_bin[2 * tid + 1] = excl + a;
    if (tid == 1023) {  // the two bins behind the last pair (max_hits = 2048: bins 2048, 2049)
      const int t0 = s_bin[2048], t1 = s_bin[2049];
      s_bin[2048] = carry + incl;
      s_bin[2049] = carry + incl + t0;
      (void) t1;
    }
  }
  __syncthreads();
  for (int r = tid; r < n_rays; r += 1024) {
    const int k = min(max(oct_start_end[2 * r + 1] - oct_start_end[2 * r], 0), max_hits);
    order[atomicAdd(&s_bin[max_hits - k], 1)] = r;
  }
}

// Strided slots -> ray-ordered compact SampleResultFlex arrays (one wave per ray, coalesced copies).  When the march
// did not store warped points (s_pts == nullptr) they are computed here, one sample per lane and all samples in
// parallel: pts = W * (x_i / z_i) at xyz = o + d * t -- the very expressions of the march (:155-169, :303), so the bits
// are the same, but off the march's sequential critical path.
__global__ __launch_bounds__(256) void pack_samples_kernel(int n_rays, const int32_t* __restrict__ pts_start_end,
                                                           const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                                                           const F2nTransInfo* __restrict__ transes, const float* __restrict__ s_pts,
                                                           const float* __restrict__ s_dt, const float* __restrict__ s_t,
                                                           const int32_t* __restrict__ s_anchors, float* __restrict__ pts,
                                                           float* __restrict__ dirs, float* __restrict__ dt, float* __restrict__ t,
                                                           int32_t* __restrict__ anchors, const int32_t* __restrict__ death_epoch,
                                                           int spec_epoch) {
  // (re-pack of a speculatively sampled batch that was already packed before the stat update: nothing to do unless a leaf
  // died since -- the same grid-uniform test as the two repair kernels)
  if (death_epoch != nullptr && *death_epoch < spec_epoch) return;
  const int lane = threadIdx.x & 63;
  const int ray = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (ray >= n_rays) return;
  const int s = pts_start_end[2 * ray], cnt = pts_start_end[2 * ray + 1] - s;
  const size_t src = (size_t) ray * F2N_MAX_SAMPLE_PER_RAY;
  const float d[3] = {rays_d[3 * ray], rays_d[3 * ray + 1], rays_d[3 * ray + 2]};
  for (int i = lane; i < 3 * cnt; i += 64) {
    if (s_pts != nullptr) pts[3 * (size_t) s + i] = s_pts[3 * src + i];
    dirs[3 * (size_t) s + i] = d[i % 3];
    const int k = i / 3, c = i - 3 * k;
    anchors[3 * (size_t) s + i] = c < 2 ? s_anchors[2 * (src + k) + c] : 0;
  }
  const float o[3] = {rays_o[3 * ray], rays_o[3 * ray + 1], rays_o[3 * ray + 2]};
  for (int i = lane; i < cnt; i += 64) {
    const float ti = s_t[src + i];
    dt[s + i] = s_dt[src + i];
    t[s + i] = ti;
    if (s_pts == nullptr) {
      const float xyz[3] = {o[0] + d[0] * ti, o[1] + d[1] * ti, o[2] + d[2] * ti};
      float w[3];
      f2n_warp(transes + s_anchors[2 * (src + i)], xyz, w);
#pragma unroll
      for (int c = 0; c < 3; c++) pts[3 * (size_t) (s + i) + c] = w[c];
    }
  }
}

// rays_d / ||rays_d|| (PersSampler.cu:319).  The reference uses torch::linalg_norm, whose summation order is an
// implementation detail of the ATen reduction; here the order is fixed -- sqrt((x*x + y*y) + z*z) -- so that the
// oracle can restate it bit for bit (everything downstream of the sampler depends on these bits).
__global__ void normalize_dirs_kernel(int n, const float* __restrict__ in, float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float x = in[3 * i], y = in[3 * i + 1], z = in[3 * i + 2];
  const float nrm = sqrtf((x * x + y * y) + z * z);
  out[3 * i] = x / nrm;
  out[3 * i + 1] = y / nrm;
  out[3 * i + 2] = z / nrm;
}

// The three ray-count-sized preparations of a GetSamples call in one launch: unit directions (above), the zeroed hit / sample
// totals the intersection and the scan add into, and the affine map of the march noise (PersSampler.cu:372-381).  They were
// three dependent launches at the head of the sampler chain, which is the longer chain of a converged training step.
__global__ void sampler_prologue_kernel(int n_rays, const float* __restrict__ in, float* __restrict__ out, int32_t* __restrict__ zero,
                                        int n_zero, int n_noise, const float* u, float fineness, float* noise_out /*may be u*/,
                                        unsigned long long key, unsigned long long seq) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_zero) zero[i] = 0;
  if (i < n_noise) {
    float ui;
    if (u != nullptr) {
      ui = u[i];
    } else {  // keyed: element i of batch `seq`'s march noise, drawn here (no rand launch at the head of the sampler chain)
      uint32_t x[4];
      f2n_philox4x32((uint32_t) (i >> 2), 0u, (uint32_t) seq, (uint32_t) (seq >> 32), (uint32_t) key, (uint32_t) (key >> 32), x);
      ui = f2n_u01(x[i & 3]);
    }
    noise_out[i] = ((ui - .5f) + 1.f) * fineness;
  }
  if (i >= n_rays) return;
  const float x = in[3 * i], y = in[3 * i + 1], z = in[3 * i + 2];
  const float nrm = sqrtf((x * x + y * y) + z * z);
  out[3 * i] = x / nrm;
  out[3 * i + 1] = y / nrm;
  out[3 * i + 2] = z / nrm;
}

// GetEdgeSamplesKernel, PersSampler.cu:436-452.  The draws either come as the reference makes them (edge_idx from randint,
// edge_coords uniform in [-1,1)) or as three uniforms in [0,1) per point (u01: idx = floor(u0 * n_edges), coords = 2u - 1 --
// one random launch instead of two).  The two warped points / transform indices of every edge point go to out_pts / out_idx
// (index i at out_idx[(2i) * idx_stride]) and, when given, to a second destination with its own index stride: a streaming
// training step appends them to the sampler's point array for the density pre-pass AND needs them at the head of the grad
// pass's point array.
__global__ void edge_samples_kernel(int n_pts, const F2nEdgePool* __restrict__ edge_pool, int n_edges,
                                    const F2nTransInfo* __restrict__ transes, const int32_t* __restrict__ edge_idx,
                                    const float* __restrict__ edge_coords, const float* __restrict__ u01,
                                    float* __restrict__ out_pts, int32_t* __restrict__ out_idx, int idx_stride,
                                    float* __restrict__ out_pts2, int32_t* __restrict__ out_idx2, int idx_stride2) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_pts) return;
  int ei;
  float c0, c1;
  if (u01 != nullptr) {
    ei = min((int) (u01[3 * i] * (float) n_edges), n_edges - 1);
    c0 = u01[3 * i + 1] * 2.f - 1.f;
    c1 = u01[3 * i + 2] * 2.f - 1.f;
  } else {
    ei = edge_idx[i];
    c0 = edge_coords[2 * i];
    c1 = edge_coords[2 * i + 1];
  }
  const F2nEdgePool* e = edge_pool + ei;
  float w[3];
#pragma unroll
  for (int c = 0; c < 3; c++) w[c] = (e->center[c] + e->dir_0[c] * c0) + e->dir_1[c] * c1;
  float a[3], b[3];
  f2n_warp(transes + e->t_idx_a, w, a);
  f2n_warp(transes + e->t_idx_b, w, b);
#pragma unroll
  for (int c = 0; c < 3; c++) {
    out_pts[6 * i + c] = a[c];
    out_pts[6 * i + 3 + c] = b[c];
  }
  out_idx[(size_t) (2 * i) * idx_stride] = e->t_idx_a;
  out_idx[(size_t) (2 * i + 1) * idx_stride] = e->t_idx_b;
  if (out_pts2 != nullptr) {
#pragma unroll
    for (int c = 0; c < 3; c++) {
      out_pts2[6 * i + c] = a[c];
      out_pts2[6 * i + 3 + c] = b[c];
    }
    out_idx2[(size_t) (2 * i) * idx_stride2] = e->t_idx_a;
    out_idx2[(size_t) (2 * i + 1) * idx_stride2] = e->t_idx_b;
  }
}

// ---------------------------------------------------------------------------------------------------
// Occupancy votes, PersSampler.cu:475-526 (integer atomic maxima).
// ---------------------------------------------------------------------------------------------------
// Votes into LDS images (mark == nullptr) are cast where they arise.  Votes into the GLOBAL buffers (trees too large for the block-local
// images: every converged scene) are not (round 6): on gfx9 a global atomic counts in vmcnt like a load, and vmcnt retires in order, so
// an atomic issued in chunk k made chunk k+1's wait for its (prefetched) samples a wait for that atomic's trip to the memory-side atomic
// unit -- the walk of the batch's longest ray paid ~1 us per chunk, 22 of mark_visit's 32 us (tools/probe/rowwalk_tail.py,
// profiles/r06_row_walks.txt).  A row now QUEUES its votes in LDS (F2N_VOTE_CAP records of 8 bytes, appended with a row ballot, no
// atomics) and casts them behind its walk; and since a maximum only ever grows, a vote the buffer's current value already covers is
// dropped after a plain load (all four words fetched together; a stale -- lower -- read costs a redundant atomic, never a lost vote): a
// leaf is voted on by hundreds of rays per batch and only the first few votes change anything, while every atomic goes through the
// memory-side unit (~21 G/s chip-wide: 3.8e5 per converged batch were 18 us).
#define F2N_VOTE_CAP 96
__device__ __forceinline__ void f2n_vote_global(int32_t* w_adder, int32_t* a_adder, int32_t* mark, int32_t* cnt, int node, int vw, int va,
                                                int visits) {
  const int cw = __hip_atomic_load(w_adder + node, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // (agent scope: read past the CU's L1)
  const int ca = __hip_atomic_load(a_adder + node, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const int cc = __hip_atomic_load(cnt + node, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const int cm = __hip_atomic_load(mark + node, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (cw < vw) atomicMax(w_adder + node, vw);
  if (ca < va) atomicMax(a_adder + node, va);
  if (cc < visits) atomicMax(cnt + node, visits);
  if (cm != 1) mark[node] = 1;
}
// the row's queued votes: lane c takes records c, c + 16, ...
__device__ __forceinline__ void f2n_votes_flush_row(const uint2* list, int n, int c, int32_t* w_adder, int32_t* a_adder, int32_t* mark, int32_t* cnt) {
  for (int k = c; k < n; k += 16) {
    const uint2 r = list[k];
    f2n_vote_global(w_adder, a_adder, mark, cnt, (int) r.x, (r.y & 1u) ? 512 : -1, (r.y & 2u) ? 32 : -1, (int) (r.y >> 2));
  }
}
__device__ __forceinline__ void f2n_vote(int32_t* w_adder, int32_t* a_adder, int32_t* mark, int32_t* cnt, int node,
                                         float w, float a, float w_thres, float a_thres, int visits) {
  atomicMax(w_adder + node, w > w_thres ? 512 : -1);  // OCC_WEIGHT_BASE
  atomicMax(a_adder + node, a > a_thres ? 32 : -1);   // OCC_ALPHA_BASE
  atomicMax(cnt + node, visits);
  if (mark != nullptr) mark[node] = 1;
}

// USE_LDS: the votes of a block's rays are first max-combined in LDS (ds_max_i32) and flushed once per block.
// A few hundred leaves receive tens of thousands of votes per batch; as global atomics those are long
// same-address dependent chains (~1 us each at the memory-side atomic unit).
// The votes of one ray (its 16-lane row): thresholds from the ray's maxima (:12-17), then one vote per run of equal leaves,
// cast by the run's last sample, which learns the run's maxima and start from a segmented max-scan over DPP row shifts.
__device__ __forceinline__ void f2n_ray_votes(int s, int e, int c, float mw, float ma, const int32_t* __restrict__ anchors,
                                              int anchor_stride, const float* weights, const float* alphas, int32_t* w_adder,
                                              int32_t* a_adder, int32_t* mark, int32_t* cnt, uint2* vote_list = nullptr) {
  // vote_list != nullptr: this row's F2N_VOTE_CAP queue slots in LDS (global vote buffers, see f2n_vote_global)
  int n_queued = 0;
  // 0.1 / 0.01 / 0.02 are double literals in the reference (:12-17): float*double, then narrowed by fminf
  const float w_thres = fminf((float) ((double) mw * 0.1), (float) 0.01);
  const float a_thres = fminf((float) ((double) ma * 0.1), (float) 0.02);
  // (round 6) A chunk's three inputs are fetched one chunk AHEAD -- a walk whose loads are issued where their values are used pays a
  // memory round trip per chunk, and the launch lasts as long as its longest ray --; the neighbours' leaves come from the row
  // itself (DPP shifts; the chunk borders from the previous chunk's last lane and the prefetched chunk's first) instead of two
  // more loads; what crosses from one chunk to the next is only ever used by lane 0, which row_ror:1 serves without an LDS trip.
  float carry_w = 0.f, carry_a = 0.f;
  int carry_start = s, carry_node = -1;
  auto rot1 = [](int v) { return __builtin_amdgcn_update_dpp(0, v, 0x121, 0xF, 0xF, false); };  // lane 0 <- lane 15
  // A window of FOUR chunks in flight: the walk's body is ~0.2 us of instructions against a ~0.9 us round trip, and with one chunk
  // ahead the longest ray paid the difference 25 times over (mark_visit 32.6 us with a 396-sample ray in the batch, 13.8 us with rays
  // clipped at 128: tools/probe/rowwalk_tail.py).  Indices are clamped into the ray: a chunk past its end is a harmless re-read.
  struct Chunk {
    int node;
    float w, a;
  };
  auto fetch_chunk = [&](int b) {
    const int j = min(b + c, e - 1);
    Chunk q;
    q.node = anchors[(size_t) j * anchor_stride + 1];
    q.w = weights[j];
    q.a = alphas[j];
    return q;
  };
  Chunk q0 = fetch_chunk(s), q1 = fetch_chunk(s + 16), q2 = fetch_chunk(s + 32), q3 = fetch_chunk(s + 48);
  for (int base = s; base < e; base += 16) {
    const int i = base + c;
    const bool in = i < e;
    const int ic = in ? i : e - 1;
    const int node = q0.node;
    const float c_w = q0.w, c_a = q0.a;
    q0 = q1;
    q1 = q2;
    q2 = q3;
    q3 = fetch_chunk(base + 64);
    const int n_node = q0.node;  // (the next chunk's leaves: lane 15's right-hand neighbour)
    // lane 0: the previous chunk's last leaf (no leaf in front of the ray's first sample); lane 15: the next chunk's first
    const int prev = __builtin_amdgcn_update_dpp(base > s ? carry_node : -1, node, 0x111, 0xF, 0xF, false);
    const int nx0 = __builtin_amdgcn_update_dpp(0, n_node, 0x12F, 0xF, 0xF, false);  // lane 15 <- lane 0
    const int nxt = __builtin_amdgcn_update_dpp(nx0, node, 0x101, 0xF, 0xF, false);
    const int next = ic + 1 < e ? nxt : -1;
    const bool head = node != prev;
    float cw = fmaxf(0.f, c_w), ca = fmaxf(0.f, c_a);  // the reference's running maxima start at 0
    int start = head ? ic : (int) 0x80000000;
    if (c == 0 && !head) {  // the run continues from the previous chunk
      cw = fmaxf(cw, carry_w);
      ca = fmaxf(ca, carry_a);
      start = carry_start;
    }
    int f = head ? 1 : 0;
#define F2N_SEGMAX_STEP(K)                                                                     \
  {                                                                                            \
    const float tw = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(cw), 0x110 + K, 0xF, 0xF, false)); \
    const float ta = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(ca), 0x110 + K, 0xF, 0xF, false)); \
    const int ts = __builtin_amdgcn_update_dpp((int) 0x80000000, start, 0x110 + K, 0xF, 0xF, false);              \
    const int tf = __builtin_amdgcn_update_dpp(1, f, 0x110 + K, 0xF, 0xF, false);                                 \
    if (c >= K && f == 0) {                                                                    \
      cw = fmaxf(cw, tw);                                                                      \
      ca = fmaxf(ca, ta);                                                                      \
      start = max(start, ts);                                                                  \
    }                                                                                          \
    if (c >= K) f |= tf;                                                                       \
  }
    F2N_SEGMAX_STEP(1)
    F2N_SEGMAX_STEP(2)
    F2N_SEGMAX_STEP(4)
    F2N_SEGMAX_STEP(8)
#undef F2N_SEGMAX_STEP
    const bool votes = in && node != next;
    if (vote_list == nullptr) {
      if (votes) f2n_vote(w_adder, a_adder, mark, cnt, node, cw, ca, w_thres, a_thres, ic - start + 1);
    } else {
      const unsigned row_bits = (unsigned) ((__ballot(votes) >> (threadIdx.x & 48)) & 0xFFFFull);  // (the rows of a wave loop independently)
      const int n_new = __popc(row_bits);
      if (n_queued + n_new > F2N_VOTE_CAP) {  // (row-uniform; a ray of more than ~F2N_VOTE_CAP leaf runs)
        f2n_votes_flush_row(vote_list, n_queued, c, w_adder, a_adder, mark, cnt);
        n_queued = 0;
      }
      if (votes)
        vote_list[n_queued + __popc(row_bits & ((1u << c) - 1u))] =
            make_uint2((unsigned) node, (cw > w_thres ? 1u : 0u) | (ca > a_thres ? 2u : 0u) | ((unsigned) (ic - start + 1) << 2));
      n_queued += n_new;
    }
    carry_w = __int_as_float(rot1(__float_as_int(cw)));
    carry_a = __int_as_float(rot1(__float_as_int(ca)));
    carry_start = rot1(start);
    carry_node = rot1(node);
  }
  if (vote_list != nullptr) f2n_votes_flush_row(vote_list, n_queued, c, w_adder, a_adder, mark, cnt);
}

// USE_LDS: block-local vote images (see mark_visit_kernel) -- set up / flushed by these two.
__device__ __forceinline__ void f2n_votes_lds_init(int32_t* s_votes, int n_nodes) {
  for (int i = threadIdx.x; i < n_nodes; i += blockDim.x) {
    s_votes[i] = -2;
    s_votes[n_nodes + i] = -2;
    s_votes[2 * n_nodes + i] = 0;
  }
  __syncthreads();
}
__device__ __forceinline__ void f2n_votes_lds_flush(const int32_t* s_votes, int n_nodes, int32_t* g_w_adder, int32_t* g_a_adder,
                                                    int32_t* g_mark, int32_t* g_cnt) {
  __syncthreads();
  for (int i = threadIdx.x; i < n_nodes; i += blockDim.x) {
    const int visits = s_votes[2 * n_nodes + i];
    if (visits > 0) {
      atomicMax(g_w_adder + i, s_votes[i]);
      atomicMax(g_a_adder + i, s_votes[n_nodes + i]);
      atomicMax(g_cnt + i, visits);
      g_mark[i] = 1;
    }
  }
}

template <bool USE_LDS>
__global__ void mark_visit_kernel(int n_rays, int n_nodes, const int32_t* __restrict__ pts_start_end,
                                  const int32_t* __restrict__ anchors, int anchor_stride, const float* __restrict__ weights,
                                  const float* __restrict__ alphas, int32_t* g_w_adder, int32_t* g_a_adder, int32_t* g_mark,
                                  int32_t* g_cnt) {
  extern __shared__ int32_t s_votes[];  // [3][n_nodes]: weight vote, alpha vote, visit count
  __shared__ uint2 s_vote_list[USE_LDS ? 1 : 16 * F2N_VOTE_CAP];  // global vote buffers: the 16 rows' queues (256-thread blocks)
  int32_t* w_adder = USE_LDS ? s_votes : g_w_adder;
  int32_t* a_adder = USE_LDS ? s_votes + n_nodes : g_a_adder;
  int32_t* cnt = USE_LDS ? s_votes + 2 * n_nodes : g_cnt;
  int32_t* mark = USE_LDS ? nullptr : g_mark;  // LDS path: "visited" <=> block-local visit count >= 1
  if (USE_LDS) f2n_votes_lds_init(s_votes, n_nodes);
  // one 16-lane row per ray (the votes are integer maxima: order-free)
  const int c = threadIdx.x & 15;
  const int ray = blockIdx.x * (blockDim.x >> 4) + (threadIdx.x >> 4);
  int s = 0, e = 0;
  if (ray < n_rays) {
    s = pts_start_end[2 * ray];
    e = pts_start_end[2 * ray + 1];
  }
  if (s < e) {
    float mw = 0.f, ma = 0.f;
    for (int i = s + c; i < e; i += 16) {
      mw = fmaxf(mw, weights[i]);
      ma = fmaxf(ma, alphas[i]);
    }
#pragma unroll
    for (int off = 8; off >= 1; off >>= 1) {
      mw = fmaxf(mw, __shfl_xor(mw, off, 16));
      ma = fmaxf(ma, __shfl_xor(ma, off, 16));
    }
    f2n_ray_votes(s, e, c, mw, ma, anchors, anchor_stride, weights, alphas, w_adder, a_adder, mark, cnt,
                  USE_LDS ? nullptr : s_vote_list + (threadIdx.x >> 4) * F2N_VOTE_CAP);
  }
  if (USE_LDS) f2n_votes_lds_flush(s_votes, n_nodes, g_w_adder, g_a_adder, g_mark, g_cnt);
}

// f2n_early_stop (Renderer.cpp:115-126) and f2n_oct_mark_visit (PersSampler.cu:475-526) in one launch: the pre-pass of a
// training step walks every ray twice in a row -- transmittance / weights / mask, then the occupancy votes over those very
// weights -- with the same row-per-ray mapping, and the second walk was a dependent launch on the chain the NEXT batch's
// sampling waits for (pre-pass -> votes -> stat update).  The row keeps the ray's maxima while it writes weights and alphas and
// re-reads its own writes for the votes (same lane, same address: no fence needed).  Arithmetic and outputs are those of the
// two kernels, bit for bit.
template <bool USE_LDS>
__global__ void early_stop_votes_kernel(int n_rays, int n_nodes, const int32_t* __restrict__ se, const float* __restrict__ f0,
                                        int f0_stride, const float* __restrict__ dt, float* weights, float* alphas,
                                        int32_t* __restrict__ mask, int32_t* __restrict__ kept,
                                        const int32_t* __restrict__ anchors, int anchor_stride, int32_t* g_w_adder,
                                        int32_t* g_a_adder, int32_t* g_mark, int32_t* g_cnt) {
  extern __shared__ int32_t s_votes[];
  __shared__ uint2 s_vote_list[USE_LDS ? 1 : 16 * F2N_VOTE_CAP];  // global vote buffers: the 16 rows' queues (256-thread blocks)
  int32_t* w_adder = USE_LDS ? s_votes : g_w_adder;
  int32_t* a_adder = USE_LDS ? s_votes + n_nodes : g_a_adder;
  int32_t* cnt_v = USE_LDS ? s_votes + 2 * n_nodes : g_cnt;
  int32_t* mark = USE_LDS ? nullptr : g_mark;
  if (USE_LDS) f2n_votes_lds_init(s_votes, n_nodes);
  const int c = threadIdx.x & 15;
  const int ray = blockIdx.x * (blockDim.x >> 4) + (threadIdx.x >> 4);
  int s = 0, e = 0;
  if (ray < n_rays) {
    s = se[2 * ray];
    e = se[2 * ray + 1];
  }
  // ---- early stop: the walk of early_stop_kernel ----
  float acc = 0.f, mw = 0.f, ma = 0.f;
  int cnt = 0;
  float n_f0 = 0.f, n_dt = 0.f;
  if (s < e) {
    const int i = min(s + c, e - 1);
    n_f0 = f0[(size_t) i * f0_stride];
    n_dt = dt[i];
  }
  for (int base = s; base < e; base += 16) {
    const int i = base + c;
    const bool in = i < e;
    const float c_f0 = n_f0, c_dt = n_dt;
    if (base + 16 < e) {
      const int j = min(i + 16, e - 1);
      n_f0 = f0[(size_t) j * f0_stride];
      n_dt = dt[j];
    }
    float sec = 0.f;
    if (in) sec = expf(c_f0 - F2N_DENSITY_SHIFT) * c_dt;
    const float alpha = 1.f - expf(-sec);
    float excl;
    f2n_row_chain1x(sec, acc, excl);
    const float trans = expf(-excl);  // exclusive cumulative density
    const int m = (in && trans > F2N_T_EPS) ? 1 : 0;
    if (in) {
      const float w = trans * alpha;
      weights[i] = w;
      alphas[i] = alpha;
      mask[i] = m;
      mw = fmaxf(mw, w);
      ma = fmaxf(ma, alpha);
    }
    cnt += m;
  }
#pragma unroll
  for (int off = 8; off >= 1; off >>= 1) {
    cnt += __shfl_xor(cnt, off, 16);
    mw = fmaxf(mw, __shfl_xor(mw, off, 16));
    ma = fmaxf(ma, __shfl_xor(ma, off, 16));
  }
  if (ray < n_rays && c == 0) kept[ray] = cnt;
  // ---- votes: the walk of mark_visit_kernel over what this row has just written ----
  if (s < e)
    f2n_ray_votes(s, e, c, mw, ma, anchors, anchor_stride, weights, alphas, w_adder, a_adder, mark, cnt_v,
                  USE_LDS ? nullptr : s_vote_list + (threadIdx.x >> 4) * F2N_VOTE_CAP);
  if (USE_LDS) f2n_votes_lds_flush(s_votes, n_nodes, g_w_adder, g_a_adder, g_mark, g_cnt);
}

// PersSampler.cu:579-593 (torch integer ops) + MarkInvalidNodes (:528-534), one node per lane.
struct F2nStatsArgs {
  int n_nodes;
  int32_t *w_adder, *a_adder, *mark, *w_stats, *a_stats;
  F2nTreeNode* nodes;
  F2nChildInfo* child_blocks;
  int reset_votes;
  int32_t* died_at;
  int epoch;
  int32_t *death_epoch, *death_epoch_host;
};

__device__ __forceinline__ void f2n_update_stats_node(int i, const F2nStatsArgs& a) {
  int32_t* __restrict__ w_adder = a.w_adder;
  int32_t* __restrict__ a_adder = a.a_adder;
  int32_t* __restrict__ mark = a.mark;
  int32_t* __restrict__ w_stats = a.w_stats;
  int32_t* __restrict__ a_stats = a.a_stats;
  F2nTreeNode* __restrict__ nodes = a.nodes;
  F2nChildInfo* __restrict__ child_blocks = a.child_blocks;
  int32_t* __restrict__ died_at = a.died_at;
  int32_t* __restrict__ death_epoch = a.death_epoch;
  int32_t* __restrict__ death_epoch_host = a.death_epoch_host;
  const int reset_votes = a.reset_votes, epoch = a.epoch, n_nodes = a.n_nodes;
  if (i >= n_nodes) return;
  const int m = mark[i];
  int st[2];
  const int add[2] = {w_adder[i], a_adder[i]};
  if (reset_votes) {  // leave the vote buffers in the state the next f2n_oct_mark_visit expects (:555-556)
    w_adder[i] = -1;
    a_adder[i] = -1;
    mark[i] = 0;
  }
  st[0] = w_stats[i];
  st[1] = a_stats[i];
#pragma unroll
  for (int k = 0; k < 2; k++) {
    const int occ = add[k] > 0 ? 1 : 0;
    int v = max(st[k], occ * add[k]);
    v += m * (1 - occ) * add[k];
    st[k] = min(max(v, -100), 1 << 20);
  }
  w_stats[i] = st[0];
  a_stats[i] = st[1];
  if (st[0] < 0 || st[1] < 0) {
    if (died_at != nullptr && nodes[i].trans_idx >= 0) {  // alive until now: stamp the death for speculative samplers (MODE 3 above)
      died_at[i] = epoch;
      *death_epoch = epoch;  // (every writer of a launch stores the same value)
      if (death_epoch_host != nullptr) *death_epoch_host = epoch;  // pinned host word: the host's "did a leaf die lately" hint
    }
    nodes[i].trans_idx = -1;
    const int pa = nodes[i].parent;
    if (child_blocks != nullptr && pa >= 0) {  // the parent's copy of this node's record
#pragma unroll
      for (int c = 0; c < 8; c++)
        if (nodes[pa].childs[c] == i) child_blocks[(size_t) pa * 8 + c].trans_idx = -1;
    }
  }
}

__global__ void update_stats_kernel(F2nStatsArgs a) { f2n_update_stats_node(blockIdx.x * blockDim.x + threadIdx.x, a); }

// The stat update and the survivor scan of a streaming training step in ONE launch (round 5, launch floor): they are independent --
// one lane per node, and one block scanning the rays' kept counts -- and sat on the step's main queue as two dependent launches
// behind the early stop.  Block 0 scans (FilterIdxBounds, Renderer.cu:20-50); the blocks behind it take 1024 nodes each.
__global__ __launch_bounds__(F2N_SCAN_THREADS) void stats_and_scan_kernel(F2nStatsArgs a, int n, const int32_t* __restrict__ counts,
                                                                          int32_t* __restrict__ start_end, int32_t* __restrict__ total,
                                                                          int32_t* __restrict__ mirror, const int32_t* __restrict__ also,
                                                                          int n_also) {
  if (blockIdx.x == 0) f2n_segment_scan_block(n, counts, start_end, total, mirror, also, n_also);
  else f2n_update_stats_node((blockIdx.x - 1) * F2N_SCAN_THREADS + threadIdx.x, a);
}

// child_blocks[u][c] <- what the DFS needs to know about child slot c of node u
__global__ void build_child_blocks_kernel(int n_nodes, const F2nTreeNode* __restrict__ nodes, F2nChildInfo* __restrict__ out) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_nodes * 8) return;
  const int u = t >> 3, c = t & 7;
  F2nChildInfo e;
  e.center[0] = e.center[1] = e.center[2] = 0.f;
  e.side_len = 0.f;
  e.child = nodes[u].childs[c];
  e.trans_idx = -1;
  e.interior = 0;
  e.pad = 0;
  if (e.child >= 0) {
    const F2nTreeNode* nd = nodes + e.child;
    e.center[0] = nd->center[0];
    e.center[1] = nd->center[1];
    e.center[2] = nd->center[2];
    e.side_len = nd->side_len;
    e.trans_idx = nd->trans_idx;
    int interior = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) interior |= nd->childs[k] >= 0;
    e.interior = interior;
  }
  out[t] = e;
}

// MarkInvisibleNodesKernel + CheckVisible, PersSampler.cu:618-661.
__global__ void mark_invisible_kernel(int n_nodes, int n_cams, F2nTreeNode* __restrict__ nodes,
                                      const float* __restrict__ intris, const float* __restrict__ w2cs,
                                      const float* __restrict__ bounds) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_nodes) return;
  const float c[3] = {nodes[i].center[0], nodes[i].center[1], nodes[i].center[2]};
  const float radius = (float) ((double) nodes[i].side_len * 0.707);
  int visible = 0;
  for (int k = 0; k < n_cams; k++) {
    const float* m = w2cs + 12 * k;
    const float* K = intris + 9 * k;
    float p[3];
#pragma unroll
    for (int r = 0; r < 3; r++) p[r] = f2n_sum4(m[4 * r] * c[0], m[4 * r + 1] * c[1], m[4 * r + 2] * c[2], m[4 * r + 3] * 1.f);
    if (-p[2] < bounds[2 * k] - radius || -p[2] > bounds[2 * k + 1] + radius) continue;
    if (f2n_norm3(p[0], p[1], p[2]) < radius) { visible++; continue; }
    const float cx = K[2], cy = K[5], fx = K[0], fy = K[4];
    const float bx = radius / -p[2] * fx, by = radius / -p[2] * fy;
    const float ix = p[0] / -p[2] * fx, iy = p[1] / -p[2] * fy;
    if (ix + bx < -cx || ix > cx + bx || iy + by < -cy || iy > cy + by) continue;
    visible++;
  }
  if (visible < 1) nodes[i].trans_idx = -1;
}

// ---------------------------------------------------------------------------------------------------
// C-ABI
// ---------------------------------------------------------------------------------------------------
// GetVisiCams (PersSampler.cpp:27-66) for a whole frontier of candidate boxes at once: block (box, camera) casts the
// camera's pixel-grid ray bundle (res_h x res_w, pixel centres given as arrays) at the box -- slab test clipped to the
// camera's [near, far] -- and reports whether ANY ray hits.  The reference evaluates this with broadcast ATen ops, one
// node at a time, inside its recursive constructor; here the octree is built level by level, so one launch serves every
// node of a depth.
__global__ __launch_bounds__(256) void visible_cams_kernel(int n_cams, const float* __restrict__ boxes,
                                                           const float* __restrict__ c2w, const float* __restrict__ bounds,
                                                           float fx, float fy, float cx, float cy, int res_h, int res_w,
                                                           const float* __restrict__ pix_i, const float* __restrict__ pix_j,
                                                           uint8_t* __restrict__ visible) {
  __shared__ int s_hit;
  const int box = blockIdx.x, cam = blockIdx.y;
  if (threadIdx.x == 0) s_hit = 0;
  __syncthreads();
  const float* P = c2w + 12 * (size_t) cam;
  const float o[3] = {P[3], P[7], P[11]};
  const float bc[3] = {boxes[4 * box], boxes[4 * box + 1], boxes[4 * box + 2]};
  const float half = boxes[4 * box + 3] * .5f;
  const float lo[3] = {bc[0] - half, bc[1] - half, bc[2] - half}, hi[3] = {bc[0] + half, bc[1] + half, bc[2] + half};
  const float b_near = bounds[2 * cam], b_far = bounds[2 * cam + 1];
  bool hit = false;
  for (int p = threadIdx.x; p < res_h * res_w && !hit; p += 256) {
    const float i = pix_i[p / res_w], j = pix_j[p % res_w];
    const float dc[3] = {(j - cx) / fx, -(i - cy) / fy, -1.f};
    float near_ = -3.4e38f, far_ = 3.4e38f;
#pragma unroll
    for (int a = 0; a < 3; a++) {
      const float d = (P[4 * a] * dc[0] + P[4 * a + 1] * dc[1]) + P[4 * a + 2] * dc[2];
      float ta = (lo[a] - o[a]) / d, tb = (hi[a] - o[a]) / d;
      // torch::nan_to_num(x, 0, 1e6, -1e6)
      ta = ta != ta ? 0.f : (ta > 3.4e38f ? 1e6f : (ta < -3.4e38f ? -1e6f : ta));
      tb = tb != tb ? 0.f : (tb > 3.4e38f ? 1e6f : (tb < -3.4e38f ? -1e6f : tb));
      far_ = fminf(far_, fmaxf(ta, tb));
      near_ = fmaxf(near_, fminf(ta, tb));
    }
    far_ = fminf(far_, b_far);
    near_ = fmaxf(near_, b_near);
    hit = far_ > near_;
  }
  if (hit) s_hit = 1;
  __syncthreads();
  if (threadIdx.x == 0) visible[(size_t) box * n_cams + cam] = (uint8_t) s_hit;
}

__global__ void march_noise_kernel(int n, const float* __restrict__ u, float fineness, float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = ((u[i] - .5f) + 1.f) * fineness;
}

// Dynamic LDS handed to every (one-wave) block of the march kernel although it uses none: it caps how many march waves a CU keeps
// resident (160 KB / this), i.e. how many wave slots and registers the latency-bound march takes from the MLP / scatter
// kernels it runs underneath.  0 = no cap.  F2N_MARCH_LDS (bytes) overrides the default for experiments.
static size_t f2n_march_lds() {
#if F2N_DEBUG_BUILD
  static const size_t bytes = []() -> size_t {  // (read once: getenv is neither cheap nor safe against a concurrent setenv)
    const char* e = getenv("F2N_MARCH_LDS");
    const long v = e != nullptr ? atol(e) : 0;
    return (size_t) (v < 0 ? 0 : (v > 160 * 1024 ? 160 * 1024 : v));
  }();
  return bytes;
#else
  return 0;
#endif
}

extern "C" {

int f2n_oct_visible_cams(void* stream, int n_boxes, int n_cams, const float* boxes, const float* c2w, const float* bounds, float fx,
                         float fy, float cx, float cy, int res_h, int res_w, const float* pix_i, const float* pix_j,
                         uint8_t* visible) {
  if (n_boxes < 0 || n_cams < 0 || res_h <= 0 || res_w <= 0) return F2N_ERR_INVALID_ARG;
  if (n_boxes == 0 || n_cams == 0) return F2N_OK;
  if (n_cams > 65535) return F2N_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(visible_cams_kernel, dim3(n_boxes, n_cams), dim3(256), 0, (hipStream_t) stream, n_cams, boxes, c2w, bounds, fx, fy,
                     cx, cy, res_h, res_w, pix_i, pix_j, visible);
  return f2n_launch_status();
}

int f2n_march_noise(void* stream, int n, const float* u, float fineness, float* out) {
  if (n < 0) return F2N_ERR_INVALID_ARG;
  if (n == 0) return F2N_OK;
  hipLaunchKernelGGL(march_noise_kernel, dim3(f2n_div_up(n, 256)), dim3(256), 0, (hipStream_t) stream, n, u, fineness, out);
  return f2n_launch_status();
}

int f2n_normalize_dirs(void* stream, int n, const float* dirs, float* out) {
  if (n < 0) return F2N_ERR_INVALID_ARG;
  if (n == 0) return F2N_OK;
  hipLaunchKernelGGL(normalize_dirs_kernel, dim3(f2n_div_up(n, 256)), dim3(256), 0, (hipStream_t) stream, n, dirs, out);
  return f2n_launch_status();
}

int f2n_sampler_prologue(void* stream, int n_rays, const float* dirs, float* out, int32_t* zero, int n_zero, int n_noise,
                         const float* u, float fineness, float* noise_out) {
  if (n_rays < 0 || n_zero < 0 || n_noise < 0 || (n_zero > 0 && zero == nullptr) || (n_noise > 0 && (u == nullptr || noise_out == nullptr)))
    return F2N_ERR_INVALID_ARG;
  const int n = n_rays > n_noise ? (n_rays > n_zero ? n_rays : n_zero) : (n_noise > n_zero ? n_noise : n_zero);
  if (n == 0) return F2N_OK;
  hipLaunchKernelGGL(sampler_prologue_kernel, dim3(f2n_div_up(n, 256)), dim3(256), 0, (hipStream_t) stream, n_rays, dirs, out, zero,
                     n_zero, n_noise, u, fineness, noise_out, 0ull, 0ull);
  return f2n_launch_status();
}

int f2n_sampler_prologue_keyed(void* stream, int n_rays, const float* dirs, float* out, int32_t* zero, int n_zero, int n_noise,
                               uint64_t key, uint64_t seq, float fineness, float* noise_out) {
  if (n_rays < 0 || n_zero < 0 || n_noise < 1 || (n_zero > 0 && zero == nullptr) || noise_out == nullptr) return F2N_ERR_INVALID_ARG;
  const int n = n_rays > n_noise ? (n_rays > n_zero ? n_rays : n_zero) : (n_noise > n_zero ? n_noise : n_zero);
  hipLaunchKernelGGL(sampler_prologue_kernel, dim3(f2n_div_up(n, 256)), dim3(256), 0, (hipStream_t) stream, n_rays, dirs, out, zero,
                     n_zero, n_noise, (const float*) nullptr, fineness, noise_out, (unsigned long long) key, (unsigned long long) seq);
  return f2n_launch_status();
}

int f2n_oct_intersect_count(void* stream, int n_rays, int max_hits, const uint8_t* search_order, const float* rays_o,
                            const float* rays_d, float near_, float far_, const void* tree_nodes, int32_t* hit_counts,
                            const void* child_blocks) {
  if (n_rays < 0 || max_hits < 0) return F2N_ERR_INVALID_ARG;
  if (n_rays == 0) return F2N_OK;
  hipLaunchKernelGGL(oct_intersect_coop_kernel<0>, dim3(f2n_div_up(n_rays, F2N_COOP_RAYS_PER_BLOCK)), dim3(256), 0,
                     (hipStream_t) stream, n_rays, max_hits, search_order, rays_o, rays_d, near_, far_,
                     (const F2nTreeNode*) tree_nodes, nullptr, hit_counts, nullptr, nullptr, nullptr, nullptr, nullptr,
                     (const F2nChildInfo*) child_blocks, nullptr, 0, nullptr, nullptr, nullptr);
  return f2n_launch_status();
}

int f2n_oct_intersect_strided(void* stream, int n_rays, int max_hits, const uint8_t* search_order, const float* rays_o,
                              const float* rays_d, float near_, float far_, const void* tree_nodes, int32_t* oct_start_end,
                              int32_t* oct_idx, float* oct_near_far, int32_t* total, int32_t* oct_trans,
                              const void* child_blocks) {
  if (n_rays < 0 || max_hits < 1) return F2N_ERR_INVALID_ARG;
  if (n_rays == 0) return F2N_OK;
  hipLaunchKernelGGL(oct_intersect_coop_kernel<2>, dim3(f2n_div_up(n_rays, F2N_COOP_RAYS_PER_BLOCK)), dim3(256), 0,
                     (hipStream_t) stream, n_rays, max_hits, search_order, rays_o, rays_d, near_, far_,
                     (const F2nTreeNode*) tree_nodes, nullptr, nullptr, oct_idx, oct_near_far, oct_start_end, total, oct_trans,
                     (const F2nChildInfo*) child_blocks, nullptr, 0, nullptr, nullptr, nullptr);
  return f2n_launch_status();
}

// Interior nodes whose records fit next to the walk's stacks (36 KB) in a CU's 160 KB of LDS; at one 256-thread block per 32
// rays a batch of 8192 rays puts one block on every CU, so the rest of the LDS stays free for the kernels it runs underneath.
#define F2N_LDS_OCT_MAX_INTERIOR 448
int f2n_oct_lds_max_interior(void) { return F2N_LDS_OCT_MAX_INTERIOR; }

int f2n_oct_intersect_strided_lds(void* stream, int n_rays, int max_hits, const uint8_t* search_order, const float* rays_o,
                                  const float* rays_d, float near_, float far_, const void* tree_nodes, int32_t* oct_start_end,
                                  int32_t* oct_idx, float* oct_near_far, int32_t* total, int32_t* oct_trans,
                                  const void* child_blocks, const int32_t* interior_nodes, const int32_t* rank_of, int n_interior) {
  if (n_rays < 0 || max_hits < 1 || child_blocks == nullptr || interior_nodes == nullptr || rank_of == nullptr) return F2N_ERR_INVALID_ARG;
  if (n_interior < 1 || n_interior > F2N_LDS_OCT_MAX_INTERIOR) return F2N_ERR_UNSUPPORTED;
  if (n_rays == 0) return F2N_OK;
  static bool attr_set[16] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return F2N_ERR_INVALID_ARG;
  if (!attr_set[dev]) {  // more than the default 64 KB of LDS (static stacks + dynamic records) per block
    if (hipFuncSetAttribute((const void*) oct_intersect_coop_kernel<2, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            F2N_LDS_OCT_MAX_INTERIOR * 8 * (int) sizeof(F2nChildInfo)) != hipSuccess)
      return F2N_ERR_UNSUPPORTED;
    attr_set[dev] = true;
  }
  hipLaunchKernelGGL((oct_intersect_coop_kernel<2, true>), dim3(f2n_div_up(n_rays, F2N_COOP_RAYS_PER_BLOCK)), dim3(256),
                     (size_t) n_interior * 8 * sizeof(F2nChildInfo), (hipStream_t) stream, n_rays, max_hits, search_order, rays_o, rays_d,
                     near_, far_, (const F2nTreeNode*) tree_nodes, nullptr, nullptr, oct_idx, oct_near_far, oct_start_end, total,
                     oct_trans, (const F2nChildInfo*) child_blocks, nullptr, 0, nullptr, nullptr, nullptr, interior_nodes, rank_of,
                     n_interior);
  return f2n_launch_status();
}

int f2n_oct_intersect_repair(void* stream, int n_rays, int max_hits, const uint8_t* search_order, const float* rays_o,
                             const float* rays_d, float near_, float far_, const void* tree_nodes, int32_t* oct_start_end,
                             int32_t* oct_idx, float* oct_near_far, int32_t* total, int32_t* oct_trans, const void* child_blocks,
                             const int32_t* died_at, int spec_epoch, const int32_t* death_epoch, int32_t* repair_flags,
                             int32_t* n_repaired) {
  if (n_rays < 0 || max_hits < 1 || died_at == nullptr || death_epoch == nullptr || repair_flags == nullptr) return F2N_ERR_INVALID_ARG;
  if (n_rays == 0) return F2N_OK;
  hipLaunchKernelGGL(oct_intersect_coop_kernel<3>, dim3(f2n_div_up(n_rays, F2N_COOP_RAYS_PER_BLOCK)), dim3(256), 0,
                     (hipStream_t) stream, n_rays, max_hits, search_order, rays_o, rays_d, near_, far_,
                     (const F2nTreeNode*) tree_nodes, nullptr, nullptr, oct_idx, oct_near_far, oct_start_end, total, oct_trans,
                     (const F2nChildInfo*) child_blocks, died_at, spec_epoch, death_epoch, repair_flags, n_repaired);
  return f2n_launch_status();
}

int f2n_segment_scan_ex(void* stream, int n, const int32_t* counts, int32_t* start_end, int32_t* total, int32_t* mirror,
                        const int32_t* also, int n_also) {
  if (n < 0 || n_also < 0 || n_also > 4 || (n_also > 0 && (also == nullptr || mirror == nullptr))) return F2N_ERR_INVALID_ARG;
  hipLaunchKernelGGL(segment_scan_kernel, dim3(1), dim3(F2N_SCAN_THREADS), 0, (hipStream_t) stream, n, counts,
                     start_end, total, mirror, also, n_also);
  return f2n_launch_status();
}

int f2n_segment_scan(void* stream, int n, const int32_t* counts, int32_t* start_end, int32_t* total) {
  return f2n_segment_scan_ex(stream, n, counts, start_end, total, nullptr, nullptr, 0);
}

int f2n_oct_intersect_fill(void* stream, int n_rays, const uint8_t* search_order, const float* rays_o,
                           const float* rays_d, float near_, float far_, const void* tree_nodes,
                           const int32_t* oct_start_end, int32_t* oct_idx, float* oct_near_far, const void* child_blocks) {
  if (n_rays < 0) return F2N_ERR_INVALID_ARG;
  if (n_rays == 0) return F2N_OK;
  hipLaunchKernelGGL(oct_intersect_coop_kernel<1>, dim3(f2n_div_up(n_rays, F2N_COOP_RAYS_PER_BLOCK)), dim3(256), 0,
                     (hipStream_t) stream, n_rays, 0, search_order, rays_o, rays_d, near_, far_,
                     (const F2nTreeNode*) tree_nodes, oct_start_end, nullptr, oct_idx, oct_near_far, nullptr, nullptr, nullptr,
                     (const F2nChildInfo*) child_blocks, nullptr, 0, nullptr, nullptr, nullptr);
  return f2n_launch_status();
}

int f2n_ray_march_count(void* stream, int n_rays, float sample_l, int scale_by_dis, const float* rays_o,
                        const float* rays_d, const float* noise, const int32_t* oct_start_end, const int32_t* oct_idx,
                        const float* oct_near_far, const void* tree_nodes, const void* transes, int32_t* pts_counts) {
  if (n_rays < 0) return F2N_ERR_INVALID_ARG;
  if (n_rays == 0) return F2N_OK;
  hipLaunchKernelGGL(ray_march_kernel<0>, dim3(f2n_div_up(n_rays, 4)), dim3(64), 0,
                     (hipStream_t) stream, n_rays, sample_l, scale_by_dis, rays_o, rays_d, noise, oct_start_end, oct_idx,
                     oct_near_far, (const F2nTreeNode*) tree_nodes, (const F2nTransInfo*) transes, nullptr, pts_counts,
                     nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0);
  return f2n_launch_status();
}

int f2n_ray_march_fill(void* stream, int n_rays, float sample_l, int scale_by_dis, const float* rays_o,
                       const float* rays_d, const float* noise, const int32_t* oct_start_end, const int32_t* oct_idx,
                       const float* oct_near_far, const void* tree_nodes, const void* transes,
                       const int32_t* pts_start_end, float* pts, float* dirs, float* dt, float* t, int32_t* anchors,
                       float* first_oct_dis) {
  if (n_rays < 0) return F2N_ERR_INVALID_ARG;
  if (n_rays == 0) return F2N_OK;
  hipLaunchKernelGGL(ray_march_kernel<1>, dim3(f2n_div_up(n_rays, 4)), dim3(64), 0,
                     (hipStream_t) stream, n_rays, sample_l, scale_by_dis, rays_o, rays_d, noise, oct_start_end, oct_idx,
                     oct_near_far, (const F2nTreeNode*) tree_nodes, (const F2nTransInfo*) transes, pts_start_end, nullptr,
                     pts, dirs, dt, t, anchors, first_oct_dis, nullptr, nullptr, nullptr, 0);
  return f2n_launch_status();
}

int f2n_ray_march_strided(void* stream, int n_rays, float sample_l, int scale_by_dis, const float* rays_o, const float* rays_d,
                          const float* noise, const int32_t* oct_start_end, const int32_t* oct_idx, const float* oct_near_far,
                          const void* tree_nodes, const void* transes, int32_t* pts_counts, float* s_pts, float* s_dt, float* s_t,
                          int32_t* s_anchors, float* first_oct_dis, const int32_t* oct_trans) {
  if (n_rays < 0) return F2N_ERR_INVALID_ARG;
  if (n_rays == 0) return F2N_OK;
  hipLaunchKernelGGL(ray_march_kernel<2>, dim3(f2n_div_up(n_rays, 4)), dim3(64), f2n_march_lds(),
                     (hipStream_t) stream, n_rays, sample_l, scale_by_dis, rays_o, rays_d, noise, oct_start_end, oct_idx,
                     oct_near_far, (const F2nTreeNode*) tree_nodes, (const F2nTransInfo*) transes, nullptr, pts_counts, s_pts,
                     nullptr, s_dt, s_t, s_anchors, first_oct_dis, oct_trans, nullptr, nullptr, 0);
  return f2n_launch_status();
}

int f2n_ray_march_repair(void* stream, int n_rays, float sample_l, int scale_by_dis, const float* rays_o, const float* rays_d,
                         const float* noise, const int32_t* oct_start_end, const int32_t* oct_idx, const float* oct_near_far,
                         const void* tree_nodes, const void* transes, int32_t* pts_counts, float* s_pts, float* s_dt, float* s_t,
                         int32_t* s_anchors, float* first_oct_dis, const int32_t* oct_trans, const int32_t* repair_flags,
                         const int32_t* death_epoch, int spec_epoch) {
  if (n_rays < 0 || repair_flags == nullptr || death_epoch == nullptr) return F2N_ERR_INVALID_ARG;
  if (n_rays == 0) return F2N_OK;
  hipLaunchKernelGGL(ray_march_kernel<2>, dim3(f2n_div_up(n_rays, 4)), dim3(64), f2n_march_lds(),
                     (hipStream_t) stream, n_rays, sample_l, scale_by_dis, rays_o, rays_d, noise, oct_start_end, oct_idx,
                     oct_near_far, (const F2nTreeNode*) tree_nodes, (const F2nTransInfo*) transes, nullptr, pts_counts, s_pts,
                     nullptr, s_dt, s_t, s_anchors, first_oct_dis, oct_trans, repair_flags, death_epoch, spec_epoch);
  return f2n_launch_status();
}

int f2n_ray_march_strided_rec(void* stream, int n_rays, int max_hits, float sample_l, int scale_by_dis, const float* rays_o,
                              const float* rays_d, const float* noise, const int32_t* oct_start_end, const int32_t* oct_idx,
                              const float* oct_near_far, const void* tree_nodes, const void* transes, int32_t* pts_counts, float* s_pts,
                              float* s_dt, float* s_t, int32_t* s_anchors, float* first_oct_dis, const int32_t* oct_trans,
                              void* leaf_state, int32_t* reached) {
  if (n_rays < 0 || leaf_state == nullptr || reached == nullptr || max_hits < 1 || max_hits > 2048) return F2N_ERR_INVALID_ARG;
  if (n_rays == 0) return F2N_OK;
  hipLaunchKernelGGL((ray_march_kernel<2, true>), dim3(f2n_div_up(n_rays, 4)), dim3(64), f2n_march_lds(),
                     (hipStream_t) stream, n_rays, sample_l, scale_by_dis, rays_o, rays_d, noise, oct_start_end, oct_idx,
                     oct_near_far, (const F2nTreeNode*) tree_nodes, (const F2nTransInfo*) transes, nullptr, pts_counts, s_pts,
                     nullptr, s_dt, s_t, s_anchors, first_oct_dis, oct_trans, nullptr, nullptr, 0, (uint2*) leaf_state, reached);
  return f2n_launch_status();
}

int f2n_ray_march_persistent(void* stream, int n_rays, int max_hits, int n_blocks, int block_waves, float sample_l, int scale_by_dis, const float* rays_o,
                             const float* rays_d, const float* noise, const int32_t* oct_start_end, const int32_t* oct_idx,
                             const float* oct_near_far, const void* tree_nodes, const void* transes, int32_t* pts_counts, float* s_pts,
                             float* s_dt, float* s_t, int32_t* s_anchors, float* first_oct_dis, const int32_t* oct_trans,
                             void* leaf_state, int32_t* reached, int32_t* order, int32_t* counter) {
  if (n_rays < 0 || n_blocks < 1 || block_waves < 1 || block_waves > 16 || max_hits < 1 || max_hits > 2048 || order == nullptr ||
      counter == nullptr || (leaf_state == nullptr) != (reached == nullptr))
    return F2N_ERR_INVALID_ARG;
  if (n_rays == 0) return F2N_OK;
  hipLaunchKernelGGL(sort_rays_by_hits_kernel, dim3(1), dim3(1024), 0, (hipStream_t) stream, n_rays, max_hits, oct_start_end, order,
                     counter);
  const int waves = min(n_blocks, f2n_div_up(n_rays, 4));
  const int blocks = f2n_div_up(waves, block_waves), threads = 64 * min(block_waves, waves);
  if (leaf_state != nullptr) {
    hipLaunchKernelGGL(ray_march_persistent_kernel<true>, dim3(blocks), dim3(threads), 0, (hipStream_t) stream, n_rays, sample_l,
                       scale_by_dis, rays_o, rays_d, noise, oct_start_end, oct_idx, oct_near_far, (const F2nTreeNode*) tree_nodes,
                       (const F2nTransInfo*) transes, pts_counts, s_pts, s_dt, s_t, s_anchors, first_oct_dis, oct_trans,
                       (uint2*) leaf_state, reached, order, counter);
  } else {
    hipLaunchKernelGGL(ray_march_persistent_kernel<false>, dim3(blocks), dim3(threads), 0, (hipStream_t) stream, n_rays, sample_l,
                       scale_by_dis, rays_o, rays_d, noise, oct_start_end, oct_idx, oct_near_far, (const F2nTreeNode*) tree_nodes,
                       (const F2nTransInfo*) transes, pts_counts, s_pts, s_dt, s_t, s_anchors, first_oct_dis, oct_trans, nullptr,
                       nullptr, order, counter);
  }
  return f2n_launch_status();
}

int f2n_ray_march_repair_tail(void* stream, int n_rays, int max_hits, float sample_l, int scale_by_dis, const float* rays_o,
                              const float* rays_d, const float* noise, const int32_t* oct_start_end, const int32_t* oct_idx,
                              const float* oct_near_far, const void* tree_nodes, const void* transes, int32_t* pts_counts, float* s_pts,
                              float* s_dt, float* s_t, int32_t* s_anchors, float* first_oct_dis, const int32_t* oct_trans,
                              void* leaf_state, int32_t* reached, const int32_t* repair_from, const int32_t* death_epoch,
                              int spec_epoch) {
  if (n_rays < 0 || leaf_state == nullptr || reached == nullptr || repair_from == nullptr || death_epoch == nullptr || max_hits < 1 ||
      max_hits > 2048)
    return F2N_ERR_INVALID_ARG;
  if (n_rays == 0) return F2N_OK;
  hipLaunchKernelGGL((ray_march_kernel<2, true>), dim3(f2n_div_up(n_rays, 4)), dim3(64), f2n_march_lds(),
                     (hipStream_t) stream, n_rays, sample_l, scale_by_dis, rays_o, rays_d, noise, oct_start_end, oct_idx,
                     oct_near_far, (const F2nTreeNode*) tree_nodes, (const F2nTransInfo*) transes, nullptr, pts_counts, s_pts,
                     nullptr, s_dt, s_t, s_anchors, first_oct_dis, oct_trans, repair_from, death_epoch, spec_epoch,
                     (uint2*) leaf_state, reached);
  return f2n_launch_status();
}

int f2n_oct_list_repair(void* stream, int n_rays, int max_hits, int32_t* oct_start_end, int32_t* oct_idx, float* oct_near_far,
                        int32_t* oct_trans, int32_t* total, const int32_t* died_at, int spec_epoch, const int32_t* death_epoch,
                        const int32_t* reached, int32_t* repair_from, int32_t* n_repaired, int32_t* n_full) {
  if (n_rays < 0 || max_hits < 1 || died_at == nullptr || death_epoch == nullptr || reached == nullptr || repair_from == nullptr ||
      n_full == nullptr || total == nullptr)
    return F2N_ERR_INVALID_ARG;
  if (n_rays == 0) return F2N_OK;
  hipLaunchKernelGGL(oct_list_repair_kernel, dim3(f2n_div_up(n_rays, 16)), dim3(256), 0, (hipStream_t) stream, n_rays, max_hits,
                     oct_start_end, oct_idx, oct_near_far, oct_trans, total, died_at, spec_epoch, death_epoch, reached, repair_from,
                     n_repaired, n_full);
  return f2n_launch_status();
}

int f2n_oct_intersect_repair_flagged(void* stream, int n_rays, int max_hits, const uint8_t* search_order, const float* rays_o,
                                     const float* rays_d, float near_, float far_, const void* tree_nodes, int32_t* oct_start_end,
                                     int32_t* oct_idx, float* oct_near_far, int32_t* total, int32_t* oct_trans,
                                     const void* child_blocks, const int32_t* death_epoch, int spec_epoch, int32_t* repair_from,
                                     const int32_t* n_full) {
  if (n_rays < 0 || max_hits < 1 || death_epoch == nullptr || repair_from == nullptr || n_full == nullptr) return F2N_ERR_INVALID_ARG;
  if (n_rays == 0) return F2N_OK;
  hipLaunchKernelGGL(oct_intersect_coop_kernel<3>, dim3(f2n_div_up(n_rays, F2N_COOP_RAYS_PER_BLOCK)), dim3(256), 0,
                     (hipStream_t) stream, n_rays, max_hits, search_order, rays_o, rays_d, near_, far_,
                     (const F2nTreeNode*) tree_nodes, nullptr, nullptr, oct_idx, oct_near_far, oct_start_end, total, oct_trans,
                     (const F2nChildInfo*) child_blocks, nullptr, spec_epoch, death_epoch, repair_from, nullptr, nullptr, nullptr, 0,
                     n_full);
  return f2n_launch_status();
}

int f2n_pack_samples_repair(void* stream, int n_rays, const int32_t* pts_start_end, const float* rays_o, const float* rays_d,
                            const void* transes, const float* s_pts, const float* s_dt, const float* s_t, const int32_t* s_anchors,
                            float* pts, float* dirs, float* dt, float* t, int32_t* anchors, const int32_t* death_epoch,
                            int spec_epoch) {
  if (n_rays < 0 || (s_pts == nullptr && (rays_o == nullptr || transes == nullptr))) return F2N_ERR_INVALID_ARG;
  if (n_rays == 0) return F2N_OK;
  hipLaunchKernelGGL(pack_samples_kernel, dim3(f2n_div_up(n_rays, 4)), dim3(256), 0, (hipStream_t) stream, n_rays, pts_start_end,
                     rays_o, rays_d, (const F2nTransInfo*) transes, s_pts, s_dt, s_t, s_anchors, pts, dirs, dt, t, anchors,
                     death_epoch, spec_epoch);
  return f2n_launch_status();
}

int f2n_pack_samples(void* stream, int n_rays, const int32_t* pts_start_end, const float* rays_o, const float* rays_d,
                     const void* transes, const float* s_pts, const float* s_dt, const float* s_t, const int32_t* s_anchors,
                     float* pts, float* dirs, float* dt, float* t, int32_t* anchors) {
  return f2n_pack_samples_repair(stream, n_rays, pts_start_end, rays_o, rays_d, transes, s_pts, s_dt, s_t, s_anchors, pts, dirs, dt, t,
                                 anchors, nullptr, 0);
}

int f2n_edge_samples(void* stream, int n_pts, const void* edge_pool, const void* transes, const int32_t* edge_idx,
                     const float* edge_coords, float* out_pts, int32_t* out_idx) {
  return f2n_edge_samples_ex(stream, n_pts, edge_pool, 0, transes, edge_idx, edge_coords, nullptr, out_pts, out_idx, 1, nullptr,
                             nullptr, 1);
}

int f2n_edge_samples_ex(void* stream, int n_pts, const void* edge_pool, int n_edges, const void* transes, const int32_t* edge_idx,
                        const float* edge_coords, const float* u01, float* out_pts, int32_t* out_idx, int idx_stride,
                        float* out_pts2, int32_t* out_idx2, int idx_stride2) {
  if (n_pts < 0 || idx_stride < 1 || idx_stride2 < 1 || (u01 != nullptr && n_edges < 1) || (out_pts2 != nullptr) != (out_idx2 != nullptr))
    return F2N_ERR_INVALID_ARG;
  if (n_pts == 0) return F2N_OK;
  if (u01 == nullptr && (edge_idx == nullptr || edge_coords == nullptr)) return F2N_ERR_INVALID_ARG;
  hipLaunchKernelGGL(edge_samples_kernel, dim3(f2n_div_up(n_pts, 64)), dim3(64), 0, (hipStream_t) stream, n_pts,
                     (const F2nEdgePool*) edge_pool, n_edges, (const F2nTransInfo*) transes, edge_idx, edge_coords, u01, out_pts,
                     out_idx, idx_stride, out_pts2, out_idx2, idx_stride2);
  return f2n_launch_status();
}

int f2n_oct_mark_visit(void* stream, int n_rays, int n_nodes, const int32_t* pts_start_end, const int32_t* anchors,
                       int anchor_stride, const float* weights, const float* alphas, int32_t* w_adder, int32_t* a_adder,
                       int32_t* mark, int32_t* visit_cnt) {
  if (n_rays < 0 || n_nodes < 1 || anchor_stride < 2) return F2N_ERR_INVALID_ARG;
  if (n_rays == 0) return F2N_OK;
  const size_t lds = sizeof(int32_t) * 3 * (size_t) n_nodes;
  if (lds <= 60 * 1024) {  // block-local vote combining in LDS (12 B per octree node; 64 KB default LDS limit)
    hipLaunchKernelGGL(mark_visit_kernel<true>, dim3(f2n_div_up(n_rays, 64)), dim3(1024), lds, (hipStream_t) stream, n_rays,
                       n_nodes, pts_start_end, anchors, anchor_stride, weights, alphas, w_adder, a_adder, mark, visit_cnt);
  } else {
    hipLaunchKernelGGL(mark_visit_kernel<false>, dim3(f2n_div_up(n_rays, 16)), dim3(256), 0, (hipStream_t) stream, n_rays,
                       n_nodes, pts_start_end, anchors, anchor_stride, weights, alphas, w_adder, a_adder, mark, visit_cnt);
  }
  return f2n_launch_status();
}

int f2n_early_stop_votes(void* stream, int n_rays, const int32_t* pts_start_end, const float* f0, int f0_stride, const float* dt,
                         float* weights, float* alphas, int32_t* mask, int32_t* kept, int n_nodes, const int32_t* anchors,
                         int anchor_stride, int32_t* w_adder, int32_t* a_adder, int32_t* mark, int32_t* visit_cnt) {
  if (n_rays < 0 || n_nodes < 1 || anchor_stride < 2 || f0_stride < 1) return F2N_ERR_INVALID_ARG;
  if (n_rays == 0) return F2N_OK;
  const size_t lds = sizeof(int32_t) * 3 * (size_t) n_nodes;
  if (lds <= 60 * 1024) {
    hipLaunchKernelGGL(early_stop_votes_kernel<true>, dim3(f2n_div_up(n_rays, 64)), dim3(1024), lds, (hipStream_t) stream, n_rays,
                       n_nodes, pts_start_end, f0, f0_stride, dt, weights, alphas, mask, kept, anchors, anchor_stride, w_adder,
                       a_adder, mark, visit_cnt);
  } else {
    hipLaunchKernelGGL(early_stop_votes_kernel<false>, dim3(f2n_div_up(n_rays, 16)), dim3(256), 0, (hipStream_t) stream, n_rays,
                       n_nodes, pts_start_end, f0, f0_stride, dt, weights, alphas, mask, kept, anchors, anchor_stride, w_adder,
                       a_adder, mark, visit_cnt);
  }
  return f2n_launch_status();
}

int f2n_oct_update_stats(void* stream, int n_nodes, int32_t* w_adder, int32_t* a_adder, int32_t* mark, int32_t* w_stats,
                         int32_t* a_stats, void* tree_nodes, void* child_blocks, int reset_votes) {
  return f2n_oct_update_stats_ex(stream, n_nodes, w_adder, a_adder, mark, w_stats, a_stats, tree_nodes, child_blocks, reset_votes,
                                 nullptr, 0, nullptr, nullptr);
}

int f2n_oct_update_stats_ex(void* stream, int n_nodes, int32_t* w_adder, int32_t* a_adder, int32_t* mark, int32_t* w_stats,
                            int32_t* a_stats, void* tree_nodes, void* child_blocks, int reset_votes, int32_t* died_at, int epoch,
                            int32_t* death_epoch, int32_t* death_epoch_host) {
  if (n_nodes < 0 || (died_at != nullptr) != (death_epoch != nullptr)) return F2N_ERR_INVALID_ARG;
  if (n_nodes == 0) return F2N_OK;
  const F2nStatsArgs a = {n_nodes, w_adder, a_adder, mark, w_stats, a_stats, (F2nTreeNode*) tree_nodes, (F2nChildInfo*) child_blocks, reset_votes,
                          died_at, epoch, death_epoch, death_epoch_host};
  hipLaunchKernelGGL(update_stats_kernel, dim3(f2n_div_up(n_nodes, 256)), dim3(256), 0, (hipStream_t) stream, a);
  return f2n_launch_status();
}

int f2n_oct_update_stats_scan(void* stream, int n_nodes, int32_t* w_adder, int32_t* a_adder, int32_t* mark, int32_t* w_stats,
                              int32_t* a_stats, void* tree_nodes, void* child_blocks, int reset_votes, int32_t* died_at, int epoch,
                              int32_t* death_epoch, int32_t* death_epoch_host, int n, const int32_t* counts, int32_t* start_end,
                              int32_t* total, int32_t* mirror, const int32_t* also, int n_also) {
  if (n_nodes < 0 || (died_at != nullptr) != (death_epoch != nullptr) || n < 0 || n_also < 0 || n_also > 4 ||
      (n_also > 0 && (also == nullptr || mirror == nullptr)))
    return F2N_ERR_INVALID_ARG;
  const F2nStatsArgs a = {n_nodes, w_adder, a_adder, mark, w_stats, a_stats, (F2nTreeNode*) tree_nodes, (F2nChildInfo*) child_blocks, reset_votes,
                          died_at, epoch, death_epoch, death_epoch_host};
  hipLaunchKernelGGL(stats_and_scan_kernel, dim3(1 + f2n_div_up(n_nodes, F2N_SCAN_THREADS)), dim3(F2N_SCAN_THREADS), 0, (hipStream_t) stream, a,
                     n, counts, start_end, total, mirror, also, n_also);
  return f2n_launch_status();
}

int f2n_oct_build_child_blocks(void* stream, int n_nodes, const void* tree_nodes, void* child_blocks) {
  if (n_nodes < 0) return F2N_ERR_INVALID_ARG;
  if (n_nodes == 0) return F2N_OK;
  hipLaunchKernelGGL(build_child_blocks_kernel, dim3(f2n_div_up((long) n_nodes * 8, 256)), dim3(256), 0, (hipStream_t) stream,
                     n_nodes, (const F2nTreeNode*) tree_nodes, (F2nChildInfo*) child_blocks);
  return f2n_launch_status();
}

int f2n_oct_mark_invisible(void* stream, int n_nodes, int n_cams, void* tree_nodes, const float* intris,
                           const float* w2cs, const float* bounds) {
  if (n_nodes < 0 || n_cams < 0) return F2N_ERR_INVALID_ARG;
  if (n_nodes == 0) return F2N_OK;
  hipLaunchKernelGGL(mark_invisible_kernel, dim3(f2n_div_up(n_nodes, 256)), dim3(256), 0, (hipStream_t) stream, n_nodes,
                     n_cams, (F2nTreeNode*) tree_nodes, intris, w2cs, bounds);
  return f2n_launch_status();
}

}  // extern "C"
