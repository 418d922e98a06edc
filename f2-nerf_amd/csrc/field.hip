// Hash3DAnchored on gfx950: the 16-level anchored hash-grid gather / scatter (Field/Hash3DAnchored.cu:11-233)
// fused with the density MLP (Field/TCNNWP.cpp call sites).  One wave64 owns 16 samples at a time; lane
// (c = sample, g = lane >> 4) gathers the four levels {2g, 2g+1, 8+2g, 9+2g} of its sample, which are exactly
// the K-slots of the MFMA fragment it has to supply (mlp_dev.h), so features go from the L2/Infinity-Cache
// resident table straight into matrix-core operands: no LDS staging, no HBM round trip of the [n,32] features.
#include <math.h>
#include <atomic>
#include <string.h>

#include "adam_dev.h"
#include "mlp_dev.h"

struct F2nHashArgs {
  const half_t* table;        // h16 [pool, 2]
  const int32_t* prim_pool;   // [16, V, 3]
  const float* bias_pool;     // [16*V, 3]
  int n_volumes;
};

struct F2nLevelTab {  // staged in LDS once per block
  float scale[F2N_N_LEVELS];
  int32_t base[F2N_N_LEVELS];   // level offset in HALVES (Hash3DAnchored.cu:37: added to a half pointer)
  uint32_t size[F2N_N_LEVELS];  // entries addressed per level
};

__device__ __forceinline__ void f2n_level_tab_fill(F2nLevelTab& s, const float* level_scale, const int32_t* local_idx,
                                                   const int32_t* local_size, int tid) {
  if (tid < F2N_N_LEVELS) {
    s.scale[tid] = level_scale[tid];
    s.base[tid] = local_idx[tid];
    s.size[tid] = (uint32_t) local_size[tid];
  }
}

// The level a lane group serves for feature-pair j = 0..3 (features sigma(g, 2j..2j+1)).
__device__ __forceinline__ int f2n_level_of(int g, int j) { return (j < 2) ? 2 * g + j : 8 + 2 * g + (j - 2); }

struct F2nCell {
  uint32_t pos[8];
  float w[8];
  uint32_t p[3];  // integer cell coordinates: with the transform index, the identity of the cell
};

// Cell lookup of one (point, level): Hash3DAnchored.cu:27-69.  p01 is the query point in [0,1] coordinates.
__device__ __forceinline__ void f2n_hash_cell(const float* p01, float mul, const int32_t* __restrict__ prim3,
                                              const float* __restrict__ bias3, uint32_t lsize, F2nCell& cell) {
  float q[3], fl[3];
#pragma unroll
  for (int k = 0; k < 3; k++) {
    q[k] = p01[k] * mul + bias3[k];
    fl[k] = floorf(q[k]);
  }
  const uint32_t px = f2n_f2u_sat(fl[0]), py = f2n_f2u_sat(fl[1]), pz = f2n_f2u_sat(fl[2]);
  const uint32_t pa = (uint32_t) prim3[0], pb = (uint32_t) prim3[1], pc = (uint32_t) prim3[2];
  cell.p[0] = px;
  cell.p[1] = py;
  cell.p[2] = pz;
  const uint32_t hx[2] = {px * pa, (px + 1u) * pa}, hy[2] = {py * pb, (py + 1u) * pb}, hz[2] = {pz * pc, (pz + 1u) * pc};
  const bool pow2 = (lsize & (lsize - 1u)) == 0u;
#pragma unroll
  for (int corner = 0; corner < 8; corner++) {  // 000,001,010,011,100,101,110,111 = (x,y,z) bits
    const uint32_t h = hx[(corner >> 2) & 1] ^ hy[(corner >> 1) & 1] ^ hz[corner & 1];
    cell.pos[corner] = pow2 ? (h & (lsize - 1u)) : (h % lsize);
  }
  const float a = q[0] - fl[0], b = q[1] - fl[1], c = q[2] - fl[2];
  cell.w[0] = (1.f - a) * (1.f - b) * (1.f - c);
  cell.w[1] = (1.f - a) * (1.f - b) * c;
  cell.w[2] = (1.f - a) * b * (1.f - c);
  cell.w[3] = (1.f - a) * b * c;
  cell.w[4] = a * (1.f - b) * (1.f - c);
  cell.w[5] = a * (1.f - b) * c;
  cell.w[6] = a * b * (1.f - c);
  cell.w[7] = a * b * c;
}

// XCD-aware level-partitioned gather.  The 8 XCDs of an MI355X have private 4 MiB L2s; a wave that gathers all 16
// levels touches the whole 17 MiB of addressed table and runs at the chip's random-access rate out of the Infinity
// Cache (~90 G 4-byte gathers/s measured, tools/xcd_probe.py).  Here the blocks of XCD x (x = blockIdx % 8, where the
// dispatcher is observed to place them) serve ONE level pair (2p, 2p+1) at a time, so an L2 only ever sees a 3 MiB slice.
// Placement is a speed assumption only: any mapping gives the same result.  Features leave through f16 "planes"
// [8][n][4] (8 B per sample and level pair, coalesced), from which the MLP kernel's lane (c,g) reads exactly its
// K-slots: plane g (features 4g..4g+3) and plane 4+g (features 16+4g..).
//
// What bounds it (tools/gather_ab.py experiments, profiles/r02_gather_experiments.txt): every one of the 128 reads of a
// sample pulls a whole 128-byte line from the XCD's L2 into a CU's L1 -- 1.04e8 lines in 0.42 ms = 31.6 TB/s of the
// ~34.5 TB/s the eight L2s deliver together (MI355X_MICROARCH.md).  The same kernel without its table reads takes
// 0.054 ms, with the eight reads of a cell redirected into ONE line 0.128 ms.  So the only lever is fewer distinct lines:
//
// Run combining (COMBINE): consecutive samples of a ray that fall into the same cell of a level (same integer cell, same
// warp) read the same eight entries; only the first lane of each such run inside the wave issues the reads, the others
// take the values over the LDS crossbar (ds_bpermute) and blend with their own weights -- bit-identical features.  The
// L1 already absorbs most of these repeats, so this alone buys little; what it makes visible is how unequal the level
// pairs are: on a converged scene (march fineness 1) pair (0,1) needs ~10 % of the lines of pair (14,15).
//
// Balanced split: with one pair pinned to each XCD the kernel lasts as long as the finest pair while the coarse pairs'
// XCDs idle.  The launcher therefore cuts the 8 x n_tiles (pair, 256-sample tile) units into 8 contiguous, equally
// EXPENSIVE shares (cost model: expected fraction of run-heading lanes per level from the march step, f2n_gather_costs)
// and hands XCD x share x as up to F2N_MAX_SEGS (pair, tile range) segments; its blocks walk the share's tiles
// block-cyclically, re-staging the hash constants when they cross into the next pair.
#define F2N_N_PARTS 8
#define F2N_MAX_SEGS 8
struct F2nGatherPlan {
  int32_t n_seg[F2N_N_PARTS];
  int32_t pair[F2N_N_PARTS][F2N_MAX_SEGS];
  int32_t t0[F2N_N_PARTS][F2N_MAX_SEGS];   // first tile of the segment
  int32_t v0[F2N_N_PARTS][F2N_MAX_SEGS];   // position of that tile in the XCD's virtual tile sequence
  int32_t n_virtual[F2N_N_PARTS];
};

// STAGED: the per-(level, transform) hash constants of the current level pair (prim_pool / bias_pool rows, 24 B each) are
// copied into LDS: 12 of the 32 vector-memory instructions a sample issues were those -- broadcast loads that cost no
// bandwidth but an address-path issue slot each.
template <bool STAGED, bool COMBINE>
__global__ __launch_bounds__(256) void hash_gather_planes_kernel(
    int n, F2nHashArgs h, const int32_t* __restrict__ local_idx, const int32_t* __restrict__ local_size,
    const float* __restrict__ level_scale, const float* __restrict__ pts, int pts_are_warped,
    const int32_t* __restrict__ volume_idx, int vol_stride, half_t* __restrict__ planes, F2nGatherPlan plan) {
  __shared__ F2nLevelTab lt;
  extern __shared__ uint32_t pb_lds[];  // STAGED: [2 levels][V][prim xyz, bias xyz]
  const int tid = threadIdx.x, lane = tid & 63;
  const int xcd = blockIdx.x % F2N_N_PARTS, q = blockIdx.x / F2N_N_PARTS, nq = gridDim.x / F2N_N_PARTS;
  f2n_level_tab_fill(lt, level_scale, local_idx, local_size, tid);
  int seg = -1, staged_pair = -1;
  for (int vt = q; vt < plan.n_virtual[xcd]; vt += nq) {
    while (seg + 1 < plan.n_seg[xcd] && vt >= plan.v0[xcd][seg + 1]) seg++;  // block-uniform
    const int part = plan.pair[xcd][seg];
    const int tile = plan.t0[xcd][seg] + (vt - plan.v0[xcd][seg]);
    // Levels {2 part, 2 part + 1}: the two feature pairs are one 8-byte plane element.  (Pairing a coarse with a fine
    // level, {part, 15 - part}, to even out L1 hit rates was measured 8 % SLOWER with the fixed split: the split 4-byte
    // stores cost more than the balance gained.)
    const int lv[2] = {2 * part, 2 * part + 1};
    if (part != staged_pair) {
      __syncthreads();  // everyone is done with the previous pair's constants (and, the first time, lt is filled)
      if (STAGED) {
        for (int i = tid; i < 2 * 3 * h.n_volumes; i += 256) {
          const int j = i >= 3 * h.n_volumes, i1 = i - j * 3 * h.n_volumes;
          const int r = i1 / 3, k = i1 - 3 * r;
          const size_t src = 3 * (size_t) lv[j] * h.n_volumes + i1;
          pb_lds[6 * (j * h.n_volumes + r) + k] = (uint32_t) h.prim_pool[src];
          pb_lds[6 * (j * h.n_volumes + r) + 3 + k] = __float_as_uint(h.bias_pool[src]);
        }
        __syncthreads();
      }
      staged_pair = part;
    }
    const int s = tile * 256 + tid;
    const bool valid = s < n;
    if (!COMBINE && !valid) continue;
    const int sc = valid ? s : n - 1;  // COMBINE: every lane takes part (tail lanes extend the last sample's run)
    float p01[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
      const float p = pts[3 * (size_t) sc + k];
      p01[k] = pts_are_warped ? (p + 1.f) * .5f : p;
    }
    const int vol = volume_idx[(size_t) sc * vol_stride];
    F2nCell cell[2];
    half2_t v[2][8];
    bool head[2] = {true, true};
#pragma unroll
    for (int j = 0; j < 2; j++) {
      const int l = lv[j];
      if (STAGED) {
        const uint32_t* row = pb_lds + 6 * (j * h.n_volumes + vol);
        const int32_t prim3[3] = {(int32_t) row[0], (int32_t) row[1], (int32_t) row[2]};
        const float bias3[3] = {__uint_as_float(row[3]), __uint_as_float(row[4]), __uint_as_float(row[5])};
        f2n_hash_cell(p01, lt.scale[l], prim3, bias3, lt.size[l], cell[j]);
      } else {
        const int tf = l * h.n_volumes + vol;
        f2n_hash_cell(p01, lt.scale[l], h.prim_pool + 3 * tf, h.bias_pool + 3 * tf, lt.size[l], cell[j]);
      }
      if (COMBINE) {  // same cell as the previous lane of the wave?  (DPP wave_shr:1; lane 0 always starts a run)
        const int q0 = __builtin_amdgcn_update_dpp(0, (int) cell[j].p[0], 0x138, 0xF, 0xF, false);
        const int q1 = __builtin_amdgcn_update_dpp(0, (int) cell[j].p[1], 0x138, 0xF, 0xF, false);
        const int q2 = __builtin_amdgcn_update_dpp(0, (int) cell[j].p[2], 0x138, 0xF, 0xF, false);
        const int qv = __builtin_amdgcn_update_dpp(-1, vol, 0x138, 0xF, 0xF, false);
        head[j] = !(lane > 0 && q0 == (int) cell[j].p[0] && q1 == (int) cell[j].p[1] && q2 == (int) cell[j].p[2] && qv == vol);
      }
      const half2_t* base = (const half2_t*) (h.table + lt.base[l]);
      if (head[j]) {
#pragma unroll
        for (int d = 0; d < 8; d++) v[j][d] = base[cell[j].pos[d]];
      }
    }
    if (COMBINE) {
#pragma unroll
      for (int j = 0; j < 2; j++) {
        const unsigned long long hm = __ballot(head[j]);
        if (hm != ~0ull) {  // wave-uniform: some lane rides on its run's first lane
          const int src = 63 - __clzll(hm & ((2ull << lane) - 1ull));
#pragma unroll
          for (int d = 0; d < 8; d++)
            v[j][d] = __builtin_bit_cast(half2_t, __shfl(__builtin_bit_cast(int, v[j][d]), src));
        }
      }
    }
    half4_t out;
#pragma unroll
    for (int j = 0; j < 2; j++) {  // same fp32 summation order as f2n_gather_frag / the oracle
      float s0 = cell[j].w[0] * (float) v[j][0][0];
      float s1 = cell[j].w[0] * (float) v[j][0][1];
#pragma unroll
      for (int d = 1; d < 8; d++) {
        s0 = s0 + cell[j].w[d] * (float) v[j][d][0];
        s1 = s1 + cell[j].w[d] * (float) v[j][d][1];
      }
      out[2 * j] = (half_t) s0;
      out[2 * j + 1] = (half_t) s1;
    }
    if (valid) *(half4_t*) (planes + ((size_t) part * n + s) * 4) = out;
  }
}

// Expected relative cost of one tile of each level pair when consecutive samples are `step01` apart in the [0,1] hash
// space: a lane heads a run (issues its 8 reads) when the step crossed a cell boundary of the level (probability
// ~1 - exp(-1.7 * cells per step): three axes), the warp changed (~4 % of steps) or the wave starts (1/64); on top, the
// part of a tile that does not depend on the reads (hashing, blending, streams: 0.13 of a full level).  Checked against
// the measured head fractions of a converged fox batch (0.09 / 0.40 / 0.87 / 1.0 at levels 0 / 8 / 12 / 15).
static void f2n_gather_costs(float step01, const float* level_scale_host, float* cost8) {
  for (int p = 0; p < F2N_N_PARTS; p++) {
    float c = 0.f;
    for (int j = 0; j < 2; j++) {
      const float cells = level_scale_host[2 * p + j] * step01;
      c += 0.13f + (1.f - (1.f - 1.f / 64.f - 0.04f) * expf(-1.7f * cells));
    }
    cost8[p] = c;
  }
}

// Cuts the (pair, tile) units into 8 contiguous shares of equal cost, finest pair first.
static void f2n_gather_plan(int n_tiles, const float* cost8 /* NULL: one pair per XCD */, F2nGatherPlan& plan) {
  memset(&plan, 0, sizeof(plan));
  if (cost8 == nullptr) {
    for (int x = 0; x < F2N_N_PARTS; x++) {
      plan.n_seg[x] = 1;
      plan.pair[x][0] = x;
      plan.n_virtual[x] = n_tiles;
    }
    return;
  }
  double total = 0;
  for (int p = 0; p < F2N_N_PARTS; p++) total += (double) cost8[p] * n_tiles;
  const double share = total / F2N_N_PARTS;
  int x = 0;
  double room = share;
  for (int p = F2N_N_PARTS - 1; p >= 0; p--) {
    int t = 0;
    while (t < n_tiles) {
      int take = (x == F2N_N_PARTS - 1) ? n_tiles - t : (int) ceil(room / cost8[p] - 1e-9);
      if (take > n_tiles - t) take = n_tiles - t;
      if (take > 0 && plan.n_seg[x] < F2N_MAX_SEGS) {
        const int k = plan.n_seg[x]++;
        plan.pair[x][k] = p;
        plan.t0[x][k] = t;
        plan.v0[x][k] = plan.n_virtual[x];
        plan.n_virtual[x] += take;
        t += take;
        room -= (double) take * cost8[p];
      }
      if (t < n_tiles || room <= 1e-9) {  // this XCD's share is full (or its segment list is): next XCD
        if (x < F2N_N_PARTS - 1) {
          x++;
          room += share;
        } else if (take <= 0) {
          break;
        }
      }
    }
  }
}


// ---------------------------------------------------------------------------------------------------
// Slice-binned gather for tables that have left the L2s (BASELINE config 5: 2^21 / 2^22 entries per level; round 4).
// With a level slice far larger than an XCD's 4 MiB L2 every hashed read of a fine level is a miss that pulls a 128-byte line
// across the fabric for 4 bytes of payload (8.35 GB per launch at 2^22: profiles/r03_big22_pmc_tcc.csv).  The mirror image of the
// owner-binned scatter reads the table ONCE instead; every intermediate stream is dense and moves through LDS at both ends:
//   (1) gather_request_kernel   block (level, chunk of 1536 samples): hash; an LDS counter per 8192-entry table slice hands every
//       corner its slot; the 13-bit entries-in-slice are laid out slice-major in LDS (exclusive scan of the counters) and leave
//       as ONE dense region of <= 12288 u16 per (level, chunk).  Side outputs: the slot of every corner (8 x u16 per sample and
//       level) and the region's slice offsets;
//   (2) gather_serve_kernel     block (level, slice): the slice (32 KB) into LDS; lane groups walk the chunks and answer the
//       slice's segment of each region in place: result[i] = slice[request[i]];
//   (3) gather_blend_kernel     block (level pair, chunk): per level the chunk's result region (<= 48 KB, dense) and its slice
//       offsets into LDS; every sample hashes again (cheaper than carrying cells through memory), picks its eight values
//       from LDS by (slice offset + slot) and blends them in the order of hash_gather_planes_kernel; one 8-byte plane store.
// Per sample and level: 16 B of requests, 32 B of results and 16 B of slots, each written once and read once (128 B against
// 8 x 128 B of line fills).  Coarse level pairs, whose working set fits the L2s whatever the table size, keep the partitioned
// gather.  Planes are bit-identical (tests/test_gpu_parity.py::test_binned_gather_equals_partitioned_gather).
// ---------------------------------------------------------------------------------------------------
#define F2N_GB_SHIFT 13  // 8192-entry table slices (32 KB in the serving block's LDS; an entry-in-slice is a u16)
#define F2N_GB_ENTRIES (1 << F2N_GB_SHIFT)
#define F2N_GB_MAX_BINS 1024
#define F2N_GB_SERVE_THREADS 512
#define F2N_GB_THREADS 512
#define F2N_GB_SPT 3                                   // samples per thread
#define F2N_GB_CHUNK (F2N_GB_THREADS * F2N_GB_SPT)     // 1536 samples per chunk
#define F2N_GB_REGION (F2N_GB_CHUNK * 8)               // requests of one (level, chunk)
#define F2N_GB_MAX_CHUNKS 1024
struct F2nGatherBins {
  uint16_t* req;    // [NL][nc][REGION]: entry-in-slice, slice-major inside a region
  uint32_t* res;    // same shape: the half2 bits of the requested entries
  uint16_t* offc;   // [NL][nc][n_bins]: where a slice's segment starts in the chunk's region
  int32_t* total;   // [NL][nc]: requests in the region
  uint16_t* slots;  // [NL][n][8]
  int n_bins, l0, nc;
};

__global__ __launch_bounds__(F2N_GB_THREADS) void gather_request_kernel(
    int n, F2nHashArgs h, const int32_t* __restrict__ local_idx, const int32_t* __restrict__ local_size,
    const float* __restrict__ level_scale, const float* __restrict__ pts, int pts_are_warped,
    const int32_t* __restrict__ volume_idx, int vol_stride, F2nGatherBins q) {
  __shared__ F2nLevelTab lt;
  __shared__ int s_cnt[F2N_GB_MAX_BINS];
  __shared__ int s_off[F2N_GB_MAX_BINS];
  __shared__ int s_wave[F2N_GB_THREADS / 64];
  __shared__ uint16_t s_pay[F2N_GB_REGION];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = blockIdx.x / q.nc, B = blockIdx.x % q.nc, l = q.l0 + li;
  f2n_level_tab_fill(lt, level_scale, local_idx, local_size, tid);
  for (int i = tid; i < F2N_GB_MAX_BINS; i += F2N_GB_THREADS) s_cnt[i] = 0;
  __syncthreads();
  uint32_t pos[F2N_GB_SPT][8], slot[F2N_GB_SPT][4];
#pragma unroll
  for (int u = 0; u < F2N_GB_SPT; u++) {
    const int s = B * F2N_GB_CHUNK + u * F2N_GB_THREADS + tid;
    if (s < n) {
      float p01[3];
#pragma unroll
      for (int k = 0; k < 3; k++) {
        const float p = pts[3 * (size_t) s + k];
        p01[k] = pts_are_warped ? (p + 1.f) * .5f : p;
      }
      const int tf = l * h.n_volumes + volume_idx[(size_t) s * vol_stride];
      F2nCell cell;
      f2n_hash_cell(p01, lt.scale[l], h.prim_pool + 3 * tf, h.bias_pool + 3 * tf, lt.size[l], cell);
#pragma unroll
      for (int d = 0; d < 8; d++) {
        pos[u][d] = cell.pos[d];
        const uint32_t got = (uint32_t) atomicAdd(&s_cnt[cell.pos[d] >> F2N_GB_SHIFT], 1);
        if (d & 1) slot[u][d >> 1] |= got << 16;
        else slot[u][d >> 1] = got;
      }
      *(uint4*) (q.slots + ((size_t) li * n + s) * 8) = make_uint4(slot[u][0], slot[u][1], slot[u][2], slot[u][3]);
    }
  }
  __syncthreads();
  // exclusive scan of the slice counters: two per thread
  const int c0 = s_cnt[2 * tid], c1 = s_cnt[2 * tid + 1];
  int incl = c0 + c1;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int up = __shfl_up(incl, o);
    if (lane >= o) incl += up;
  }
  if (lane == 63) s_wave[wave] = incl;
  __syncthreads();
  int before = 0;
#pragma unroll
  for (int w = 0; w < F2N_GB_THREADS / 64; w++) before += w < wave ? s_wave[w] : 0;
  const int excl = before + incl - (c0 + c1);
  s_off[2 * tid] = excl;
  s_off[2 * tid + 1] = excl + c0;
  if (2 * tid < q.n_bins) {
    ((uint32_t*) (q.offc + ((size_t) li * q.nc + B) * q.n_bins))[tid] = (uint32_t) excl | (uint32_t) (excl + c0) << 16;
  }
  __syncthreads();
  int total = 0;
#pragma unroll
  for (int w = 0; w < F2N_GB_THREADS / 64; w++) total += s_wave[w];
  if (tid == 0) q.total[li * q.nc + B] = total;
#pragma unroll
  for (int u = 0; u < F2N_GB_SPT; u++) {
    const int s = B * F2N_GB_CHUNK + u * F2N_GB_THREADS + tid;
    if (s < n) {
#pragma unroll
      for (int d = 0; d < 8; d++) {
        const uint32_t got = (d & 1) ? slot[u][d >> 1] >> 16 : slot[u][d >> 1] & 0xFFFFu;
        s_pay[s_off[pos[u][d] >> F2N_GB_SHIFT] + got] = (uint16_t) (pos[u][d] & (F2N_GB_ENTRIES - 1));
      }
    }
  }
  __syncthreads();
  uint32_t* dst = (uint32_t*) (q.req + ((size_t) li * q.nc + B) * F2N_GB_REGION);
  const uint32_t* src = (const uint32_t*) s_pay;
  for (int i = tid; i < (total + 1) / 2; i += F2N_GB_THREADS) dst[i] = src[i];
}

// A group of G lanes per chunk segment, FOUR requests per lane (one aligned 8-byte load, one 16-byte store; the launcher picks
// G by the mean segment length 12288 / n_bins), U segments of a group in flight: the kernel is a stream of short dependent
// load -> LDS -> store chains over data other XCDs wrote a moment ago, so its rate is the bytes it keeps in flight.
__device__ __forceinline__ void f2n_serve4(const uint32_t* s_slice, uint2 ld, uint32_t* res, int idx, int first, int last) {
  const uint32_t e[4] = {ld.x & 0xFFFFu, ld.x >> 16, ld.y & 0xFFFFu, ld.y >> 16};
  uint32_t v[4];
#pragma unroll
  for (int k = 0; k < 4; k++) v[k] = s_slice[e[k] & (F2N_GB_ENTRIES - 1)];  // (neighbouring segments' entries at the edges)
  if (idx >= first && idx + 4 <= last) {
    *(uint4*) (res + idx) = make_uint4(v[0], v[1], v[2], v[3]);
  } else {
#pragma unroll
    for (int k = 0; k < 4; k++)
      if (idx + k >= first && idx + k < last) res[idx + k] = v[k];
  }
}

template <int G, int U>
__global__ __launch_bounds__(F2N_GB_SERVE_THREADS) void gather_serve_kernel(const half_t* __restrict__ table,
                                                                            const int32_t* __restrict__ local_idx, F2nGatherBins q) {
  __shared__ uint32_t s_slice[F2N_GB_ENTRIES];
  __shared__ uint32_t s_meta[F2N_GB_MAX_CHUNKS];
  constexpr int NG = F2N_GB_SERVE_THREADS / G;
  const int tid = threadIdx.x, sub = tid % G, grp = tid / G;
  // Consecutive blocks go to consecutive XCDs: XCD x serves the slices [x, x + 1) * n_bins / 8 of a level in order, so that the
  // segments of adjacent slices -- neighbours inside every chunk's region, a few dozen bytes each -- share their 128-byte lines
  // in ONE L2 instead of being fetched by several.
  const int li = blockIdx.x / q.n_bins, r = blockIdx.x % q.n_bins, l = q.l0 + li;
  const int bin = (r % F2N_N_PARTS) * (q.n_bins / F2N_N_PARTS) + r / F2N_N_PARTS;
  const uint32_t* slice = (const uint32_t*) (table + local_idx[l]) + (size_t) bin * F2N_GB_ENTRIES;
  for (int i = tid; i < F2N_GB_ENTRIES / 4; i += F2N_GB_SERVE_THREADS) ((uint4*) s_slice)[i] = ((const uint4*) slice)[i];
  // (offset, count) of this slice's segment in every chunk's region: two u16 out of the chunk's offset row -- rows are 2 n_bins
  // bytes apart, but the blocks of adjacent slices run on this XCD at the same time and find the lines in its L2
  for (int B = tid; B < q.nc; B += F2N_GB_SERVE_THREADS) {
    const uint16_t* row = q.offc + ((size_t) li * q.nc + B) * q.n_bins;
    const uint32_t off = row[bin], end = bin + 1 < q.n_bins ? (uint32_t) row[bin + 1] : (uint32_t) q.total[li * q.nc + B];
    s_meta[B] = off << 16 | (end - off);
  }
  __syncthreads();
  const size_t region0 = (size_t) li * q.nc * F2N_GB_REGION;
  for (int B0 = grp; B0 < q.nc; B0 += NG * U) {
    uint2 ld[U];
    size_t base[U];
    int first[U], last[U], idx[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int B = B0 + NG * u;
      const uint32_t m = B < q.nc ? s_meta[B] : 0u;
      first[u] = (int) (m >> 16);
      last[u] = first[u] + (int) (m & 0xFFFFu);
      base[u] = region0 + (size_t) B * F2N_GB_REGION;
      idx[u] = (first[u] & ~3) + 4 * sub;
      ld[u] = idx[u] < last[u] ? *(const uint2*) (q.req + base[u] + idx[u]) : make_uint2(0u, 0u);
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      if (idx[u] < last[u]) f2n_serve4(s_slice, ld[u], q.res + base[u], idx[u], first[u], last[u]);
      for (int i = idx[u] + 4 * G; i < last[u]; i += 4 * G)
        f2n_serve4(s_slice, *(const uint2*) (q.req + base[u] + i), q.res + base[u], i, first[u], last[u]);
    }
  }
}

__global__ __launch_bounds__(F2N_GB_THREADS) void gather_blend_kernel(
    int n, F2nHashArgs h, const int32_t* __restrict__ local_idx, const int32_t* __restrict__ local_size,
    const float* __restrict__ level_scale, const float* __restrict__ pts, int pts_are_warped,
    const int32_t* __restrict__ volume_idx, int vol_stride, half_t* __restrict__ planes, F2nGatherBins q) {
  __shared__ F2nLevelTab lt;
  __shared__ uint32_t s_res[F2N_GB_REGION];
  __shared__ uint16_t s_off[F2N_GB_MAX_BINS];
  const int tid = threadIdx.x;
  const int pi = blockIdx.x / q.nc, B = blockIdx.x % q.nc, part = q.l0 / 2 + pi;
  f2n_level_tab_fill(lt, level_scale, local_idx, local_size, tid);
  float p01[F2N_GB_SPT][3];
  int vol[F2N_GB_SPT];
  half4_t out[F2N_GB_SPT];
#pragma unroll
  for (int u = 0; u < F2N_GB_SPT; u++) {
    const int s = B * F2N_GB_CHUNK + u * F2N_GB_THREADS + tid;
    const int sc = s < n ? s : n - 1;
#pragma unroll
    for (int k = 0; k < 3; k++) {
      const float p = pts[3 * (size_t) sc + k];
      p01[u][k] = pts_are_warped ? (p + 1.f) * .5f : p;
    }
    vol[u] = volume_idx[(size_t) sc * vol_stride];
  }
#pragma unroll
  for (int j = 0; j < 2; j++) {
    const int li = 2 * pi + j, l = q.l0 + li;
    __syncthreads();  // lt filled / the previous level's region is no longer read
    const int total = q.total[li * q.nc + B];
    const uint4* src = (const uint4*) (q.res + ((size_t) li * q.nc + B) * F2N_GB_REGION);
    for (int i = tid; i < (total + 3) / 4; i += F2N_GB_THREADS) ((uint4*) s_res)[i] = src[i];
    if (2 * tid < q.n_bins) ((uint32_t*) s_off)[tid] = ((const uint32_t*) (q.offc + ((size_t) li * q.nc + B) * q.n_bins))[tid];
    __syncthreads();
#pragma unroll
    for (int u = 0; u < F2N_GB_SPT; u++) {
      const int s = B * F2N_GB_CHUNK + u * F2N_GB_THREADS + tid;
      if (s < n) {
        const int tf = l * h.n_volumes + vol[u];
        F2nCell cell;
        f2n_hash_cell(p01[u], lt.scale[l], h.prim_pool + 3 * tf, h.bias_pool + 3 * tf, lt.size[l], cell);
        const uint4 sl = *(const uint4*) (q.slots + ((size_t) li * n + s) * 8);
        const uint32_t packed[4] = {sl.x, sl.y, sl.z, sl.w};
        half2_t v[8];
#pragma unroll
        for (int d = 0; d < 8; d++) {
          const uint32_t got = (d & 1) ? packed[d >> 1] >> 16 : packed[d >> 1] & 0xFFFFu;
          v[d] = __builtin_bit_cast(half2_t, s_res[s_off[cell.pos[d] >> F2N_GB_SHIFT] + got]);
        }
        float s0 = cell.w[0] * (float) v[0][0];  // same fp32 summation order as hash_gather_planes_kernel / the oracle
        float s1 = cell.w[0] * (float) v[0][1];
#pragma unroll
        for (int d = 1; d < 8; d++) {
          s0 = s0 + cell.w[d] * (float) v[d][0];
          s1 = s1 + cell.w[d] * (float) v[d][1];
        }
        out[u][2 * j] = (half_t) s0;
        out[u][2 * j + 1] = (half_t) s1;
      }
    }
  }
#pragma unroll
  for (int u = 0; u < F2N_GB_SPT; u++) {
    const int s = B * F2N_GB_CHUNK + u * F2N_GB_THREADS + tid;
    if (s < n) *(half4_t*) (planes + ((size_t) part * n + s) * 4) = out[u];
  }
}

// (pair, tile) units of the pairs [0, n_pairs) dealt to the 8 XCD shares of the partitioned gather in pair-major order.
static void f2n_gather_plan_coarse(int n_tiles, int n_pairs, F2nGatherPlan& plan) {
  memset(&plan, 0, sizeof(plan));
  const long units = (long) n_pairs * n_tiles, share = (units + F2N_N_PARTS - 1) / F2N_N_PARTS;
  for (int x = 0; x < F2N_N_PARTS; x++) {
    long u = (long) x * share;
    const long u_end = u + share < units ? u + share : units;
    while (u < u_end && plan.n_seg[x] < F2N_MAX_SEGS) {
      const int p = (int) (u / n_tiles), t = (int) (u % n_tiles);
      const long take = (u_end - u) < (n_tiles - t) ? (u_end - u) : (n_tiles - t);
      const int k = plan.n_seg[x]++;
      plan.pair[x][k] = p;
      plan.t0[x][k] = t;
      plan.v0[x][k] = plan.n_virtual[x];
      plan.n_virtual[x] += (int) take;
      u += take;
    }
  }
}

// Gathers this lane's four levels of one sample: returns the X row fragment (8 halves).
__device__ __forceinline__ half8_t f2n_gather_frag(const F2nHashArgs& h, const F2nLevelTab& lt, const float* p01, int vol,
                                                   int g, bool valid) {
  half8_t xf;
  F2nCell cell[4];
  half2_t v[4][8];
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const int l = f2n_level_of(g, j);
    const int tf = l * h.n_volumes + vol;
    f2n_hash_cell(p01, lt.scale[l], h.prim_pool + 3 * tf, h.bias_pool + 3 * tf, lt.size[l], cell[j]);
    const half2_t* base = (const half2_t*) (h.table + lt.base[l]);
#pragma unroll
    for (int d = 0; d < 8; d++) v[j][d] = base[cell[j].pos[d]];  // 32 independent 4-byte gathers in flight
  }
#pragma unroll
  for (int j = 0; j < 4; j++) {
    float s0 = cell[j].w[0] * (float) v[j][0][0];
    float s1 = cell[j].w[0] * (float) v[j][0][1];
#pragma unroll
    for (int d = 1; d < 8; d++) {
      s0 = s0 + cell[j].w[d] * (float) v[j][d][0];
      s1 = s1 + cell[j].w[d] * (float) v[j][d][1];
    }
    xf[2 * j] = valid ? (half_t) s0 : (half_t) 0.f;
    xf[2 * j + 1] = valid ? (half_t) s1 : (half_t) 0.f;
  }
  return xf;
}

// DPP row shifts inside the 16-lane row that holds the 16 samples of one level group.
template <int K>
__device__ __forceinline__ float f2n_row_shr(float v) {  // lane c reads lane c-K of its row; lanes c < K read 0
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x110 + K, 0xF, 0xF, false));
}
template <int K>
__device__ __forceinline__ int f2n_row_shr_i(int v, int fill) {
  return __builtin_amdgcn_update_dpp(fill, v, 0x110 + K, 0xF, 0xF, false);
}
__device__ __forceinline__ int f2n_row_shl1_i(int v, int fill) {  // lane c reads lane c+1; lane 15 reads `fill`
  return __builtin_amdgcn_update_dpp(fill, v, 0x101, 0xF, 0xF, false);
}

// Run combining.  The 16 lanes of a row hold 16 consecutive samples of a ray, a few 1/100 apart in [0,1] warp space,
// while a cell of level l is 2^-(3+7l/15) wide: at the coarse levels most of a row sits in one or two cells.  Runs of
// equal (cell, transform) are summed inside the row -- a segmented Hillis-Steele scan over DPP row shifts, in fp32 --
// and only the last lane of each run owns a contribution: v[2d + ch] = run total for corner d, channel ch (the
// reference rounds every addend to f16 and every partial sum; this is the same sum with fewer roundings).
// Returns true on the last lane of a run.  EVERY lane of the wave must call this (lanes without a sample pass 0).
__device__ __forceinline__ bool f2n_combine_runs(const F2nCell& cell, int vol, int c, float g0, float g1, float* v) {
  const bool same_as_prev = c > 0 && f2n_row_shr_i<1>((int) cell.p[0], -1) == (int) cell.p[0] &&
                            f2n_row_shr_i<1>((int) cell.p[1], -1) == (int) cell.p[1] &&
                            f2n_row_shr_i<1>((int) cell.p[2], -1) == (int) cell.p[2] && f2n_row_shr_i<1>(vol, -1) == vol;
  const int head = same_as_prev ? 0 : 1;
  const bool tail = f2n_row_shl1_i(head, 1) != 0;
#pragma unroll
  for (int d = 0; d < 8; d++) {
    v[2 * d] = g0 * cell.w[d];
    v[2 * d + 1] = g1 * cell.w[d];
  }
  // no run longer than one sample in this wave (the fine levels).  (Skipping the scan below also when only a few lanes continue a
  // run -- up to 24 of 64 -- was measured and changes nothing: profiles/r04_pipeline_experiments.txt item 7.)
  if (__ballot(same_as_prev) == 0ull) return tail;
  int f = head;
#define F2N_SEG_STEP(K)                                   \
  {                                                       \
    const int tf_ = f2n_row_shr_i<K>(f, 1);               \
    const bool take = c >= K && f == 0;                   \
    _Pragma("unroll") for (int i = 0; i < 16; i++) {      \
      const float t_ = f2n_row_shr<K>(v[i]);              \
      v[i] = take ? v[i] + t_ : v[i];                     \
    }                                                     \
    if (c >= K) f |= tf_;                                 \
  }
  F2N_SEG_STEP(1)
  F2N_SEG_STEP(2)
  F2N_SEG_STEP(4)
  F2N_SEG_STEP(8)
#undef F2N_SEG_STEP
  return tail;
}

// Direct scatter of this lane's four levels with packed-f16 global atomics (== the reference's atomicAdd(__half2*),
// Hash3DAnchored.cu:150-151): gx[2j + ch] is the f16 (loss-scaled) gradient of feature sigma(g, 2j+ch).  The chip
// retires ~21 G lane-atomics/s in total whatever their flavour, scope, placement or locality (one XCD alone reaches
// 16.5 G/s: a shared, memory-side unit -- tools/atomic_probe.py), so this path is only used for small batches; large
// ones go through the owner-binned pipeline below.  Contributions that round to (0, 0) are not issued: adding zero is
// a no-op.
__device__ __forceinline__ void f2n_scatter_frag(const F2nHashArgs& h, const F2nLevelTab& lt, half_t* __restrict__ grad_table,
                                                 const float* p01, int vol, int g, int c, half8_t gx) {
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const float g0 = (float) gx[2 * j], g1 = (float) gx[2 * j + 1];
    if (__ballot(g0 != 0.f || g1 != 0.f) == 0ull) continue;  // :149, wave-uniform
    const int l = f2n_level_of(g, j);
    const int tf = l * h.n_volumes + vol;
    F2nCell cell;
    f2n_hash_cell(p01, lt.scale[l], h.prim_pool + 3 * tf, h.bias_pool + 3 * tf, lt.size[l], cell);
    float v[16];
#if F2N_REFERENCE_NUMERICS
    // The reference's arithmetic, addend by addend (Hash3DAnchored.cu:145-153): every (sample, corner) product is rounded to
    // f16 on its own and added by its own packed-f16 atomic, i.e. the running sums are f16 and their order is the order of
    // arrival.  (The product build sums runs of equal cells in fp32 first: fewer roundings, no order dependence.)
#pragma unroll
    for (int d = 0; d < 8; d++) {
      v[2 * d] = g0 * cell.w[d];
      v[2 * d + 1] = g1 * cell.w[d];
    }
    if (g0 != 0.f || g1 != 0.f) {
#else
    if (f2n_combine_runs(cell, vol, c, g0, g1, v)) {
#endif
      half2_t* base = (half2_t*) (grad_table + lt.base[l]);
#pragma unroll
      for (int d = 0; d < 8; d++) {
        const half2_t val = {(half_t) v[2 * d], (half_t) v[2 * d + 1]};
        if ((float) val[0] != 0.f || (float) val[1] != 0.f)
          __builtin_amdgcn_global_atomic_fadd_v2f16((__attribute__((address_space(1))) half2_t*) (base + cell.pos[d]), val);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// Owner-binned scatter (large batches).  Measured on MI355X (tools/atomic_probe.py, tools/lds_atomic_probe.py):
// global atomics cap at ~21 G/s chip-wide; plain read-modify-write out of an XCD's own L2 runs at ~135 G/s; LDS
// atomics run at 0.33 lane-ops/clk/CU for ds_add_f32 / ds_pk_add_f16 but >= 1.4 for ds_add_u32 / u64 / f64.  So
// contributions are not applied where they are produced.  (1) hash_bin_kernel: block (level l, chunk B) walks its
// samples, combines runs, and appends every non-zero contribution as an 8-byte record {entry index inside its
// 4096-entry slice, packed f16 pair} to the private queue segment (l, slice, B) -- slots come from LDS integer
// counters, so no global atomic and no synchronisation between blocks.  (2) hash_bin_accumulate_kernel: one owner
// block per 4096-entry slice of the TABLE sums every record that lands there into an LDS image (round 6: a packed fixed-point image,
// one ds_add_u64 per record -- THE IMAGE below; rounds 1-5: 64 KB of fp64 pairs, two ds_add_f64 per record)
// and adds the image to the f16 gradient table with plain loads and stores; nobody else touches those entries.
// Level l addresses table pairs [l*E/2, l*E/2 + E) (E = entries per level, the reference's 50% level overlap,
// Hash3DAnchored.cpp:60-70), so a table slice receives records from at most two levels.
// A full segment (the hot slices of a big table's coarse levels) sends the record to its producer block's overflow list (F2nBinQueues).
// ---------------------------------------------------------------------------------------------------
#define F2N_BIN_SHIFT 12
#define F2N_BIN_ENTRIES (1 << F2N_BIN_SHIFT)
#define F2N_BIN_NB 128        // sample chunks (producer blocks) per level: the launch shape, and the layout for large batches
// How many of them are USED is decided on the device from the true sample count (it may still be a device-side value when
// the kernels are launched): an owner block visits 2 x nb queue segments, each a dependent fabric round trip however few
// records it holds, so with the ~2.6e5 samples of an ExpRunner::Train batch 128 chunks left the owners latency-bound on
// mostly empty segments (0.09 ms, the same as for 8e5 samples).  Fewer, fuller segments: the queue memory of a (level,
// slice) is split into nb segments of capacity cap * 128 / nb; producer blocks with B >= nb exit at once.
__device__ __forceinline__ int f2n_bin_nb(int n_true, int force = 0) {
  return force > 0 ? force : n_true > 393216 ? 128 : n_true > 131072 ? 64 : 32;
}
#define F2N_BIN_MAX_BINS 1024  // tables up to 2^22 entries per level (BASELINE config 5)
#define F2N_BIN_OVF_CAP 4096   // records per overflow list (32 KB; 64 MB of address space for the 2048 lists, touched only where used)
#define F2N_BIN_MAX_CHUNK 16384  // samples per producer block (the compacted index list lives in LDS)

// Diagnostic counters (f2n_debug_counters): [0] scatter records that found their queue segment AND their block's overflow list full
// and fell back to the packed-f16 global atomic -- the one place where the owner-binned scatter's result depends on the order of
// arrival; [1] slices summed on the owners' fp64 route; [2] (debug variant) the largest per-slice sum of |addend|; [3] records that
// travelled through an overflow list.
__device__ int f2n_dbg_counters[8];

struct F2nBinQueues {
  uint2* rec;      // [16 levels][n_bins][NB][cap]
  int32_t* cnt;    // [16 levels][n_bins][NB]
  int cap, n_bins;
  int nb_force;  // 0, or the chunk count to use whatever the sample count (F2N_BIN_NB: measurement knob)
  // OVERFLOW LISTS (round 6).  A record that finds its segment full -- the coarse levels of a big table put a chunk's records into a
  // few hundred hot slices -- goes to the private list of its producer block (level, chunk) with its full position instead of to a
  // packed-f16 atomic (whose sum depends on the order of arrival: ~130 records per step at 2^22 entries per level).  A level whose
  // producers overflowed in THIS launch carries the launch's stamp in ovf_any[level] (no zeroing pass); only then do that level's
  // owners read the lists' lengths and pick their slice's records out of the non-empty ones.
  uint2* ovf_rec;    // [16 levels][F2N_BIN_NB][F2N_BIN_OVF_CAP]  {entry index inside the LEVEL, packed f16 pair}
  int32_t* ovf_cnt;  // [16 levels][F2N_BIN_NB]
  int32_t* ovf_any;  // [16 levels]
  int stamp;         // this launch (never 0)
  int force_f64;  // != 0: the owners sum every slice on their fp64 route (F2N_OWNER_F64: test knob of the debug variant)
  int dissect;    // debug variant, timing only (F2N_BIN_DISSECT; results are garbage): 1 = the producers do everything but the record stores, 2 = neither slot atomics nor stores, +4 = no owner launch, +8 = 4-byte stores, +16 = 6-byte records
  int producer_major;  // record layout: 1 = [level][chunk][slice][slot] (a producer block's 128 segments contiguous), 0 = [level][slice][chunk][slot] (an owner's, rounds 1-5)
};

// OVF: with overflow lists (F2nBinQueues) -- tables of more than 2^19 entries per level, where segments do fill up (f2n_binned_scatter picks the instantiation); the
// benched 2^19 tables keep the code without them (their trainings never filled a segment, and the lists' code, cold as it is, cost the
// converged step ~5 us: 0.678-0.682 against 0.667-0.674 ms).
template <bool OVF>
__global__ __launch_bounds__(256) void hash_bin_kernel(int n, int chunk, F2nHashArgs h, const int32_t* __restrict__ local_idx,
                                                       const int32_t* __restrict__ local_size,
                                                       const float* __restrict__ level_scale, const float* __restrict__ pts,
                                                       int pts_are_warped, const int32_t* __restrict__ volume_idx, int vol_stride,
                                                       const half_t* __restrict__ gx, long gx_sample_stride,
                                                       long gx_pair_stride, const uint16_t* __restrict__ nz_mask,
                                                       F2nBinQueues q, half_t* __restrict__ grad_table,
                                                       const int32_t* __restrict__ n_dev, int n_off) {
  F2N_RAISE_PRIO();
  if (n_dev != nullptr) n = min(n, *n_dev + n_off);  // the row count is still on the device
  const int l = blockIdx.x % F2N_N_LEVELS, B = blockIdx.x / F2N_N_LEVELS;
  const int nb = f2n_bin_nb(n, q.nb_force);
  if (B >= nb) return;  // (block-uniform, before any barrier)
  chunk = (((n + nb - 1) / nb) + 255) & ~255;  // split what there really is over the producer blocks in use
  const int cap_nb = q.cap * (F2N_BIN_NB / nb);
  __shared__ F2nLevelTab lt;
  __shared__ int s_cnt[F2N_BIN_MAX_BINS];
  __shared__ uint16_t s_idx[F2N_BIN_MAX_CHUNK];  // offsets (inside the chunk) of the samples with a non-zero gradient
  __shared__ int s_wave_tot[4], s_n_nz, s_ovf;
  const int tid = threadIdx.x, c = tid & 15;
  f2n_level_tab_fill(lt, level_scale, local_idx, local_size, tid);
  for (int i = tid; i < q.n_bins; i += 256) s_cnt[i] = 0;
  if (tid == 0) s_ovf = 0;
  __syncthreads();
  const int s_begin = B * chunk, s_end = min(n, s_begin + chunk);
  const half_t* gl = gx + (size_t) (l >> 1) * gx_pair_stride + 2 * (l & 1);
  half2_t* tab = (half2_t*) (grad_table + lt.base[l]);
  // segment (l, bin, B) = my_rec + bin * bin_stride
  uint2* my_rec = q.producer_major ? q.rec + ((size_t) l * nb + B) * q.n_bins * cap_nb : q.rec + ((size_t) l * q.n_bins * nb + B) * cap_nb;
  const size_t bin_stride = q.producer_major ? (size_t) cap_nb : (size_t) nb * cap_nb;
  // Samples whose whole f16 gradient is zero (most of them while the loss-scaled gradients sit at the f16 underflow
  // boundary) are dropped up front: nz_mask has one bit per sample (16-sample words, written by the MLP backward).
  // Order is preserved, so runs of equal cells stay adjacent.
  int n_nz = s_end - s_begin;
  if (nz_mask != nullptr && n_nz > 0) {
    if (tid == 0) s_n_nz = 0;
    __syncthreads();
    const int n_words = (s_end - s_begin + 15) / 16;  // s_begin is a multiple of 256
    for (int w0 = 0; w0 < n_words; w0 += 256) {
      const int w = w0 + tid;
      unsigned m = w < n_words ? nz_mask[s_begin / 16 + w] : 0u;
      if (w < n_words && s_begin + 16 * w + 16 > s_end) m &= (1u << (s_end - s_begin - 16 * w)) - 1u;
      const int cnt = __popc(m);
      int incl = cnt;  // block-wide exclusive prefix of the popcounts
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const int up = __shfl_up(incl, off);
        if ((tid & 63) >= off) incl += up;
      }
      if ((tid & 63) == 63) s_wave_tot[tid >> 6] = incl;
      __syncthreads();
      int base = s_n_nz;
      for (int k = 0; k < (tid >> 6); k++) base += s_wave_tot[k];
      int dst = base + incl - cnt;
      while (m) {
        const int b = __ffs(m) - 1;
        m &= m - 1u;
        s_idx[dst++] = (uint16_t) (16 * w + b);
      }
      __syncthreads();
      if (tid == 255) s_n_nz = base + incl;
      __syncthreads();
    }
    n_nz = s_n_nz;
  }
  // software pipeline: the next tile's gradient pair, point and transform index are in flight while this tile is
  // hashed and appended (three dependent HBM round trips per tile otherwise)
  struct Tile {
    half2_t gpair;
    float p[3];
    int vol;
  };
  auto load_tile = [&](int i0, Tile& t) {
    const int i = min(i0 + tid, n_nz - 1);
    const int sc = s_begin + (nz_mask != nullptr ? (int) s_idx[i] : i);
    t.gpair = *(const half2_t*) (gl + (size_t) sc * gx_sample_stride);
#pragma unroll
    for (int k = 0; k < 3; k++) t.p[k] = pts[3 * (size_t) sc + k];
    t.vol = volume_idx[(size_t) sc * vol_stride];
  };
  Tile cur, nxt;
  if (n_nz > 0) load_tile(0, cur);
  for (int i0 = 0; i0 < n_nz; i0 += 256) {
    if (i0 + 256 < n_nz) load_tile(i0 + 256, nxt);
    const bool valid = i0 + tid < n_nz;
    const float g0 = valid ? (float) cur.gpair[0] : 0.f, g1 = valid ? (float) cur.gpair[1] : 0.f;
    if (__ballot(g0 != 0.f || g1 != 0.f) != 0ull) {  // Hash3DAnchored.cu:149, wave-uniform
      float p01[3];
#pragma unroll
      for (int k = 0; k < 3; k++) p01[k] = pts_are_warped ? (cur.p[k] + 1.f) * .5f : cur.p[k];
      const int vol = cur.vol;
      const int tf = l * h.n_volumes + vol;
      F2nCell cell;
      f2n_hash_cell(p01, lt.scale[l], h.prim_pool + 3 * tf, h.bias_pool + 3 * tf, lt.size[l], cell);
      float v[16];
      if (f2n_combine_runs(cell, vol, c, g0, g1, v)) {
#pragma unroll
        for (int d = 0; d < 8; d++) {
          const half2_t val = {(half_t) v[2 * d], (half_t) v[2 * d + 1]};
          const uint32_t bits = __builtin_bit_cast(uint32_t, val);
          if ((bits & 0x7fff7fffu) != 0u) {  // not (+-0, +-0)
            const uint32_t pos = cell.pos[d];
            const int bin = (int) (pos >> F2N_BIN_SHIFT);
#if F2N_DEBUG_BUILD
            if ((q.dissect & 3) == 2) {
              if ((pos ^ bits) == 0x12345u) s_cnt[bin] = 1;  // (keeps the record's operands alive)
              continue;
            }
#endif
            const int slot = atomicAdd(&s_cnt[bin], 1);
#if F2N_DEBUG_BUILD
            if ((q.dissect & 3) == 1) {
              if (((unsigned) slot ^ pos ^ bits) == 0x12345u) s_cnt[bin] = 1;
              continue;
            }
            if (q.dissect & 8) {  // half the bytes, as many stores: 4-byte "records" 4 bytes apart
              if (slot < cap_nb) ((uint32_t*) (my_rec + (size_t) bin * bin_stride))[slot] = bits ^ pos;
              continue;
            }
            if (q.dissect & 16) {  // 6-byte records 6 bytes apart, as three 2-byte stores
              if (slot < cap_nb) {
                uint16_t* r6 = (uint16_t*) (my_rec + (size_t) bin * bin_stride) + 3 * (size_t) slot;
                r6[0] = (uint16_t) (pos & (F2N_BIN_ENTRIES - 1));
                r6[1] = (uint16_t) bits;
                r6[2] = (uint16_t) (bits >> 16);
              }
              continue;
            }
#endif
            if (slot < cap_nb) {
              uint2 r;
              r.x = pos & (F2N_BIN_ENTRIES - 1);
              r.y = bits;
              my_rec[(uint32_t) bin * (uint32_t) bin_stride + (uint32_t) slot] = r;  // (a segment index inside the block's region fits 32 bits: one scalar base, one vector offset)
            } else {
              const int os = OVF ? atomicAdd(&s_ovf, 1) : F2N_BIN_OVF_CAP;
              if (OVF && os < F2N_BIN_OVF_CAP) {
                q.ovf_rec[((size_t) l * F2N_BIN_NB + B) * F2N_BIN_OVF_CAP + os] = uint2{pos, bits};
              } else {  // (an overflow list full as well: the one order-dependent addition that is left, counted)
                atomicAdd(&f2n_dbg_counters[0], 1);
                __builtin_amdgcn_global_atomic_fadd_v2f16((__attribute__((address_space(1))) half2_t*) (tab + pos), val);
              }
            }
          }
        }
      }
    }
    cur = nxt;
  }
  __syncthreads();
  for (int i = tid; i < q.n_bins; i += 256) q.cnt[((size_t) l * q.n_bins + i) * nb + B] = min(s_cnt[i], cap_nb);
  if (OVF && tid == 0) {
    q.ovf_cnt[l * F2N_BIN_NB + B] = min(s_ovf, F2N_BIN_OVF_CAP);
    if (s_ovf > 0) {
      q.ovf_any[l] = q.stamp;  // (every writer of a launch stores the same value)
      atomicAdd(&f2n_dbg_counters[3], min(s_ovf, F2N_BIN_OVF_CAP));
    }
  }
}

// ADAM (round 6; f2n_field_bwd_step_tail): the owner does not write its slice's sums to the gradient table for a streaming
// optimiser pass behind the step -- it steps the slice's parameters itself (adam_fused_kernel's table arithmetic, element for
// element: gradient = the f16 value the table WOULD hold; widening, / 128, Adam, f16 refresh) and leaves the gradient table zero.
// Every slice of the active prefix is stepped, also the ones nothing landed in (Adam moves a parameter on its moments alone).
struct F2nOwnerAdam {
  float2* param;       // fp32 master [entries][2]
  float2* exp_avg;
  float2* exp_avg_sq;
  half2_t* param_h;    // the f16 working copy the gather reads
  F2nAdamCoef k;
  const int32_t* skip;  // flags[2] of the step's finiteness check: != 0 -> the iteration is dropped (ExpRunner.cpp:131-134)
};

// one round of the owner's Adam: F2N_OWNER_U entries per thread (old gradient pair, master pair, both moments: 7 registers per entry --
// four entries, so that a round in flight under the record reads still leaves the kernel at four waves per SIMD)
#define F2N_OWNER_U 4
#define F2N_OWNER_ROUNDS (F2N_BIN_ENTRIES / (256 * F2N_OWNER_U))
struct F2nOwnerRound {
  half2_t old[F2N_OWNER_U];
  float2 p[F2N_OWNER_U], m[F2N_OWNER_U], v[F2N_OWNER_U];
};
// (`second_moment` false: everything but exp_avg_sq -- what is fetched in front of the record reads; the rest follows behind them)
__device__ __forceinline__ void f2n_owner_fetch(F2nOwnerRound& r, const half2_t* __restrict__ tab, const F2nOwnerAdam& ad, size_t e_base, int e0, bool skip,
                                                bool first_part = true, bool second_moment = true) {
#pragma unroll
  for (int u = 0; u < F2N_OWNER_U; u++) {
    if (first_part) r.old[u] = tab[e0 + 256 * u];
    if (!skip) {
      if (first_part) {
        r.p[u] = ad.param[e_base + e0 + 256 * u];
        r.m[u] = ad.exp_avg[e_base + e0 + 256 * u];
      }
      if (second_moment) r.v[u] = ad.exp_avg_sq[e_base + e0 + 256 * u];
    }
  }
}

// THE IMAGE (round 6): one 64-bit integer per table entry, both channels of a record in ONE ds_add_u64 -- channel 0 in the low 32 bits,
// channel 1 in the high 32, each a signed count of 2^-24 (every f16 is a whole multiple of 2^-24, so every addend is exact; integer sums
// are order-free; a borrow out of the low field is undone when the fields are taken apart, as long as both TRUE sums fit 32 bits).
// 32 KB instead of the 64 KB of two fp64 sums per entry: four owner blocks per CU instead of two -- twice the record reads in flight,
// which is what the owner's time is made of -- and half the LDS atomics.  The fields hold |sum| < 128 (loss-scaled f16 gradients sit
// far below).  Guarantee, not hope: every lane keeps the sum of |addend| it has cast; if the block's total could reach the field's
// range (>= 96, a quarter of slack for the fp32 rounding of that running sum; NaN / Inf land here too) NO entry of the image is trusted
// and the slice is summed again one channel at a time in a 4096-entry fp64 image (the arithmetic of rounds 1-5: fp64 sums of f16 addends
// are exact too, so both routes end in the same bits) -- counted in f2n_debug_counters()[1].
#define F2N_OWNER_FIXED_ONE 16777216.f  // 2^24
#define F2N_OWNER_MAG_LIMIT 96.f

template <bool ADAM, bool OVF>
__global__ __launch_bounds__(256, 4) void hash_bin_accumulate_kernel(F2nBinQueues q, int slices_per_half_level,
                                                                  half_t* __restrict__ grad_table, int n,
                                                                  const int32_t* __restrict__ n_dev, int n_off, int first_slice,
                                                                  F2nOwnerAdam ad) {
  F2N_RAISE_PRIO();
  __shared__ unsigned long long s_acc[F2N_BIN_ENTRIES];  // 32 KB: the fixed-point image of this block's table slice
  __shared__ int s_total;
  __shared__ float s_mag[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // table slice g <- (level l1 = g / H, local slice g - l1*H) and (level l1 - 1, local slice g - l1*H + H)
  const int H = slices_per_half_level, g = first_slice + (int) blockIdx.x;  // (a launch may cover a bucket of the table's slices)
  // (dispatching the 2H light slices -- the table's first and last half level take records from one level only -- last, as the stragglers of
  // a launch ~6 % larger than the chip holds: measured, nothing, 0.677-0.680 against 0.676-0.680 ms per converged step)
  const int l1 = g / H, b1 = g - l1 * H;
  if (n_dev != nullptr) n = min(n, *n_dev + n_off);
  const int nb = f2n_bin_nb(n, q.nb_force), cap_nb = q.cap * (F2N_BIN_NB / nb);  // the producers' choice (same function of the same count)
  // 2 sources x nb segments; wave w owns segments w, w+4, ...: lane j keeps the length of segment w + 4j
  const int seg = wave + 4 * lane;                       // 0 .. 255, of which 0 .. 2*nb-1 exist
  const int src = seg / nb, B = seg % nb;
  const int l = l1 - src, bl = b1 + src * H;
  const bool live = seg < 2 * nb && l >= 0 && l < F2N_N_LEVELS && bl < q.n_bins;
  const size_t my_seg = ((size_t) (live ? l : 0) * q.n_bins + (live ? bl : 0)) * nb + (live ? B : 0);
  const int my_cnt = live ? q.cnt[my_seg] : 0;
  // where that segment's records are (F2nBinQueues::producer_major), in records / cap_nb
  const unsigned my_rseg = q.producer_major ? (unsigned) (((live ? l : 0) * nb + (live ? B : 0)) * q.n_bins + (live ? bl : 0)) : (unsigned) my_seg;
  if (tid == 0) s_total = 0;
  __syncthreads();
  int wsum = my_cnt;
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) wsum += __shfl_xor(wsum, off);
  if (lane == 0 && wsum > 0) atomicAdd(&s_total, wsum);
  __syncthreads();
  const int total = s_total;
  if (total == 0 && !ADAM) return;  // block-uniform: nothing landed in this slice
  // ADAM: the first round's table / parameter / moment operands travel while the records are read and summed
  const bool skip = ADAM && ad.skip != nullptr && *ad.skip != 0;
  half2_t* tab = (half2_t*) grad_table + (size_t) g * F2N_BIN_ENTRIES;
  F2nOwnerRound ro0;
  if (ADAM) f2n_owner_fetch(ro0, tab, ad, (size_t) g * F2N_BIN_ENTRIES, tid, skip, true, false);
  bool packed = total != 0;  // the image holds this slice's sums (block-uniform)
  // overflow lists (F2nBinQueues): only when a producer of one of this slice's two levels overflowed in this launch
  const bool ovf = OVF && ((l1 < F2N_N_LEVELS && q.ovf_any[min(l1, F2N_N_LEVELS - 1)] == q.stamp) || (l1 >= 1 && q.ovf_any[max(l1 - 1, 0)] == q.stamp));
  auto for_overflow = [&](auto&& f) {
    if (!OVF || !ovf) return;  // (block-uniform)
    const int my_ocnt = live ? q.ovf_cnt[l * F2N_BIN_NB + B] : 0;  // lane j of wave w: the list of segment w + 4j's producer block
    unsigned long long m = __ballot(my_ocnt > 0);
    while (m != 0ull) {
      const int j = __ffsll((long long) m) - 1;
      m &= m - 1ull;
      const int cnt = __shfl(my_ocnt, j), bj = __shfl(bl, j);
      const uint2* r = q.ovf_rec + (size_t) __shfl(l * F2N_BIN_NB + B, j) * F2N_BIN_OVF_CAP;
      for (int i = lane; i < cnt; i += 64) {
        const uint2 rec = r[i];
        if ((int) (rec.x >> F2N_BIN_SHIFT) == bj) f(uint2{rec.x & (F2N_BIN_ENTRIES - 1), rec.y});
      }
    }
  };
  if (total != 0) {
    for (int i = tid; i < F2N_BIN_ENTRIES; i += 256) s_acc[i] = 0ull;
    __syncthreads();
    float mag = 0.f;
    auto add = [&](uint2 rec) {
      if (rec.y != 0u) {  // a stored record is never (+0, +0); padding is
        const half2_t val = __builtin_bit_cast(half2_t, rec.y);
        const float f0 = (float) val[0], f1 = (float) val[1];
        mag += fabsf(f0) + fabsf(f1);
        // (clamped: the conversion stays defined for addends the fields cannot hold -- those slices are summed again below)
        const int i0 = (int) fminf(fmaxf(f0 * F2N_OWNER_FIXED_ONE, -2147483648.f), 2147483520.f);
        const int i1 = (int) fminf(fmaxf(f1 * F2N_OWNER_FIXED_ONE, -2147483648.f), 2147483520.f);
        atomicAdd(&s_acc[rec.x], (unsigned long long) (((long long) i1 << 32) + (long long) i0));
      }
    };
    // Eight segments at a time, the first 256 records of each with FOUR coalesced 8-byte loads per lane, all 32 of them issued
    // before any record is added (the records were written by other XCDs a moment ago -- every read is a fabric round trip, so as
    // many as possible are kept in flight): a segment holds ~150 records at the 2.6e5 samples of an ExpRunner::Train batch.
    // (Measured: the kernel's time is the volume of these reads -- 76 us with them, 13 us without, the LDS adds hidden underneath --
    // not their scheduling: profiles/r04_pipeline_experiments.txt item 7.)
    for (int sg = 0; sg < nb / 2; sg += 8) {
      uint2 rec[32];
      int longest = 0;
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const int cnt = __shfl(my_cnt, sg + u);
        const uint2* r = q.rec + (size_t) __shfl(my_rseg, sg + u) * cap_nb;
        longest = max(longest, cnt);
#pragma unroll
        for (int k = 0; k < 4; k++) rec[4 * u + k] = lane + 64 * k < cnt ? r[lane + 64 * k] : uint2{0u, 0u};
      }
#pragma unroll
      for (int u = 0; u < 32; u++) add(rec[u]);
      if (longest > 256) {  // long segments: the rest (wave-uniform)
        for (int u = 0; u < 8; u++) {
          const int cnt = __shfl(my_cnt, sg + u);
          const uint2* r = q.rec + (size_t) __shfl(my_rseg, sg + u) * cap_nb;
          for (int i = lane + 256; i < cnt; i += 64) add(r[i]);
        }
      }
    }
    for_overflow(add);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) mag += __shfl_xor(mag, off);
    if (lane == 0) s_mag[wave] = mag;
    __syncthreads();
    const float mag_all = (s_mag[0] + s_mag[1]) + (s_mag[2] + s_mag[3]);
#if F2N_DEBUG_BUILD  // how close a run's slices come to the limit: f2n_debug_counters()[2] = the largest total, as float bits
    if (tid == 0 && mag_all == mag_all) atomicMax(&f2n_dbg_counters[2], __float_as_int(mag_all));
#endif
    if (!(mag_all < F2N_OWNER_MAG_LIMIT) || q.force_f64 != 0) {
      // ---- a field could have overflowed: the slice again, one channel at a time, exact fp64 sums folded into the gradient table ----
      packed = false;
      if (tid == 0) atomicAdd(&f2n_dbg_counters[1], 1);
      double* s_f64 = (double*) s_acc;
      half_t* tab_h = (half_t*) tab;
      for (int ch = 0; ch < 2; ch++) {
        __syncthreads();
        for (int i = tid; i < F2N_BIN_ENTRIES; i += 256) s_f64[i] = 0.0;
        __syncthreads();
        for (int sg = 0; sg < nb / 2; sg++) {
          const int cnt = __shfl(my_cnt, sg);
          const uint2* r = q.rec + (size_t) __shfl(my_rseg, sg) * cap_nb;
          for (int i = lane; i < cnt; i += 64) {
            const uint2 rec = r[i];
            const half2_t val = __builtin_bit_cast(half2_t, rec.y);
            atomicAdd(&s_f64[rec.x], (double) val[ch]);
          }
        }
        for_overflow([&](uint2 rec) { atomicAdd(&s_f64[rec.x], (double) __builtin_bit_cast(half2_t, rec.y)[ch]); });
        __syncthreads();
        for (int e = tid; e < F2N_BIN_ENTRIES; e += 256) {
          const double a = s_f64[e];
          if (a != 0.0) tab_h[2 * e + ch] = (half_t) (float) ((double) (float) tab_h[2 * e + ch] + a);
        }
      }
      __threadfence_block();
      __syncthreads();
      if (ADAM) {  // the gradient pairs this block has just written (its own stores: visible to it)
#pragma unroll
        for (int u = 0; u < F2N_OWNER_U; u++) ro0.old[u] = tab[tid + 256 * u];
      }
    }
  }  // total != 0
  // the two sums of entry e (exact; what the fp64 pair of rounds 1-5 held)
  auto sums_of = [&](int e, double& a0, double& a1) {
    a0 = 0.0;
    a1 = 0.0;
    if (packed) {
      const long long t = (long long) s_acc[e];
      const int lo = (int) (unsigned) (unsigned long long) t;
      const long long hi = (t - (long long) lo) >> 32;
      a0 = (double) lo * (1.0 / 16777216.0);
      a1 = (double) (int) hi * (1.0 / 16777216.0);
    }
  };
  if (!ADAM) {
    if (!packed) return;
    for (int e0 = tid; e0 < F2N_BIN_ENTRIES; e0 += 256 * 8) {  // eight independent table reads in flight per thread
      half2_t old[8];
#pragma unroll
      for (int u = 0; u < 8; u++) old[u] = tab[e0 + 256 * u];
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const int e = e0 + 256 * u;
        double a0, a1;
        sums_of(e, a0, a1);
        if (a0 != 0.0 || a1 != 0.0)
          tab[e] = half2_t{(half_t) (float) ((double) (float) old[u][0] + a0), (half_t) (float) ((double) (float) old[u][1] + a1)};
      }
    }
    return;
  }
  // ---- ADAM: the slice's 4096 entries in rounds of F2N_OWNER_U per thread; round 0's operands were fetched before the records (ro0
  // above), round r + 1's travel while round r is stepped ----
  const size_t e_base = (size_t) g * F2N_BIN_ENTRIES;
  auto step_round = [&](const F2nOwnerRound& ro, int e0) {
#pragma unroll
    for (int u = 0; u < F2N_OWNER_U; u++) {
      const int e = e0 + 256 * u;
      double a0, a1;
      sums_of(e, a0, a1);
      half2_t gh = ro.old[u];  // the value the gradient table holds behind the plain owner
      if (a0 != 0.0 || a1 != 0.0)
        gh = half2_t{(half_t) (float) ((double) (float) ro.old[u][0] + a0), (half_t) (float) ((double) (float) ro.old[u][1] + a1)};
      if (__builtin_bit_cast(uint32_t, ro.old[u]) != 0u) tab[e] = half2_t{(half_t) 0.f, (half_t) 0.f};  // zero_grad (fallback atomics landed here)
      if (!skip) {
        float m0 = ro.m[u].x, m1 = ro.m[u].y, v0 = ro.v[u].x, v1 = ro.v[u].y;
        const float p0 = f2n_adam_update(ro.p[u].x, (float) gh[0] * ad.k.grad_scale, m0, v0, ad.k);
        const float p1 = f2n_adam_update(ro.p[u].y, (float) gh[1] * ad.k.grad_scale, m1, v1, ad.k);
        ad.param[e_base + e] = make_float2(p0, p1);
        ad.exp_avg[e_base + e] = make_float2(m0, m1);
        ad.exp_avg_sq[e_base + e] = make_float2(v0, v1);
        ad.param_h[e_base + e] = half2_t{(half_t) p0, (half_t) p1};
      }
    }
  };
  f2n_owner_fetch(ro0, tab, ad, e_base, tid, skip, false, true);
  F2nOwnerRound ro1;
#pragma unroll
  for (int r = 0; r < F2N_OWNER_ROUNDS; r += 2) {
    f2n_owner_fetch(ro1, tab, ad, e_base, tid + 256 * F2N_OWNER_U * (r + 1), skip);
    step_round(ro0, tid + 256 * F2N_OWNER_U * r);
    if (r + 2 < F2N_OWNER_ROUNDS) f2n_owner_fetch(ro0, tab, ad, e_base, tid + 256 * F2N_OWNER_U * (r + 2), skip);
    step_round(ro1, tid + 256 * F2N_OWNER_U * (r + 1));
  }
}

__device__ __forceinline__ void f2n_load_point(const float* __restrict__ pts, int s, bool warped, float* p01) {
#pragma unroll
  for (int k = 0; k < 3; k++) {
    const float p = pts[3 * (size_t) s + k];
    p01[k] = warped ? (p + 1.f) * .5f : p;  // Hash3DAnchored.cpp:91
  }
}

// X row fragment from a row-major [n][32] array (h16 or fp32).
__device__ __forceinline__ half8_t f2n_load_xfrag_h(const half_t* __restrict__ x, int s, int g, bool valid) {
  if (!valid) return half8_t{0, 0, 0, 0, 0, 0, 0, 0};
  return f2n_rowfrag(x, F2N_D_IN, s, 0, g);
}
__device__ __forceinline__ half8_t f2n_load_xfrag_f32(const float* __restrict__ x, int s, int g, bool valid) {
  half8_t r = {0, 0, 0, 0, 0, 0, 0, 0};
  if (valid) {
    const float4_t lo = *(const float4_t*) (x + (size_t) s * F2N_D_IN + 4 * g);
    const float4_t hi = *(const float4_t*) (x + (size_t) s * F2N_D_IN + 16 + 4 * g);
#pragma unroll
    for (int i = 0; i < 4; i++) {
      r[i] = (half_t) lo[i];
      r[4 + i] = (half_t) hi[i];
    }
  }
  return r;
}
__device__ __forceinline__ void f2n_store_xfrag_h(half_t* __restrict__ x, int s, int g, half8_t xf) {
  half_t* p = x + (size_t) s * F2N_D_IN + 4 * g;
  *(half4_t*) p = __builtin_shufflevector(xf, xf, 0, 1, 2, 3);
  *(half4_t*) (p + 16) = __builtin_shufflevector(xf, xf, 4, 5, 6, 7);
}

// ---------------------------------------------------------------------------------------------------
// Forward: hash gather (optional) -> MLP (optional).  4 waves per block, each wave strides over
// 16-sample blocks.
// ---------------------------------------------------------------------------------------------------
#define F2N_FWD_THREADS 256

template <int NH, bool DO_HASH, bool DO_MLP>
__global__ __launch_bounds__(F2N_FWD_THREADS) void field_fwd_kernel(
    int n, F2nHashArgs h, const int32_t* __restrict__ local_idx, const int32_t* __restrict__ local_size,
    const float* __restrict__ level_scale, const float* __restrict__ pts, int pts_are_warped,
    const int32_t* __restrict__ volume_idx, int vol_stride, const float* __restrict__ x_f32,
    const half_t* __restrict__ params, float* __restrict__ out_feat_f32, half_t* __restrict__ out_feat_h,
    float* __restrict__ out_f0, half_t* __restrict__ save_x, const half_t* __restrict__ x_planes,
    const half_t* __restrict__ x_cache, const int32_t* __restrict__ src_rows, const int32_t* __restrict__ n_dev) {
  F2N_RAISE_PRIO();
  __shared__ F2nLevelTab lt;
  const int tid = threadIdx.x, lane = tid & 63, c = lane & 15, g = lane >> 4;
  const int n_alloc = n;  // the plane stride stays the allocated row count
  if (n_dev != nullptr) n = min(n, *n_dev);  // the sample count is still on the device (f2n_field_fwd_cached_dyn)
  if (DO_HASH) {
    f2n_level_tab_fill(lt, level_scale, local_idx, local_size, tid);
    __syncthreads();
  }
  F2nMlpFwdW<NH> w;
  if (DO_MLP) w.load(params, c, g);
  const int n_blocks = (n + 15) / 16;
  const int wave_global = blockIdx.x * (F2N_FWD_THREADS / 64) + (tid >> 6);
  const int wave_stride = gridDim.x * (F2N_FWD_THREADS / 64);
  for (int blk = wave_global; blk < n_blocks; blk += wave_stride) {
    const int s = blk * 16 + c;
    const bool valid = s < n;
    const int sc = valid ? s : n - 1;
    half8_t xf;
    if (DO_HASH) {
      float p01[3];
      f2n_load_point(pts, sc, pts_are_warped != 0, p01);
      const int vol = volume_idx[(size_t) sc * vol_stride];
      xf = f2n_gather_frag(h, lt, p01, vol, g, valid);
    } else if (x_planes != nullptr) {  // features gathered by hash_gather_planes_kernel
      xf = f2n_cat(*(const half4_t*) (x_planes + ((size_t) g * n_alloc + sc) * 4),
                   *(const half4_t*) (x_planes + ((size_t) (4 + g) * n_alloc + sc) * 4));
      if (!valid) xf = half8_t{0, 0, 0, 0, 0, 0, 0, 0};
    } else if (x_cache != nullptr) {  // h16 feature rows of an earlier query of the same table (f2n_field_fwd_cached)
      xf = f2n_load_xfrag_h(x_cache, src_rows != nullptr ? src_rows[sc] : sc, g, valid);
    } else {
      xf = f2n_load_xfrag_f32(x_f32, sc, g, valid);
    }
    if (save_x != nullptr && valid) f2n_store_xfrag_h(save_x, s, g, xf);
    if (DO_MLP) {
      const float4_t o = w.forward(xf);  // lane (c = sample, g): outputs 4g..4g+3
      if (valid) {
        half4_t oh;
        float4_t of;
#pragma unroll
        for (int r = 0; r < 4; r++) {
          oh[r] = (half_t) o[r];  // output precision is f16 (TCNNWP.cpp:143-144)
          of[r] = (float) oh[r];
        }
        if (out_feat_f32 != nullptr) *(float4_t*) (out_feat_f32 + (size_t) s * F2N_D_OUT + 4 * g) = of;
        if (out_feat_h != nullptr) *(half4_t*) (out_feat_h + (size_t) s * F2N_D_OUT + 4 * g) = oh;
        if (out_f0 != nullptr && g == 0) out_f0[s] = of[0];
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// Backward: MLP backward (recompute + both orientations, mlp_dev.h) chained into the hash scatter.
// One wave handles 16-sample tiles; the weight gradients contract a tile's 16 samples with K = 16 MFMAs
// (v_mfma_f32_16x16x16_f16), so nothing but the accumulators is carried between tiles and the register budget allows
// two or three resident blocks per CU.
// ---------------------------------------------------------------------------------------------------
#define F2N_BWD_THREADS 256

template <int NH>
union F2nBwdSmem {
  F2nMlpLds<NH> w;
  float acc[2 * (F2N_D_HID * F2N_D_IN + (NH == 2 ? F2N_D_HID * F2N_D_HID : 0) + F2N_D_OUT * F2N_D_HID)];  // two images, see f2n_mlp_flush_dw
};

// HASH: 0 = MLP only (dL/dx to fp32), 1 = chained into the packed-f16 atomic scatter (small batches), 2 = dL/dx leaves as
// f16 planes + non-zero mask for the owner-binned scatter.  BPC = resident blocks per CU the register budget is cut for.
template <int NH, int HASH, int BPC>
__global__ __launch_bounds__(F2N_BWD_THREADS, BPC) void field_bwd_kernel(
    int n, F2nHashArgs h, const int32_t* __restrict__ local_idx, const int32_t* __restrict__ local_size,
    const float* __restrict__ level_scale, const float* __restrict__ pts, int pts_are_warped,
    const int32_t* __restrict__ volume_idx, int vol_stride, const half_t* __restrict__ params,
    const half_t* __restrict__ x_h, const float* __restrict__ x_f32, const float* __restrict__ dy, float loss_scale,
    float* __restrict__ dparams, float* __restrict__ dx_f32, half_t* __restrict__ grad_table, half_t* __restrict__ dx_planes,
    uint16_t* __restrict__ nz_mask, const int32_t* __restrict__ n_dev, int n_off) {
  F2N_RAISE_PRIO();
  const int n_alloc = n;  // plane stride / mask words: the allocated row count
  if (n_dev != nullptr) n = min(n, *n_dev + n_off);  // the row count is still on the device (f2n_field_bwd_dyn)
  __shared__ F2nBwdSmem<NH> sm;
  __shared__ F2nLevelTab lt;
  const int tid = threadIdx.x, lane = tid & 63, c = lane & 15, g = lane >> 4;
  if (HASH == 1) f2n_level_tab_fill(lt, level_scale, local_idx, local_size, tid);
  f2n_mlp_lds_fill<NH>(sm.w, params, tid, F2N_BWD_THREADS);
  __syncthreads();
  const half8_t idf[2] = {f2n_identity_frag(0, c, g), f2n_identity_frag(1, c, g)};
  F2nMlpGradAcc<NH> acc;
  acc.zero();
  const int wave_global = blockIdx.x * (F2N_BWD_THREADS / 64) + (tid >> 6);
  const int wave_stride = gridDim.x * (F2N_BWD_THREADS / 64);
  const float inv_scale = 1.f / loss_scale;
  // One 16-sample tile per round, the next tile's inputs in flight in registers (see shade_bwd_kernel: with one or two
  // waves per SIMD a load issued at its point of use exposes its whole latency).
  struct In {
    half8_t xf;
    float4_t d4;
  };
  auto fetch = [&](int tile, In& o) {
    const int s = tile * 16 + c;
    const int sc = s < n ? s : n - 1;
    o.xf = (x_h != nullptr) ? f2n_load_xfrag_h(x_h, sc, g, true) : f2n_load_xfrag_f32(x_f32, sc, g, true);
    o.d4 = *(const float4_t*) (dy + (size_t) sc * F2N_D_OUT + 4 * g);
  };
  const int n_tiles = (n + 15) / 16;
  In cur;
  if (wave_global < n_tiles) fetch(wave_global, cur);
  for (int tile = wave_global; tile < n_tiles; tile += wave_stride) {
    In nxt;
    fetch(tile + wave_stride < n_tiles ? tile + wave_stride : tile, nxt);  // last round: a harmless re-read
    __builtin_amdgcn_sched_barrier(0);
    {
      F2nHalfBwd<NH> hb;
      int lds_off = 0;  // weight fragments are read from LDS where they are used, not hoisted (see shade_bwd_kernel)
      asm volatile("" : "+v"(lds_off));
      const F2nMlpLds<NH>& wl = *(const F2nMlpLds<NH>*) ((const char*) &sm.w + lds_off);
      const int s = tile * 16 + c;
      const bool valid = s < n;
      const int sc = valid ? s : n - 1;
      const half8_t xf = valid ? cur.xf : half8_t{0, 0, 0, 0, 0, 0, 0, 0};
      half8_t dyf = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
      for (int r = 0; r < 4; r++)  // f16 cast by autograd, then *scale (TCNNWP.cpp:174)
        dyf[r] = valid ? (half_t) ((float) (half_t) cur.d4[r] * loss_scale) : (half_t) 0.f;
      f2n_mlp_half_bwd<NH, 2>(wl, xf, [&](half8_t, half8_t) { return dyf; }, idf, c, g, hb);
      if (valid) {
        if (dx_f32 != nullptr) {
#pragma unroll
          for (int ft = 0; ft < 2; ft++) {
            float4_t v;
#pragma unroll
            for (int r = 0; r < 4; r++) v[r] = hb.dxT[ft][r] * inv_scale;  // TCNNWP.cpp:231
            *(float4_t*) (dx_f32 + (size_t) s * F2N_D_IN + 16 * ft + 4 * g) = v;
          }
        }
      }
      if (HASH != 0) {
        half8_t gx = f2n_pack<false>(hb.dxT[0], hb.dxT[1]);  // (dL/dx * 128) -> f16, Hash3DAnchored.cu:220
        if (HASH == 2) {  // large batches: the owner-binned scatter consumes f16 planes [8][n][4]
          bool nz = false;
#pragma unroll
          for (int e = 0; e < 8; e++) nz |= (float) gx[e] != 0.f;
          const unsigned long long bal = __ballot(nz && valid);  // bit 16g + c: lane (c, g) of this tile
          if (valid) {
            *(half4_t*) (dx_planes + ((size_t) g * n_alloc + s) * 4) = __builtin_shufflevector(gx, gx, 0, 1, 2, 3);
            *(half4_t*) (dx_planes + ((size_t) (4 + g) * n_alloc + s) * 4) = __builtin_shufflevector(gx, gx, 4, 5, 6, 7);
          }
          if (lane == 0)  // one 16-bit word per tile: sample c has a non-zero gradient
            nz_mask[tile] = (uint16_t) ((bal | (bal >> 16) | (bal >> 32) | (bal >> 48)) & 0xffffull);
        } else {  // every lane takes part in the row-level combining; out-of-range samples carry zero gradient
          float p01[3];
          f2n_load_point(pts, sc, pts_are_warped != 0, p01);
          const int vol = volume_idx[(size_t) sc * vol_stride];
          if (!valid) gx = half8_t{0, 0, 0, 0, 0, 0, 0, 0};
          f2n_scatter_frag(h, lt, grad_table, p01, vol, g, c, gx);
        }
      }
      f2n_mlp_accumulate_dw_half<NH>(hb, acc);
    }
    cur = nxt;
  }
  __syncthreads();  // everyone is done with the LDS weights: reuse the space for the block reduction
  f2n_mlp_flush_dw<NH>(acc, sm.acc, dparams, c, g, tid, F2N_BWD_THREADS);
}

// Stand-alone hash scatter (seam-level f2n_hash_bwd): grad_in is h16 [n,32].
__global__ __launch_bounds__(256) void hash_bwd_kernel(int n, F2nHashArgs h, const int32_t* __restrict__ local_idx,
                                                       const int32_t* __restrict__ local_size,
                                                       const float* __restrict__ level_scale, const float* __restrict__ pts,
                                                       int pts_are_warped, const int32_t* __restrict__ volume_idx,
                                                       int vol_stride, const half_t* __restrict__ grad_in,
                                                       half_t* __restrict__ grad_table) {
  __shared__ F2nLevelTab lt;
  const int tid = threadIdx.x, lane = tid & 63, c = lane & 15, g = lane >> 4;
  f2n_level_tab_fill(lt, level_scale, local_idx, local_size, tid);
  __syncthreads();
  const int n_blocks = (n + 15) / 16;
  const int wave_global = blockIdx.x * 4 + (tid >> 6);
  const int wave_stride = gridDim.x * 4;
  for (int blk = wave_global; blk < n_blocks; blk += wave_stride) {
    const int s = blk * 16 + c;
    const bool valid = s < n;
    const int sc = valid ? s : n - 1;
    float p01[3];
    f2n_load_point(pts, sc, pts_are_warped != 0, p01);
    const int vol = volume_idx[(size_t) sc * vol_stride];
    half8_t gx = f2n_rowfrag(grad_in, F2N_D_IN, sc, 0, g);
    if (!valid) gx = half8_t{0, 0, 0, 0, 0, 0, 0, 0};
    f2n_scatter_frag(h, lt, grad_table, p01, vol, g, c, gx);
  }
}

__global__ void mlp_init_kernel(uint64_t seed, int d_in, int d_hidden, int n_hidden, float* __restrict__ params) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int n0 = d_hidden * d_in, n1 = (n_hidden - 1) * d_hidden * d_hidden, no = F2N_D_OUT * d_hidden;
  if (i >= n0 + n1 + no) return;
  int fan_in, fan_out;
  if (i < n0) { fan_in = d_in; fan_out = d_hidden; }
  else if (i < n0 + n1) { fan_in = d_hidden; fan_out = d_hidden; }
  else { fan_in = d_hidden; fan_out = F2N_D_OUT; }
  // splitmix64 counter hash -> U[0,1) -> Xavier uniform
  uint64_t zz = seed + 0x9e3779b97f4a7c15ull * (uint64_t) (i + 1);
  zz = (zz ^ (zz >> 30)) * 0xbf58476d1ce4e5b9ull;
  zz = (zz ^ (zz >> 27)) * 0x94d049bb133111ebull;
  zz = zz ^ (zz >> 31);
  const float u = (float) (zz >> 40) * (1.f / 16777216.f);
  const float scale = sqrtf(6.f / (float) (fan_in + fan_out));
  params[i] = (u * 2.f - 1.f) * scale;
}

__global__ void to_h16_kernel(int n, const float* __restrict__ in, half_t* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = (half_t) in[i];
}

static inline unsigned f2n_wave_grid(int n_units, int waves_per_block) {
  // enough blocks to fill 256 CUs several times over, capped so that each wave still amortises its setup
  long blocks = ((long) n_units + waves_per_block - 1) / waves_per_block;
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  return (unsigned) blocks;
}

// Backward kernels hold their weight-gradient accumulators in registers (1 wave per SIMD, 1 block per CU) and pay
// one LDS + global-atomic flush of all parameters per BLOCK: exactly one resident block per CU minimises that.
static inline unsigned f2n_bwd_grid(int n_super, int blocks_per_cu = 1) {
  long blocks = ((long) n_super + 3) / 4;
  if (blocks > 256 * blocks_per_cu) blocks = 256 * blocks_per_cu;
  if (blocks < 1) blocks = 1;
  return (unsigned) blocks;
}

static inline bool f2n_mlp_shape_ok(int d_in, int d_hidden, int n_hidden) {
  return d_in == F2N_D_IN && d_hidden == F2N_D_HID && (n_hidden == 1 || n_hidden == 2);
}

#define F2N_BIN_MIN_N 32768  // below this the direct atomics cost less than the two extra launches

struct F2nBucketHook {
  f2n_bucket_fn fn;
  void* user;
  int n;
  const void* table;  // the gradient table the hook is for (nullptr: any)
};
static F2nBucketHook g_bucket_hook[16];

// Owner-binned scatter of f16 gradients gx (pair (l, ch) of sample s at gx[s*ss + (l>>1)*ps + 2*(l&1) + ch]).
static int f2n_binned_scatter(hipStream_t st, int n, const F2nHashArgs& h, const int32_t* local_idx, const int32_t* local_size,
                              const float* level_scale, const float* pts, int warped, const int32_t* volume_idx, int vol_stride,
                              const half_t* gx, long ss, long ps, const uint16_t* nz_mask, half_t* grad_table, int level_entries,
                              const int32_t* n_dev = nullptr, int n_off = 0, const F2nOwnerAdam* adam = nullptr,
                              hipEvent_t wait_before_owners = nullptr, int* adam_applied = nullptr) {
  if (adam_applied != nullptr) *adam_applied = 0;
  F2nBinQueues q;
#if F2N_DEBUG_BUILD  // measurement knob of the debug variant: the producer chunk count whatever the sample count
  static const int nb_force = []() {
    const char* e = getenv("F2N_BIN_NB");
    const int v = e != nullptr ? atoi(e) : 0;
    return (v == 32 || v == 64 || v == 128) ? v : 0;
  }();
  q.nb_force = nb_force;
  static const int force_f64 = []() {
    const char* e = getenv("F2N_OWNER_F64");
    return e != nullptr && atoi(e) != 0 ? 1 : 0;
  }();
  q.force_f64 = force_f64;
  static const int dissect = []() {
    const char* e = getenv("F2N_BIN_DISSECT");
    return e != nullptr ? atoi(e) : 0;
  }();
  q.dissect = dissect;
  static const int layout = []() {
    const char* e = getenv("F2N_BIN_LAYOUT");
    return e != nullptr ? atoi(e) : 1;
  }();
  q.producer_major = layout;
#else
  q.nb_force = 0;
  q.force_f64 = 0;
  q.dissect = 0;
  q.producer_major = 1;
#endif


  q.n_bins = level_entries >> F2N_BIN_SHIFT;
  const int chunk = (((n + F2N_BIN_NB - 1) / F2N_BIN_NB) + 255) & ~255;
  q.cap = (int) (1.25 * 8.0 * (double) chunk / (double) q.n_bins) + 64;
  const size_t n_seg = (size_t) F2N_N_LEVELS * q.n_bins * F2N_BIN_NB;
  q.rec = (uint2*) f2n_ws_get(F2N_WS_BIN_REC, n_seg * q.cap * sizeof(uint2));
  q.cnt = (int32_t*) f2n_ws_get(F2N_WS_BIN_CNT, n_seg * sizeof(int32_t));
  const size_t n_lists = (size_t) F2N_N_LEVELS * F2N_BIN_NB;
  q.ovf_rec = (uint2*) f2n_ws_get(F2N_WS_BIN_OVF, n_lists * F2N_BIN_OVF_CAP * sizeof(uint2) + (n_lists + F2N_N_LEVELS) * sizeof(int32_t));
  if (q.rec == nullptr || q.cnt == nullptr || q.ovf_rec == nullptr) return F2N_ERR_INVALID_ARG;
  q.ovf_cnt = (int32_t*) (q.ovf_rec + n_lists * F2N_BIN_OVF_CAP);
  q.ovf_any = q.ovf_cnt + n_lists;
  static std::atomic<int> g_scatter_stamp{0};
  q.stamp = (g_scatter_stamp.fetch_add(1) & 0x3fffffff) + 1;  // (the workspace starts out zeroed: never a launch's stamp)
  const bool ovf = level_entries > (1 << 19);  // (see hash_bin_kernel)
  if (ovf)
    hipLaunchKernelGGL(hash_bin_kernel<true>, dim3(F2N_N_LEVELS * F2N_BIN_NB), dim3(256), 0, st, n, chunk, h, local_idx, local_size,
                       level_scale, pts, warped, volume_idx, vol_stride, gx, ss, ps, nz_mask, q, grad_table, n_dev, n_off);
  else
    hipLaunchKernelGGL(hash_bin_kernel<false>, dim3(F2N_N_LEVELS * F2N_BIN_NB), dim3(256), 0, st, n, chunk, h, local_idx, local_size,
                       level_scale, pts, warped, volume_idx, vol_stride, gx, ss, ps, nz_mask, q, grad_table, n_dev, n_off);
#define F2N_LAUNCH_OWNERS(ADAM_, GRID, FIRST, AD)                                                                                         \
  do {                                                                                                                                    \
    if (ovf) hipLaunchKernelGGL((hash_bin_accumulate_kernel<ADAM_, true>), dim3(GRID), dim3(256), 0, st, q, H, grad_table, n, n_dev, n_off, FIRST, AD);  \
    else hipLaunchKernelGGL((hash_bin_accumulate_kernel<ADAM_, false>), dim3(GRID), dim3(256), 0, st, q, H, grad_table, n, n_dev, n_off, FIRST, AD);     \
  } while (0)
#if F2N_DEBUG_BUILD
  if (q.dissect & 4) return f2n_launch_status();  // (timing only: the producers alone)
#endif
  const int H = q.n_bins / 2;  // table slices per half level; the table spans (16 + 1) half levels
  // Data-parallel runs ask for the owner launch in BUCKETS of table slices (f2n_set_scatter_buckets): after each bucket's launch the
  // host is called back and starts that range's all-reduce while the next bucket's owners still run.
  const int S = (F2N_N_LEVELS + 1) * H;
  int dev = 0;
  (void) hipGetDevice(&dev);
  const F2nBucketHook hook = (dev >= 0 && dev < 16) ? g_bucket_hook[dev] : F2nBucketHook{nullptr, nullptr, 0, nullptr};
  if (hook.fn != nullptr && hook.n > 1 && S >= hook.n && (hook.table == nullptr || hook.table == (const void*) grad_table)) {
    for (int b = 0; b < hook.n; b++) {
      const int g0 = (int) ((long) b * S / hook.n), g1 = (int) ((long) (b + 1) * S / hook.n);
      F2N_LAUNCH_OWNERS(false, g1 - g0, g0, F2nOwnerAdam{});
      const int rc = f2n_launch_status();
      if (rc != F2N_OK) return rc;
      hook.fn(hook.user, b, hook.n);
    }
    return F2N_OK;
  }
  if (adam != nullptr) {  // (no bucket hook on this table: checked above)
    if (wait_before_owners != nullptr && hipStreamWaitEvent(st, wait_before_owners, 0) != hipSuccess) return F2N_ERR_INVALID_ARG;
    F2N_LAUNCH_OWNERS(true, S, 0, *adam);
    if (adam_applied != nullptr) *adam_applied = 1;
    return f2n_launch_status();
  }
  F2N_LAUNCH_OWNERS(false, S, 0, F2nOwnerAdam{});
#undef F2N_LAUNCH_OWNERS
  return f2n_launch_status();
}

// The binned path needs the reference's table layout -- local_idx[l] = l * E halves, local_size[l] = E entries with E a
// power of two -- and whole 4096-entry slices per half level.
static inline bool f2n_use_bins(int n, int level_entries) {
  if (F2N_REFERENCE_NUMERICS) return false;  // per-addend f16 atomics in arrival order, as the reference (f2n_scatter_frag)
  return n >= F2N_BIN_MIN_N && n <= F2N_BIN_NB * (F2N_BIN_MAX_CHUNK - 256) && level_entries >= 2 * F2N_BIN_ENTRIES && (level_entries & (level_entries - 1)) == 0 &&
         (level_entries >> F2N_BIN_SHIFT) <= F2N_BIN_MAX_BINS;
}

extern "C" {

int f2n_set_scatter_buckets(int n_buckets, f2n_bucket_fn fn, void* user) { return f2n_set_scatter_buckets_for(n_buckets, fn, user, nullptr); }

int f2n_set_scatter_buckets_for(int n_buckets, f2n_bucket_fn fn, void* user, const void* grad_table_h16) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16 || n_buckets < 0 || n_buckets > 64) return F2N_ERR_INVALID_ARG;
  g_bucket_hook[dev] = F2nBucketHook{n_buckets > 1 ? fn : nullptr, user, n_buckets > 1 ? n_buckets : 0, grad_table_h16};
  return F2N_OK;
}

int f2n_debug_counters(int32_t* out8_host, int reset) {
  if (out8_host != nullptr && hipMemcpyFromSymbol(out8_host, HIP_SYMBOL(f2n_dbg_counters), 8 * sizeof(int32_t)) != hipSuccess)
    return F2N_ERR_INVALID_ARG;
  if (reset) {
    const int32_t zero[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(f2n_dbg_counters), zero, sizeof(zero)) != hipSuccess) return F2N_ERR_INVALID_ARG;
  }
  return F2N_OK;
}

int f2n_mlp_n_params(int d_in, int d_hidden, int n_hidden) {
  if (d_in <= 0 || d_hidden <= 0 || n_hidden < 1) return F2N_ERR_INVALID_ARG;
  return d_hidden * d_in + (n_hidden - 1) * d_hidden * d_hidden + F2N_D_OUT * d_hidden;
}

int f2n_mlp_init_params(void* stream, uint64_t seed, int d_in, int d_hidden, int n_hidden, float* params_f32) {
  const int n = f2n_mlp_n_params(d_in, d_hidden, n_hidden);
  if (n < 0) return n;
  hipLaunchKernelGGL(mlp_init_kernel, dim3(f2n_div_up(n, 256)), dim3(256), 0, (hipStream_t) stream, seed, d_in, d_hidden,
                     n_hidden, params_f32);
  return f2n_launch_status();
}

int f2n_params_to_h16(void* stream, int n, const float* params_f32, void* params_h) {
  if (n < 0) return F2N_ERR_INVALID_ARG;
  if (n == 0) return F2N_OK;
  hipLaunchKernelGGL(to_h16_kernel, dim3(f2n_div_up(n, 256)), dim3(256), 0, (hipStream_t) stream, n, params_f32,
                     (half_t*) params_h);
  return f2n_launch_status();
}

int f2n_hash_fwd(void* stream, int n, int n_volumes, const void* table_h, const int32_t* prim_pool,
                 const int32_t* local_idx, const int32_t* local_size, const float* bias_pool, const float* level_scale,
                 const float* pts, int pts_are_warped, const int32_t* volume_idx, int vol_stride, void* out_h) {
  if (n < 0 || n_volumes <= 0 || vol_stride < 1) return F2N_ERR_INVALID_ARG;
  if (n == 0) return F2N_OK;
  F2nHashArgs h = {(const half_t*) table_h, prim_pool, bias_pool, n_volumes};
  hipLaunchKernelGGL((field_fwd_kernel<1, true, false>), dim3(f2n_wave_grid((n + 15) / 16, 4)), dim3(F2N_FWD_THREADS), 0,
                     (hipStream_t) stream, n, h, local_idx, local_size, level_scale, pts, pts_are_warped, volume_idx,
                     vol_stride, nullptr, nullptr, nullptr, nullptr, nullptr, (half_t*) out_h, nullptr, nullptr, nullptr, nullptr);
  return f2n_launch_status();
}

int f2n_hash_bwd(void* stream, int n, int n_volumes, const int32_t* prim_pool, const int32_t* local_idx,
                 const int32_t* local_size, const float* bias_pool, const float* level_scale, const float* pts,
                 int pts_are_warped, const int32_t* volume_idx, int vol_stride, const void* grad_in_h, void* grad_table_h,
                 int level_entries) {
  if (n < 0 || n_volumes <= 0 || vol_stride < 1 || level_entries < 0) return F2N_ERR_INVALID_ARG;
  if (n == 0) return F2N_OK;
  F2nHashArgs h = {nullptr, prim_pool, bias_pool, n_volumes};
  if (f2n_use_bins(n, level_entries))  // row-major [n][32]: pair (l, ch) at 32*s + 2*l + ch
    return f2n_binned_scatter((hipStream_t) stream, n, h, local_idx, local_size, level_scale, pts, pts_are_warped, volume_idx,
                              vol_stride, (const half_t*) grad_in_h, F2N_D_IN, 4, nullptr, (half_t*) grad_table_h, level_entries);
  hipLaunchKernelGGL(hash_bwd_kernel, dim3(f2n_wave_grid((n + 15) / 16, 4)), dim3(256), 0, (hipStream_t) stream, n, h,
                     local_idx, local_size, level_scale, pts, pts_are_warped, volume_idx, vol_stride,
                     (const half_t*) grad_in_h, (half_t*) grad_table_h);
  return f2n_launch_status();
}

int f2n_mlp_fwd(void* stream, int n, int d_in, int d_hidden, int n_hidden, const void* params_h, const float* x, void* out_h) {
  if (n < 0) return F2N_ERR_INVALID_ARG;
  if (!f2n_mlp_shape_ok(d_in, d_hidden, n_hidden)) {  // not one of the two shipped networks: the general kernels (mlp_generic.hip)
    if (!f2n_mlpg_shape_ok(d_in, d_hidden, n_hidden)) return F2N_ERR_UNSUPPORTED;
    return n == 0 ? F2N_OK : f2n_mlpg_fwd(stream, n, d_in, d_hidden, n_hidden, params_h, x, out_h);
  }
  if (n == 0) return F2N_OK;
  F2nHashArgs h = {nullptr, nullptr, nullptr, 1};
  const dim3 grid(f2n_wave_grid((n + 15) / 16, 4)), block(F2N_FWD_THREADS);
  if (n_hidden == 1)
    hipLaunchKernelGGL((field_fwd_kernel<1, false, true>), grid, block, 0, (hipStream_t) stream, n, h, nullptr, nullptr, nullptr,
                       nullptr, 0, nullptr, 1, x, (const half_t*) params_h, nullptr, (half_t*) out_h, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
  else
    hipLaunchKernelGGL((field_fwd_kernel<2, false, true>), grid, block, 0, (hipStream_t) stream, n, h, nullptr, nullptr, nullptr,
                       nullptr, 0, nullptr, 1, x, (const half_t*) params_h, nullptr, (half_t*) out_h, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
  return f2n_launch_status();
}

int f2n_mlp_bwd(void* stream, int n, int d_in, int d_hidden, int n_hidden, float loss_scale, const void* params_h,
                const float* x, const float* dy, float* dparams_f32_scaled, float* dx_f32) {
  if (n < 0 || !(loss_scale > 0.f) || ((uintptr_t) params_h & 15)) return F2N_ERR_INVALID_ARG;  // 16-byte loads of the weights
  if (!f2n_mlp_shape_ok(d_in, d_hidden, n_hidden)) {
    if (!f2n_mlpg_shape_ok(d_in, d_hidden, n_hidden)) return F2N_ERR_UNSUPPORTED;
    return n == 0 ? F2N_OK : f2n_mlpg_bwd(stream, n, d_in, d_hidden, n_hidden, loss_scale, params_h, x, dy, dparams_f32_scaled, dx_f32);
  }
  if (n == 0) return F2N_OK;
  F2nHashArgs h = {nullptr, nullptr, nullptr, 1};
  // resident blocks per CU by register budget: the one-hidden-layer kernel needs ~150 registers (3), the two-layer one 244 (2)
  const dim3 grid(f2n_bwd_grid((n + 31) / 32, n_hidden == 1 ? 3 : 2)), block(F2N_BWD_THREADS);
  const int n_params = f2n_mlp_n_params(d_in, d_hidden, n_hidden);
  float* partials = (float*) f2n_ws_get(F2N_WS_FIELD_DW, sizeof(float) * (size_t) grid.x * n_params);
  if (partials == nullptr) return F2N_ERR_INVALID_ARG;
  if (n_hidden == 1)
    hipLaunchKernelGGL((field_bwd_kernel<1, 0, 3>), grid, block, 0, (hipStream_t) stream, n, h, nullptr, nullptr, nullptr, nullptr,
                       0, nullptr, 1, (const half_t*) params_h, nullptr, x, dy, loss_scale, partials, dx_f32, nullptr, nullptr, nullptr, nullptr, 0);
  else
    hipLaunchKernelGGL((field_bwd_kernel<2, 0, 2>), grid, block, 0, (hipStream_t) stream, n, h, nullptr, nullptr, nullptr, nullptr,
                       0, nullptr, 1, (const half_t*) params_h, nullptr, x, dy, loss_scale, partials, dx_f32, nullptr, nullptr, nullptr, nullptr, 0);
  int rc = f2n_launch_status();
  if (rc != F2N_OK) return rc;
  return f2n_reduce_partials(stream, n_params, (int) grid.x, partials, dparams_f32_scaled);
}

int f2n_field_fwd(void* stream, int n, int n_volumes, const void* table_h, const int32_t* prim_pool,
                  const int32_t* local_idx, const int32_t* local_size, const float* bias_pool, const float* level_scale,
                  const float* pts_warped, const int32_t* volume_idx, int vol_stride, const void* mlp_params_h,
                  float* out_feat_f32, float* out_f0, void* save_x_h) {
  if (n < 0 || n_volumes <= 0 || vol_stride < 1) return F2N_ERR_INVALID_ARG;
  if (n == 0) return F2N_OK;
  F2nHashArgs h = {(const half_t*) table_h, prim_pool, bias_pool, n_volumes};
  if (n < F2N_PARTITION_MIN_N) {  // small batches: one launch, features stay in registers between gather and MFMA
    hipLaunchKernelGGL((field_fwd_kernel<1, true, true>), dim3(f2n_wave_grid((n + 15) / 16, 4)), dim3(F2N_FWD_THREADS), 0,
                       (hipStream_t) stream, n, h, local_idx, local_size, level_scale, pts_warped, 1, volume_idx, vol_stride,
                       nullptr, (const half_t*) mlp_params_h, out_feat_f32, nullptr, out_f0, (half_t*) save_x_h, nullptr, nullptr, nullptr, nullptr);
    return f2n_launch_status();
  }
  // large batches: XCD-aware level-partitioned gather into f16 planes (64 B/sample of internal workspace), then the
  // same MLP kernel reads its MFMA K-slots from the planes.  ~130 B/sample of extra streaming buys ~3x on the gathers.
  half_t* planes = (half_t*) f2n_ws_get(F2N_WS_FIELD_PLANES, sizeof(half_t) * 32 * (size_t) n);
  if (planes == nullptr) return F2N_ERR_INVALID_ARG;
  int rc = f2n_hash_gather_planes(stream, n, n_volumes, table_h, prim_pool, local_idx, local_size, bias_pool, level_scale,
                                  pts_warped, 1, volume_idx, vol_stride, planes);
  if (rc != F2N_OK) return rc;
  return f2n_field_mlp_planes(stream, n, planes, mlp_params_h, out_feat_f32, out_f0, save_x_h);
}

// The one-kernel gather -> MLP of the small-batch path at ANY size (the north star's "features staged on-chip as the MFMA tile"):
// a wave gathers all 16 levels of its 16 samples and feeds the matrix cores from registers, no planes in HBM.  What it gives
// up is the XCD-aware level partition (every wave touches all 16 levels: each L2 sees the whole table).  Kept as the A/B
// comparator of the partitioned pipeline (profiles/r04_fused_gather_ab.txt), not used by the host at n >= F2N_PARTITION_MIN_N.
int f2n_field_fwd_fused(void* stream, int n, int n_volumes, const void* table_h, const int32_t* prim_pool, const int32_t* local_idx,
                        const int32_t* local_size, const float* bias_pool, const float* level_scale, const float* pts_warped,
                        const int32_t* volume_idx, int vol_stride, const void* mlp_params_h, float* out_feat_f32, float* out_f0,
                        void* save_x_h) {
  if (n < 0 || n_volumes <= 0 || vol_stride < 1) return F2N_ERR_INVALID_ARG;
  if (n == 0) return F2N_OK;
  F2nHashArgs h = {(const half_t*) table_h, prim_pool, bias_pool, n_volumes};
  hipLaunchKernelGGL((field_fwd_kernel<1, true, true>), dim3(f2n_wave_grid((n + 15) / 16, 4)), dim3(F2N_FWD_THREADS), 0,
                     (hipStream_t) stream, n, h, local_idx, local_size, level_scale, pts_warped, 1, volume_idx, vol_stride, nullptr,
                     (const half_t*) mlp_params_h, out_feat_f32, nullptr, out_f0, (half_t*) save_x_h, nullptr, nullptr, nullptr, nullptr);
  return f2n_launch_status();
}

int f2n_gather_plan_query(int n_tiles, float step01, const float* level_scale_host, int32_t* out, float* cost8_out) {
  if (n_tiles < 0 || out == nullptr) return F2N_ERR_INVALID_ARG;
  F2nGatherPlan plan;
  float cost8[F2N_N_PARTS];
  const bool balanced = step01 > 0.f && level_scale_host != nullptr;
  if (balanced) f2n_gather_costs(step01, level_scale_host, cost8);
  f2n_gather_plan(n_tiles, balanced ? cost8 : nullptr, plan);
  for (int x = 0; x < F2N_N_PARTS; x++) {
    int32_t* o = out + x * (1 + 3 * F2N_MAX_SEGS);
    o[0] = plan.n_seg[x];
    for (int k = 0; k < F2N_MAX_SEGS; k++) {
      const bool live = k < plan.n_seg[x];
      const int end_v = !live ? 0 : (k + 1 < plan.n_seg[x] ? plan.v0[x][k + 1] : plan.n_virtual[x]);
      o[1 + 3 * k] = live ? plan.pair[x][k] : -1;
      o[2 + 3 * k] = live ? plan.t0[x][k] : 0;
      o[3 + 3 * k] = live ? end_v - plan.v0[x][k] : 0;  // tiles in the segment
    }
    if (cost8_out != nullptr) cost8_out[x] = balanced ? cost8[x] : 1.f;
  }
  return F2N_OK;
}

int f2n_hash_gather_planes(void* stream, int n, int n_volumes, const void* table_h, const int32_t* prim_pool,
                           const int32_t* local_idx, const int32_t* local_size, const float* bias_pool, const float* level_scale,
                           const float* pts, int pts_are_warped, const int32_t* volume_idx, int vol_stride, void* planes_h) {
  return f2n_hash_gather_planes_balanced(stream, n, n_volumes, table_h, prim_pool, local_idx, local_size, bias_pool, level_scale,
                                         pts, pts_are_warped, volume_idx, vol_stride, planes_h, 0.f, nullptr);
}

// The launcher's choice of kernel variant: bit 0 = hash constants staged in LDS, bit 1 = run combining + cost-balanced split.
static int f2n_gather_variant(int n, int n_volumes, float step01, const float* level_scale_host, float* cost8) {
  const size_t stage_bytes = (size_t) 2 * n_volumes * 6 * sizeof(uint32_t);
  const bool staged = stage_bytes <= 20000 && n >= 64 * 256;  // keeps 8 blocks per CU resident; not worth the copy for small batches
  bool balanced = step01 > 0.f && level_scale_host != nullptr;
  if (balanced) {
    // Run combining and pair switching cost ~10 % of a tile (more registers: 6 instead of 8 waves per SIMD; re-staging;
    // an L2 refill per switch): worth it only when the balanced share is well below the costliest pair's load.  On a
    // fresh scene (fineness 16) the model predicts 0.90 and the balanced kernel measured 5 % SLOWER; on a converged one
    // 0.53 predicted, 0.57 measured (tools/gather_ab.py).
    f2n_gather_costs(step01, level_scale_host, cost8);
    float sum = 0.f, mx = 0.f;
    for (int p = 0; p < F2N_N_PARTS; p++) {
      sum += cost8[p];
      mx = fmaxf(mx, cost8[p]);
    }
    balanced = sum / F2N_N_PARTS < 0.8f * mx;
  }
  return (staged ? 1 : 0) | (balanced ? 2 : 0);
}

int f2n_hash_gather_variant(int n, int n_volumes, float step01, const float* level_scale_host) {
  if (n < 0 || n_volumes <= 0) return F2N_ERR_INVALID_ARG;
  float cost8[F2N_N_PARTS];
  return f2n_gather_variant(n, n_volumes, step01, level_scale_host, cost8);
}

int f2n_hash_gather_planes_balanced(void* stream, int n, int n_volumes, const void* table_h, const int32_t* prim_pool,
                                    const int32_t* local_idx, const int32_t* local_size, const float* bias_pool,
                                    const float* level_scale, const float* pts, int pts_are_warped, const int32_t* volume_idx,
                                    int vol_stride, void* planes_h, float step01, const float* level_scale_host) {
  if (n < 0 || n_volumes <= 0 || vol_stride < 1) return F2N_ERR_INVALID_ARG;
  if (n == 0) return F2N_OK;
  F2nHashArgs h = {(const half_t*) table_h, prim_pool, bias_pool, n_volumes};
  long per_part = ((long) n + 255) / 256;
  if (per_part > 256) per_part = 256;  // 32 CUs per XCD x 8 resident 256-thread blocks
  const size_t stage_bytes = (size_t) 2 * n_volumes * 6 * sizeof(uint32_t);
  F2nGatherPlan plan;
  float cost8[F2N_N_PARTS];
  const int variant = f2n_gather_variant(n, n_volumes, step01, level_scale_host, cost8);
  const bool staged = (variant & 1) != 0, balanced = (variant & 2) != 0;
  f2n_gather_plan((n + 255) / 256, balanced ? cost8 : nullptr, plan);
#define F2N_LAUNCH_GATHER(ST, CB)                                                                                              \
  hipLaunchKernelGGL((hash_gather_planes_kernel<ST, CB>), dim3((unsigned) (F2N_N_PARTS * per_part)), dim3(256),                \
                     ST ? stage_bytes : 0, (hipStream_t) stream, n, h, local_idx, local_size, level_scale, pts, pts_are_warped, \
                     volume_idx, vol_stride, (half_t*) planes_h, plan)
  if (staged && balanced) F2N_LAUNCH_GATHER(true, true);
  else if (staged) F2N_LAUNCH_GATHER(true, false);
  else if (balanced) F2N_LAUNCH_GATHER(false, true);
  else F2N_LAUNCH_GATHER(false, false);
#undef F2N_LAUNCH_GATHER
  return f2n_launch_status();
}

int f2n_hash_gather_planes_binned(void* stream, int n, int n_volumes, const void* table_h, const int32_t* prim_pool,
                                  const int32_t* local_idx, const int32_t* local_size, const float* bias_pool,
                                  const float* level_scale, const float* pts, int pts_are_warped, const int32_t* volume_idx,
                                  int vol_stride, void* planes_h, int level_entries, int first_binned_pair) {
  const int p0 = first_binned_pair;
  if (n < 0 || n_volumes <= 0 || vol_stride < 1 || p0 < 0 || p0 > F2N_N_PARTS) return F2N_ERR_INVALID_ARG;
  if (level_entries < F2N_N_PARTS * F2N_GB_ENTRIES || (level_entries & (level_entries - 1)) != 0 || (level_entries >> F2N_GB_SHIFT) > F2N_GB_MAX_BINS)
    return F2N_ERR_UNSUPPORTED;  // (the reference's table layout: local_size[l] = E, a power of two, whole 8192-entry slices)
  if (n == 0) return F2N_OK;
  hipStream_t st = (hipStream_t) stream;
  F2nHashArgs h = {(const half_t*) table_h, prim_pool, bias_pool, n_volumes};
  const int n_tiles = (n + 255) / 256;
  if (p0 > 0) {  // the coarse pairs: the partitioned gather, its (pair, tile) units dealt to the eight XCD shares
    long per_part = n_tiles;
    if (per_part > 256) per_part = 256;
    F2nGatherPlan plan;
    f2n_gather_plan_coarse(n_tiles, p0, plan);
    hipLaunchKernelGGL((hash_gather_planes_kernel<false, false>), dim3((unsigned) (F2N_N_PARTS * per_part)), dim3(256), 0, st, n, h,
                       local_idx, local_size, level_scale, pts, pts_are_warped, volume_idx, vol_stride, (half_t*) planes_h, plan);
    const int rc = f2n_launch_status();
    if (rc != F2N_OK) return rc;
  }
  if (p0 == F2N_N_PARTS) return F2N_OK;
  F2nGatherBins q;
  q.l0 = 2 * p0;
  const int nl = F2N_N_LEVELS - q.l0;
  q.n_bins = level_entries >> F2N_GB_SHIFT;
  q.nc = (n + F2N_GB_CHUNK - 1) / F2N_GB_CHUNK;
  if (q.nc > F2N_GB_MAX_CHUNKS) return F2N_ERR_UNSUPPORTED;
  const size_t n_regions = (size_t) nl * q.nc;
  auto up = [](size_t b) { return (b + 255) & ~(size_t) 255; };
  const size_t b_res = up(n_regions * F2N_GB_REGION * sizeof(uint32_t)), b_req = up(n_regions * F2N_GB_REGION * sizeof(uint16_t));
  const size_t b_offc = up(n_regions * q.n_bins * sizeof(uint16_t));
  const size_t b_total = up(n_regions * sizeof(int32_t)), b_slots = up((size_t) nl * n * 8 * sizeof(uint16_t));
  char* ws = (char*) f2n_ws_get(F2N_WS_GATHER_BINS, b_res + b_req + b_offc + b_total + b_slots);
  if (ws == nullptr) return F2N_ERR_INVALID_ARG;
  q.res = (uint32_t*) ws;
  q.req = (uint16_t*) (ws + b_res);
  q.offc = (uint16_t*) (ws + b_res + b_req);
  q.total = (int32_t*) (ws + b_res + b_req + b_offc);
  q.slots = (uint16_t*) (ws + b_res + b_req + b_offc + b_total);
  hipLaunchKernelGGL(gather_request_kernel, dim3(nl * q.nc), dim3(F2N_GB_THREADS), 0, st, n, h, local_idx, local_size, level_scale,
                     pts, pts_are_warped, volume_idx, vol_stride, q);
  if (F2N_GB_REGION / q.n_bins > 40)  // mean segment length: 48 at 2^21 entries per level, 24 at 2^22
    hipLaunchKernelGGL((gather_serve_kernel<16, 4>), dim3(nl * q.n_bins), dim3(F2N_GB_SERVE_THREADS), 0, st, (const half_t*) table_h, local_idx, q);
  else
    hipLaunchKernelGGL((gather_serve_kernel<8, 4>), dim3(nl * q.n_bins), dim3(F2N_GB_SERVE_THREADS), 0, st, (const half_t*) table_h, local_idx, q);
  hipLaunchKernelGGL(gather_blend_kernel, dim3((F2N_N_PARTS - p0) * q.nc), dim3(F2N_GB_THREADS), 0, st, n, h, local_idx, local_size,
                     level_scale, pts, pts_are_warped, volume_idx, vol_stride, (half_t*) planes_h, q);
  return f2n_launch_status();
}

int f2n_field_mlp_planes(void* stream, int n, const void* planes_h, const void* mlp_params_h, float* out_feat_f32, float* out_f0,
                         void* save_x_h) {
  if (n < 0 || (n > 0 && planes_h == nullptr)) return F2N_ERR_INVALID_ARG;
  if (n == 0) return F2N_OK;
  F2nHashArgs h = {nullptr, nullptr, nullptr, 1};
  hipLaunchKernelGGL((field_fwd_kernel<1, false, true>), dim3(f2n_wave_grid((n + 15) / 16, 4)), dim3(F2N_FWD_THREADS), 0,
                     (hipStream_t) stream, n, h, nullptr, nullptr, nullptr, nullptr, 0, nullptr, 1, nullptr,
                     (const half_t*) mlp_params_h, out_feat_f32, nullptr, out_f0, (half_t*) save_x_h, (const half_t*) planes_h,
                     nullptr, nullptr, nullptr);
  return f2n_launch_status();
}

int f2n_field_fwd_cached(void* stream, int n, int n_cache, const int32_t* src_rows, const void* x_cache_h,
                         const void* mlp_params_h, float* out_feat_f32, float* out_f0, void* save_x_h) {
  return f2n_field_fwd_cached_dyn(stream, n, nullptr, n_cache, src_rows, x_cache_h, mlp_params_h, out_feat_f32, out_f0, save_x_h);
}

int f2n_field_fwd_cached_dyn(void* stream, int n_max, const int32_t* n_dev, int n_cache, const int32_t* src_rows,
                             const void* x_cache_h, const void* mlp_params_h, float* out_feat_f32, float* out_f0, void* save_x_h) {
  const int n = n_max;
  if (n < 0 || n_cache < 0 || (n > 0 && x_cache_h == nullptr) || (src_rows == nullptr && n > n_cache)) return F2N_ERR_INVALID_ARG;
  if (n == 0) return F2N_OK;
  F2nHashArgs h = {nullptr, nullptr, nullptr, 1};
  hipLaunchKernelGGL((field_fwd_kernel<1, false, true>), dim3(f2n_wave_grid((n + 15) / 16, 4)), dim3(F2N_FWD_THREADS), 0,
                     (hipStream_t) stream, n, h, nullptr, nullptr, nullptr, nullptr, 0, nullptr, 1, nullptr,
                     (const half_t*) mlp_params_h, out_feat_f32, nullptr, out_f0, (half_t*) save_x_h, nullptr,
                     (const half_t*) x_cache_h, src_rows, n_dev);
  return f2n_launch_status();
}

int f2n_field_bwd(void* stream, int n, int n_volumes, const int32_t* prim_pool, const int32_t* local_idx,
                  const int32_t* local_size, const float* bias_pool, const float* level_scale, const float* pts_warped,
                  const int32_t* volume_idx, int vol_stride, const void* mlp_params_h, const void* saved_x_h,
                  const float* dfeat, float loss_scale, float* dparams_f32_scaled, void* grad_table_h, int level_entries) {
  return f2n_field_bwd_dyn(stream, n, nullptr, 0, n_volumes, prim_pool, local_idx, local_size, bias_pool, level_scale, pts_warped,
                           volume_idx, vol_stride, mlp_params_h, saved_x_h, dfeat, loss_scale, dparams_f32_scaled, grad_table_h,
                           level_entries, 0);
}

// f2n_field_bwd_dyn, and -- with a tail -- f2n_field_bwd_step_tail (include/f2n_abi.h): the rest of the training step re-ordered
// around it.
static int f2n_field_bwd_impl(void* stream, void* tail_stream, int n_max, const int32_t* n_dev, int n_off, int n_volumes,
                              const int32_t* prim_pool, const int32_t* local_idx, const int32_t* local_size, const float* bias_pool,
                              const float* level_scale, const float* pts_warped, const int32_t* volume_idx, int vol_stride,
                              const void* mlp_params_h, const void* saved_x_h, const float* dfeat, float loss_scale,
                              float* dparams_f32_scaled, void* grad_table_h, int level_entries, int defer_reduce,
                              const F2nStepTail* tail, int* table_stepped) {
  const int n = n_max;
  if (table_stepped != nullptr) *table_stepped = 0;
  if (n < 0 || n_volumes <= 0 || vol_stride < 1 || !(loss_scale > 0.f) || level_entries < 0 || ((uintptr_t) mlp_params_h & 15))
    return F2N_ERR_INVALID_ARG;
  if (tail != nullptr && (tail->flags == nullptr || tail->n_groups < 0 || tail->n_groups > 4 || tail->step < 1 || tail->n_table < 0 ||
                          (tail->n_table > 0 && (tail->table_param == nullptr || tail->table_exp_avg == nullptr ||
                                                 tail->table_exp_avg_sq == nullptr || tail->table_param_h == nullptr))))
    return F2N_ERR_INVALID_ARG;
  if (n == 0 && tail == nullptr) return F2N_OK;
  F2nHashArgs h = {nullptr, prim_pool, bias_pool, n_volumes};
  const bool bins = n > 0 && f2n_use_bins(n, level_entries);
  int rc = F2N_OK;
  const int n_params = f2n_mlp_n_params(F2N_D_IN, F2N_D_HID, 1);
  half_t* dx_planes = nullptr;
  uint16_t* nz_mask = nullptr;
  if (n > 0) {
    const unsigned blocks = f2n_bwd_grid((n + 31) / 32, bins ? 3 : 2);  // 154 / 186 registers: three / two resident blocks per CU
    float* partials = (float*) f2n_ws_get(F2N_WS_FIELD_DW, sizeof(float) * (size_t) blocks * n_params);
    if (partials == nullptr) return F2N_ERR_INVALID_ARG;
    if (bins) {  // planes [8][n][4] followed by one non-zero bit per sample
      const size_t plane_bytes = sizeof(half_t) * 32 * (size_t) n;
      dx_planes = (half_t*) f2n_ws_get(F2N_WS_FIELD_PLANES, plane_bytes + sizeof(uint16_t) * ((size_t) n / 16 + 2));
      if (dx_planes == nullptr) return F2N_ERR_INVALID_ARG;
      nz_mask = (uint16_t*) ((char*) dx_planes + plane_bytes);
    }
#define F2N_LAUNCH_FIELD_BWD(HASH, BPC)                                                                                      \
  hipLaunchKernelGGL((field_bwd_kernel<1, HASH, BPC>), dim3(blocks), dim3(F2N_BWD_THREADS), 0, (hipStream_t) stream, n, h,  \
                     local_idx, local_size, level_scale, pts_warped, 1, volume_idx, vol_stride, (const half_t*) mlp_params_h, \
                     (const half_t*) saved_x_h, nullptr, dfeat, loss_scale, partials, nullptr, (half_t*) grad_table_h,       \
                     dx_planes, nz_mask, n_dev, n_off)
    if (!bins) F2N_LAUNCH_FIELD_BWD(1, 2);
    else F2N_LAUNCH_FIELD_BWD(2, 3);
#undef F2N_LAUNCH_FIELD_BWD
    rc = f2n_launch_status();
    if (rc != F2N_OK) return rc;
    if (tail == nullptr && !defer_reduce) {
      if (dx_planes != nullptr) {
        rc = f2n_binned_scatter((hipStream_t) stream, n, h, local_idx, local_size, level_scale, pts_warped, 1, volume_idx, vol_stride,
                                dx_planes, 4, 4 * (long) n, nz_mask, (half_t*) grad_table_h, level_entries, n_dev, n_off);
        if (rc != F2N_OK) return rc;
      }
      return f2n_reduce_partials(stream, n_params, (int) blocks, partials, dparams_f32_scaled);
    }
    rc = f2n_defer_reduction(n_params, (int) blocks, partials, dparams_f32_scaled);
    if (rc != F2N_OK) return rc;
  }
  if (tail == nullptr) {  // f2n_field_bwd_dyn with defer_reduce: the scatter, nothing else
    if (dx_planes == nullptr) return F2N_OK;
    return f2n_binned_scatter((hipStream_t) stream, n, h, local_idx, local_size, level_scale, pts_warped, 1, volume_idx, vol_stride, dx_planes,
                              4, 4 * (long) n, nz_mask, (half_t*) grad_table_h, level_entries, n_dev, n_off);
  }
  // ---- the step's tail: the MLPs' gradients are complete (every deferring backward has been queued on `stream`) ----
  hipStream_t st = (hipStream_t) stream, ts = tail_stream != nullptr ? (hipStream_t) tail_stream : st;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  if (ts != st) {
    int dev = 0;
    static hipEvent_t evs[16][2];
    static bool made[16];
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return F2N_ERR_INVALID_ARG;
    if (!made[dev]) {
      if (hipEventCreateWithFlags(&evs[dev][0], hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&evs[dev][1], hipEventDisableTiming) != hipSuccess)
        return F2N_ERR_UNSUPPORTED;
      made[dev] = true;
    }
    ev_fork = evs[dev][0];
    ev_join = evs[dev][1];
    if (hipEventRecord(ev_fork, st) != hipSuccess || hipStreamWaitEvent(ts, ev_fork, 0) != hipSuccess) return F2N_ERR_INVALID_ARG;
  }
  rc = f2n_reduce_deferred(ts);
  if (rc != F2N_OK) return rc;
  if (tail->after_reduce != nullptr) tail->after_reduce(tail->after_reduce_user, (void*) ts);  // (data-parallel: the small buffers' exchange)
  rc = f2n_nonfinite_flags_ex(ts, tail->n_flags_a, tail->flags_grad_a, tail->n_flags_b, tail->flags_grad_b, tail->flags, tail->flags_mirror);
  if (rc != F2N_OK) return rc;
  const int32_t* skip = tail->flags + 2;
  rc = f2n_adam_fused(ts, tail->n_groups, tail->groups, 0, nullptr, nullptr, 1.f, nullptr, nullptr, nullptr, tail->step, tail->lr, tail->beta1,
                      tail->beta2, tail->eps, /*zero_grad=*/1, skip);
  if (rc != F2N_OK) return rc;
  if (ts != st && hipEventRecord(ev_join, ts) != hipSuccess) return F2N_ERR_INVALID_ARG;
  int by_owners = 0;
  if (dx_planes != nullptr) {
    // the owners step the table when its active prefix is exactly the slices they own (17 half levels of the reference's layout)
    const long S = (long) (F2N_N_LEVELS + 1) * ((level_entries >> F2N_BIN_SHIFT) / 2);
    F2nOwnerAdam ad;
    ad.param = (float2*) tail->table_param;
    ad.exp_avg = (float2*) tail->table_exp_avg;
    ad.exp_avg_sq = (float2*) tail->table_exp_avg_sq;
    ad.param_h = (half2_t*) tail->table_param_h;
    ad.k = f2n_adam_coef(tail->step, tail->lr, tail->beta1, tail->beta2, tail->eps, 0.f, tail->table_grad_scale);
    ad.skip = skip;
    const bool can = !tail->leave_table_to_caller && tail->n_table > 0 && (long) tail->n_table == S * 2 * F2N_BIN_ENTRIES;
    rc = f2n_binned_scatter(st, n, h, local_idx, local_size, level_scale, pts_warped, 1, volume_idx, vol_stride, dx_planes, 4, 4 * (long) n, nz_mask,
                            (half_t*) grad_table_h, level_entries, n_dev, n_off, can ? &ad : nullptr, ev_join, &by_owners);
    if (rc != F2N_OK) return rc;
  }
  if (!by_owners) {  // small batches, odd tables, a bucket hook: the ordinary table pass behind the scatter
    if (ts != st && hipStreamWaitEvent(st, ev_join, 0) != hipSuccess) return F2N_ERR_INVALID_ARG;
    if (tail->n_table > 0 && !tail->leave_table_to_caller) {
      rc = f2n_adam_fused(st, 0, nullptr, tail->n_table, tail->table_param, grad_table_h, tail->table_grad_scale, tail->table_exp_avg,
                          tail->table_exp_avg_sq, tail->table_param_h, tail->step, tail->lr, tail->beta1, tail->beta2, tail->eps, /*zero_grad=*/1, skip);
      if (rc != F2N_OK) return rc;
    }
  }
  if (table_stepped != nullptr) *table_stepped = by_owners;
  return F2N_OK;
}

int f2n_field_bwd_dyn(void* stream, int n_max, const int32_t* n_dev, int n_off, int n_volumes, const int32_t* prim_pool,
                      const int32_t* local_idx, const int32_t* local_size, const float* bias_pool, const float* level_scale,
                      const float* pts_warped, const int32_t* volume_idx, int vol_stride, const void* mlp_params_h,
                      const void* saved_x_h, const float* dfeat, float loss_scale, float* dparams_f32_scaled, void* grad_table_h,
                      int level_entries, int defer_reduce) {
  return f2n_field_bwd_impl(stream, nullptr, n_max, n_dev, n_off, n_volumes, prim_pool, local_idx, local_size, bias_pool, level_scale, pts_warped,
                            volume_idx, vol_stride, mlp_params_h, saved_x_h, dfeat, loss_scale, dparams_f32_scaled, grad_table_h, level_entries,
                            defer_reduce, nullptr, nullptr);
}

int f2n_field_bwd_step_tail(void* stream, void* tail_stream, int n_max, const int32_t* n_dev, int n_off, int n_volumes,
                            const int32_t* prim_pool, const int32_t* local_idx, const int32_t* local_size, const float* bias_pool,
                            const float* level_scale, const float* pts_warped, const int32_t* volume_idx, int vol_stride,
                            const void* mlp_params_h, const void* saved_x_h, const float* dfeat, float loss_scale,
                            float* dparams_f32_scaled, void* grad_table_h, int level_entries, const F2nStepTail* tail,
                            int* table_stepped_by_owners) {
  if (tail == nullptr) return F2N_ERR_INVALID_ARG;
  return f2n_field_bwd_impl(stream, tail_stream, n_max, n_dev, n_off, n_volumes, prim_pool, local_idx, local_size, bias_pool, level_scale,
                            pts_warped, volume_idx, vol_stride, mlp_params_h, saved_x_h, dfeat, loss_scale, dparams_f32_scaled, grad_table_h,
                            level_entries, 1, tail, table_stepped_by_owners);
}

}  // extern "C"
