// Shared device-side definitions for the gfx950 kernels of the f2-nerf hot path.
// Everything in csrc/ is built with -ffp-contract=off: the oracle (oracle/f2n_oracle.c) fixes the fp32
// operation order, and integer outputs (sample counts, leaf lists, hash cells) depend on it.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/f2n_abi.h"
#ifndef F2N_DEBUG_BUILD
#define F2N_DEBUG_BUILD 0  // 1: the debug variant of the library (include/f2n_debug.h): debugging launches + measurement knobs
#endif
#if F2N_DEBUG_BUILD
#include "../../include/f2n_debug.h"
#endif

#define F2N_WAVE 64

typedef _Float16 half_t;
typedef half_t half2_t __attribute__((ext_vector_type(2)));
typedef half_t half4_t __attribute__((ext_vector_type(4)));
typedef half_t half8_t __attribute__((ext_vector_type(8)));
typedef float float4_t __attribute__((ext_vector_type(4)));

// Byte-compatible with PtsSampler/PersSampler.h:15-37 (checked against the reference build in
// tests/test_oracle_vs_ref.py::test_layout_facts).
struct alignas(32) F2nTreeNode {
  float center[3];       // @0
  float side_len;        // @12
  int32_t parent;        // @16
  int32_t childs[8];     // @20
  uint8_t is_leaf_node;  // @52
  uint8_t pad0[3];
  int32_t trans_idx;     // @56
  uint8_t pad1[4];
};
struct alignas(32) F2nTransInfo {
  float w2xz[12][2][4];  // @0
  float weight[3][12];   // @384
  float center[3];       // @528
  float dis_summary;     // @540
};
struct alignas(32) F2nEdgePool {
  int32_t t_idx_a, t_idx_b;
  float center[3], dir_0[3], dir_1[3];
  uint8_t pad[20];
};
// Derived acceleration structure for the octree DFS (not a reference type): entry [u][c] describes child slot c of node u,
// so that expanding a node is ONE 256-byte read instead of "8 child indices, then 8 child nodes".
struct alignas(32) F2nChildInfo {
  float center[3];
  float side_len;
  int32_t child;      // node index, -1 = no child in this slot
  int32_t trans_idx;  // the child's trans_idx
  int32_t interior;   // the child has at least one child of its own
  int32_t pad;
};
static_assert(sizeof(F2nChildInfo) == 32, "layout");
static_assert(sizeof(F2nTreeNode) == 64 && sizeof(F2nTransInfo) == 544 && sizeof(F2nEdgePool) == 64, "layout");

#if defined(__HIPCC__)
// Wave issue priority of the compute-stream kernels.  The next batch's ray march runs on a second queue underneath the
// backward; it is VALU-bound (two waves per SIMD issuing back to back) and, at equal priority, takes every other issue slot
// from the MLP / compositing / scatter kernels it overlaps -- which are the step's critical path, while the march has slack.
// Measured (profiles/r02_notes: s_setprio(2) on the compute-stream kernels): fresh scene unchanged (1.287 vs 1.28 ms), converged
// scene WORSE (0.962 vs 0.93 ms: there the sampler's queue is as long as the compute queue, and the march slowed from 0.35 to
// 0.38 ms) -- left at the default priority.
#define F2N_RAISE_PRIO() ((void) 0)
// Sum over the 16 lanes of a DPP row, result in every lane (quad butterflies, row_half_mirror, row_mirror): four VALU
// instructions, no LDS round trip (a __shfl_xor ladder is four dependent ds_bpermute, ~100 cycles each for a lone wave).
__device__ __forceinline__ float f2n_row16_allsum(float v) {
  v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
  v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
  v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x141, 0xF, 0xF, true));  // row_half_mirror
  v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x140, 0xF, 0xF, true));  // row_mirror
  return v;
}
#endif

static inline int f2n_launch_status() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? F2N_OK : -(1000 + (int) e);
}

static inline unsigned f2n_div_up(long a, long b) { return (unsigned) ((a + b - 1) / b); }

// workspace.hip: internal per-device scratch + the partial-sum reduction used by the backward kernels
void* f2n_ws_get(int slot, size_t bytes);
int f2n_reduce_partials(void* stream, int n, int n_blocks, const float* partials, float* out);
// Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11): the counter-based generator
// behind the keyed draws of the host (host/KeyedDraws.h) where a kernel makes its own uniforms instead of reading a rand launch's
// output: key = (seed ^ purpose), counter = (element / 4, 0, sequence number).
__device__ __forceinline__ void f2n_philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t out[4]) {
#pragma unroll
  for (int r = 0; r < 10; r++) {
    const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
    c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
__device__ __forceinline__ float f2n_u01(uint32_t x) { return (float) (x >> 8) * (1.f / 16777216.f); }  // 24 bits: [0, 1)

int f2n_defer_reduction(int n, int n_blocks, const float* partials, float* out);  // folded later by f2n_reduce_deferred
#define F2N_WS_FIELD_DW 0
#define F2N_WS_SHADE_DW 1
#define F2N_WS_SHADE_EMB 2
#define F2N_WS_FIELD_PLANES 3
#define F2N_WS_LOSS 4
#define F2N_WS_BIN_REC 5
#define F2N_WS_BIN_CNT 6
#define F2N_WS_GATHER_CTR 7
#define F2N_WS_MLPG_W 8     // mlp_generic.hip: padded / transposed weights of the call
#define F2N_WS_MLPG_ACTS 9  // ... saved activations and hidden gradients (tile-transposed f16)
#define F2N_WS_MLPG_DW 10   // ... per-block partial weight gradients
#define F2N_WS_GATHER_BINS 11  // field.hip: request / result queues and slot lists of the slice-binned gather (big tables)
#define F2N_WS_BIN_OVF 12      // field.hip: the scatter's overflow lists (records that found their queue segment full)
#define F2N_WS_SLOTS 13
// mlp_generic.hip: the tcnn FullyFusedMLP shapes the two specialised kernels do not cover
bool f2n_mlpg_shape_ok(int d_in, int d_hidden, int n_hidden);
int f2n_mlpg_fwd(void* stream, int n, int d_in, int d_hidden, int n_hidden, const void* params_h, const float* x, void* out_h);
int f2n_mlpg_bwd(void* stream, int n, int d_in, int d_hidden, int n_hidden, float loss_scale, const void* params_h, const float* x,
                 const float* dy, float* dparams_f32_scaled, float* dx_f32);
#define F2N_PARTITION_MIN_N 8192  // below this the single fused gather+MLP launch wins (an unpartitioned gather runs at 91 G reads/s, a partitioned one at 250 G/s)

// ---- reductions in the order Eigen's scalar fixed-size unrollers use (n/2 | n - n/2 recursive split) ----
__device__ __forceinline__ float f2n_sum3(float a, float b, float c) { return a + (b + c); }
__device__ __forceinline__ float f2n_sum4(float a, float b, float c, float d) { return (a + b) + (c + d); }
__device__ __forceinline__ float f2n_sum12(const float* e) {
  return ((e[0] + (e[1] + e[2])) + (e[3] + (e[4] + e[5]))) + ((e[6] + (e[7] + e[8])) + (e[9] + (e[10] + e[11])));
}
__device__ __forceinline__ float f2n_norm3(float x, float y, float z) { return sqrtf(f2n_sum3(x * x, y * y, z * z)); }

// float -> u32 with the saturating semantics of v_cvt_u32_f32 / CUDA cvt.rzi.u32.f32, spelled out so the
// compiler cannot treat the out-of-range cast as undefined behaviour.
__device__ __forceinline__ uint32_t f2n_f2u_sat(float f) {
  if (!(f > 0.f)) return 0u;
  if (f >= 4294967296.f) return 0xffffffffu;
  return (uint32_t) f;
}

// Warp of one point by one leaf's perspective transform (PersSampler.cu:155-169) and its Jacobian (:171-187).
__device__ __forceinline__ void f2n_proj(const float* __restrict__ m /*2x4*/, const float* p, float& x, float& z) {
  x = f2n_sum4(m[0] * p[0], m[1] * p[1], m[2] * p[2], m[3] * 1.f);
  z = f2n_sum4(m[4] * p[0], m[5] * p[1], m[6] * p[2], m[7] * 1.f);
}

__device__ __forceinline__ void f2n_warp(const F2nTransInfo* __restrict__ tr, const float* p, float* out) {
  float v[12];
#pragma unroll
  for (int i = 0; i < 12; i++) {
    float x, z;
    f2n_proj(&tr->w2xz[i][0][0], p, x, z);
    v[i] = x / z;
  }
#pragma unroll
  for (int r = 0; r < 3; r++) {
    float e[12];
#pragma unroll
    for (int i = 0; i < 12; i++) e[i] = tr->weight[r][i] * v[i];
    out[r] = f2n_sum12(e);
  }
}

__device__ __forceinline__ void f2n_warp_jac(const F2nTransInfo* __restrict__ tr, const float* p, float jac[3][3]) {
  float tj[12][3];
#pragma unroll
  for (int i = 0; i < 12; i++) {
    float x, z;
    f2n_proj(&tr->w2xz[i][0][0], p, x, z);
    float d0 = 1 / z;
    float d1 = -x / (z * z);
#pragma unroll
    for (int c = 0; c < 3; c++) tj[i][c] = d0 * tr->w2xz[i][0][c] + d1 * tr->w2xz[i][1][c];
  }
#pragma unroll
  for (int r = 0; r < 3; r++)
#pragma unroll
    for (int c = 0; c < 3; c++) {
      float e[12];
#pragma unroll
      for (int i = 0; i < 12; i++) e[i] = tr->weight[r][i] * tj[i][c];
      jac[r][c] = f2n_sum12(e);
    }
}
