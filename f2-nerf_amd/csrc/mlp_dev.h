// Register-resident fully-fused MLP on the gfx950 matrix cores (v_mfma_f32_16x16x32_f16).
//
// Replaces tiny-cuda-nn's FullyFusedMLP (Field/TCNNWP.cpp:79-243 call sites) for the two networks of the
// reference configs: 32 -> 64 (-> 64) -> 16, ReLU, no bias, fp16 weights and inter-layer activations, fp32
// accumulation.  This is NOT a WMMA-shaped tiling recompiled for AMD: there is no LDS staging of activations
// and no block-level barrier in the forward path.  One wave64 owns 16 samples at a time and the whole layer
// chain lives in its VGPRs:
//
//   mfma(A,B): D[i][j] = sum_k A[i][k] * B[k][j];  lane l = (c = l & 15, g = l >> 4)
//     A operand: lane holds A[c][slot(g, 0..7)]      B operand: lane holds B[slot(g, 0..7)][c]
//     D result : lane holds D[4g + r][c], r = 0..3
//
//   "row fragment" of a matrix M[rows][K] (tile row0, k0): lane (c,g) holds M[row0 + c][k0 + sigma(g,e)],
//   e = 0..7, with the K-slot map  sigma(g,e) = 16*(e>>2) + 4g + (e&3).
//   Because A and B use the SAME slot map the contraction is independent of how the hardware labels K.
//   sigma is chosen so that the D tiles of one MFMA ARE the row fragment of the next one: the four fp32
//   results a lane holds for output tile 2q (rows 4g..4g+3) and tile 2q+1 fill slots e = 0..3 and 4..7 of the
//   K-block q of the following layer.  Hence:
//
//     sample-column orientation:  T[neuron][sample] = mfma(A = W row fragment, B = X row fragment)
//        -> relu -> cvt f16 -> is the X row fragment of the next layer (no transpose, no LDS).
//     sample-row orientation:     T'[sample][neuron] = mfma(A = X row fragment, B = W row fragment)
//        -> lanes hold 4 consecutive SAMPLES of one neuron: exactly the row fragment needed when the
//           contraction runs over samples (weight gradients).  Any tile can be re-oriented with one MFMA
//           against an identity fragment (exact for f16 data).
#pragma once
#include "f2n_dev.h"

#define F2N_D_IN 32
#define F2N_D_HID 64
#define F2N_D_OUT 16

__device__ __forceinline__ float4_t f2n_mfma(half8_t a, half8_t b, float4_t c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}

// ---- the accumulator of the forward contractions --------------------------------------------------------------------------
// tiny-cuda-nn is not in the reference tree, so the accumulator type of its FullyFusedMLP forward cannot be pinned (DESIGN.md
// section 6).  The product build accumulates in fp32 (what the MFMA does).  A build with -DF2N_REFERENCE_NUMERICS=1 (the
// `refnum` variant of f2-nerf_amd/build.py: libf2n_hip_refnum.so) takes the OTHER plausible reading, the oracle's accumulator
// mode 1 (oracle/f2n_oracle.c: or_mlp_dot): a binary16 accumulator fragment of m16n16k16 tiles -- the 16 products of a k-block
// are summed in fp32 (one v_mfma_f32_16x16x16_f16 into a zero accumulator) and the running sum is rounded to f16 after every
// block.  It exists to A/B whole trainings (bench.py: psnr_numerics_ab), not to be fast.
#ifndef F2N_REFERENCE_NUMERICS
#define F2N_REFERENCE_NUMERICS 0
#endif
__device__ __forceinline__ float4_t f2n_round_h4(float4_t v) {
  float4_t r;
#pragma unroll
  for (int i = 0; i < 4; i++) r[i] = (float) (half_t) v[i];
  return r;
}
// acc + sum over the 32-wide K-block the two fragments hold (forward / activation-recompute contractions only)
__device__ __forceinline__ float4_t f2n_mfma_fwd(half8_t a, half8_t b, float4_t acc) {
#if F2N_REFERENCE_NUMERICS
  const float4_t z = {0.f, 0.f, 0.f, 0.f};
  // slots e = 0..3 of a fragment are k = 4g..4g+3 of the block's first 16 columns, e = 4..7 those of its second 16
  const half4_t a0 = __builtin_shufflevector(a, a, 0, 1, 2, 3), a1 = __builtin_shufflevector(a, a, 4, 5, 6, 7);
  const half4_t b0 = __builtin_shufflevector(b, b, 0, 1, 2, 3), b1 = __builtin_shufflevector(b, b, 4, 5, 6, 7);
  float4_t blk = __builtin_amdgcn_mfma_f32_16x16x16f16(a0, b0, z, 0, 0, 0);
#pragma unroll
  for (int i = 0; i < 4; i++) acc[i] = acc[i] + blk[i];
  acc = f2n_round_h4(acc);
  blk = __builtin_amdgcn_mfma_f32_16x16x16f16(a1, b1, z, 0, 0, 0);
#pragma unroll
  for (int i = 0; i < 4; i++) acc[i] = acc[i] + blk[i];
  return f2n_round_h4(acc);
#else
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
#endif
}

__device__ __forceinline__ half8_t f2n_cat(half4_t lo, half4_t hi) {
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

// Row fragment straight from memory (global or LDS): M is row-major with leading dimension ld (halves).
__device__ __forceinline__ half8_t f2n_rowfrag(const half_t* M, int ld, int row, int k0, int g) {
  const half_t* p = M + (size_t) row * ld + k0 + 4 * g;
  return f2n_cat(*(const half4_t*) p, *(const half4_t*) (p + 16));
}

// Two fp32 D tiles -> f16 row fragment of the next contraction (optionally through ReLU).  The ReLU is applied AFTER the
// rounding, on packed halves (v_pk_max_f16: one instruction per two values instead of one v_max_f32 each; rounding to nearest
// is monotone and keeps the sign, and max(-0, +0) = +0 on this ISA, so relu(round(x)) has the bits of round(relu(x))).
template <bool RELU>
__device__ __forceinline__ half8_t f2n_pack(float4_t a, float4_t b) {
  half8_t r;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    r[i] = (half_t) a[i];
    r[4 + i] = (half_t) b[i];
  }
  if (RELU) {
    const half8_t z = {0, 0, 0, 0, 0, 0, 0, 0};
    r = __builtin_elementwise_max(r, z);
  }
  return r;
}

template <bool RELU>
__device__ __forceinline__ half4_t f2n_cvt4(float4_t a) {
  half4_t r;
#pragma unroll
  for (int i = 0; i < 4; i++) r[i] = (half_t) a[i];
  if (RELU) {
    const half4_t z = {0, 0, 0, 0};
    r = __builtin_elementwise_max(r, z);
  }
  return r;
}

// D tile masked by the sign of a pre-activation tile (ReLU backward), both in the same orientation.
__device__ __forceinline__ float4_t f2n_relu_mask(float4_t grad, float4_t pre) {
  float4_t r;
#pragma unroll
  for (int i = 0; i < 4; i++) r[i] = pre[i] > 0.f ? grad[i] : 0.f;
  return r;
}

// Identity row fragment: I_ft[c][k] = (k == 16*ft + c).  mfma(A = X fragment, B = this) re-orients a tile.
__device__ __forceinline__ half8_t f2n_identity_frag(int ft, int c, int g) {
  half8_t r;
#pragma unroll
  for (int e = 0; e < 8; e++) {
    const int k = 16 * (e >> 2) + 4 * g + (e & 3);
    r[e] = (k == 16 * ft + c) ? (half_t) 1.f : (half_t) 0.f;
  }
  return r;
}

// Weight fragments of one MLP held in registers for the forward chain.
template <int NH>
struct F2nMlpFwdW {
  half8_t w0[4];                   // layer 0: 64x32, tiles of 16 neurons
  half8_t w1[NH == 2 ? 8 : 1];     // layer 1: 64x64, [tile][k-block]
  half8_t wo[2];                   // output: 16x64, [k-block]

  __device__ __forceinline__ void load(const half_t* __restrict__ params, int c, int g) {
    const half_t* p = params;
#pragma unroll
    for (int t = 0; t < 4; t++) w0[t] = f2n_rowfrag(p, F2N_D_IN, 16 * t + c, 0, g);
    p += F2N_D_HID * F2N_D_IN;
    if (NH == 2) {
#pragma unroll
      for (int t = 0; t < 4; t++)
#pragma unroll
        for (int q = 0; q < 2; q++) w1[t * 2 + q] = f2n_rowfrag(p, F2N_D_HID, 16 * t + c, 32 * q, g);
      p += F2N_D_HID * F2N_D_HID;
    }
#pragma unroll
    for (int q = 0; q < 2; q++) wo[q] = f2n_rowfrag(p, F2N_D_HID, c, 32 * q, g);
  }

  // xf: row fragment of X[sample][32].  Returns O^T tile: lane (c = sample, g) holds outputs 4g..4g+3 (fp32,
  // not yet rounded to the f16 output precision).
  __device__ __forceinline__ float4_t forward(half8_t xf) const {
    const float4_t z = {0.f, 0.f, 0.f, 0.f};
    float4_t t[4];
#pragma unroll
    for (int i = 0; i < 4; i++) t[i] = f2n_mfma_fwd(w0[i], xf, z);
    half8_t h0 = f2n_pack<true>(t[0], t[1]), h1 = f2n_pack<true>(t[2], t[3]);
    if (NH == 2) {
#pragma unroll
      for (int i = 0; i < 4; i++) {
        t[i] = f2n_mfma_fwd(w1[i * 2], h0, z);
        t[i] = f2n_mfma_fwd(w1[i * 2 + 1], h1, t[i]);
      }
      h0 = f2n_pack<true>(t[0], t[1]);
      h1 = f2n_pack<true>(t[2], t[3]);
    }
    float4_t o = f2n_mfma_fwd(wo[0], h0, z);
    return f2n_mfma_fwd(wo[1], h1, o);
  }
};

// The same forward chain with the weight fragments in LDS instead of registers: [fragment][lane] x 16 bytes, so a fragment
// is one conflict-free ds_read_b128 per lane right before the MFMA that consumes it.  The two networks' fragments cost 80
// VGPRs in the register-resident form (152 registers, three waves per SIMD in the fused field + colour forward); here they
// cost 20 KB of LDS per block and the kernel keeps twice the waves resident to hide its tile's dependency chain.  `w` must
// already include the lane offset and be opaque to the optimiser (else the reads are hoisted back into registers).
template <int NH>
struct F2nMlpFwdWLds {
  static constexpr int N_FRAG = 4 + (NH == 2 ? 8 : 0) + 2;
  static __device__ __forceinline__ void fill(half8_t* dst /*[N_FRAG][64]*/, const half_t* __restrict__ params, int lane) {
    F2nMlpFwdW<NH> w;
    w.load(params, lane & 15, lane >> 4);
#pragma unroll
    for (int t = 0; t < 4; t++) dst[t * 64 + lane] = w.w0[t];
    if (NH == 2) {
#pragma unroll
      for (int t = 0; t < 8; t++) dst[(4 + t) * 64 + lane] = w.w1[t];
    }
#pragma unroll
    for (int q = 0; q < 2; q++) dst[(N_FRAG - 2 + q) * 64 + lane] = w.wo[q];
  }
  static __device__ __forceinline__ float4_t forward(const half8_t* w, half8_t xf) {
    const float4_t z = {0.f, 0.f, 0.f, 0.f};
    float4_t t[4];
#pragma unroll
    for (int i = 0; i < 4; i++) t[i] = f2n_mfma_fwd(w[i * 64], xf, z);
    half8_t h0 = f2n_pack<true>(t[0], t[1]), h1 = f2n_pack<true>(t[2], t[3]);
    if (NH == 2) {
#pragma unroll
      for (int i = 0; i < 4; i++) {
        t[i] = f2n_mfma_fwd(w[(4 + i * 2) * 64], h0, z);
        t[i] = f2n_mfma_fwd(w[(4 + i * 2 + 1) * 64], h1, t[i]);
      }
      h0 = f2n_pack<true>(t[0], t[1]);
      h1 = f2n_pack<true>(t[2], t[3]);
    }
    float4_t o = f2n_mfma_fwd(w[(N_FRAG - 2) * 64], h0, z);
    return f2n_mfma_fwd(w[(N_FRAG - 1) * 64], h1, o);
  }
};

// ---------------------------------------------------------------------------------------------------
// Backward: weights and their transposes live in LDS (padded rows: +4 halves keeps ds_read_b64 fragment
// reads conflict-free), weight-gradient accumulators live in registers for the whole kernel.
// ---------------------------------------------------------------------------------------------------
#define F2N_LD32 36  // leading dimension of [*][32] LDS matrices
#define F2N_LD64 68  // leading dimension of [*][64] LDS matrices

template <int NH>
struct F2nMlpLds {
  half_t w0[F2N_D_HID * F2N_LD32];                      // W0   [64][32]
  half_t w0t[F2N_D_IN * F2N_LD64];                      // W0^T [32][64]
  half_t w1[NH == 2 ? F2N_D_HID * F2N_LD64 : 4];        // W1   [64][64]
  half_t w1t[NH == 2 ? F2N_D_HID * F2N_LD64 : 4];       // W1^T [64][64]
  half_t wot[F2N_D_HID * F2N_LD32];                     // Wo^T [64][16 | 16 zeros]
};

// params must be 16-byte aligned (every MLP parameter block of the host is a tensor allocation).  All global loads are
// issued before the first LDS write: a 256-thread block with nothing else resident pays ONE memory round trip here,
// not one per element (the element-wise version of this fill was ~15 % of the backward kernels' time).
template <int NH>
__device__ __forceinline__ void f2n_mlp_lds_fill(F2nMlpLds<NH>& s, const half_t* __restrict__ params, int tid, int nthreads) {
  constexpr int C0 = F2N_D_HID * F2N_D_IN / 8;                  // 8-half chunks of W0 rows
  constexpr int C1 = NH == 2 ? F2N_D_HID * F2N_D_HID / 8 : 0;  // ... of W1 rows
  constexpr int CO = F2N_D_OUT * F2N_D_HID / 8;                 // ... of Wo rows
  constexpr int ROUNDS = (C0 + C1 + CO + 255) / 256;            // nthreads >= 256 at every call site
  half8_t v[ROUNDS];
#pragma unroll
  for (int it = 0; it < ROUNDS; it++) {
    const int i = tid + it * nthreads;
    v[it] = *(const half8_t*) (params + 8 * (size_t) (i < C0 + C1 + CO ? i : 0));
  }
#pragma unroll
  for (int it = 0; it < ROUNDS; it++) {
    const int i = tid + it * nthreads;
    const half4_t lo = __builtin_shufflevector(v[it], v[it], 0, 1, 2, 3), hi = __builtin_shufflevector(v[it], v[it], 4, 5, 6, 7);
    if (i < C0) {
      const int r = i / (F2N_D_IN / 8), k = (i % (F2N_D_IN / 8)) * 8;
      *(half4_t*) (s.w0 + r * F2N_LD32 + k) = lo;
      *(half4_t*) (s.w0 + r * F2N_LD32 + k + 4) = hi;
#pragma unroll
      for (int e = 0; e < 8; e++) s.w0t[(k + e) * F2N_LD64 + r] = v[it][e];
    } else if (i < C0 + C1) {
      const int j = i - C0, r = j / (F2N_D_HID / 8), k = (j % (F2N_D_HID / 8)) * 8;
      *(half4_t*) (s.w1 + r * F2N_LD64 + k) = lo;
      *(half4_t*) (s.w1 + r * F2N_LD64 + k + 4) = hi;
#pragma unroll
      for (int e = 0; e < 8; e++) s.w1t[(k + e) * F2N_LD64 + r] = v[it][e];
    } else if (i < C0 + C1 + CO) {
      const int j = i - C0 - C1, o = j / (F2N_D_HID / 8), k = (j % (F2N_D_HID / 8)) * 8;
#pragma unroll
      for (int e = 0; e < 8; e++) s.wot[(k + e) * F2N_LD32 + o] = v[it][e];
    }
  }
  for (int i = tid; i < F2N_D_HID * 16; i += nthreads) s.wot[(i >> 4) * F2N_LD32 + 16 + (i & 15)] = (half_t) 0.f;
}

template <int NH>
struct F2nMlpGradAcc {
  float4_t dwo[4];                    // [16 o][64 j]: tile t -> j = 16t + c, o = 4g + r
  float4_t dw1[NH == 2 ? 16 : 1];     // [64][64]: tile (to, ti) -> j_in = 16ti + c, j_out = 16to + 4g + r
  float4_t dw0[8];                    // [64][32]: tile (t, ft) -> i = 16ft + c, j = 16t + 4g + r

  __device__ __forceinline__ void zero() {
    const float4_t z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 4; i++) dwo[i] = z;
#pragma unroll
    for (int i = 0; i < (NH == 2 ? 16 : 1); i++) dw1[i] = z;
#pragma unroll
    for (int i = 0; i < 8; i++) dw0[i] = z;
  }
};

// Everything the backward of one 16-sample half produces in the sample-ROW orientation (inputs to the
// sample-contracted weight-gradient MFMAs) plus dX^T in the sample-COLUMN orientation.
template <int NH>
struct F2nHalfBwd {
  float4_t dxT[2];                  // dX^T tiles ft: lane (c = sample, g) holds features 16ft + 4g + r (scaled)
  // sample-row tiles, already rounded to f16 (that is the precision they are consumed in): 2 VGPRs per tile
  half4_t dyR;                      // dY      [sample][o]      lane (c = o,      g): samples 4g + r
  half4_t xR[2];                    // X       [sample][16ft+c]
  half4_t hlR[4];                   // H_last  [sample][16t+c]  (post-ReLU)
  half4_t glR[4];                   // G_last  [sample][16t+c]  (masked hidden gradient of the last hidden layer)
  half4_t h0R[NH == 2 ? 4 : 1];     // H_0     (NH == 2 only)
  half4_t g0R[NH == 2 ? 4 : 1];     // G_0     (NH == 2 only)
};

// xf: X row fragment.  dy_fn(hl0, hl1) returns the dY row fragment (K = output index, slots >= 16 zero, already
// loss-scaled f16) given the post-ReLU activations of the last hidden layer as row fragments -- a caller whose dY
// depends on the network output (the colour path: sigmoid derivative) computes it from them instead of running the
// forward chain a second time; a caller with a given dY ignores the arguments.
template <int NH, int N_DX_TILES, typename DyFn>
__device__ __forceinline__ void f2n_mlp_half_bwd(const F2nMlpLds<NH>& s, half8_t xf, DyFn dy_fn, const half8_t* idf /*[2]*/,
                                                 int c, int g, F2nHalfBwd<NH>& out) {
  const float4_t z = {0.f, 0.f, 0.f, 0.f};
  // ---- sample-column orientation: recompute pre-activations, then the hidden-gradient chain ----
  float4_t t0[4], t1[4];
#pragma unroll
  for (int t = 0; t < 4; t++) t0[t] = f2n_mfma_fwd(f2n_rowfrag(s.w0, F2N_LD32, 16 * t + c, 0, g), xf, z);
  half8_t h0f[2] = {f2n_pack<true>(t0[0], t0[1]), f2n_pack<true>(t0[2], t0[3])};
  if (NH == 2) {
#pragma unroll
    for (int t = 0; t < 4; t++) {
      t1[t] = f2n_mfma_fwd(f2n_rowfrag(s.w1, F2N_LD64, 16 * t + c, 0, g), h0f[0], z);
      t1[t] = f2n_mfma_fwd(f2n_rowfrag(s.w1, F2N_LD64, 16 * t + c, 32, g), h0f[1], t1[t]);
    }
  }
  half8_t hlf[2] = {h0f[0], h0f[1]};  // post-ReLU activations of the LAST hidden layer, as row fragments
  if (NH == 2) {
    hlf[0] = f2n_pack<true>(t1[0], t1[1]);
    hlf[1] = f2n_pack<true>(t1[2], t1[3]);
  }
  const half8_t dyf = dy_fn(hlf[0], hlf[1]);
  float4_t gl[4];
#pragma unroll
  for (int t = 0; t < 4; t++) {
    gl[t] = f2n_mfma(f2n_rowfrag(s.wot, F2N_LD32, 16 * t + c, 0, g), dyf, z);
    gl[t] = f2n_relu_mask(gl[t], NH == 2 ? t1[t] : t0[t]);
  }
  half8_t gff[2] = {f2n_pack<false>(gl[0], gl[1]), f2n_pack<false>(gl[2], gl[3])};  // G_last row fragments
  half8_t glf[2] = {gff[0], gff[1]};
  if (NH == 2) {
    float4_t g0[4];
#pragma unroll
    for (int t = 0; t < 4; t++) {
      g0[t] = f2n_mfma(f2n_rowfrag(s.w1t, F2N_LD64, 16 * t + c, 0, g), glf[0], z);
      g0[t] = f2n_mfma(f2n_rowfrag(s.w1t, F2N_LD64, 16 * t + c, 32, g), glf[1], g0[t]);
      g0[t] = f2n_relu_mask(g0[t], t0[t]);
    }
    gff[0] = f2n_pack<false>(g0[0], g0[1]);
    gff[1] = f2n_pack<false>(g0[2], g0[3]);
  }
#pragma unroll
  for (int ft = 0; ft < N_DX_TILES; ft++) {
    out.dxT[ft] = f2n_mfma(f2n_rowfrag(s.w0t, F2N_LD64, 16 * ft + c, 0, g), gff[0], z);
    out.dxT[ft] = f2n_mfma(f2n_rowfrag(s.w0t, F2N_LD64, 16 * ft + c, 32, g), gff[1], out.dxT[ft]);
  }
  // ---- sample-row orientation: the SAME f16 quantities with samples in the register index, obtained by re-orienting the
  // packed fragments with one identity MFMA per 16-neuron tile (exact for f16 data) -- not by recomputing the chain with
  // swapped operands, which cost 24 weight-fragment LDS reads and 24 MFMAs per 16 samples instead of 16 MFMAs and none ----
#pragma unroll
  for (int t = 0; t < 4; t++) {
    out.hlR[t] = f2n_cvt4<false>(f2n_mfma(hlf[t >> 1], idf[t & 1], z));
    out.glR[t] = f2n_cvt4<false>(f2n_mfma(glf[t >> 1], idf[t & 1], z));
    if (NH == 2) {
      out.h0R[t] = f2n_cvt4<false>(f2n_mfma(h0f[t >> 1], idf[t & 1], z));
      out.g0R[t] = f2n_cvt4<false>(f2n_mfma(gff[t >> 1], idf[t & 1], z));
    }
  }
  out.dyR = f2n_cvt4<false>(f2n_mfma(dyf, idf[0], z));
  out.xR[0] = f2n_cvt4<false>(f2n_mfma(xf, idf[0], z));
  out.xR[1] = f2n_cvt4<false>(f2n_mfma(xf, idf[1], z));
}

// Weight-gradient MFMAs of ONE 16-sample half: v_mfma_f32_16x16x16_f16 contracts exactly the 16 samples a half's
// sample-row tiles hold (lane (c = neuron, g): samples 4g..4g+3 = K-slots 4g..4g+3 of both operands), so a half's tiles
// are consumed as soon as they exist instead of being held until its partner half is done (K = 32 needs both): ~40
// fewer live registers in the backward loops.
__device__ __forceinline__ float4_t f2n_mfma16(half4_t a, half4_t b, float4_t c) {
  return __builtin_amdgcn_mfma_f32_16x16x16f16(a, b, c, 0, 0, 0);
}

template <int NH>
__device__ __forceinline__ void f2n_mlp_accumulate_dw_half(const F2nHalfBwd<NH>& a, F2nMlpGradAcc<NH>& acc) {
#pragma unroll
  for (int t = 0; t < 4; t++) acc.dwo[t] = f2n_mfma16(a.dyR, a.hlR[t], acc.dwo[t]);
  if (NH == 2) {
#pragma unroll
    for (int to = 0; to < 4; to++)
#pragma unroll
      for (int ti = 0; ti < 4; ti++) acc.dw1[to * 4 + ti] = f2n_mfma16(a.glR[to], a.h0R[ti], acc.dw1[to * 4 + ti]);
#pragma unroll
    for (int t = 0; t < 4; t++)
#pragma unroll
      for (int ft = 0; ft < 2; ft++) acc.dw0[t * 2 + ft] = f2n_mfma16(a.g0R[t], a.xR[ft], acc.dw0[t * 2 + ft]);
  } else {
#pragma unroll
    for (int t = 0; t < 4; t++)
#pragma unroll
      for (int ft = 0; ft < 2; ft++) acc.dw0[t * 2 + ft] = f2n_mfma16(a.glR[t], a.xR[ft], acc.dw0[t * 2 + ft]);
  }
}

// Block-level reduction of the per-wave accumulators through LDS, then the block's partial parameter gradient is
// written with plain coalesced stores to partials[blockIdx.x][n_params]; f2n_reduce_partials sums the blocks afterwards.
// The waves take turns on the LDS image with plain read-add-write (wave 0 stores): ds_add_f32 runs at 0.33 lane-ops/clk
// per CU on gfx950 (tools/lds_atomic_probe), which made the atomic version of this flush ~35 us for the colour network;
// the turn order also makes the sum deterministic.  s_acc holds TWO images of n_params floats (even / odd waves: half the
// turns); no initialisation needed.  The caller must have a __syncthreads() between the last use of whatever s_acc
// aliases and this call.
template <int NH>
__device__ __forceinline__ void f2n_mlp_flush_dw(const F2nMlpGradAcc<NH>& acc, float* s_acc, float* __restrict__ partials,
                                                 int c, int g, int tid, int nthreads) {
  const int off1 = F2N_D_HID * F2N_D_IN;
  const int offo = off1 + (NH == 2 ? F2N_D_HID * F2N_D_HID : 0);
  const int n_params = offo + F2N_D_OUT * F2N_D_HID;
  const int wave = tid >> 6, n_waves = nthreads >> 6;
  for (int w = 0; w < n_waves; w += 2) {
    if ((wave & ~1) == w) {
      const bool first = w == 0;
      float* img = s_acc + (wave & 1) * n_params;
      auto put = [&](int idx, float v) { img[idx] = first ? v : img[idx] + v; };
#pragma unroll
      for (int t = 0; t < 4; t++)
#pragma unroll
        for (int ft = 0; ft < 2; ft++)
#pragma unroll
          for (int r = 0; r < 4; r++) put((16 * t + 4 * g + r) * F2N_D_IN + 16 * ft + c, acc.dw0[t * 2 + ft][r]);
      if (NH == 2) {
#pragma unroll
        for (int to = 0; to < 4; to++)
#pragma unroll
          for (int ti = 0; ti < 4; ti++)
#pragma unroll
            for (int r = 0; r < 4; r++) put(off1 + (16 * to + 4 * g + r) * F2N_D_HID + 16 * ti + c, acc.dw1[to * 4 + ti][r]);
      }
#pragma unroll
      for (int t = 0; t < 4; t++)
#pragma unroll
        for (int r = 0; r < 4; r++) put(offo + (4 * g + r) * F2N_D_HID + 16 * t + c, acc.dwo[t][r]);
    }
    __syncthreads();
  }
  float* dst = partials + (size_t) blockIdx.x * n_params;
  for (int i = tid; i < n_params; i += nthreads) dst[i] = s_acc[i] + s_acc[n_params + i];
}
