// SHShader on gfx950: real spherical harmonics (Shader/SHShader.cu:10-118) + appearance embedding
// (Utils/CustomOps/Scatter.cu:10-40) + colour MLP 32->64->64->16 + scaled sigmoid (Shader/SHShader.cpp:23-29),
// fused: the MLP input row fragment is assembled in registers from feat/app_emb/dir and never stored as fp32.
#include "mlp_dev.h"

// Degree-4 real SH basis in the reference's polynomial forms and operation order (SHShader.cu:25-50).
__device__ __forceinline__ void f2n_sh16(float x, float y, float z, float* o) {
  const float xy = x * y, xz = x * z, yz = y * z, x2 = x * x, y2 = y * y, z2 = z * z;
  o[0] = 0.28209479177387814f;
  o[1] = -0.48860251190291987f * y;
  o[2] = 0.48860251190291987f * z;
  o[3] = -0.48860251190291987f * x;
  o[4] = 1.0925484305920792f * xy;
  o[5] = -1.0925484305920792f * yz;
  o[6] = 0.94617469575755997f * z2 - 0.31539156525251999f;
  o[7] = -1.0925484305920792f * xz;
  o[8] = 0.54627421529603959f * x2 - 0.54627421529603959f * y2;
  o[9] = 0.59004358992664352f * y * (-3.0f * x2 + y2);
  o[10] = 2.8906114426405538f * xy * z;
  o[11] = 0.45704579946446572f * y * (1.0f - 5.0f * z2);
  o[12] = 0.3731763325901154f * z * (5.0f * z2 - 3.0f);
  o[13] = 0.45704579946446572f * x * (1.0f - 5.0f * z2);
  o[14] = 1.4453057213202769f * z * (x2 - y2);
  o[15] = 0.59004358992664352f * x * (-x2 + 3.0f * y2);
}

// Degrees 5..8 (SHShader.cu:51-102): only the stand-alone encoding seam offers them -- the fused colour path is built for
// the 16 + 16 inputs every shipped config uses.  Same polynomial forms and operation order as the reference.
__device__ __forceinline__ void f2n_sh_high(int degree, float x, float y, float z, float* o) {
  const float xy = x * y, xz = x * z, yz = y * z, x2 = x * x, y2 = y * y, z2 = z * z;
  const float x4 = x2 * x2, y4 = y2 * y2, z4 = z2 * z2;
  const float x6 = x4 * x2, y6 = y4 * y2, z6 = z4 * z2;
  if (degree <= 4) return;
  o[16] = 2.5033429417967046f*xy*(x2 - y2);
  o[17] = 1.7701307697799304f*yz*(-3.0f*x2 + y2);
  o[18] = 0.94617469575756008f*xy*(7.0f*z2 - 1.0f);
  o[19] = 0.66904654355728921f*yz*(3.0f - 7.0f*z2);
  o[20] = -3.1735664074561294f*z2 + 3.7024941420321507f*z4 + 0.31735664074561293f;
  o[21] = 0.66904654355728921f*xz*(3.0f - 7.0f*z2);
  o[22] = 0.47308734787878004f*(x2 - y2)*(7.0f*z2 - 1.0f);
  o[23] = 1.7701307697799304f*xz*(-x2 + 3.0f*y2);
  o[24] = -3.7550144126950569f*x2*y2 + 0.62583573544917614f*x4 + 0.62583573544917614f*y4;
  if (degree <= 5) return;
  o[25] = 0.65638205684017015f*y*(10.0f*x2*y2 - 5.0f*x4 - y4);
  o[26] = 8.3026492595241645f*xy*z*(x2 - y2);
  o[27] = -0.48923829943525038f*y*(3.0f*x2 - y2)*(9.0f*z2 - 1.0f);
  o[28] = 4.7935367849733241f*xy*z*(3.0f*z2 - 1.0f);
  o[29] = 0.45294665119569694f*y*(14.0f*z2 - 21.0f*z4 - 1.0f);
  o[30] = 0.1169503224534236f*z*(-70.0f*z2 + 63.0f*z4 + 15.0f);
  o[31] = 0.45294665119569694f*x*(14.0f*z2 - 21.0f*z4 - 1.0f);
  o[32] = 2.3967683924866621f*z*(x2 - y2)*(3.0f*z2 - 1.0f);
  o[33] = -0.48923829943525038f*x*(x2 - 3.0f*y2)*(9.0f*z2 - 1.0f);
  o[34] = 2.0756623148810411f*z*(-6.0f*x2*y2 + x4 + y4);
  o[35] = 0.65638205684017015f*x*(10.0f*x2*y2 - x4 - 5.0f*y4);
  if (degree <= 6) return;
  o[36] = 1.3663682103838286f*xy*(-10.0f*x2*y2 + 3.0f*x4 + 3.0f*y4);
  o[37] = 2.3666191622317521f*yz*(10.0f*x2*y2 - 5.0f*x4 - y4);
  o[38] = 2.0182596029148963f*xy*(x2 - y2)*(11.0f*z2 - 1.0f);
  o[39] = -0.92120525951492349f*yz*(3.0f*x2 - y2)*(11.0f*z2 - 3.0f);
  o[40] = 0.92120525951492349f*xy*(-18.0f*z2 + 33.0f*z4 + 1.0f);
  o[41] = 0.58262136251873131f*yz*(30.0f*z2 - 33.0f*z4 - 5.0f);
  o[42] = 6.6747662381009842f*z2 - 20.024298714302954f*z4 + 14.684485723822165f*z6 - 0.31784601133814211f;
  o[43] = 0.58262136251873131f*xz*(30.0f*z2 - 33.0f*z4 - 5.0f);
  o[44] = 0.46060262975746175f*(x2 - y2)*(11.0f*z2*(3.0f*z2 - 1.0f) - 7.0f*z2 + 1.0f);
  o[45] = -0.92120525951492349f*xz*(x2 - 3.0f*y2)*(11.0f*z2 - 3.0f);
  o[46] = 0.50456490072872406f*(11.0f*z2 - 1.0f)*(-6.0f*x2*y2 + x4 + y4);
  o[47] = 2.3666191622317521f*xz*(10.0f*x2*y2 - x4 - 5.0f*y4);
  o[48] = 10.247761577878714f*x2*y4 - 10.247761577878714f*x4*y2 + 0.6831841051919143f*x6 - 0.6831841051919143f*y6;
  if (degree <= 7) return;
  o[49] = 0.70716273252459627f*y*(-21.0f*x2*y4 + 35.0f*x4*y2 - 7.0f*x6 + y6);
  o[50] = 5.2919213236038001f*xy*z*(-10.0f*x2*y2 + 3.0f*x4 + 3.0f*y4);
  o[51] = -0.51891557872026028f*y*(13.0f*z2 - 1.0f)*(-10.0f*x2*y2 + 5.0f*x4 + y4);
  o[52] = 4.1513246297620823f*xy*z*(x2 - y2)*(13.0f*z2 - 3.0f);
  o[53] = -0.15645893386229404f*y*(3.0f*x2 - y2)*(13.0f*z2*(11.0f*z2 - 3.0f) - 27.0f*z2 + 3.0f);
  o[54] = 0.44253269244498261f*xy*z*(-110.0f*z2 + 143.0f*z4 + 15.0f);
  o[55] = 0.090331607582517306f*y*(-135.0f*z2 + 495.0f*z4 - 429.0f*z6 + 5.0f);
  o[56] = 0.068284276912004949f*z*(315.0f*z2 - 693.0f*z4 + 429.0f*z6 - 35.0f);
  o[57] = 0.090331607582517306f*x*(-135.0f*z2 + 495.0f*z4 - 429.0f*z6 + 5.0f);
  o[58] = 0.07375544874083044f*z*(x2 - y2)*(143.0f*z2*(3.0f*z2 - 1.0f) - 187.0f*z2 + 45.0f);
  o[59] = -0.15645893386229404f*x*(x2 - 3.0f*y2)*(13.0f*z2*(11.0f*z2 - 3.0f) - 27.0f*z2 + 3.0f);
  o[60] = 1.0378311574405206f*z*(13.0f*z2 - 3.0f)*(-6.0f*x2*y2 + x4 + y4);
  o[61] = -0.51891557872026028f*x*(13.0f*z2 - 1.0f)*(-10.0f*x2*y2 + x4 + 5.0f*y4);
  o[62] = 2.6459606618019f*z*(15.0f*x2*y4 - 15.0f*x4*y2 + x6 - y6);
  o[63] = 0.70716273252459627f*x*(-35.0f*x2*y4 + 21.0f*x4*y2 - x6 + 7.0f*y6);
}

__global__ void sh_encode_kernel(int n, int degree, const float* __restrict__ dirs, float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float sh[64];
  const float x = dirs[3 * i], y = dirs[3 * i + 1], z = dirs[3 * i + 2];
  f2n_sh16(x, y, z, sh);
  if (degree > 4) f2n_sh_high(degree, x, y, z, sh);
  const int w = degree * degree;
  for (int k = 0; k < w; k++) out[(size_t) i * w + k] = sh[k];
}

// ScatterIdxKernal, Scatter.cu:110-120: broadcast a per-ray value to the ray's samples (one wave per ray).
__global__ void scatter_idx_kernel(int n_rays, const int32_t* __restrict__ start_end, const int32_t* __restrict__ ray_val,
                                   int32_t* __restrict__ out) {
  const int ray = blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);
  if (ray >= n_rays) return;
  const int s = start_end[2 * ray], e = start_end[2 * ray + 1], v = ray_val[ray];
  for (int i = s + (threadIdx.x & 63); i < e; i += 64) out[i] = v;
}

// Row fragment of the colour-MLP input for sample s: slots 0..3 = shading features 4g..4g+3
// ([1 | feat[1:16]] + app_emb, Renderer.cpp:181-187), slots 4..7 = SH coefficients 4g..4g+3.
__device__ __forceinline__ half8_t f2n_shade_input_frag(const float* __restrict__ feat, const float* __restrict__ dirs,
                                                        const float* __restrict__ app_emb, const int32_t* __restrict__ sample_emb_idx,
                                                        int s, int g, bool valid) {
  half8_t xf = {0, 0, 0, 0, 0, 0, 0, 0};
  if (!valid) return xf;
  float4_t f = *(const float4_t*) (feat + (size_t) s * F2N_D_OUT + 4 * g);
  if (g == 0) f[0] = 1.f;
  if (app_emb != nullptr) {
    const float4_t e = *(const float4_t*) (app_emb + (size_t) sample_emb_idx[s] * 16 + 4 * g);
#pragma unroll
    for (int r = 0; r < 4; r++) f[r] = f[r] + e[r];
  }
  float sh[16];
  f2n_sh16(dirs[3 * (size_t) s], dirs[3 * (size_t) s + 1], dirs[3 * (size_t) s + 2], sh);
#pragma unroll
  for (int r = 0; r < 4; r++) {
    xf[r] = (half_t) f[r];
    const float v = (g == 0) ? sh[r] : (g == 1) ? sh[4 + r] : (g == 2) ? sh[8 + r] : sh[12 + r];
    xf[4 + r] = (half_t) v;
  }
  return xf;
}

__device__ __forceinline__ half8_t f2n_load_xfrag(const half_t* __restrict__ x, int s, int g, bool valid) {
  if (!valid) return half8_t{0, 0, 0, 0, 0, 0, 0, 0};
  return f2n_rowfrag(x, F2N_D_IN, s, 0, g);
}

#define F2N_SHADE_EPS 1e-3f

// The colour network's three outputs of sample c sit in lane (c, g = 0) as o[0..2].  The scaled sigmoid behind them
// (SHShader.cpp:28: expf + an IEEE division, ~25 instructions) is evaluated ONCE per wave instead of three times: lane
// (c, g) takes channel g from lane c and stores its own result (g < 3).  Same arithmetic per value, so the same bits.
__device__ __forceinline__ float f2n_channel_of_group(const float4_t& o, int c, int g) {
  const float v1 = __shfl(o[1], c), v2 = __shfl(o[2], c);
  return g == 0 ? o[0] : g == 1 ? v1 : v2;
}

__global__ __launch_bounds__(256) void shade_fwd_kernel(int n, const float* __restrict__ feat, const float* __restrict__ dirs,
                                                        const float* __restrict__ app_emb,
                                                        const int32_t* __restrict__ sample_emb_idx,
                                                        const half_t* __restrict__ params, float* __restrict__ rgb,
                                                        half_t* __restrict__ save_x, const int32_t* __restrict__ n_dev) {
  F2N_RAISE_PRIO();
  const int tid = threadIdx.x, lane = tid & 63, c = lane & 15, g = lane >> 4;
  if (n_dev != nullptr) n = min(n, *n_dev);  // the sample count is still on the device (see f2n_shade_fwd_dyn)
  F2nMlpFwdW<2> w;
  w.load(params, c, g);
  const int n_blocks = (n + 15) / 16;
  const int wave_global = blockIdx.x * 4 + (tid >> 6), wave_stride = gridDim.x * 4;
  for (int blk = wave_global; blk < n_blocks; blk += wave_stride) {
    const int s = blk * 16 + c;
    const bool valid = s < n;
    const half8_t xf = f2n_shade_input_frag(feat, dirs, app_emb, sample_emb_idx, s, g, valid);
    if (save_x != nullptr && valid) {
      half_t* p = save_x + (size_t) s * F2N_D_IN + 4 * g;
      *(half4_t*) p = __builtin_shufflevector(xf, xf, 0, 1, 2, 3);
      *(half4_t*) (p + 16) = __builtin_shufflevector(xf, xf, 4, 5, 6, 7);
    }
    const float4_t o = w.forward(xf);
    const float ov = (float) (half_t) f2n_channel_of_group(o, c, g);  // f16 output precision, then fp32 torch ops (SHShader.cpp:28)
    const float col = (1.f + 2.f * F2N_SHADE_EPS) / (1.f + expf(-ov)) - F2N_SHADE_EPS;
    if (valid && g < 3) rgb[3 * (size_t) s + g] = col;
  }
}

// Field MLP + colour path of the grad pass in ONE kernel (Renderer.cpp:152-189 for the surviving samples): the density
// network's D tiles -- lane (c = sample, g) holds outputs 4g..4g+3 -- ARE the K-slots 4g..4g+3 of the colour network's input
// fragment, so `feat` [M,16] never exists in memory: x_cache rows (h16 hash features of the pre-pass, through src_rows) ->
// field MLP -> f16 rounding -> [1 | feat[1:]] + app_emb -> | SH4(dir) -> colour MLP -> sigmoid.  Written: the compact density
// pre-activation f0 [M] (compositing), the two networks' h16 inputs (their backward passes recompute everything else) and
// rgb.  Bit-identical to f2n_field_fwd_cached followed by f2n_shade_fwd (same fragments, same MFMA chains, same roundings).
// Inputs of the next tile are in flight while the current one goes through its 20 MFMAs.
__global__ __launch_bounds__(256) void field_shade_fwd_kernel(int n, const int32_t* __restrict__ n_dev, const int32_t* __restrict__ src_rows,
                                                              const half_t* __restrict__ x_cache, const half_t* __restrict__ field_params,
                                                              const float* __restrict__ dirs, const float* __restrict__ app_emb,
                                                              const int32_t* __restrict__ sample_emb_idx,
                                                              const half_t* __restrict__ color_params, float* __restrict__ out_f0,
                                                              half_t* __restrict__ save_field_x, half_t* __restrict__ save_shade_x,
                                                              float* __restrict__ rgb, int n_extra,
                                                              const half_t* __restrict__ x_extra, float* __restrict__ feat_extra,
                                                              half_t* __restrict__ save_x_extra) {
  F2N_RAISE_PRIO();
  const int tid = threadIdx.x, lane = tid & 63, c = lane & 15, g = lane >> 4;
  if (n_dev != nullptr) n = min(n, *n_dev);
  // both networks' weight fragments live in LDS (F2nMlpFwdWLds: 152 -> ~80 registers, twice the resident waves)
  __shared__ half8_t s_wf[F2nMlpFwdWLds<1>::N_FRAG * 64];
  __shared__ half8_t s_wc[F2nMlpFwdWLds<2>::N_FRAG * 64];
  if ((tid >> 6) == 0) F2nMlpFwdWLds<1>::fill(s_wf, field_params, lane);
  if ((tid >> 6) == 1) F2nMlpFwdWLds<2>::fill(s_wc, color_params, lane);
  __syncthreads();
  int w_off = lane;
  asm volatile("" : "+v"(w_off));  // opaque: the fragment reads stay at their point of use
  const half8_t* wf = s_wf + w_off;
  const half8_t* wc = s_wc + w_off;
  const int wave_global = blockIdx.x * 4 + (tid >> 6), wave_stride = gridDim.x * 4;
  // "extra" rows (the 2E edge samples of the TV loss, Renderer.cpp:159-166): field MLP only, on their own cached rows; their
  // 16 outputs are wanted as fp32 rows (the loss reads them).  Exactly f2n_field_fwd_cached's arithmetic; riding here saves
  // that launch.  The LAST waves of the grid take them, the first ones start on the survivors' tiles at once.
  {
    const int n_etiles = (n_extra + 15) / 16;
    for (int tile = wave_stride - 1 - wave_global; tile < n_etiles; tile += wave_stride) {
      const int s = tile * 16 + c;
      const bool valid = s < n_extra;
      const half8_t xf = valid ? f2n_rowfrag(x_extra, F2N_D_IN, s, 0, g) : half8_t{0, 0, 0, 0, 0, 0, 0, 0};
      if (save_x_extra != nullptr && valid) {
        half_t* p = save_x_extra + (size_t) s * F2N_D_IN + 4 * g;
        *(half4_t*) p = __builtin_shufflevector(xf, xf, 0, 1, 2, 3);
        *(half4_t*) (p + 16) = __builtin_shufflevector(xf, xf, 4, 5, 6, 7);
      }
      const float4_t o = F2nMlpFwdWLds<1>::forward(wf, xf);
      if (valid) {
        float4_t of;
#pragma unroll
        for (int r = 0; r < 4; r++) of[r] = (float) (half_t) o[r];  // output precision is f16 (TCNNWP.cpp:143-144)
        *(float4_t*) (feat_extra + (size_t) s * F2N_D_OUT + 4 * g) = of;
      }
    }
  }
  const int n_tiles = (n + 15) / 16;
  struct In {
    half8_t xf;
    float4_t e;
    float d[3];
  };
  struct Idx {
    int row, img;
  };
  const bool emb = app_emb != nullptr;
  auto fetch_idx = [&](int tile) {
    const int s = min(tile * 16 + c, n - 1);
    Idx r;
    r.row = src_rows != nullptr ? src_rows[s] : s;
    r.img = emb ? sample_emb_idx[s] : 0;
    return r;
  };
  auto fetch = [&](int tile, const Idx& ix) {
    const int s = min(tile * 16 + c, n - 1);
    In r;
    r.xf = f2n_rowfrag(x_cache, F2N_D_IN, ix.row, 0, g);
    r.e = emb ? *(const float4_t*) (app_emb + (size_t) ix.img * 16 + 4 * g) : float4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 3; k++) r.d[k] = dirs[3 * (size_t) s + k];
    return r;
  };
  if (wave_global >= n_tiles) return;
  Idx ix_next = fetch_idx(wave_global);
  In cur = fetch(wave_global, ix_next);
  ix_next = fetch_idx(min(wave_global + wave_stride, n_tiles - 1));
  for (int tile = wave_global; tile < n_tiles; tile += wave_stride) {
    const int t1 = min(tile + wave_stride, n_tiles - 1), t2 = min(tile + 2 * wave_stride, n_tiles - 1);
    const In nxt = fetch(t1, ix_next);  // (past the end: a harmless re-read of the last tile)
    ix_next = fetch_idx(t2);
    __builtin_amdgcn_sched_barrier(0);
    const int s = tile * 16 + c;
    const bool valid = s < n;
    const half8_t xf = valid ? cur.xf : half8_t{0, 0, 0, 0, 0, 0, 0, 0};
    if (save_field_x != nullptr && valid) {
      half_t* p = save_field_x + (size_t) s * F2N_D_IN + 4 * g;
      *(half4_t*) p = __builtin_shufflevector(xf, xf, 0, 1, 2, 3);
      *(half4_t*) (p + 16) = __builtin_shufflevector(xf, xf, 4, 5, 6, 7);
    }
    const float4_t o = F2nMlpFwdWLds<1>::forward(wf, xf);  // lane (c = sample, g): field outputs 4g..4g+3
    float4_t f;
#pragma unroll
    for (int r = 0; r < 4; r++) f[r] = (float) (half_t) o[r];  // f16 output precision (TCNNWP.cpp:143-144), widened (:112)
    if (valid && g == 0 && out_f0 != nullptr) out_f0[s] = f[0];
    if (g == 0) f[0] = 1.f;  // Renderer.cpp:181-182
    if (emb) {
#pragma unroll
      for (int r = 0; r < 4; r++) f[r] = f[r] + cur.e[r];  // Scatter.cu:10-18
    }
    float sh[16];
    f2n_sh16(cur.d[0], cur.d[1], cur.d[2], sh);
    half8_t xs;
#pragma unroll
    for (int r = 0; r < 4; r++) {
      xs[r] = valid ? (half_t) f[r] : (half_t) 0.f;
      const float v = (g == 0) ? sh[r] : (g == 1) ? sh[4 + r] : (g == 2) ? sh[8 + r] : sh[12 + r];
      xs[4 + r] = valid ? (half_t) v : (half_t) 0.f;
    }
    if (save_shade_x != nullptr && valid) {
      half_t* p = save_shade_x + (size_t) s * F2N_D_IN + 4 * g;
      *(half4_t*) p = __builtin_shufflevector(xs, xs, 0, 1, 2, 3);
      *(half4_t*) (p + 16) = __builtin_shufflevector(xs, xs, 4, 5, 6, 7);
    }
    const float4_t oc = F2nMlpFwdWLds<2>::forward(wc, xs);
    const float ov = (float) (half_t) f2n_channel_of_group(oc, c, g);
    const float col = (1.f + 2.f * F2N_SHADE_EPS) / (1.f + expf(-ov)) - F2N_SHADE_EPS;
    if (valid && g < 3) rgb[3 * (size_t) s + g] = col;
    cur = nxt;
  }
}

// Appearance-embedding gradient addends on a 2^-38 grid (order-free integer sums; see shade_bwd_kernel).  |v| is clamped to 2^14, so
// an addend is at most 2^52 and a cell takes 1024 of them without wrapping -- a block adds one addend per 16-sample tile and image,
// at most ~200 per cell at the largest batches (round-4 advisor: at 2^-48 two clamped addends wrapped).  A larger or non-finite addend
// only occurs when the colour network's backward has produced non-finite values, i.e. next to non-finite MLP gradients: that step is
// dropped by the finiteness flags over the two MLPs' gradients (the same rows feed the field MLP's backward), so the clamp / the zero
// a NaN maps to is never applied.  Scenes with more than 240 images (the per-block LDS image is 128 B per image) take the global
// path below, whose float atomics add in arrival order: "same seed, same bits" holds up to 240 images (the shipped scenes: 50-185).
#define F2N_EMB_FIXED_SCALE 274877906944.0          // 2^38
#define F2N_EMB_FIXED_INV (1.0 / 274877906944.0)
__device__ __forceinline__ unsigned long long f2n_emb_fixed(float v) {
  const float cl = v == v ? fminf(fmaxf(v, -16384.f), 16384.f) : 0.f;
  return (unsigned long long) __double2ll_rn((double) cl * F2N_EMB_FIXED_SCALE);
}

union F2nShadeSmem {
  F2nMlpLds<2> w;
  float acc[2 * (F2N_D_HID * F2N_D_IN + F2N_D_HID * F2N_D_HID + F2N_D_OUT * F2N_D_HID)];  // two images, see f2n_mlp_flush_dw
};

__global__ __launch_bounds__(256, 2) void shade_bwd_kernel(int n, const float* __restrict__ drgb,
                                                        const int32_t* __restrict__ sample_emb_idx,
                                                        const half_t* __restrict__ params, const half_t* __restrict__ x_h,
                                                        float loss_scale, float* __restrict__ dfeat,
                                                        float* __restrict__ dparams, int n_emb,
                                                        float* __restrict__ emb_partials, const float* __restrict__ df0,
                                                        const int32_t* __restrict__ n_dev, float* __restrict__ emb_global) {
  F2N_RAISE_PRIO();
  // emb_partials: per-block LDS image of the appearance-embedding gradient, flushed as a partial (n_emb <= 240).  The image
  // is 64-bit FIXED POINT (2^-38 units, ds_add_u64): every addend is rounded to the grid on its own and integer sums do not
  // depend on the order in which the block's four waves arrive -- with ds_add_f32 two trainings from one seed parted at
  // iteration 2, in this gradient (tools/determinism_probe.py, round 4);
  // emb_global: more images than fit into LDS -- row sums go straight to the gradient with global atomics, as the
  // reference's ScatterAddFuncBackward does (Scatter.cu:20-40)
  if (n_dev != nullptr) n = min(n, *n_dev);
  __shared__ F2nShadeSmem sm;
  extern __shared__ unsigned long long s_emb[];  // [n_emb * 16] per-block appearance-embedding gradient, fixed point
  const int tid = threadIdx.x, lane = tid & 63, c = lane & 15, g = lane >> 4;
  f2n_mlp_lds_fill<2>(sm.w, params, tid, 256);
  const bool do_emb = emb_partials != nullptr || emb_global != nullptr;
  const bool emb_lds = emb_partials != nullptr;
  if (emb_lds)
    for (int i = tid; i < n_emb * 16; i += 256) s_emb[i] = 0ull;
  __syncthreads();
  const half8_t idf[2] = {f2n_identity_frag(0, c, g), f2n_identity_frag(1, c, g)};
  // forward output layer fragments (needed to recompute o for the sigmoid derivative)
  const half_t* po = params + F2N_D_HID * F2N_D_IN + F2N_D_HID * F2N_D_HID;
  const half8_t wo[2] = {f2n_rowfrag(po, F2N_D_HID, c, 0, g), f2n_rowfrag(po, F2N_D_HID, c, 32, g)};
  F2nMlpGradAcc<2> acc;
  acc.zero();
  const int wave_global = blockIdx.x * 4 + (tid >> 6), wave_stride = gridDim.x * 4;
  const float inv_scale = 1.f / loss_scale;
  const float4_t z = {0.f, 0.f, 0.f, 0.f};
  // Per-sample inputs are fetched one tile AHEAD into registers: with two waves per SIMD a load issued where its value
  // is needed exposes most of its latency -- three dependent round trips per tile (x, d rgb, image index) were ~80 % of
  // the first version's time.  Register budget (254 per lane, two resident blocks per CU): 112 weight-gradient
  // accumulators, one tile's fragments, nothing hoisted -- a lone wave issues one instruction per ~4 cycles whatever the
  // SIMD could take, so the second resident wave is worth more than anything a 512-register body can keep in registers
  // (measured: 0.121 -> 0.097 ms for 8e5 samples, tools/mlp_bwd_bench.py).
  struct In {
    half8_t xf;
    float d[3];
    int img;
    float df0;  // the density path's gradient of dfeat[:,0] (compact array from f2n_composite_bwd), merged into the row store
  };
  auto fetch = [&](int tile, In& o) {
    const int s = tile * 16 + c;
    const int sc = s < n ? s : n - 1;  // always in range: no branch around the loads
    o.xf = f2n_rowfrag(x_h, F2N_D_IN, sc, 0, g);
#pragma unroll
    for (int r = 0; r < 3; r++) o.d[r] = drgb[3 * (size_t) sc + r];
    o.img = do_emb ? sample_emb_idx[sc] : -1;
    o.df0 = df0 != nullptr ? df0[sc] : 0.f;
  };
  // One 16-sample tile per round (the weight gradients contract a tile's 16 samples with K = 16 MFMAs, so nothing is
  // carried from one tile to the next but the accumulators), the next tile's inputs in flight.
  const int n_tiles = (n + 15) / 16;
  In cur;
  if (wave_global < n_tiles) fetch(wave_global, cur);
  for (int tile = wave_global; tile < n_tiles; tile += wave_stride) {
    In nxt;
    fetch(tile + wave_stride < n_tiles ? tile + wave_stride : tile, nxt);  // last round: a harmless re-read
    __builtin_amdgcn_sched_barrier(0);  // keep the loads up here; their s_waitcnt lands at the bottom of the round
    {
      F2nHalfBwd<2> hb;
      // the weight fragments are re-read from LDS where they are used: an address the compiler cannot prove loop-invariant
      // keeps it from hoisting ~100 registers of fragments out of the loop (and then spilling them)
      int lds_off = 0;
      asm volatile("" : "+v"(lds_off));
      const F2nMlpLds<2>& wl = *(const F2nMlpLds<2>*) ((const char*) &sm.w + lds_off);
      const int s = tile * 16 + c;
      const bool valid = s < n;
      const half8_t xf = valid ? cur.xf : half8_t{0, 0, 0, 0, 0, 0, 0, 0};
      const float d0 = cur.d[0], d1 = cur.d[1], d2 = cur.d[2];
      // d(rgb)/d(o) needs the network output o: computed inside the backward's own forward recomputation from the
      // last hidden layer's activations (no second forward chain)
      auto dy_fn = [&](half8_t h0, half8_t h1) {
        float4_t o = f2n_mfma(wo[0], h0, z);
        o = f2n_mfma(wo[1], h1, o);
        // (evaluating the three channels on three lane groups with shuffles, as the forward kernels do, was measured
        // SLOWER here -- shade_bwd 0.102 -> 0.122 ms: four LDS round trips on the dependent chain between two MFMA stages,
        // with two waves per SIMD to hide them)
        half8_t dyf = {0, 0, 0, 0, 0, 0, 0, 0};
        const float dv[3] = {d0, d1, d2};
#pragma unroll
        for (int r = 0; r < 3; r++) {
          const float ov = (float) (half_t) o[r];
          const float e = expf(-ov);
          // e = +inf (output below ~-88.7): the quotient is inf / inf = NaN where the derivative's limit is 0.  The reference lets
          // the NaN through (ATen's backward of SHShader.cpp:27-28) and tcnn then drops the step; the product -- here, in the
          // taped path (SHShader::Query in host/Renderer.cpp: the clamp at -80) and in the oracle -- takes the limit.  The reference-numerics build keeps the NaN.
          const float dsig = (!F2N_REFERENCE_NUMERICS && e > 3.0e38f) ? 0.f : (1.f + 2.f * F2N_SHADE_EPS) * e / ((1.f + e) * (1.f + e));
          const half_t v = (half_t) ((float) (half_t) (dv[r] * dsig) * loss_scale);
          dyf[r] = (valid && g == 0) ? v : (half_t) 0.f;
        }
        return dyf;
      };
      f2n_mlp_half_bwd<2, 1>(wl, xf, dy_fn, idf, c, g, hb);
      // d(shading_feat) = dX[:, 0:16]: lane (c = sample, g) holds features 4g..4g+3
      float4_t dsf;
#pragma unroll
      for (int r = 0; r < 4; r++) dsf[r] = valid ? hb.dxT[0][r] * inv_scale : 0.f;
      if (valid) {
        float* p = dfeat + (size_t) s * F2N_D_OUT + 4 * g;
        if (g == 0 && df0 != nullptr) {  // column 0 belongs to the density path (constant-1 shading input): merged here
          float4_t row = dsf;  // (dsf[0] itself still feeds the appearance-embedding gradient below)
          row[0] = cur.df0;
          *(float4_t*) p = row;
        } else if (g == 0) {  // ... or left untouched for whoever writes it in place
          p[1] = dsf[1];
          p[2] = dsf[2];
          p[3] = dsf[3];
        } else {
          *(float4_t*) p = dsf;
        }
      }
      if (do_emb) {
        // ScatterAdd backward (Scatter.cu:20-40): per-image sum over samples, accumulated in LDS.  The 16 samples
        // of a tile usually share a ray, hence an image: reduce across the 16 sample lanes first.
        const int img = (valid && cur.img >= 0 && cur.img < n_emb) ? cur.img : -1;  // an index outside the table adds nothing
        // sample 0's image, without an LDS round trip: every row of 16 lanes holds the same 16 samples
        const int img0 = __builtin_amdgcn_readfirstlane(img);
        const bool uniform = __all(img == img0);
        auto add_emb = [&](int im, int col, float v) {
          if (emb_lds) atomicAdd(&s_emb[im * 16 + col], f2n_emb_fixed(v));
          else atomicAdd(&emb_global[im * 16 + col], v);
        };
        if (uniform) {
          if (img0 >= 0) {
#pragma unroll
            for (int r = 0; r < 4; r++) {
              const float v = f2n_row16_allsum(dsf[r]);
              if (c == 0) add_emb(img0, 4 * g + r, v);
            }
          }
        } else if (img >= 0) {
#pragma unroll
          for (int r = 0; r < 4; r++) add_emb(img, 4 * g + r, dsf[r]);
        }
      }
      f2n_mlp_accumulate_dw_half<2>(hb, acc);
    }
    cur = nxt;
  }
  __syncthreads();
  f2n_mlp_flush_dw<2>(acc, sm.acc, dparams, c, g, tid, 256);
  if (emb_lds) {  // s_emb is complete since the __syncthreads() above
    float* dst = emb_partials + (size_t) blockIdx.x * n_emb * 16;
    for (int i = tid; i < n_emb * 16; i += 256) dst[i] = (float) ((double) (long long) s_emb[i] * F2N_EMB_FIXED_INV);
  }
}

static inline unsigned f2n_shade_grid(int n_units, int per_block) {
  long blocks = ((long) n_units + per_block - 1) / per_block;
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  return (unsigned) blocks;
}

extern "C" {

int f2n_sh_encode(void* stream, int n, int degree, const float* dirs, float* out) {
  if (n < 0) return F2N_ERR_INVALID_ARG;
  if (degree < 1 || degree > 8) return F2N_ERR_UNSUPPORTED;
  if (n == 0) return F2N_OK;
  hipLaunchKernelGGL(sh_encode_kernel, dim3(f2n_div_up(n, 256)), dim3(256), 0, (hipStream_t) stream, n, degree, dirs, out);
  return f2n_launch_status();
}

int f2n_scatter_idx(void* stream, int n_rays, const int32_t* start_end, const int32_t* ray_val, int32_t* out) {
  if (n_rays < 0) return F2N_ERR_INVALID_ARG;
  if (n_rays == 0) return F2N_OK;
  hipLaunchKernelGGL(scatter_idx_kernel, dim3(f2n_div_up(n_rays, 4)), dim3(256), 0, (hipStream_t) stream, n_rays, start_end,
                     ray_val, out);
  return f2n_launch_status();
}

int f2n_shade_fwd(void* stream, int n, const float* feat, const float* dirs, const float* app_emb,
                  const int32_t* sample_emb_idx, const void* mlp_params_h, float* rgb, void* save_x_h) {
  return f2n_shade_fwd_dyn(stream, n, nullptr, feat, dirs, app_emb, sample_emb_idx, mlp_params_h, rgb, save_x_h);
}

int f2n_shade_fwd_dyn(void* stream, int n_max, const int32_t* n_dev, const float* feat, const float* dirs, const float* app_emb,
                      const int32_t* sample_emb_idx, const void* mlp_params_h, float* rgb, void* save_x_h) {
  const int n = n_max;
  if (n < 0 || (app_emb != nullptr && sample_emb_idx == nullptr)) return F2N_ERR_INVALID_ARG;
  if (n == 0) return F2N_OK;
  hipLaunchKernelGGL(shade_fwd_kernel, dim3(f2n_shade_grid((n + 15) / 16, 4)), dim3(256), 0, (hipStream_t) stream, n, feat,
                     dirs, app_emb, sample_emb_idx, (const half_t*) mlp_params_h, rgb, (half_t*) save_x_h, n_dev);
  return f2n_launch_status();
}

int f2n_field_shade_fwd_dyn(void* stream, int n_max, const int32_t* n_dev, const int32_t* src_rows, const void* x_cache_h,
                            const void* field_params_h, const float* dirs, const float* app_emb, const int32_t* sample_emb_idx,
                            const void* color_params_h, float* out_f0, void* save_field_x_h, void* save_shade_x_h, float* rgb) {
  return f2n_field_shade_fwd_extra(stream, n_max, n_dev, src_rows, x_cache_h, field_params_h, dirs, app_emb, sample_emb_idx,
                                   color_params_h, out_f0, save_field_x_h, save_shade_x_h, rgb, 0, nullptr, nullptr, nullptr);
}

int f2n_field_shade_fwd_extra(void* stream, int n_max, const int32_t* n_dev, const int32_t* src_rows, const void* x_cache_h,
                              const void* field_params_h, const float* dirs, const float* app_emb, const int32_t* sample_emb_idx,
                              const void* color_params_h, float* out_f0, void* save_field_x_h, void* save_shade_x_h, float* rgb,
                              int n_extra, const void* x_extra_h, float* feat_extra, void* save_x_extra_h) {
  const int n = n_max;
  if (n < 0 || n_extra < 0 || (n > 0 && (x_cache_h == nullptr || rgb == nullptr)) || (app_emb != nullptr && sample_emb_idx == nullptr) ||
      (n_extra > 0 && (x_extra_h == nullptr || feat_extra == nullptr)))
    return F2N_ERR_INVALID_ARG;
  if (n == 0 && n_extra == 0) return F2N_OK;
  const int tiles = (n + 15) / 16 + (n_extra + 15) / 16;
  hipLaunchKernelGGL(field_shade_fwd_kernel, dim3(f2n_shade_grid(tiles, 4)), dim3(256), 0, (hipStream_t) stream, n, n_dev,
                     src_rows, (const half_t*) x_cache_h, (const half_t*) field_params_h, dirs, app_emb, sample_emb_idx,
                     (const half_t*) color_params_h, out_f0, (half_t*) save_field_x_h, (half_t*) save_shade_x_h, rgb, n_extra,
                     (const half_t*) x_extra_h, feat_extra, (half_t*) save_x_extra_h);
  return f2n_launch_status();
}

int f2n_shade_bwd(void* stream, int n, const float* drgb, const int32_t* sample_emb_idx, const void* mlp_params_h,
                  const void* saved_x_h, float loss_scale, float* dfeat, float* dparams_f32_scaled, float* dapp_emb,
                  int n_emb, const float* df0) {
  return f2n_shade_bwd_dyn(stream, n, nullptr, drgb, sample_emb_idx, mlp_params_h, saved_x_h, loss_scale, dfeat, dparams_f32_scaled,
                           dapp_emb, n_emb, df0, 0);
}

int f2n_shade_bwd_dyn(void* stream, int n_max, const int32_t* n_dev, const float* drgb, const int32_t* sample_emb_idx,
                      const void* mlp_params_h, const void* saved_x_h, float loss_scale, float* dfeat, float* dparams_f32_scaled,
                      float* dapp_emb, int n_emb, const float* df0, int defer_reduce) {
  const int n = n_max;
  if (n < 0 || !(loss_scale > 0.f) || (dapp_emb != nullptr && (sample_emb_idx == nullptr || n_emb < 1)) || ((uintptr_t) mlp_params_h & 15))
    return F2N_ERR_INVALID_ARG;
  const bool emb_in_lds = dapp_emb != nullptr && n_emb <= 240;  // per-block LDS accumulator: 128 B (16 x int64) per image next to 57 KB of weights / reduction images
  if (n == 0) return F2N_OK;
  unsigned blocks = f2n_shade_grid((n + 31) / 32, 4);
  if (blocks > 512) blocks = 512;  // two resident blocks per CU (254 registers per lane)
  // (fewer blocks for smaller batches -- 256 / 384 at the 2.6e5 samples of a converged step, so that a block would fit next to the
  // sampler's march waves -- measured: nothing, 0.700-0.703 against 0.701-0.714 ms per step; profiles/r06_backward_grid_ab.txt)
  const int n_params = F2N_D_HID * F2N_D_IN + F2N_D_HID * F2N_D_HID + F2N_D_OUT * F2N_D_HID;
  float* partials = (float*) f2n_ws_get(F2N_WS_SHADE_DW, sizeof(float) * (size_t) blocks * n_params);
  float* emb_partials = nullptr;
  size_t dyn_lds = 0;
  if (emb_in_lds) {
    emb_partials = (float*) f2n_ws_get(F2N_WS_SHADE_EMB, sizeof(float) * (size_t) blocks * n_emb * 16);
    dyn_lds = sizeof(unsigned long long) * (size_t) n_emb * 16;
    if (emb_partials == nullptr) return F2N_ERR_INVALID_ARG;
  }
  if (partials == nullptr) return F2N_ERR_INVALID_ARG;
  if (dyn_lds > 4096) {  // 57 KB of static LDS + the per-image accumulator can pass 64 KB (a gfx950 workgroup may own all 160 KB)
    static bool raised = false;
    if (!raised) {
      if (hipFuncSetAttribute((const void*) shade_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 240 * 16 * (int) sizeof(unsigned long long)) != hipSuccess)
        return F2N_ERR_UNSUPPORTED;
      raised = true;
    }
  }
  hipLaunchKernelGGL(shade_bwd_kernel, dim3(blocks), dim3(256), dyn_lds, (hipStream_t) stream, n, drgb, sample_emb_idx,
                     (const half_t*) mlp_params_h, (const half_t*) saved_x_h, loss_scale, dfeat, partials, n_emb, emb_partials, df0,
                     n_dev, (dapp_emb != nullptr && !emb_in_lds) ? dapp_emb : nullptr);
  int rc = f2n_launch_status();
  if (rc != F2N_OK) return rc;
  if (defer_reduce) {  // folded into their destinations by f2n_reduce_deferred, together with the field network's
    rc = f2n_defer_reduction(n_params, (int) blocks, partials, dparams_f32_scaled);
    if (rc != F2N_OK || !emb_in_lds) return rc;
    return f2n_defer_reduction(n_emb * 16, (int) blocks, emb_partials, dapp_emb);
  }
  rc = f2n_reduce_partials(stream, n_params, (int) blocks, partials, dparams_f32_scaled);
  if (rc != F2N_OK || !emb_in_lds) return rc;
  return f2n_reduce_partials(stream, n_emb * 16, (int) blocks, emb_partials, dapp_emb);
}

}  // extern "C"
