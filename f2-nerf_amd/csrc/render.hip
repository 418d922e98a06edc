// Renderer glue on gfx950: early-stop pre-pass, sample compaction, volume-rendering compositing forward /
// backward, WeightVar loss and the seam-level FlexOps.  Semantics: Renderer/Renderer.cpp:105-208,
// Renderer/Renderer.cu:8-50, Utils/CustomOps/{FlexOps.cu:5-93, CustomOps.cu:12-80, CustomOps.cpp:9-18}.
// The per-ray fp32 accumulations are sequential left-to-right walks, one ray per lane, because that order is
// part of the parity contract; everything that the reference spreads over a dozen ATen
// element-wise launches is done inside the same walk.
#include "f2n_dev.h"

#define F2N_DENSITY_SHIFT 3.f  // Renderer.cpp:101-104
#define F2N_T_EPS 1e-4f        // early-stop threshold, Renderer.cpp:125
#define F2N_T_BIAS 1e-2f       // sampled_t = t + 1e-2, Renderer.cpp:118,197

// Renderer.cpp:115-126
__global__ void early_stop_kernel(int n_rays, const int32_t* __restrict__ se, const float* __restrict__ f0, int f0_stride,
                                  const float* __restrict__ dt, float* __restrict__ weights, float* __restrict__ alphas,
                                  int32_t* __restrict__ mask, int32_t* __restrict__ kept) {
  const int ray = blockIdx.x * blockDim.x + threadIdx.x;
  if (ray >= n_rays) return;
  const int s = se[2 * ray], e = se[2 * ray + 1];
  float acc = 0.f;
  int cnt = 0;
  for (int i = s; i < e; i++) {
    const float sigma = expf(f0[(size_t) i * f0_stride] - F2N_DENSITY_SHIFT);
    const float sec = sigma * dt[i];
    const float alpha = 1.f - expf(-sec);
    const float trans = expf(-acc);  // exclusive cumulative density
    acc += sec;
    weights[i] = trans * alpha;
    alphas[i] = alpha;
    const int m = trans > F2N_T_EPS ? 1 : 0;
    mask[i] = m;
    cnt += m;
  }
  kept[ray] = cnt;
}

// Renderer.cpp:128-135: order-preserving compaction of masked samples; one wave per ray.  The mask of a ray is
// (numerically) a prefix, but nothing here relies on it: positions come from a wave-level ballot prefix count.
__global__ __launch_bounds__(256) void compact_kernel(int n_rays, const int32_t* __restrict__ old_se,
                                                      const int32_t* __restrict__ new_se, const int32_t* __restrict__ mask,
                                                      const float* __restrict__ pts, const float* __restrict__ dirs,
                                                      const float* __restrict__ dt, const float* __restrict__ t,
                                                      const int32_t* __restrict__ anchors, float* __restrict__ o_pts,
                                                      float* __restrict__ o_dirs, float* __restrict__ o_dt, float* __restrict__ o_t,
                                                      int32_t* __restrict__ o_anchors, int32_t* __restrict__ o_src) {
  const int lane = threadIdx.x & 63;
  const int ray = blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);
  if (ray >= n_rays) return;
  const int s = old_se[2 * ray], e = old_se[2 * ray + 1];
  int dst = new_se[2 * ray];
  for (int base = s; base < e; base += 64) {
    const int i = base + lane;
    const bool keep = i < e && mask[i] != 0;
    const unsigned long long bal = __ballot(keep);
    if (keep) {
      const int k = dst + __popcll(bal & ((1ull << lane) - 1ull));
#pragma unroll
      for (int c = 0; c < 3; c++) {
        o_pts[3 * (size_t) k + c] = pts[3 * (size_t) i + c];
        o_dirs[3 * (size_t) k + c] = dirs[3 * (size_t) i + c];
        o_anchors[3 * (size_t) k + c] = anchors[3 * (size_t) i + c];
      }
      o_dt[k] = dt[i];
      o_t[k] = t[i];
      if (o_src != nullptr) o_src[k] = i;
    }
    dst += __popcll(bal);
  }
}

// Renderer.cpp:190-208 forward.
__global__ void composite_fwd_kernel(int n_rays, const int32_t* __restrict__ se, const float* __restrict__ feat,
                                     const float* __restrict__ dt, const float* __restrict__ t, const float* __restrict__ rgb,
                                     const float* __restrict__ bg, float* __restrict__ colors, float* __restrict__ disparity,
                                     float* __restrict__ depth, float* __restrict__ weights) {
  const int ray = blockIdx.x * blockDim.x + threadIdx.x;
  if (ray >= n_rays) return;
  const int s = se[2 * ray], e = se[2 * ray + 1];
  float acc = 0.f, col[3] = {0.f, 0.f, 0.f}, disp = 0.f, dep = 0.f;
  for (int i = s; i < e; i++) {
    const float sigma = expf(feat[(size_t) i * 16] - F2N_DENSITY_SHIFT);
    const float sec = sigma * dt[i];
    const float alpha = 1.f - expf(-sec);
    const float trans = expf(-acc);
    acc += sec;
    const float w = trans * alpha;
    weights[i] = w;
    const float tt = t[i] + F2N_T_BIAS;
#pragma unroll
    for (int c = 0; c < 3; c++) col[c] += w * rgb[3 * (size_t) i + c];
    disp += w / tt;
    dep += w * tt;
  }
  const float last_trans = expf(-acc);
#pragma unroll
  for (int c = 0; c < 3; c++) colors[3 * ray + c] = col[c] + last_trans * bg[3 * ray + c];
  disparity[ray] = disp;
  depth[ray] = dep / (1.f - last_trans + 1e-4f);
}

// Backward of the compositing chain (FlexOps backward kernels FlexOps.cu:17-26,42-53,75-93; TruncExp backward
// CustomOps.cpp:15-18; GradientScaling backward CustomOps.cu:68-80) in two walks per ray: a forward walk to
// rebuild T_i / w_i and the ray totals, and a reverse walk carrying the suffix sum of d(acc).
__global__ void composite_bwd_kernel(int n_rays, const int32_t* __restrict__ se, const float* __restrict__ feat,
                                     const float* __restrict__ dt, const float* __restrict__ t, const float* __restrict__ rgb,
                                     const float* __restrict__ bg, const float* __restrict__ dcolors,
                                     const float* __restrict__ ddisparity, const float* __restrict__ ddepth,
                                     const float* __restrict__ dweights, float gs_progress, float* __restrict__ drgb,
                                     float* __restrict__ dfeat) {
  const int ray = blockIdx.x * blockDim.x + threadIdx.x;
  if (ray >= n_rays) return;
  const int s = se[2 * ray], e = se[2 * ray + 1];
  if (s >= e) return;
  float dC[3] = {0.f, 0.f, 0.f};
  if (dcolors != nullptr) { dC[0] = dcolors[3 * ray]; dC[1] = dcolors[3 * ray + 1]; dC[2] = dcolors[3 * ray + 2]; }
  const float dDisp = ddisparity != nullptr ? ddisparity[ray] : 0.f;
  const float dDep = ddepth != nullptr ? ddepth[ray] : 0.f;
  // walk 1: totals
  float acc = 0.f, dep_sum = 0.f;
  for (int i = s; i < e; i++) {
    const float sec = expf(feat[(size_t) i * 16] - F2N_DENSITY_SHIFT) * dt[i];
    const float w = expf(-acc) * (1.f - expf(-sec));
    acc += sec;
    dep_sum += w * (t[i] + F2N_T_BIAS);
  }
  const float total = acc;
  const float last_trans = expf(-total);
  const float denom = 1.f - last_trans + 1e-4f;
  // d(last_trans): colors = .. + last_trans*bg ; depth = dep_sum / (1 - last_trans + 1e-4)
  const float d_last = (dC[0] * bg[3 * ray] + dC[1] * bg[3 * ray + 1] + dC[2] * bg[3 * ray + 2]) +
                       dDep * dep_sum / (denom * denom);
  const float d_total = -last_trans * d_last;  // last_trans = exp(-sum sec): every sec_i receives this
  const float dDepW = dDep / denom;
  // walk 2: reverse, suffix carries sum_{j>i} d(acc_j)
  float suffix = 0.f;
  for (int i = e - 1; i >= s; i--) {
    const float x = feat[(size_t) i * 16] - F2N_DENSITY_SHIFT;
    const float sigma = expf(x);
    const float sec = sigma * dt[i];
    acc -= sec;  // exclusive cumulative density of sample i (rebuilt backwards)
    const float trans = expf(-acc);
    const float ems = expf(-sec);
    const float alpha = 1.f - ems;
    const float w = trans * alpha;
    const float tt = t[i] + F2N_T_BIAS;
    const float c0 = rgb[3 * (size_t) i], c1 = rgb[3 * (size_t) i + 1], c2 = rgb[3 * (size_t) i + 2];
    float dw = (dC[0] * c0 + dC[1] * c1 + dC[2] * c2) + dDisp / tt + dDepW * tt;
    if (dweights != nullptr) dw += dweights[i];
    // w = trans*alpha ; trans = exp(-acc_excl) ; alpha = 1 - exp(-sec)
    const float d_acc = -dw * w;
    const float d_sec = dw * trans * ems + suffix + d_total;
    suffix += d_acc;
    float d_sigma = d_sec * dt[i];
    float g0 = dC[0] * w, g1 = dC[1] * w, g2 = dC[2] * w;
    if (gs_progress < 1.f) {  // CustomOps.cu:68-80
      const float a = ((float) (i - s) + .5f) / (float) (e - s);
      const float sc = gs_progress + (1.f - gs_progress) * a * a;
      d_sigma *= sc;
      g0 *= sc; g1 *= sc; g2 *= sc;
    }
    drgb[3 * (size_t) i] = g0;
    drgb[3 * (size_t) i + 1] = g1;
    drgb[3 * (size_t) i + 2] = g2;
    // TruncExp backward: grad * exp(clamp(x, -100, 5))
    dfeat[(size_t) i * 16] = d_sigma * expf(fminf(fmaxf(x, -100.f), 5.f));
  }
}

// CustomOps.cu:12-66
__device__ __forceinline__ void f2n_wv_stats(const float* __restrict__ w, int n, float& mean, float& wsum) {
  float m = 0.f, ws = 1e-6f;
  for (int i = 0; i < n; i++) {
    m += w[i] * ((float) i / 16.f);
    ws += w[i];
  }
  mean = m / ws;
  wsum = ws;
}

__global__ void weight_var_fwd_kernel(int n_rays, const float* __restrict__ weights, const int32_t* __restrict__ se,
                                      float* __restrict__ out) {
  const int ray = blockIdx.x * blockDim.x + threadIdx.x;
  if (ray >= n_rays) return;
  const int s = se[2 * ray], e = se[2 * ray + 1];
  if (s >= e) { out[ray] = 0.f; return; }
  float mean, ws;
  f2n_wv_stats(weights + s, e - s, mean, ws);
  float var = 0.f;
  for (int i = 0; i + s < e; i++) {
    const float b = (float) i / 16.f - mean;
    var += weights[i + s] * b * b;
  }
  out[ray] = var;
}

__global__ void weight_var_bwd_kernel(int n_rays, const float* __restrict__ weights, const int32_t* __restrict__ se,
                                      const float* __restrict__ dvars, float* __restrict__ dw) {
  const int ray = blockIdx.x * blockDim.x + threadIdx.x;
  if (ray >= n_rays) return;
  const int s = se[2 * ray], e = se[2 * ray + 1];
  if (s >= e) return;
  float mean, ws;
  f2n_wv_stats(weights + s, e - s, mean, ws);
  float tmp = 0.f;
  for (int i = 0; i + s < e; i++) {
    const float b = (float) i / 16.f - mean;
    tmp += weights[i + s] * 2.f * b;
  }
  const float dv = dvars[ray];
  for (int i = 0; i + s < e; i++) {
    const float b = (float) i / 16.f - mean;
    const float g = (b * b + tmp * -((float) i / 16.f) / ws);
    dw[i + s] = dv * g;
  }
}

// FlexOps.cu:5-93
__global__ void flex_sum_fwd_kernel(int n, int vec, const float* __restrict__ val, const int32_t* __restrict__ se,
                                    float* __restrict__ sum) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  for (int j = 0; j < vec; j++) {
    float s = 0.f;
    for (int i = se[2 * r]; i < se[2 * r + 1]; i++) s += val[(size_t) i * vec + j];
    sum[(size_t) r * vec + j] = s;
  }
}
__global__ void flex_sum_bwd_kernel(int n, int vec, const float* __restrict__ dsum, const int32_t* __restrict__ se,
                                    float* __restrict__ dval) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  for (int j = 0; j < vec; j++) {
    const float f = dsum[(size_t) r * vec + j];
    for (int i = se[2 * r]; i < se[2 * r + 1]; i++) dval[(size_t) i * vec + j] = f;
  }
}
__global__ void flex_acc_fwd_kernel(int n, int include_this, const float* __restrict__ val, const int32_t* __restrict__ se,
                                    float* __restrict__ sum) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  float s = 0.f;
  for (int i = se[2 * r]; i < se[2 * r + 1]; i++) {
    if (include_this) { s += val[i]; sum[i] = s; }
    else { sum[i] = s; s += val[i]; }
  }
}
__global__ void flex_acc_bwd_kernel(int n, int include_this, const float* __restrict__ dsum, const int32_t* __restrict__ se,
                                    float* __restrict__ dval) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  float wp = 0.f;
  for (int i = se[2 * r + 1] - 1; i >= se[2 * r]; i--) {
    if (include_this) { wp += dsum[i]; dval[i] = wp; }
    else { dval[i] = wp; wp += dsum[i]; }
  }
}

#define F2N_RAY_LAUNCH(kernel, n_rays, ...)                                                                          \
  do {                                                                                                               \
    if ((n_rays) < 0) return F2N_ERR_INVALID_ARG;                                                                    \
    if ((n_rays) == 0) return F2N_OK;                                                                                \
    hipLaunchKernelGGL(kernel, dim3(f2n_div_up((n_rays), 64)), dim3(64), 0, (hipStream_t) stream, (n_rays), __VA_ARGS__); \
    return f2n_launch_status();                                                                                      \
  } while (0)

extern "C" {

int f2n_early_stop(void* stream, int n_rays, const int32_t* pts_start_end, const float* f0, int f0_stride, const float* dt,
                   float* weights, float* alphas, int32_t* mask, int32_t* kept) {
  if (f0_stride < 1) return F2N_ERR_INVALID_ARG;
  F2N_RAY_LAUNCH(early_stop_kernel, n_rays, pts_start_end, f0, f0_stride, dt, weights, alphas, mask, kept);
}

int f2n_compact_samples(void* stream, int n_rays, const int32_t* old_start_end, const int32_t* new_start_end,
                        const int32_t* mask, const float* pts, const float* dirs, const float* dt, const float* t,
                        const int32_t* anchors, float* o_pts, float* o_dirs, float* o_dt, float* o_t, int32_t* o_anchors) {
  if (n_rays < 0) return F2N_ERR_INVALID_ARG;
  if (n_rays == 0) return F2N_OK;
  hipLaunchKernelGGL(compact_kernel, dim3(f2n_div_up(n_rays, 4)), dim3(256), 0, (hipStream_t) stream, n_rays, old_start_end,
                     new_start_end, mask, pts, dirs, dt, t, anchors, o_pts, o_dirs, o_dt, o_t, o_anchors, nullptr);
  return f2n_launch_status();
}

int f2n_compact_samples_src(void* stream, int n_rays, const int32_t* old_start_end, const int32_t* new_start_end,
                            const int32_t* mask, const float* pts, const float* dirs, const float* dt, const float* t,
                            const int32_t* anchors, float* o_pts, float* o_dirs, float* o_dt, float* o_t, int32_t* o_anchors,
                            int32_t* o_src) {
  if (n_rays < 0 || o_src == nullptr) return F2N_ERR_INVALID_ARG;
  if (n_rays == 0) return F2N_OK;
  hipLaunchKernelGGL(compact_kernel, dim3(f2n_div_up(n_rays, 4)), dim3(256), 0, (hipStream_t) stream, n_rays, old_start_end,
                     new_start_end, mask, pts, dirs, dt, t, anchors, o_pts, o_dirs, o_dt, o_t, o_anchors, o_src);
  return f2n_launch_status();
}

int f2n_composite_fwd(void* stream, int n_rays, const int32_t* pts_start_end, const float* feat, const float* dt,
                      const float* t, const float* rgb, const float* bg, float* colors, float* disparity, float* depth,
                      float* weights) {
  F2N_RAY_LAUNCH(composite_fwd_kernel, n_rays, pts_start_end, feat, dt, t, rgb, bg, colors, disparity, depth, weights);
}

int f2n_composite_bwd(void* stream, int n_rays, const int32_t* pts_start_end, const float* feat, const float* dt,
                      const float* t, const float* rgb, const float* bg, const float* dcolors, const float* ddisparity,
                      const float* ddepth, const float* dweights, float grad_scaling_progress, float* drgb, float* dfeat) {
  F2N_RAY_LAUNCH(composite_bwd_kernel, n_rays, pts_start_end, feat, dt, t, rgb, bg, dcolors, ddisparity, ddepth, dweights,
                 grad_scaling_progress, drgb, dfeat);
}

int f2n_weight_var_fwd(void* stream, int n_rays, const float* weights, const int32_t* pts_start_end, float* out_vars) {
  F2N_RAY_LAUNCH(weight_var_fwd_kernel, n_rays, weights, pts_start_end, out_vars);
}

int f2n_weight_var_bwd(void* stream, int n_rays, const float* weights, const int32_t* pts_start_end, const float* dvars,
                       float* dweights) {
  F2N_RAY_LAUNCH(weight_var_bwd_kernel, n_rays, weights, pts_start_end, dvars, dweights);
}

int f2n_flex_sum_fwd(void* stream, int n_rays, int vec, const float* val, const int32_t* start_end, float* sum) {
  if (vec < 1) return F2N_ERR_INVALID_ARG;
  F2N_RAY_LAUNCH(flex_sum_fwd_kernel, n_rays, vec, val, start_end, sum);
}
int f2n_flex_sum_bwd(void* stream, int n_rays, int vec, const float* dsum, const int32_t* start_end, float* dval) {
  if (vec < 1) return F2N_ERR_INVALID_ARG;
  F2N_RAY_LAUNCH(flex_sum_bwd_kernel, n_rays, vec, dsum, start_end, dval);
}
int f2n_flex_acc_fwd(void* stream, int n_rays, int include_this, const float* val, const int32_t* start_end, float* sum) {
  F2N_RAY_LAUNCH(flex_acc_fwd_kernel, n_rays, include_this, val, start_end, sum);
}
int f2n_flex_acc_bwd(void* stream, int n_rays, int include_this, const float* dsum, const int32_t* start_end, float* dval) {
  F2N_RAY_LAUNCH(flex_acc_bwd_kernel, n_rays, include_this, dsum, start_end, dval);
}

}  // extern "C"
