// Fused Adam for the hot path's parameter groups (Field/Hash3DAnchored.cpp:124-150, Shader/SHShader.cpp:44-56,
// Renderer/Renderer.cpp:238-258): LibTorch's Adam::step arithmetic
//   exp_avg = exp_avg*b1 + (1-b1)*g ; exp_avg_sq = exp_avg_sq*b2 + (1-b2)*g*g ;
//   denom = sqrt(exp_avg_sq)/sqrt(1-b2^t) + eps ; p += -(lr/(1-b1^t)) * exp_avg/denom      (L2 decay: g += wd*p)
// in ONE streaming pass that also does what the reference spends separate full-table passes on: the fp16->fp32
// widening and /128 of the hash gradient (Hash3DAnchored.cu:232), the fp32->fp16 refresh of the working table
// (Hash3DAnchored.cu:186 / TCNNWP.cpp:111) and the re-zeroing of the gradient buffer (Hash3DAnchored.cu:222).
#include "f2n_dev.h"

struct F2nAdamCoef {
  float lr_over_bc1, sqrt_bc2, beta1, beta2, one_m_beta1, one_m_beta2, eps, weight_decay, grad_scale;
};

__device__ __forceinline__ float f2n_adam_update(float p, float g, float& m, float& v, const F2nAdamCoef& k) {
  if (k.weight_decay != 0.f) g = g + k.weight_decay * p;
  m = m * k.beta1 + k.one_m_beta1 * g;
  v = v * k.beta2 + k.one_m_beta2 * g * g;
  const float denom = sqrtf(v) / k.sqrt_bc2 + k.eps;
  return p + (-k.lr_over_bc1) * (m / denom);
}

// grad_round_h16: reproduce the two binary16 roundings the reference applies to MLP parameter gradients
// (tcnn writes dL/dparams in param precision while still loss-scaled, Field/TCNNWP.cpp:214-215; autograd then
// casts the unscaled fp32 gradient back to the f16 dtype of the Function's `params` input, :111,:242).
__global__ void adam_kernel(int n, float* __restrict__ param, float* __restrict__ grad, float* __restrict__ exp_avg,
                            float* __restrict__ exp_avg_sq, F2nAdamCoef k, int grad_round_h16, half_t* __restrict__ param_h,
                            int zero_grad, const int32_t* __restrict__ skip_flag) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (skip_flag != nullptr && *skip_flag != 0) {  // non-finite gradients: the iteration is dropped (ExpRunner.cpp:131-134)
    if (zero_grad) grad[i] = 0.f;
    return;
  }
  float g = grad[i];
  if (zero_grad) grad[i] = 0.f;  // optimizer.zero_grad() of the next iteration (ExpRunner.cpp:135), fused
  if (grad_round_h16) g = (float) (half_t) ((float) (half_t) g * k.grad_scale);
  else g = g * k.grad_scale;
  float m = exp_avg[i], v = exp_avg_sq[i];
  const float p = f2n_adam_update(param[i], g, m, v, k);
  param[i] = p;
  exp_avg[i] = m;
  exp_avg_sq[i] = v;
  if (param_h != nullptr) param_h[i] = (half_t) p;
}

// 4 entries per lane: 16-B fp32 and 8-B fp16 accesses.
__global__ void adam_h16grad_kernel(int n4, float4_t* __restrict__ param, half4_t* __restrict__ grad, float4_t* __restrict__ exp_avg,
                                    float4_t* __restrict__ exp_avg_sq, F2nAdamCoef k, half4_t* __restrict__ param_h, int zero_grad,
                                    const int32_t* __restrict__ skip_flag) {
  const int stride = gridDim.x * blockDim.x;
  if (skip_flag != nullptr && *skip_flag != 0) {  // dropped iteration: leave parameters and moments, only clear the gradient
    if (zero_grad)
      for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) grad[i] = half4_t{0, 0, 0, 0};
    return;
  }
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    const half4_t gh = grad[i];
    float4_t p = param[i], m = exp_avg[i], v = exp_avg_sq[i];
    half4_t ph;
#pragma unroll
    for (int c = 0; c < 4; c++) {
      float mm = m[c], vv = v[c];
      p[c] = f2n_adam_update(p[c], (float) gh[c] * k.grad_scale, mm, vv, k);
      m[c] = mm;
      v[c] = vv;
      ph[c] = (half_t) p[c];
    }
    param[i] = p;
    exp_avg[i] = m;
    exp_avg_sq[i] = v;
    param_h[i] = ph;
    if (zero_grad) grad[i] = half4_t{0, 0, 0, 0};
  }
}

// ---------------------------------------------------------------------------------------------------
// The loss of ExpRunner::Train (ExpRunner.cpp:95-120) and its gradient with respect to everything the renderer
// produced, in one pass: every term is a mean of an element-wise function, so the gradients are element-wise too.
//   color = mean sqrt((pred-gt)^2 + 1e-4)      disp = mean disparity^2
//   tv    = mean (edge_feat[:,0,:] - edge_feat[:,1,:])^2      var = mean sqrt(weight_var + 1e-2)
//   loss  = color + var_w*var + disp_w*disp + tv_w*tv ;  mse = mean (pred-gt)^2 (reported only)
// Partial sums go through fixed-size per-block slots reduced in a fixed order: the reported values are deterministic.
// ---------------------------------------------------------------------------------------------------
#define F2N_LOSS_BLOCKS 64
#define F2N_LOSS_TERMS 5  // color, var, disp, tv, mse

struct F2nLossArgs {
  int n_rays, n_edge, feat_dim;
  const float *pred, *gt, *disp, *var, *edge;
  float var_w, disp_w, tv_w;
  float *dcolors, *ddisp, *dvar, *dedge;
};

__global__ __launch_bounds__(256) void train_loss_kernel(F2nLossArgs a, float* __restrict__ partials) {
  __shared__ float s_red[F2N_LOSS_TERMS][256];
  float acc[F2N_LOSS_TERMS] = {0.f, 0.f, 0.f, 0.f, 0.f};
  const int tid = blockIdx.x * blockDim.x + threadIdx.x, stride = gridDim.x * blockDim.x;
  const int n_col = 3 * a.n_rays;
  const float inv_col = 1.f / (float) max(n_col, 1), inv_ray = 1.f / (float) max(a.n_rays, 1);
  for (int i = tid; i < n_col; i += stride) {
    const float d = a.pred[i] - a.gt[i];
    const float r = sqrtf(d * d + 1e-4f);
    acc[0] += r;
    acc[4] += d * d;
    if (a.dcolors != nullptr) a.dcolors[i] = d / r * inv_col;
  }
  if (a.var != nullptr)
    for (int i = tid; i < a.n_rays; i += stride) {
      const float r = sqrtf(a.var[i] + 1e-2f);
      acc[1] += r;
      if (a.dvar != nullptr) a.dvar[i] = a.var_w * .5f / r * inv_ray;
    }
  if (a.disp != nullptr)
    for (int i = tid; i < a.n_rays; i += stride) {
      const float v = a.disp[i];
      acc[2] += v * v;
      if (a.ddisp != nullptr) a.ddisp[i] = a.disp_w * 2.f * v * inv_ray;
    }
  if (a.edge != nullptr) {
    const int n_tv = a.n_edge * a.feat_dim;
    const float inv_tv = 1.f / (float) max(n_tv, 1);
    for (int i = tid; i < n_tv; i += stride) {
      const int e = i / a.feat_dim, f = i - e * a.feat_dim;
      const size_t i0 = ((size_t) 2 * e) * a.feat_dim + f, i1 = i0 + a.feat_dim;
      const float d = a.edge[i0] - a.edge[i1];
      acc[3] += d * d;
      if (a.dedge != nullptr) {
        const float g = a.tv_w * 2.f * d * inv_tv;
        a.dedge[i0] = g;
        a.dedge[i1] = -g;
      }
    }
  }
#pragma unroll
  for (int k = 0; k < F2N_LOSS_TERMS; k++) s_red[k][threadIdx.x] = acc[k];
  __syncthreads();
  for (int off = 128; off >= 1; off >>= 1) {
    if ((int) threadIdx.x < off)
#pragma unroll
      for (int k = 0; k < F2N_LOSS_TERMS; k++) s_red[k][threadIdx.x] += s_red[k][threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x < F2N_LOSS_TERMS) partials[blockIdx.x * F2N_LOSS_TERMS + threadIdx.x] = s_red[threadIdx.x][0];
}

__global__ __launch_bounds__(64) void train_loss_finalize_kernel(F2nLossArgs a, int n_blocks, const float* __restrict__ partials,
                                                                 float* __restrict__ out) {
  float t[F2N_LOSS_TERMS];
#pragma unroll
  for (int k = 0; k < F2N_LOSS_TERMS; k++) {  // one wave: lane b owns block b's partial, fixed butterfly order
    float s = 0.f;
    for (int b = threadIdx.x; b < n_blocks; b += 64) s += partials[b * F2N_LOSS_TERMS + k];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off);
    t[k] = s;
  }
  if (threadIdx.x != 0) return;
  const float n_col = (float) max(3 * a.n_rays, 1), n_ray = (float) max(a.n_rays, 1), n_tv = (float) max(a.n_edge * a.feat_dim, 1);
  const float color = t[0] / n_col, var = a.var != nullptr ? t[1] / n_ray : 0.f, disp = a.disp != nullptr ? t[2] / n_ray : 0.f;
  const float tv = a.edge != nullptr ? t[3] / n_tv : 0.f;
  out[0] = color + var * a.var_w + disp * a.disp_w + tv * a.tv_w;
  out[1] = color;
  out[2] = var;
  out[3] = disp;
  out[4] = tv;
  out[5] = t[4] / n_col;
  out[6] = 0.f;
  out[7] = 0.f;
}

// TCNNWP.cpp:234-240: are the (loss-scaled) parameter gradients of the two MLPs finite?  One block, no atomics, no
// pre-zeroing: flags = {a has a non-finite value, b has one, either}.
__global__ __launch_bounds__(1024) void nonfinite_flags_kernel(int n_a, const float* __restrict__ a, int n_b,
                                                               const float* __restrict__ b, int32_t* __restrict__ flags) {
  __shared__ int s_bad[2];
  if (threadIdx.x < 2) s_bad[threadIdx.x] = 0;
  __syncthreads();
  bool bad_a = false, bad_b = false;
  for (int i = threadIdx.x; i < n_a; i += blockDim.x) bad_a |= !isfinite(a[i]);
  for (int i = threadIdx.x; i < n_b; i += blockDim.x) bad_b |= !isfinite(b[i]);
  if (bad_a) s_bad[0] = 1;
  if (bad_b) s_bad[1] = 1;
  __syncthreads();
  if (threadIdx.x == 0) {
    flags[0] = s_bad[0];
    flags[1] = s_bad[1];
    flags[2] = s_bad[0] | s_bad[1];
  }
}

static F2nAdamCoef f2n_adam_coef(int step, float lr, float beta1, float beta2, float eps, float wd, float grad_scale) {
  const double bc1 = 1.0 - pow((double) beta1, (double) step);
  const double bc2 = 1.0 - pow((double) beta2, (double) step);
  F2nAdamCoef k;
  k.lr_over_bc1 = (float) ((double) lr / bc1);
  k.sqrt_bc2 = (float) sqrt(bc2);
  k.beta1 = beta1;
  k.beta2 = beta2;
  k.one_m_beta1 = (float) (1.0 - (double) beta1);
  k.one_m_beta2 = (float) (1.0 - (double) beta2);
  k.eps = eps;
  k.weight_decay = wd;
  k.grad_scale = grad_scale;
  return k;
}

extern "C" {

int f2n_adam_step(void* stream, int n, float* param, float* grad, float grad_scale, int grad_round_h16, float* exp_avg,
                  float* exp_avg_sq, int step, float lr, float beta1, float beta2, float eps, float weight_decay,
                  void* param_h_or_null, int zero_grad, const int32_t* skip_flag) {
  if (n < 0 || step < 1) return F2N_ERR_INVALID_ARG;
  if (n == 0) return F2N_OK;
  const F2nAdamCoef k = f2n_adam_coef(step, lr, beta1, beta2, eps, weight_decay, grad_scale);
  hipLaunchKernelGGL(adam_kernel, dim3(f2n_div_up(n, 256)), dim3(256), 0, (hipStream_t) stream, n, param, grad, exp_avg,
                     exp_avg_sq, k, grad_round_h16, (half_t*) param_h_or_null, zero_grad, skip_flag);
  return f2n_launch_status();
}

int f2n_adam_step_h16grad(void* stream, int n, float* param, void* grad_h, float grad_scale, float* exp_avg, float* exp_avg_sq,
                          int step, float lr, float beta1, float beta2, float eps, float weight_decay, void* param_h,
                          int zero_grad, const int32_t* skip_flag) {
  if (n < 0 || step < 1 || (n & 3) != 0 || param_h == nullptr) return F2N_ERR_INVALID_ARG;
  if (n == 0) return F2N_OK;
  const F2nAdamCoef k = f2n_adam_coef(step, lr, beta1, beta2, eps, weight_decay, grad_scale);
  const int n4 = n / 4;
  unsigned blocks = f2n_div_up(n4, 256);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(adam_h16grad_kernel, dim3(blocks), dim3(256), 0, (hipStream_t) stream, n4, (float4_t*) param,
                     (half4_t*) grad_h, (float4_t*) exp_avg, (float4_t*) exp_avg_sq, k, (half4_t*) param_h, zero_grad,
                     skip_flag);
  return f2n_launch_status();
}

int f2n_train_loss(void* stream, int n_rays, const float* pred_colors, const float* gt_colors, const float* disparity,
                   const float* sampled_var, int n_edge, int feat_dim, const float* edge_feats, float var_w, float disp_w,
                   float tv_w, float* out_losses, float* dcolors, float* ddisparity, float* dvar, float* dedge_feats) {
  if (n_rays < 0 || n_edge < 0 || feat_dim < 0 || out_losses == nullptr) return F2N_ERR_INVALID_ARG;
  if (n_rays > 0 && (pred_colors == nullptr || gt_colors == nullptr)) return F2N_ERR_INVALID_ARG;
  float* partials = (float*) f2n_ws_get(F2N_WS_LOSS, sizeof(float) * F2N_LOSS_BLOCKS * F2N_LOSS_TERMS);
  if (partials == nullptr) return F2N_ERR_INVALID_ARG;
  F2nLossArgs a = {n_rays, n_edge, feat_dim, pred_colors, gt_colors, disparity, sampled_var,
                   (n_edge > 0 && feat_dim > 0) ? edge_feats : nullptr, var_w, disp_w, tv_w, dcolors, ddisparity, dvar, dedge_feats};
  hipLaunchKernelGGL(train_loss_kernel, dim3(F2N_LOSS_BLOCKS), dim3(256), 0, (hipStream_t) stream, a, partials);
  hipLaunchKernelGGL(train_loss_finalize_kernel, dim3(1), dim3(64), 0, (hipStream_t) stream, a, F2N_LOSS_BLOCKS, partials, out_losses);
  return f2n_launch_status();
}

int f2n_nonfinite_flags(void* stream, int n_a, const float* a, int n_b, const float* b, int32_t* flags) {
  if (n_a < 0 || n_b < 0 || flags == nullptr) return F2N_ERR_INVALID_ARG;
  hipLaunchKernelGGL(nonfinite_flags_kernel, dim3(1), dim3(1024), 0, (hipStream_t) stream, n_a, a, n_b, b, flags);
  return f2n_launch_status();
}

int f2n_abi_version(void) { return 4; }
const char* f2n_build_info(void) { return "f2n_hip gfx950 (hipcc, -ffp-contract=off), wave64, mfma_f32_16x16x32_f16"; }

}  // extern "C"
