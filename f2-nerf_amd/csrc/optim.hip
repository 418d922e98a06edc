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
__global__ void adam_kernel(int n, float* __restrict__ param, const float* __restrict__ grad, float* __restrict__ exp_avg,
                            float* __restrict__ exp_avg_sq, F2nAdamCoef k, int grad_round_h16, half_t* __restrict__ param_h) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float g = grad[i];
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
                                    float4_t* __restrict__ exp_avg_sq, F2nAdamCoef k, half4_t* __restrict__ param_h, int zero_grad) {
  const int stride = gridDim.x * blockDim.x;
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

int f2n_adam_step(void* stream, int n, float* param, const float* grad, float grad_scale, int grad_round_h16, float* exp_avg,
                  float* exp_avg_sq, int step, float lr, float beta1, float beta2, float eps, float weight_decay,
                  void* param_h_or_null) {
  if (n < 0 || step < 1) return F2N_ERR_INVALID_ARG;
  if (n == 0) return F2N_OK;
  const F2nAdamCoef k = f2n_adam_coef(step, lr, beta1, beta2, eps, weight_decay, grad_scale);
  hipLaunchKernelGGL(adam_kernel, dim3(f2n_div_up(n, 256)), dim3(256), 0, (hipStream_t) stream, n, param, grad, exp_avg,
                     exp_avg_sq, k, grad_round_h16, (half_t*) param_h_or_null);
  return f2n_launch_status();
}

int f2n_adam_step_h16grad(void* stream, int n, float* param, void* grad_h, float grad_scale, float* exp_avg, float* exp_avg_sq,
                          int step, float lr, float beta1, float beta2, float eps, float weight_decay, void* param_h,
                          int zero_grad) {
  if (n < 0 || step < 1 || (n & 3) != 0 || param_h == nullptr) return F2N_ERR_INVALID_ARG;
  if (n == 0) return F2N_OK;
  const F2nAdamCoef k = f2n_adam_coef(step, lr, beta1, beta2, eps, weight_decay, grad_scale);
  const int n4 = n / 4;
  unsigned blocks = f2n_div_up(n4, 256);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(adam_h16grad_kernel, dim3(blocks), dim3(256), 0, (hipStream_t) stream, n4, (float4_t*) param,
                     (half4_t*) grad_h, (float4_t*) exp_avg, (float4_t*) exp_avg_sq, k, (half4_t*) param_h, zero_grad);
  return f2n_launch_status();
}

int f2n_abi_version(void) { return 1; }
const char* f2n_build_info(void) { return "f2n_hip gfx950 (hipcc, -ffp-contract=off), wave64, mfma_f32_16x16x32_f16"; }

}  // extern "C"
