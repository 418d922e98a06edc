// What the optimiser's kernels (optim.hip) and the scatter's owner -- which applies the hash table's Adam to the slice it has just
// summed (field.hip, round 6) -- share: the step's scalars and the per-element update, LibTorch's Adam::step arithmetic.
#pragma once
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

// the scalars of torch::optim::Adam::step as LibTorch forms them (optim.hip)
F2nAdamCoef f2n_adam_coef(int step, float lr, double beta1, double beta2, float eps, float wd, float grad_scale);
