// Fused Adam for the hot path's parameter groups (Field/Hash3DAnchored.cpp:124-150, Shader/SHShader.cpp:44-56,
// Renderer/Renderer.cpp:238-258): LibTorch's Adam::step arithmetic
//   exp_avg = exp_avg*b1 + (1-b1)*g ; exp_avg_sq = exp_avg_sq*b2 + (1-b2)*g*g ;
//   denom = sqrt(exp_avg_sq)/sqrt(1-b2^t) + eps ; p += -(lr/(1-b1^t)) * exp_avg/denom      (L2 decay: g += wd*p)
// in ONE streaming pass that also does what the reference spends separate full-table passes on: the fp16->fp32
// widening and /128 of the hash gradient (Hash3DAnchored.cu:232), the fp32->fp16 refresh of the working table
// (Hash3DAnchored.cu:186 / TCNNWP.cpp:111) and the re-zeroing of the gradient buffer (Hash3DAnchored.cu:222).
#include "adam_dev.h"


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

// The last block to arrive adds the blocks' partial sums up (one wave, lane b owns block b's partial, fixed butterfly order:
// the result does not depend on which block that is) and writes the eight outputs -- the second launch this used to be cost a
// dependent boundary on the step's critical queue for a microsecond of work.  `arrived` lives behind the partials in the
// workspace, is zero when the workspace is created and wraps back to zero with the last arrival (atomicInc).
__global__ __launch_bounds__(256) void train_loss_kernel(F2nLossArgs a, float* __restrict__ partials, unsigned* __restrict__ arrived,
                                                         float* __restrict__ out) {
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
  __shared__ int s_last;
  __threadfence();  // this block's partials are visible device-wide before its arrival is
  __syncthreads();
  if (threadIdx.x == 0) s_last = atomicInc(arrived, gridDim.x - 1) == gridDim.x - 1;
  __syncthreads();
  if (!s_last || threadIdx.x >= 64) return;
  __threadfence();
  const int n_blocks = gridDim.x;
  float t[F2N_LOSS_TERMS];
#pragma unroll
  for (int k = 0; k < F2N_LOSS_TERMS; k++) {  // one wave: lane b owns block b's partial, fixed butterfly order
    float s = 0.f;
    for (int b = threadIdx.x; b < n_blocks; b += 64)  // (device-scope loads: the other blocks' stores, not this CU's L1)
      s += __hip_atomic_load(partials + b * F2N_LOSS_TERMS + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off);
    t[k] = s;
  }
  if (threadIdx.x != 0) return;
  const float f_col = (float) max(3 * a.n_rays, 1), f_ray = (float) max(a.n_rays, 1), f_tv = (float) max(a.n_edge * a.feat_dim, 1);
  const float color = t[0] / f_col, var = a.var != nullptr ? t[1] / f_ray : 0.f, disp = a.disp != nullptr ? t[2] / f_ray : 0.f;
  const float tv = a.edge != nullptr ? t[3] / f_tv : 0.f;
  out[0] = color + var * a.var_w + disp * a.disp_w + tv * a.tv_w;
  out[1] = color;
  out[2] = var;
  out[3] = disp;
  out[4] = tv;
  out[5] = t[4] / f_col;
  out[6] = 0.f;
  out[7] = 0.f;
}

// TCNNWP.cpp:234-240: are the (loss-scaled) parameter gradients of the two MLPs finite?  One block, no atomics, no
// pre-zeroing: flags = {a has a non-finite value, b has one, either}.
__global__ __launch_bounds__(1024) void nonfinite_flags_kernel(int n_a, const float* __restrict__ a, int n_b,
                                                               const float* __restrict__ b, int32_t* __restrict__ flags,
                                                               int32_t* __restrict__ mirror) {
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
    if (mirror != nullptr) {  // mapped host memory: the host's copy without a copy launch behind the optimiser
      mirror[0] = s_bad[0];
      mirror[1] = s_bad[1];
      mirror[2] = s_bad[0] | s_bad[1];
    }
  }
}

// The small parameter groups of one iteration (field MLP, colour MLP, appearance embedding: ~11 k values) in ONE
// single-block launch: finiteness flags of the checked groups (the layout of f2n_nonfinite_flags: group 0, group 1, either),
// then Adam on every group, predicated on those flags (and on an optional external one).  Same per-element arithmetic as
// adam_kernel; replaces four launches of a few microseconds each on the per-iteration floor.
#define F2N_ADAM_MAX_GROUPS 4
struct F2nAdamGroupDev {
  float *param, *grad, *exp_avg, *exp_avg_sq;
  half_t* param_h;
  int n, grad_round_h16, check_finite;
  F2nAdamCoef k;
};
struct F2nAdamGroupsDev {
  F2nAdamGroupDev g[F2N_ADAM_MAX_GROUPS];
};

__global__ __launch_bounds__(1024) void adam_small_groups_kernel(F2nAdamGroupsDev gs, int n_groups, int zero_grad,
                                                                 int32_t* __restrict__ flags, const int32_t* __restrict__ skip_in) {
  bool bad[F2N_ADAM_MAX_GROUPS];
#pragma unroll
  for (int q = 0; q < F2N_ADAM_MAX_GROUPS; q++) {
    bool b = false;
    if (q < n_groups && gs.g[q].check_finite)
      for (int i = threadIdx.x; i < gs.g[q].n; i += blockDim.x) b |= !isfinite(gs.g[q].grad[i]);
    bad[q] = __syncthreads_or(b ? 1 : 0) != 0;  // block-wide: every thread knows the group's flag
  }
  bool any_bad = false;
#pragma unroll
  for (int q = 0; q < F2N_ADAM_MAX_GROUPS; q++) any_bad |= bad[q];
  if (flags != nullptr && threadIdx.x == 0) {
    flags[0] = bad[0] ? 1 : 0;
    flags[1] = bad[1] ? 1 : 0;
    flags[2] = any_bad ? 1 : 0;
  }
  const bool skip = any_bad || (skip_in != nullptr && *skip_in != 0);
#pragma unroll
  for (int q = 0; q < F2N_ADAM_MAX_GROUPS; q++) {
    if (q >= n_groups) continue;
    const F2nAdamGroupDev& G = gs.g[q];
    for (int i = threadIdx.x; i < G.n; i += blockDim.x) {
      if (skip) {  // dropped iteration (ExpRunner.cpp:131-134): parameters and moments stay, the gradient is consumed
        if (zero_grad) G.grad[i] = 0.f;
        continue;
      }
      float g = G.grad[i];
      if (zero_grad) G.grad[i] = 0.f;
      if (G.grad_round_h16) g = (float) (half_t) ((float) (half_t) g * G.k.grad_scale);
      else g = g * G.k.grad_scale;
      float m = G.exp_avg[i], v = G.exp_avg_sq[i];
      const float p = f2n_adam_update(G.param[i], g, m, v, G.k);
      G.param[i] = p;
      G.exp_avg[i] = m;
      G.exp_avg_sq[i] = v;
      if (G.param_h != nullptr) G.param_h[i] = (half_t) p;
    }
  }
}

// Every parameter group of an iteration in ONE launch: the first `small_blocks` blocks step the small fp32 groups (flat index
// over the concatenated groups), the others the hash table (the adam_h16grad_kernel body).  All of it is predicated on a flag
// that a previous launch computed (f2n_nonfinite_flags: flags[2]), so nothing here waits for anything block-wide -- the
// single-block flags-then-step launch this replaces took 21 us in front of the 41 us table pass, on the step's tail.
__global__ __launch_bounds__(256) void adam_fused_kernel(F2nAdamGroupsDev gs, int n_groups, int small_blocks, int n4,
                                                         float4_t* __restrict__ param, half4_t* __restrict__ grad,
                                                         float4_t* __restrict__ exp_avg, float4_t* __restrict__ exp_avg_sq,
                                                         F2nAdamCoef k, half4_t* __restrict__ param_h, int zero_grad,
                                                         const int32_t* __restrict__ skip_flag) {
  const bool skip = skip_flag != nullptr && *skip_flag != 0;
  if ((int) blockIdx.x < small_blocks) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
#pragma unroll
    for (int q = 0; q < F2N_ADAM_MAX_GROUPS; q++) {
      if (q >= n_groups) break;
      const F2nAdamGroupDev& G = gs.g[q];
      if (i < G.n) {
        float g = G.grad[i];
        if (zero_grad) G.grad[i] = 0.f;
        if (!skip) {
          if (G.grad_round_h16) g = (float) (half_t) ((float) (half_t) g * G.k.grad_scale);
          else g = g * G.k.grad_scale;
          float m = G.exp_avg[i], v = G.exp_avg_sq[i];
          const float p = f2n_adam_update(G.param[i], g, m, v, G.k);
          G.param[i] = p;
          G.exp_avg[i] = m;
          G.exp_avg_sq[i] = v;
          if (G.param_h != nullptr) G.param_h[i] = (half_t) p;
        }
        return;
      }
      i -= G.n;
    }
    return;
  }
  const int stride = (gridDim.x - small_blocks) * blockDim.x;
  const int first = (blockIdx.x - small_blocks) * blockDim.x + threadIdx.x;
  if (skip) {  // dropped iteration: leave parameters and moments, only clear the gradient
    if (zero_grad)
      for (int i = first; i < n4; i += stride) grad[i] = half4_t{0, 0, 0, 0};
    return;
  }
  for (int i = first; i < n4; i += stride) {
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

// The scalars of torch::optim::Adam::step as LibTorch forms them (adam.cpp): betas are doubles (AdamOptions), `1 - beta` and the bias
// corrections `1 - pow(beta, step)` are double expressions, step_size = lr / bias_correction1 as well; each is narrowed to float only
// where it meets a float tensor (mul_(beta1), add_(grad, 1 - beta1), addcmul_(..., 1 - beta2), div by sqrt(bias_correction2),
// addcdiv_(..., -step_size)).
F2nAdamCoef f2n_adam_coef(int step, float lr, double beta1, double beta2, float eps, float wd, float grad_scale) {
  const double bc1 = 1.0 - pow(beta1, (double) step);
  const double bc2 = 1.0 - pow(beta2, (double) step);
  F2nAdamCoef k;
  k.lr_over_bc1 = (float) ((double) lr / bc1);
  k.sqrt_bc2 = (float) sqrt(bc2);
  k.beta1 = (float) beta1;
  k.beta2 = (float) beta2;
  k.one_m_beta1 = (float) (1.0 - beta1);
  k.one_m_beta2 = (float) (1.0 - beta2);
  k.eps = eps;
  k.weight_decay = wd;
  k.grad_scale = grad_scale;
  return k;
}

#if F2N_DEBUG_BUILD  // (include/f2n_debug.h: the debug variant of the library only)
// Leaves `value`-derived garbage in 64 KB of LDS and ~100 vector registers of every CU (f2n_debug_pollute): what a co-tenant's
// kernels do to the state a kernel finds when it starts.  A kernel that reads LDS or registers it never wrote then depends on it.
__global__ __launch_bounds__(256) void debug_pollute_kernel(unsigned value, unsigned* __restrict__ sink) {
  extern __shared__ unsigned s_junk[];
  unsigned r[96];
#pragma unroll
  for (int i = 0; i < 96; i++) r[i] = value * 2654435761u + (unsigned) i * 40503u + threadIdx.x;
  for (int i = threadIdx.x; i < 16 * 1024; i += 256) s_junk[i] = value ^ (0x9E3779B9u * (unsigned) i);
  __syncthreads();
  unsigned acc = s_junk[(threadIdx.x * 61u) & (16 * 1024 - 1)];
#pragma unroll
  for (int i = 0; i < 96; i++) acc = acc * 31u + r[i];  // (keeps the registers live)
  if (sink != nullptr && acc == 0x12345678u) sink[0] = acc;
}

// One wave that spins for `ticks` of the 100 MHz constant clock: a delay on a stream (f2n_debug_spin; the race amplifier of
// Renderer's F2N_DEBUG_SIDE_DELAY).
__global__ void debug_spin_kernel(long long ticks) {
  const long long t0 = (long long) wall_clock64();
  while ((long long) wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}
#endif

extern "C" {

int f2n_adam_coefficients(int step, float lr, double beta1, double beta2, float eps, float weight_decay, float grad_scale, float* out9) {
  if (step < 1 || out9 == nullptr) return F2N_ERR_INVALID_ARG;
  const F2nAdamCoef k = f2n_adam_coef(step, lr, beta1, beta2, eps, weight_decay, grad_scale);
  const float v[9] = {k.lr_over_bc1, k.sqrt_bc2, k.beta1, k.beta2, k.one_m_beta1, k.one_m_beta2, k.eps, k.weight_decay, k.grad_scale};
  for (int i = 0; i < 9; i++) out9[i] = v[i];
  return F2N_OK;
}

int f2n_adam_step(void* stream, int n, float* param, float* grad, float grad_scale, int grad_round_h16, float* exp_avg,
                  float* exp_avg_sq, int step, float lr, double beta1, double beta2, float eps, float weight_decay,
                  void* param_h_or_null, int zero_grad, const int32_t* skip_flag) {
  if (n < 0 || step < 1) return F2N_ERR_INVALID_ARG;
  if (n == 0) return F2N_OK;
  const F2nAdamCoef k = f2n_adam_coef(step, lr, beta1, beta2, eps, weight_decay, grad_scale);
  hipLaunchKernelGGL(adam_kernel, dim3(f2n_div_up(n, 256)), dim3(256), 0, (hipStream_t) stream, n, param, grad, exp_avg,
                     exp_avg_sq, k, grad_round_h16, (half_t*) param_h_or_null, zero_grad, skip_flag);
  return f2n_launch_status();
}

int f2n_adam_step_h16grad(void* stream, int n, float* param, void* grad_h, float grad_scale, float* exp_avg, float* exp_avg_sq,
                          int step, float lr, double beta1, double beta2, float eps, float weight_decay, void* param_h,
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

int f2n_adam_small_groups(void* stream, int n_groups, const F2nAdamGroup* groups, int step, float lr, double beta1, double beta2,
                          float eps, int zero_grad, int32_t* flags, const int32_t* skip_flag) {
  if (n_groups < 1 || n_groups > F2N_ADAM_MAX_GROUPS || groups == nullptr || step < 1) return F2N_ERR_INVALID_ARG;
  F2nAdamGroupsDev gs = {};
  for (int q = 0; q < n_groups; q++) {
    const F2nAdamGroup& g = groups[q];
    if (g.n < 0 || (g.n > 0 && (g.param == nullptr || g.grad == nullptr || g.exp_avg == nullptr || g.exp_avg_sq == nullptr)))
      return F2N_ERR_INVALID_ARG;
    if (g.check_finite && q > 1) return F2N_ERR_UNSUPPORTED;  // the flag layout names groups 0 and 1
    gs.g[q].param = g.param;
    gs.g[q].grad = g.grad;
    gs.g[q].exp_avg = g.exp_avg;
    gs.g[q].exp_avg_sq = g.exp_avg_sq;
    gs.g[q].param_h = (half_t*) g.param_h;
    gs.g[q].n = g.n;
    gs.g[q].grad_round_h16 = g.grad_round_h16;
    gs.g[q].check_finite = g.check_finite;
    gs.g[q].k = f2n_adam_coef(step, lr, beta1, beta2, eps, g.weight_decay, g.grad_scale);
  }
  hipLaunchKernelGGL(adam_small_groups_kernel, dim3(1), dim3(1024), 0, (hipStream_t) stream, gs, n_groups, zero_grad, flags, skip_flag);
  return f2n_launch_status();
}

int f2n_adam_fused(void* stream, int n_groups, const F2nAdamGroup* groups, int n_table, float* table_param, void* table_grad_h,
                   float table_grad_scale, float* table_exp_avg, float* table_exp_avg_sq, void* table_param_h, int step, float lr,
                   double beta1, double beta2, float eps, int zero_grad, const int32_t* skip_flag) {
  if (n_groups < 0 || n_groups > F2N_ADAM_MAX_GROUPS || (n_groups > 0 && groups == nullptr) || step < 1 || n_table < 0 ||
      (n_table & 3) != 0 || (n_table > 0 && (table_param == nullptr || table_grad_h == nullptr || table_param_h == nullptr)))
    return F2N_ERR_INVALID_ARG;
  F2nAdamGroupsDev gs = {};
  long n_small = 0;
  for (int q = 0; q < n_groups; q++) {
    const F2nAdamGroup& g = groups[q];
    if (g.n < 0 || (g.n > 0 && (g.param == nullptr || g.grad == nullptr || g.exp_avg == nullptr || g.exp_avg_sq == nullptr)))
      return F2N_ERR_INVALID_ARG;
    gs.g[q].param = g.param;
    gs.g[q].grad = g.grad;
    gs.g[q].exp_avg = g.exp_avg;
    gs.g[q].exp_avg_sq = g.exp_avg_sq;
    gs.g[q].param_h = (half_t*) g.param_h;
    gs.g[q].n = g.n;
    gs.g[q].grad_round_h16 = g.grad_round_h16;
    gs.g[q].check_finite = 0;
    gs.g[q].k = f2n_adam_coef(step, lr, beta1, beta2, eps, g.weight_decay, g.grad_scale);
    n_small += g.n;
  }
  const int small_blocks = (int) ((n_small + 255) / 256);
  const int n4 = n_table / 4;
  unsigned table_blocks = n4 > 0 ? f2n_div_up(n4, 256) : 0;
  if (table_blocks > 8192) table_blocks = 8192;
  if (small_blocks + table_blocks == 0) return F2N_OK;
  const F2nAdamCoef k = f2n_adam_coef(step, lr, beta1, beta2, eps, 0.f, table_grad_scale);
  hipLaunchKernelGGL(adam_fused_kernel, dim3(small_blocks + table_blocks), dim3(256), 0, (hipStream_t) stream, gs, n_groups, small_blocks,
                     n4, (float4_t*) table_param, (half4_t*) table_grad_h, (float4_t*) table_exp_avg, (float4_t*) table_exp_avg_sq, k,
                     (half4_t*) table_param_h, zero_grad, skip_flag);
  return f2n_launch_status();
}

int f2n_train_loss(void* stream, int n_rays, const float* pred_colors, const float* gt_colors, const float* disparity,
                   const float* sampled_var, int n_edge, int feat_dim, const float* edge_feats, float var_w, float disp_w,
                   float tv_w, float* out_losses, float* dcolors, float* ddisparity, float* dvar, float* dedge_feats) {
  if (n_rays < 0 || n_edge < 0 || feat_dim < 0 || out_losses == nullptr) return F2N_ERR_INVALID_ARG;
  if (n_rays > 0 && (pred_colors == nullptr || gt_colors == nullptr)) return F2N_ERR_INVALID_ARG;
  float* partials = (float*) f2n_ws_get(F2N_WS_LOSS, sizeof(float) * (F2N_LOSS_BLOCKS * F2N_LOSS_TERMS + 1));  // (+ arrival counter)
  if (partials == nullptr) return F2N_ERR_INVALID_ARG;
  F2nLossArgs a = {n_rays, n_edge, feat_dim, pred_colors, gt_colors, disparity, sampled_var,
                   (n_edge > 0 && feat_dim > 0) ? edge_feats : nullptr, var_w, disp_w, tv_w, dcolors, ddisparity, dvar, dedge_feats};
  hipLaunchKernelGGL(train_loss_kernel, dim3(F2N_LOSS_BLOCKS), dim3(256), 0, (hipStream_t) stream, a, partials,
                     (unsigned*) (partials + F2N_LOSS_BLOCKS * F2N_LOSS_TERMS), out_losses);
  return f2n_launch_status();
}

int f2n_nonfinite_flags_ex(void* stream, int n_a, const float* a, int n_b, const float* b, int32_t* flags, int32_t* mirror) {
  if (n_a < 0 || n_b < 0 || flags == nullptr) return F2N_ERR_INVALID_ARG;
  hipLaunchKernelGGL(nonfinite_flags_kernel, dim3(1), dim3(1024), 0, (hipStream_t) stream, n_a, a, n_b, b, flags, mirror);
  return f2n_launch_status();
}

int f2n_nonfinite_flags(void* stream, int n_a, const float* a, int n_b, const float* b, int32_t* flags) {
  return f2n_nonfinite_flags_ex(stream, n_a, a, n_b, b, flags, nullptr);
}

#if F2N_DEBUG_BUILD
int f2n_debug_pollute(void* stream, unsigned value) {
  hipLaunchKernelGGL(debug_pollute_kernel, dim3(1024), dim3(256), 64 * 1024, (hipStream_t) stream, value, (unsigned*) nullptr);
  return f2n_launch_status();
}

int f2n_debug_spin(void* stream, int microseconds) {
  if (microseconds <= 0) return F2N_OK;
  hipLaunchKernelGGL(debug_spin_kernel, dim3(1), dim3(64), 0, (hipStream_t) stream, (long long) microseconds * 100);
  return f2n_launch_status();
}
#endif

int f2n_abi_version(void) { return 13; }
#ifndef F2N_REFERENCE_NUMERICS
#define F2N_REFERENCE_NUMERICS 0
#endif
const char* f2n_build_info(void) {
  return F2N_REFERENCE_NUMERICS
             ? "f2n_hip gfx950 (hipcc, -ffp-contract=off), wave64, REFERENCE-NUMERICS build: f16 MLP forward accumulator (k-blocks "
               "of 16), hash gradient by per-addend packed-f16 atomics"
             : "f2n_hip gfx950 (hipcc, -ffp-contract=off), wave64, mfma_f32_16x16x32_f16";
}
/* 0: product numerics (fp32 MFMA accumulation, owner-binned fp32/fp64 hash-gradient sums); 1: the reference-numerics build */
int f2n_numerics_mode(void) { return F2N_REFERENCE_NUMERICS; }

}  // extern "C"
