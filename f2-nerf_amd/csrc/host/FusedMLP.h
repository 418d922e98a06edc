// FusedMLP: the MI355X replacement of the reference's TCNNWP wrapper around tcnn::cpp::Module
// (src/Field/TCNNWP.h/.cpp): "FullyFusedMLP", ReLU, no output activation, fp16 weights, manual loss scale 128
// with the NaN flag / halving of TCNNWP.cpp:234-240.  Kernels: csrc/mlp_dev.h through f2n_mlp_*.
#pragma once
#include "Field.h"

namespace f2n {

class FusedMLP : public Field {
 public:
  FusedMLP(GlobalDataPool* global_data_pool, int d_in, int d_out, int d_hidden, int n_hidden_layers);
  Tensor Query(const Tensor& pts) override;  // [n, d_in] fp32 -> [n, d_out] fp32 (autograd-enabled)
  void InitParams();
  void SyncHalf();        // params_ (fp32 master) -> params_h_ (TCNNWP.cpp:111; normally done by the optimiser step)
  void ZeroGrad();
  Tensor GradUnscaled();  // the gradient as the reference's autograd would deliver it (two fp16 roundings)
  // non-finite gradient -> backward_nan_ = true and loss_scale_ halved (floor 1); one host read-back
  bool CheckGradFinite();

  int d_in_, d_out_, d_hidden_, n_hidden_layers_;
  int n_params_;
  Tensor params_;        // fp32 master, requires_grad
  Tensor params_h_;      // fp16 working copy used by the kernels
  Tensor grad_scaled_;   // fp32 accumulator of loss_scale * dL/dparams
  float loss_scale_ = 128.f;
};

}  // namespace f2n
