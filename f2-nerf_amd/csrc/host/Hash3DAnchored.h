// Hash3DAnchored host side (mirrors src/Field/Hash3DAnchored.h/.cpp).
#pragma once
#include "FusedMLP.h"

namespace f2n {

#define N_CHANNELS 2
#define N_LEVELS 16
#define RES_FINE_POW_2 10.f
#define RES_BASE_POW_2 3.f

class Hash3DAnchored : public Field {
 public:
  explicit Hash3DAnchored(GlobalDataPool* global_data_pool);
  // points: warped coords [n,3]; anchors: [n] (trans idx) or the sampler's [n,3] anchors read in place
  Tensor AnchoredQuery(const Tensor& points, const Tensor& anchors) override;
  // no-grad density pre-activation only (channel 0) for Renderer's early-stop pre-pass
  // keep_features: also keep the gathered h16 hash features [n,32] of every point for AnchoredQueryReuse
  Tensor QueryDensityPreAct(const Tensor& points, const Tensor& anchors, bool keep_features = false);
  // AnchoredQuery whose first n_reuse points are points src_rows[i] of the preceding QueryDensityPreAct (same table,
  // same coordinates): their 128 gathers are not repeated.  Bit-identical to AnchoredQuery(points, anchors).
  Tensor AnchoredQueryReuse(const Tensor& points, const Tensor& anchors, const Tensor& src_rows, int n_reuse);
  // The kernels behind the autograd node, callable directly (the fused train step does): feat [n,16] fp32 and
  // saved_x [n,32] h16 are written; BackwardRaw scatters into grad_h_ / mlp_->grad_scaled_.
  // f0_cached (optional): compact density pre-activations [n_reuse] of the rows served from the pre-pass cache
  void ForwardRaw(const Tensor& points, const Tensor& anchors, int stride, const Tensor& src_rows, int n_reuse, Tensor& feat,
                  Tensor& saved_x, Tensor* f0_cached = nullptr);
  void BackwardRaw(const Tensor& points, const Tensor& anchors, int stride, const Tensor& saved_x, const Tensor& dfeat);

  int LoadStates(const std::vector<Tensor>& states, int idx) override;
  std::vector<Tensor> States() override;
  std::vector<ParamGroup> OptimParamGroups() override;
  void Reset() override;
  void SyncHalf();
  void ZeroGrad();
  // The fused gather + MLP kernels exist for the field network of the shipped configs (32 -> 64 -> 16).  Any other
  // field.mlp_hidden_dim / field.n_hidden_layers (TCNNWP.cpp:86-92) runs unfused: f2n_hash_fwd -> the general MLP kernels
  // (csrc/mlp_generic.hip through FusedMLP::Query) -> f2n_hash_bwd, on the autograd tape (no feature cache, no streaming step).
  bool fused_ok_ = true;
  Tensor HashEncode(const Tensor& points, const Tensor& anchors);  // [n,32] fp32 features, differentiable w.r.t. the table
  Tensor TableGradUnscaled();  // fp32 [pool,2] = grad_h / 128 (Hash3DAnchored.cu:232)

  int pool_size_;
  int mlp_hidden_dim_, mlp_out_dim_, n_hidden_layers_;
  Tensor feat_pool_;       // [pool_size_, 2] fp32 master
  Tensor feat_pool_h_;     // fp16 working table (what the kernels gather from)
  Tensor grad_h_;          // fp16 gradient table, loss-scaled by 128 (packed-f16 atomics)
  Tensor prim_pool_;       // [16, V, 3] int32
  Tensor bias_pool_;       // [16*V, 3]
  Tensor feat_local_idx_, feat_local_size_, level_scale_;
  bool grad_clean_ = false;  // grad_h_ is known to be all zero
  Tensor prepass_x_;       // h16 [n,32] features of the last QueryDensityPreAct(keep_features = true), or undefined
  std::unique_ptr<FusedMLP> mlp_;
  int n_volumes_;
  std::vector<float> level_scale_host_;  // the 16 level scales again, for the gather's cost model (host side)
  float march_step_warped_ = 0.f;        // sample_l (x typical distance stretch): warped-space step at fineness 1
  bool balance_gather_ = true;           // field.balance_gather=false: one level pair per XCD (A/B measurements)
  int64_t active_halves_;  // halves [0, active) are the only ones any level can address (level-overlap quirk)
};

}  // namespace f2n
