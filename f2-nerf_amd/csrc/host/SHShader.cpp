// SHShader host logic (mirrors src/Shader/SHShader.cpp; kernels: csrc/shade.hip): the colour network behind the Shader plugin
// surface -- SH encoding, the op-by-op Query of the reference, the fused QueryFromField of the Renderer's taped path, states and
// optimiser groups.  Split out of Renderer.cpp in round 5.
#include "Renderer.h"

#include <hip/hip_runtime_api.h>

#include <cstdio>
#include <cstdlib>
#include <mutex>

namespace f2n {

using torch::autograd::AutogradContext;
using torch::autograd::variable_list;

// ---------------------------------------------------------------------------------------------------------
// SHShader
// ---------------------------------------------------------------------------------------------------------
SHShader::SHShader(GlobalDataPool* gdp) {  // SHShader.cpp:9-20
  global_data_pool_ = gdp;
  gdp->shader_ = this;
  const auto& c = gdp->config_;
  d_in_ = c.Int("shader.d_in");
  d_out_ = c.Int("shader.d_out");
  degree_ = c.Int("shader.degree");
  d_hidden_ = c.Int("shader.d_hidden");
  n_hiddens_ = c.Int("shader.n_hiddens");
  TORCH_CHECK(degree_ >= 1 && degree_ <= 8, "SH degree ", degree_, " is not supported (1..8, SHShader.cu:51-102)");
  TORCH_CHECK(d_in_ == 16 + degree_ * degree_, "shader.d_in must be 16 shading features + degree^2 SH coefficients (SHShader.cpp:24-25)");
  mlp_ = std::make_unique<FusedMLP>(gdp, d_in_, d_out_, d_hidden_, n_hiddens_);
  fused_ok_ = degree_ == 4 && d_hidden_ == 64 && n_hiddens_ == 2;
}

Tensor SHShader::SHEncode(const Tensor& dirs) {  // SHShader.cu:108-118
  Tensor d = dirs.contiguous();
  CheckDev(d, torch::kFloat32, "dirs");
  const int n = d.size(0);
  Tensor out = torch::empty({n, degree_ * degree_}, DevF32());
  F2N_CALL(f2n_sh_encode(CurStream(), n, degree_, F32P(d), F32P(out)));
  return out;
}

Tensor SHShader::Query(const Tensor& feats, const Tensor& dirs) {  // SHShader.cpp:23-29, op by op
  Tensor enc = SHEncode(dirs);
  Tensor input = torch::cat({feats, enc}, -1);
  Tensor output = mlp_->Query(input);
  const float eps = 1e-3f;
  // (the reference's expression, SHShader.cpp:27-28, on an output clamped at -80: below ~-88.7 exp(-output) is +inf in fp32 and
  // ATen's backward of 1 / (1 + e) * e forms 0 * inf = NaN -- the gradient's true limit there is 0, which the clamp delivers;
  // values are unchanged for every output >= -80.  Without it a wide colour network at the reference's learning rate 1e-2 ran
  // into a permanent "Nan!" skip after ~20 iterations (tools/debug_generic.py).)
  return (1.f + 2.f * eps) / (1.f + torch::exp(-output.clamp_min(-80.f))) - eps;
}

namespace {

struct ShadeFunction : public torch::autograd::Function<ShadeFunction> {
  static variable_list forward(AutogradContext* ctx, Tensor field_feats, Tensor color_params, Tensor app_emb, Tensor dirs,
                               Tensor sample_emb_idx, int64_t shader_ptr, int64_t emb_grad_ptr) {
    auto* sh = reinterpret_cast<SHShader*>(shader_ptr);
    Tensor feats = field_feats.contiguous();
    CheckDev(feats, torch::kFloat32, "field feats");
    TORCH_CHECK(feats.size(1) == 16 && sh->degree_ == 4 && sh->n_hiddens_ == 2, "fused shading needs 16 feats + SH4 + 2 hidden");
    const int n = feats.size(0);
    const bool emb = app_emb.defined() && sample_emb_idx.defined() && app_emb.numel() > 0 && sample_emb_idx.numel() > 0;
    Tensor rgb = torch::empty({n, 3}, DevF32());
    Tensor saved_x = torch::empty({n, 32}, DevF16());
    F2N_TIMED_CALL("shade_fwd", f2n_shade_fwd(CurStream(), n, F32P(feats), F32P(dirs), emb ? F32P(app_emb) : nullptr,
                           emb ? I32P(sample_emb_idx) : nullptr, VoidP(sh->mlp_->params_h_), F32P(rgb), VoidP(saved_x)));
    ctx->saved_data["shader"] = shader_ptr;
    ctx->saved_data["emb_grad"] = emb_grad_ptr;
    ctx->saved_data["emb"] = emb;
    ctx->save_for_backward({saved_x, sample_emb_idx});
    return {rgb};
  }
  static variable_list backward(AutogradContext* ctx, variable_list grad_output) {
    auto* sh = reinterpret_cast<SHShader*>(ctx->saved_data["shader"].toInt());
    auto* emb_grad = reinterpret_cast<Tensor*>(ctx->saved_data["emb_grad"].toInt());
    const bool emb = ctx->saved_data["emb"].toBool();
    auto saved = ctx->get_saved_variables();
    Tensor drgb = grad_output[0].contiguous();
    const int n = saved[0].size(0);
    Tensor dfeat = torch::zeros({n, 16}, DevF32());
    F2N_TIMED_CALL("shade_bwd", f2n_shade_bwd(CurStream(), n, F32P(drgb), emb ? I32P(saved[1]) : nullptr, VoidP(sh->mlp_->params_h_),
                           VoidP(saved[0]), sh->mlp_->loss_scale_, F32P(dfeat), F32P(sh->mlp_->grad_scaled_),
                           (emb && emb_grad != nullptr) ? F32P(*emb_grad) : nullptr,
                           (emb && emb_grad != nullptr) ? (int) emb_grad->size(0) : 0, nullptr));
    return {dfeat, Tensor(), Tensor(), Tensor(), Tensor(), Tensor(), Tensor()};
  }
};

}  // namespace

Tensor SHShader::QueryFromField(const Tensor& field_feats, const Tensor& dirs, const Tensor& app_emb,
                                const Tensor& sample_emb_idx, Tensor* app_emb_grad) {
  if (!fused_ok_) {
    // op by op, as the reference: shading_feat = [1 | feat[1:]] (+ app_emb[img], CustomOps::ScatterAdd) -> SHShader::Query
    // (Renderer.cpp:181-188).  The appearance embedding's gradient arrives through autograd here (app_emb.grad): the caller
    // moves it into the optimiser's buffer (ExpRunner::TrainStepAutograd).
    Tensor shading = torch::cat({torch::ones_like(field_feats.index({Slc(), Slc(0, 1)})), field_feats.index({Slc(), Slc(1, 16)})}, 1);
    const bool emb = app_emb.defined() && sample_emb_idx.defined() && app_emb.numel() > 0 && sample_emb_idx.numel() > 0;
    if (emb) shading = shading + app_emb.index_select(0, sample_emb_idx.to(torch::kInt64));
    return Query(shading, dirs.contiguous());
  }
  return ShadeFunction::apply(field_feats, mlp_->params_, app_emb, dirs.contiguous(), sample_emb_idx,
                              reinterpret_cast<int64_t>(this), reinterpret_cast<int64_t>(app_emb_grad))[0];
}

int SHShader::LoadStates(const std::vector<Tensor>& states, int idx) {
  torch::NoGradGuard g;
  mlp_->params_.copy_(states[idx++].to(torch::kCUDA).to(torch::kFloat32));
  mlp_->SyncHalf();
  return idx;
}
std::vector<Tensor> SHShader::States() { return {mlp_->params_.detach()}; }
std::vector<ParamGroup> SHShader::OptimParamGroups() {  // SHShader.cpp:44-56
  ParamGroup g;
  g.name = "color_mlp";
  g.param = mlp_->params_;
  g.grad = mlp_->grad_scaled_;
  g.param_h = mlp_->params_h_;
  g.weight_decay = 1e-6f;
  g.grad_round_h16 = true;
  g.grad_scale = -1.f;
  return {g};
}
void SHShader::Reset() { mlp_->InitParams(); }

std::unique_ptr<Shader> ConstructShader(GlobalDataPool* gdp) {  // ShaderFactory.cpp:8-17
  const std::string type = gdp->config_.Str("shader.type");
  TORCH_CHECK(type == "SHShader", "unknown shader.type: ", type);
  return std::make_unique<SHShader>(gdp);
}

}  // namespace f2n
