// FusedMLP + Hash3DAnchored host logic.
#include "Hash3DAnchored.h"

#include <cstdlib>

namespace f2n {

using torch::autograd::AutogradContext;
using torch::autograd::variable_list;

// ---------------------------------------------------------------------------------------------------------
// FusedMLP
// ---------------------------------------------------------------------------------------------------------
FusedMLP::FusedMLP(GlobalDataPool* gdp, int d_in, int d_out, int d_hidden, int n_hidden_layers) {
  global_data_pool_ = gdp;
  d_in_ = d_in; d_out_ = d_out; d_hidden_ = d_hidden; n_hidden_layers_ = n_hidden_layers;
  n_params_ = f2n_mlp_n_params(d_in, d_hidden, n_hidden_layers);
  TORCH_CHECK(n_params_ > 0 && d_out <= F2N_MLP_OUT_PAD, "unsupported MLP shape");
  params_ = torch::zeros({n_params_}, DevF32());
  params_h_ = torch::zeros({n_params_}, DevF16());
  grad_scaled_ = torch::zeros({n_params_}, DevF32());
  InitParams();
  params_.requires_grad_(true);
}

void FusedMLP::InitParams() {
  torch::NoGradGuard g;
  const uint64_t seed = 19970826;  // TCNNWP.cpp:96
  F2N_CALL(f2n_mlp_init_params(CurStream(), seed + (uint64_t) n_hidden_layers_, d_in_, d_hidden_, n_hidden_layers_,
                               F32P(params_)));
  SyncHalf();
}

void FusedMLP::SyncHalf() {
  F2N_CALL(f2n_params_to_h16(CurStream(), n_params_, F32P(params_), VoidP(params_h_)));
}

void FusedMLP::ZeroGrad() { grad_scaled_.zero_(); }

Tensor FusedMLP::GradUnscaled() {
  return ((grad_scaled_.to(torch::kFloat16).to(torch::kFloat32) / loss_scale_).to(torch::kFloat16)).to(torch::kFloat32);
}

bool FusedMLP::CheckGradFinite() {  // TCNNWP.cpp:234-240
  bool finite = torch::all(torch::isfinite(grad_scaled_)).item<bool>();
  if (!finite) {
    global_data_pool_->backward_nan_ = true;
    loss_scale_ = std::max(loss_scale_ / 2.f, 1.f);
  }
  return finite;
}

namespace {

struct MlpFunction : public torch::autograd::Function<MlpFunction> {
  static variable_list forward(AutogradContext* ctx, Tensor x, Tensor params, int64_t mlp_ptr) {
    auto* mlp = reinterpret_cast<FusedMLP*>(mlp_ptr);
    ctx->saved_data["mlp"] = mlp_ptr;
    ctx->save_for_backward({x});
    const int n = x.size(0);
    Tensor out = torch::empty({n, F2N_MLP_OUT_PAD}, DevF16());
    F2N_TIMED_CALL("mlp_fwd", f2n_mlp_fwd(CurStream(), n, mlp->d_in_, mlp->d_hidden_, mlp->n_hidden_layers_, VoidP(mlp->params_h_),
                         F32P(x), VoidP(out)));
    return {out.to(torch::kFloat32)};
  }
  static variable_list backward(AutogradContext* ctx, variable_list grad_output) {
    auto* mlp = reinterpret_cast<FusedMLP*>(ctx->saved_data["mlp"].toInt());
    Tensor x = ctx->get_saved_variables()[0];
    Tensor dy = grad_output[0].contiguous();
    const int n = x.size(0);
    Tensor dx = torch::empty({n, mlp->d_in_}, DevF32());
    F2N_TIMED_CALL("mlp_bwd", f2n_mlp_bwd(CurStream(), n, mlp->d_in_, mlp->d_hidden_, mlp->n_hidden_layers_, mlp->loss_scale_,
                         VoidP(mlp->params_h_), F32P(x), F32P(dy), F32P(mlp->grad_scaled_), F32P(dx)));
    return {dx, Tensor(), Tensor()};  // parameter gradient is delivered through mlp->grad_scaled_
  }
};

}  // namespace

Tensor FusedMLP::Query(const Tensor& pts) {  // TCNNWP.cpp:102-113 (no 128-row padding needed here)
  Tensor x = pts.contiguous();
  CheckDev(x, torch::kFloat32, "mlp input");
  TORCH_CHECK(x.size(1) == d_in_, "mlp input width");
  Tensor out = MlpFunction::apply(x, params_, reinterpret_cast<int64_t>(this))[0];
  return out.index({Slc(), Slc(0, d_out_)}).contiguous();
}

// ---------------------------------------------------------------------------------------------------------
// Hash3DAnchored
// ---------------------------------------------------------------------------------------------------------
Hash3DAnchored::Hash3DAnchored(GlobalDataPool* gdp) {  // Hash3DAnchored.cpp:19-82
  global_data_pool_ = gdp;
  gdp->scene_field_ = this;
  const auto& c = gdp->config_;
  pool_size_ = (1 << c.Int("field.log2_table_size")) * N_LEVELS;
  mlp_hidden_dim_ = c.Int("field.mlp_hidden_dim");
  mlp_out_dim_ = c.Int("field.mlp_out_dim");
  n_hidden_layers_ = c.Int("field.n_hidden_layers");
  n_volumes_ = gdp->n_volumes_;
  feat_pool_ = (torch::rand({pool_size_, N_CHANNELS}, DevF32()) * .2f - 1.f) * 1e-4f;
  feat_pool_.requires_grad_(true);
  feat_pool_h_ = torch::zeros({pool_size_, N_CHANNELS}, DevF16());
  grad_h_ = torch::zeros({pool_size_, N_CHANNELS}, DevF16());
  // Per-(level, warp) hash primes in [2^28, 2^30) by rejection sampling and random biases in [100, 1100)
  // (Hash3DAnchored.cpp:40-66); both are part of the serialised state, so LoadStates replaces them on resume.
  {
    std::vector<int> small;  // primes up to sqrt(2^30)
    std::vector<char> sieve(32769, 1);
    for (int i = 2; i <= 32768; i++) {
      if (!sieve[i]) continue;
      small.push_back(i);
      for (int64_t k = (int64_t) i * i; k <= 32768; k += i) sieve[k] = 0;
    }
    const int64_t need = int64_t(3) * N_LEVELS * n_volumes_;
    std::vector<int> chosen;
    chosen.reserve(need);
    while ((int64_t) chosen.size() < need) {
      Tensor cand = torch::randint(1 << 28, 1 << 30, {std::max<int64_t>(4096, 24 * (need - (int64_t) chosen.size()))}, CpuI32());
      const int* cd = cand.data_ptr<int>();
      for (int64_t k = 0; k < cand.numel() && (int64_t) chosen.size() < need; k++) {
        bool prime = true;
        for (int p : small) {
          if ((int64_t) p * p > cd[k]) break;
          if (cd[k] % p == 0) { prime = false; break; }
        }
        if (prime) chosen.push_back(cd[k]);
      }
    }
    prim_pool_ = torch::from_blob(chosen.data(), {N_LEVELS, n_volumes_, 3}, CpuI32()).clone().to(torch::kCUDA).contiguous();
  }
  if (c.Bool("field.rand_bias")) bias_pool_ = (torch::rand({N_LEVELS * n_volumes_, 3}, DevF32()) * 1000.f + 100.f).contiguous();
  else bias_pool_ = torch::zeros({N_LEVELS * n_volumes_, 3}, DevF32());
  int local_size = pool_size_ / N_LEVELS;
  local_size = (local_size >> 4) << 4;
  feat_local_size_ = torch::full({N_LEVELS}, local_size, DevI32());
  feat_local_idx_ = (torch::arange(N_LEVELS, DevI32()) * local_size).contiguous();
  std::vector<float> scales(N_LEVELS);
  for (int l = 0; l < N_LEVELS; l++)  // Hash3DAnchored.cu:28, evaluated once on the host
    scales[l] = exp2f((RES_FINE_POW_2 - RES_BASE_POW_2) * float(l) / float(N_LEVELS - 1) + RES_BASE_POW_2);
  level_scale_ = torch::from_blob(scales.data(), {N_LEVELS}, CpuF32()).to(torch::kCUDA).contiguous();
  level_scale_host_ = scales;
  march_step_warped_ = c.Has("pts_sampler.sample_l") ? c.Float("pts_sampler.sample_l") : 0.f;
  if (c.Has("pts_sampler.scale_by_dis") && c.Bool("pts_sampler.scale_by_dis")) march_step_warped_ *= 1.25f;  // typical stretch
  balance_gather_ = !(c.Has("field.balance_gather") && !c.Bool("field.balance_gather"));
  // level l addresses halves [l*local, l*local + 2*local): the union is [0, (N_LEVELS+1)*local)
  active_halves_ = std::min<int64_t>(int64_t(N_LEVELS + 1) * local_size, int64_t(pool_size_) * N_CHANNELS);
  mlp_ = std::make_unique<FusedMLP>(gdp, N_LEVELS * N_CHANNELS, mlp_out_dim_, mlp_hidden_dim_, n_hidden_layers_);
  fused_ok_ = mlp_hidden_dim_ == 64 && n_hidden_layers_ == 1;
  SyncHalf();
}

void Hash3DAnchored::SyncHalf() {  // Hash3DAnchored.cu:186 (done once here; afterwards by the optimiser step)
  torch::NoGradGuard g;
  F2N_CALL(f2n_params_to_h16(CurStream(), pool_size_ * N_CHANNELS, F32P(feat_pool_), VoidP(feat_pool_h_)));
  mlp_->SyncHalf();
}

void Hash3DAnchored::ZeroGrad() {
  if (!grad_clean_) grad_h_.zero_();  // the fused optimiser step already re-zeroes the table gradient it consumed
  grad_clean_ = true;
  mlp_->ZeroGrad();
}

Tensor Hash3DAnchored::TableGradUnscaled() { return grad_h_.to(torch::kFloat32) / 128.f; }

namespace {

struct AnchorView {
  Tensor t;
  int stride;
};
AnchorView ViewAnchors(const Tensor& anchors) {
  Tensor a = anchors.contiguous();
  CheckDev(a, torch::kInt32, "anchors");
  if (a.dim() == 2) return {a, (int) a.size(1)};
  return {a, 1};
}

struct FieldFunction : public torch::autograd::Function<FieldFunction> {
  // Samples [0, n_reuse) take their hash features from the pre-pass cache (row src_rows[i]); the rest are gathered.
  static variable_list forward(AutogradContext* ctx, Tensor feat_pool, Tensor mlp_params, Tensor points, Tensor anchors,
                               Tensor src_rows, int64_t n_reuse, int64_t field_ptr) {
    auto* f = reinterpret_cast<Hash3DAnchored*>(field_ptr);
    const int n = points.size(0);
    AnchorView av = ViewAnchors(anchors);
    Tensor feat = torch::empty({n, F2N_MLP_OUT_PAD}, DevF32());
    Tensor saved_x = torch::empty({n, N_LEVELS * N_CHANNELS}, DevF16());
    f->ForwardRaw(points, av.t, av.stride, src_rows, (int) n_reuse, feat, saved_x);
    ctx->saved_data["field"] = field_ptr;
    ctx->save_for_backward({points, av.t, saved_x});
    ctx->saved_data["stride"] = (int64_t) av.stride;
    return {feat};
  }
  static variable_list backward(AutogradContext* ctx, variable_list grad_output) {
    auto* f = reinterpret_cast<Hash3DAnchored*>(ctx->saved_data["field"].toInt());
    auto saved = ctx->get_saved_variables();
    f->BackwardRaw(saved[0], saved[1], (int) ctx->saved_data["stride"].toInt(), saved[2], grad_output[0].contiguous());
    // gradients live in f->grad_h_ (fp16, x128) and f->mlp_->grad_scaled_: consumed by the fused optimiser step
    return {Tensor(), Tensor(), Tensor(), Tensor(), Tensor(), Tensor(), Tensor()};
  }
};

}  // namespace

namespace {

// Hash3DAnchoredFunction of the reference (Hash3DAnchored.cu:160-233) as its own autograd node, for the unfused field path.
struct HashEncodeFunction : public torch::autograd::Function<HashEncodeFunction> {
  static variable_list forward(AutogradContext* ctx, Tensor feat_pool, Tensor points, Tensor anchors, int64_t field_ptr) {
    auto* f = reinterpret_cast<Hash3DAnchored*>(field_ptr);
    AnchorView av = ViewAnchors(anchors);
    const int n = points.size(0);
    Tensor out_h = torch::empty({n, N_LEVELS * N_CHANNELS}, DevF16());
    F2N_TIMED_CALL("hash_fwd", f2n_hash_fwd(CurStream(), n, f->n_volumes_, VoidP(f->feat_pool_h_), I32P(f->prim_pool_), I32P(f->feat_local_idx_),
                          I32P(f->feat_local_size_), F32P(f->bias_pool_), F32P(f->level_scale_), F32P(points), /*warped=*/1,
                          I32P(av.t), av.stride, VoidP(out_h)));
    ctx->saved_data["field"] = field_ptr;
    ctx->saved_data["stride"] = (int64_t) av.stride;
    ctx->save_for_backward({points, av.t});
    return {out_h.to(torch::kFloat32)};
  }
  static variable_list backward(AutogradContext* ctx, variable_list grad_output) {
    auto* f = reinterpret_cast<Hash3DAnchored*>(ctx->saved_data["field"].toInt());
    auto saved = ctx->get_saved_variables();
    const int n = saved[0].size(0);
    Tensor gin = (grad_output[0] * 128.f).to(torch::kFloat16).contiguous();  // Hash3DAnchored.cu:220
    f->grad_clean_ = false;
    F2N_TIMED_CALL("hash_bwd", f2n_hash_bwd(CurStream(), n, f->n_volumes_, I32P(f->prim_pool_), I32P(f->feat_local_idx_), I32P(f->feat_local_size_),
                          F32P(f->bias_pool_), F32P(f->level_scale_), F32P(saved[0]), /*warped=*/1, I32P(saved[1]),
                          (int) ctx->saved_data["stride"].toInt(), VoidP(gin), VoidP(f->grad_h_), f->pool_size_ / N_LEVELS));
    return {Tensor(), Tensor(), Tensor(), Tensor()};  // the table gradient lives in f->grad_h_ (fp16, x128)
  }
};

}  // namespace

Tensor Hash3DAnchored::HashEncode(const Tensor& points, const Tensor& anchors) {
  Tensor pts = points.contiguous();
  CheckDev(pts, torch::kFloat32, "points");
  return HashEncodeFunction::apply(feat_pool_, pts, anchors, reinterpret_cast<int64_t>(this))[0];
}

void Hash3DAnchored::ForwardRaw(const Tensor& points, const Tensor& anchors, int stride, const Tensor& src_rows, int n_reuse,
                                Tensor& feat, Tensor& saved_x, Tensor* f0_cached) {
  const int n = points.size(0);
  const int n_cached = n_reuse, n_tail = n - n_cached;
  if (n_cached > 0) {
    TORCH_CHECK(prepass_x_.defined() && src_rows.numel() >= n_cached, "no pre-pass feature cache for this query");
    F2N_TIMED_CALL("field_fwd_cached", f2n_field_fwd_cached(CurStream(), n_cached, (int) prepass_x_.size(0), I32P(src_rows),
                           VoidP(prepass_x_), VoidP(mlp_->params_h_), F32P(feat), f0_cached != nullptr ? F32P(*f0_cached) : nullptr,
                           VoidP(saved_x)));
  }
  if (n_tail > 0)
    F2N_TIMED_CALL("field_fwd", f2n_field_fwd(CurStream(), n_tail, n_volumes_, VoidP(feat_pool_h_), I32P(prim_pool_),
                           I32P(feat_local_idx_), I32P(feat_local_size_), F32P(bias_pool_), F32P(level_scale_),
                           F32P(points) + 3 * (int64_t) n_cached, I32P(anchors) + (int64_t) stride * n_cached, stride,
                           VoidP(mlp_->params_h_), F32P(feat) + (int64_t) F2N_MLP_OUT_PAD * n_cached, nullptr,
                           static_cast<void*>(saved_x.data_ptr<at::Half>() + (int64_t) N_LEVELS * N_CHANNELS * n_cached)));
}

void Hash3DAnchored::BackwardRaw(const Tensor& points, const Tensor& anchors, int stride, const Tensor& saved_x, const Tensor& dfeat) {
  const int n = points.size(0);
  grad_clean_ = false;
  F2N_TIMED_CALL("field_bwd", f2n_field_bwd(CurStream(), n, n_volumes_, I32P(prim_pool_), I32P(feat_local_idx_),
                         I32P(feat_local_size_), F32P(bias_pool_), F32P(level_scale_), F32P(points), I32P(anchors), stride,
                         VoidP(mlp_->params_h_), VoidP(saved_x), F32P(dfeat), mlp_->loss_scale_, F32P(mlp_->grad_scaled_),
                         VoidP(grad_h_), pool_size_ / N_LEVELS));
}

Tensor Hash3DAnchored::AnchoredQuery(const Tensor& points, const Tensor& anchors) {  // Hash3DAnchored.cpp:84-99
  Tensor pts = points.contiguous();
  CheckDev(pts, torch::kFloat32, "points");
  if (!fused_ok_) return mlp_->Query(HashEncode(pts, anchors));  // hash -> fp32 -> tcnn MLP, as the reference spells it (:91-98)
  Tensor feat = FieldFunction::apply(feat_pool_, mlp_->params_, pts, anchors, torch::empty({0}, DevI32()), (int64_t) 0,
                                     reinterpret_cast<int64_t>(this))[0];
  return mlp_out_dim_ == F2N_MLP_OUT_PAD ? feat : feat.index({Slc(), Slc(0, mlp_out_dim_)}).contiguous();
}

Tensor Hash3DAnchored::AnchoredQueryReuse(const Tensor& points, const Tensor& anchors, const Tensor& src_rows, int n_reuse) {
  Tensor pts = points.contiguous();
  CheckDev(pts, torch::kFloat32, "points");
  Tensor rows = src_rows.contiguous();
  CheckDev(rows, torch::kInt32, "src_rows");
  TORCH_CHECK(n_reuse >= 0 && n_reuse <= pts.size(0) && rows.numel() >= n_reuse, "bad reuse range");
  if (!fused_ok_) {  // (no feature cache on the unfused path)
    prepass_x_ = Tensor();
    return AnchoredQuery(pts, anchors);
  }
  Tensor feat = FieldFunction::apply(feat_pool_, mlp_->params_, pts, anchors, rows.slice(0, 0, n_reuse), (int64_t) n_reuse,
                                     reinterpret_cast<int64_t>(this))[0];
  prepass_x_ = Tensor();  // one consumer per pre-pass; the table may change after this step
  return mlp_out_dim_ == F2N_MLP_OUT_PAD ? feat : feat.index({Slc(), Slc(0, mlp_out_dim_)}).contiguous();
}

Tensor Hash3DAnchored::QueryDensityPreAct(const Tensor& points, const Tensor& anchors, bool keep_features) {
  torch::NoGradGuard g;
  Tensor pts = points.contiguous();
  CheckDev(pts, torch::kFloat32, "points");
  AnchorView av = ViewAnchors(anchors);
  const int n = pts.size(0);
  if (!fused_ok_) {
    prepass_x_ = Tensor();
    return mlp_->Query(HashEncode(pts, anchors)).select(1, 0).contiguous();
  }
  Tensor f0 = torch::empty({n}, DevF32());
  prepass_x_ = keep_features ? torch::empty({n, N_LEVELS * N_CHANNELS}, DevF16()) : Tensor();
#if F2N_DEBUG_BUILD
  static const bool force_fused = []() {  // measurement knob: the one-kernel gather -> MLP at every size (profiles/r04_fused_gather_ab.txt)
    const char* e = std::getenv("F2N_FUSED_GATHER");
    return e != nullptr && e[0] == '1';
  }();
#else
  constexpr bool force_fused = false;  // (measured to lose at every training size: profiles/r04_fused_gather_ab.txt)
#endif
  if (force_fused) {
    F2N_TIMED_CALL("field_prepass_fused", f2n_field_fwd_fused(CurStream(), n, n_volumes_, VoidP(feat_pool_h_), I32P(prim_pool_),
                           I32P(feat_local_idx_), I32P(feat_local_size_), F32P(bias_pool_), F32P(level_scale_), F32P(pts), I32P(av.t),
                           av.stride, VoidP(mlp_->params_h_), nullptr, F32P(f0), keep_features ? VoidP(prepass_x_) : nullptr));
  } else if (n >= 32768) {  // the two kernels of the large-batch path, issued (and timed) separately
    Tensor planes = torch::empty({8, n, 4}, DevF16());
    // consecutive samples of a ray are one march step apart: sample_l * fineness in warped space (PersSampler.cu:262-270,
    // stretched by the distance scaling where it is on), half of that in the [0,1] space the grid hashes (:91)
    const float step01 = march_step_warped_ * global_data_pool_->ray_march_fineness_ * .5f;
    // Tables that have left the L2s (2^21 entries per level and more: BASELINE config 5) take the slice-binned gather for the level
    // pairs whose working set exceeds an L2: from level pair 1 on, for tables of 2^20 entries per level and more (round 5: at 2^20,
    // wanjinyou_big.yaml's own size, the binned pipeline takes the step from 1.37 to 1.31 ms -- profiles/r05_big20_binned_ab.txt)
#if F2N_DEBUG_BUILD  // (measurement knobs of the debug variant: first binned pair, 8 = off; smallest table that takes the binned path)
    static const int binned_p0 = []() {
      const char* e = std::getenv("F2N_BINNED_GATHER_P0");
      const int v = e != nullptr ? std::atoi(e) : 1;
      return v < 0 ? 0 : (v > 8 ? 8 : v);
    }();
    static const int binned_min_log2 = []() {
      const char* e = std::getenv("F2N_BINNED_GATHER_MIN_LOG2");
      return e != nullptr ? std::atoi(e) : 20;
    }();
#else
    constexpr int binned_p0 = 1, binned_min_log2 = 20;
#endif
    const int level_entries = (int) (pool_size_ / N_LEVELS);
    if (binned_p0 < 8 && level_entries >= (1 << binned_min_log2) && level_entries <= (1 << 22) && n >= 65536 && n <= 1536 * 1024) {
      F2N_TIMED_CALL("hash_gather", f2n_hash_gather_planes_binned(CurStream(), n, n_volumes_, VoidP(feat_pool_h_), I32P(prim_pool_),
                             I32P(feat_local_idx_), I32P(feat_local_size_), F32P(bias_pool_), F32P(level_scale_), F32P(pts), 1,
                             I32P(av.t), av.stride, VoidP(planes), level_entries, binned_p0));
    } else {
      F2N_TIMED_CALL("hash_gather", f2n_hash_gather_planes_balanced(CurStream(), n, n_volumes_, VoidP(feat_pool_h_), I32P(prim_pool_),
                             I32P(feat_local_idx_), I32P(feat_local_size_), F32P(bias_pool_), F32P(level_scale_), F32P(pts), 1,
                             I32P(av.t), av.stride, VoidP(planes), balance_gather_ ? step01 : 0.f, level_scale_host_.data()));
    }
    F2N_TIMED_CALL("field_mlp_prepass", f2n_field_mlp_planes(CurStream(), n, VoidP(planes), VoidP(mlp_->params_h_), nullptr, F32P(f0),
                           keep_features ? VoidP(prepass_x_) : nullptr));
  } else {
    F2N_TIMED_CALL("field_prepass", f2n_field_fwd(CurStream(), n, n_volumes_, VoidP(feat_pool_h_), I32P(prim_pool_), I32P(feat_local_idx_),
                           I32P(feat_local_size_), F32P(bias_pool_), F32P(level_scale_), F32P(pts), I32P(av.t), av.stride,
                           VoidP(mlp_->params_h_), nullptr, F32P(f0), keep_features ? VoidP(prepass_x_) : nullptr));
  }
  return f0;
}

int Hash3DAnchored::LoadStates(const std::vector<Tensor>& states, int idx) {  // Hash3DAnchored.cpp:101-110
  torch::NoGradGuard g;
  feat_pool_.copy_(states[idx++].to(torch::kCUDA).to(torch::kFloat32).reshape({pool_size_, N_CHANNELS}));
  prim_pool_ = states[idx++].clone().to(torch::kCUDA).to(torch::kInt32).contiguous();
  bias_pool_ = states[idx++].clone().to(torch::kCUDA).to(torch::kFloat32).contiguous();
  n_volumes_ = states[idx++].item<int>();
  TORCH_CHECK(prim_pool_.numel() == int64_t(N_LEVELS) * n_volumes_ * 3 && bias_pool_.numel() == int64_t(N_LEVELS) * n_volumes_ * 3,
              "prime/bias pools do not match n_volumes");
  mlp_->params_.copy_(states[idx++].to(torch::kCUDA).to(torch::kFloat32));
  SyncHalf();
  return idx;
}

std::vector<Tensor> Hash3DAnchored::States() {  // Hash3DAnchored.cpp:112-122
  return {feat_pool_.detach(), prim_pool_, bias_pool_, torch::full({1}, n_volumes_, CpuI32()), mlp_->params_.detach()};
}

std::vector<ParamGroup> Hash3DAnchored::OptimParamGroups() {  // Hash3DAnchored.cpp:124-150
  ParamGroup table;
  table.name = "feat_pool";
  table.param = feat_pool_;
  table.grad = grad_h_;
  table.param_h = feat_pool_h_;
  table.grad_scale = 1.f / 128.f;
  table.grad_is_h16 = true;
  table.active = active_halves_;
  ParamGroup mlp;
  mlp.name = "field_mlp";
  mlp.param = mlp_->params_;
  mlp.grad = mlp_->grad_scaled_;
  mlp.param_h = mlp_->params_h_;
  mlp.weight_decay = 1e-6f;
  mlp.grad_round_h16 = true;
  mlp.grad_scale = -1.f;  // resolved at step time from mlp_->loss_scale_ (it can halve on NaN)
  return {table, mlp};
}

void Hash3DAnchored::Reset() {  // Hash3DAnchored.cpp:152-155
  torch::NoGradGuard g;
  feat_pool_.uniform_(-1e-2f, 1e-2f);
  mlp_->InitParams();
  SyncHalf();
}

std::unique_ptr<Field> ConstructField(GlobalDataPool* gdp) {  // FieldFactory.cpp:7-16
  const std::string type = gdp->config_.Str("field.type");
  TORCH_CHECK(type == "Hash3DAnchored", "unknown field.type: ", type);
  return std::make_unique<Hash3DAnchored>(gdp);
}

}  // namespace f2n
