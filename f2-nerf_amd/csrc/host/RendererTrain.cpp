// Renderer::TrainForwardBackward: one training iteration's forward, loss and backward without the autograd tape (what
// ExpRunner::TrainStep runs; Renderer::Render in Renderer.cpp is the taped plugin entry point of the reference and its cross-check).
// Split out of Renderer.cpp in round 5.
#include "Renderer.h"

#include <hip/hip_runtime_api.h>

#include <cstdio>
#include <cstdlib>
#include <mutex>

namespace f2n {

using torch::autograd::AutogradContext;
using torch::autograd::variable_list;

// One training iteration's forward AND backward without the autograd tape: the same kernels as Render() + the loss
// of ExpRunner::Train (ExpRunner.cpp:95-120) + the backward chain, issued back to back.  Every gradient buffer of the
// chain is written exactly once by the kernel that owns it (composite_bwd: dfeat[:,0] and drgb; shade_bwd:
// dfeat[:,1:16]; the loss kernel: the edge rows of dfeat), so there is no zero-fill, no gradient accumulation pass and
// no slice/cat copy: ~100 ATen launches and ~0.5 GB of HBM traffic per step less than the taped version.
TrainOutputs Renderer::TrainForwardBackward(const Tensor& rays_o, const Tensor& rays_d, const Tensor& bounds,
                                            const Tensor& gt_colors, const Tensor& emb_idx, float var_w, float disp_w, float tv_w) {
  auto* gdp = global_data_pool_;
  auto* field = static_cast<Hash3DAnchored*>(scene_field_.get());
  auto* shader = static_cast<SHShader*>(shader_.get());
  torch::NoGradGuard no_grad;
  const int n_rays = rays_o.size(0);
  Tensor gt = gt_colors.contiguous();
  CheckDev(gt, torch::kFloat32, "gt_colors");
  TORCH_CHECK(gt.numel() == (int64_t) n_rays * 3, "gt_colors must be [n_rays,3]");
  RenderFront fr = SampleAndFilter(rays_o, rays_d, bounds, emb_idx, async_count_);
  void* st = CurStream();
  TrainOutputs out;
  out.losses = torch::empty({8}, DevF32());
  if (fr.empty) {  // no samples at all: the colour is the background, nothing depends on the parameters (Renderer.cpp:83-97)
    Tensor bg = fr.bg_color.contiguous();
    F2N_CALL(f2n_train_loss(st, n_rays, F32P(bg), F32P(gt), nullptr, nullptr, 0, 0, nullptr, 0.f, 0.f, 0.f, F32P(out.losses),
                            nullptr, nullptr, nullptr, nullptr));
    return out;
  }
  SampleResultFlex& es = fr.es;
  const int n_kept = fr.n_kept, n_edge = fr.n_edge, n = n_kept + 2 * n_edge;
  TORCH_CHECK(FusedPathOk(), "the untaped training step needs the shipped network shapes (ExpRunner::TrainStep takes the taped path otherwise)");

  // Row layout of the field's arrays: [survivors | edge samples], or -- when the survivor count is still on the device
  // (fr.dyn: n_kept is then the capacity) -- [edge samples | survivors] so that every offset is known on the host.
  const int64_t so = fr.dyn ? 2 * (int64_t) n_edge : 0, eo = fr.dyn ? 0 : n_kept;
  const int32_t* n_dev = fr.dyn ? I32P(fr.n_kept_dev) : nullptr;
  // ---- forward ----
  Tensor feat = torch::empty({fr.dyn ? std::max(2 * n_edge, 1) : n, F2N_MLP_OUT_PAD}, DevF32());
  Tensor field_x = torch::empty({n, N_LEVELS * N_CHANNELS}, DevF16());
  // the density pre-activations of the surviving samples also leave as a compact array: compositing then reads 4 B per
  // sample instead of one 64-byte line of `feat` per sample, and its backward writes a compact d f0 that the colour
  // backward merges into the dfeat rows it writes anyway (column 0 written in place was a read-modify-write of every line)
  Tensor f0c = torch::empty({std::max(n_kept, 1)}, DevF32()), df0c = torch::empty({std::max(n_kept, 1)}, DevF32());
  Tensor rgb = torch::empty({std::max(n_kept, 1), 3}, DevF32()), shade_x = torch::empty({std::max(n_kept, 1), 32}, DevF16());
  Tensor app = fr.emb ? app_emb_ : Tensor();
  if (fr.dyn) {
    // streaming step: field MLP (cached hash features) and colour path of the survivors in one launch -- their `feat` rows
    // are never written (only the 2E edge rows of `feat` exist: the TV loss reads them); the synchronous path below keeps
    // the two separate kernels and is what tests compare this with
    TORCH_CHECK(field->prepass_x_.defined(), "no pre-pass feature cache for this query");
    const bool edges_ride = n_edge > 0 && fr.edge_cache_row >= 0;  // their hash features are in the pre-pass cache too
    if (n_edge > 0 && !edges_ride)
      F2N_TIMED_CALL("field_fwd", f2n_field_fwd(st, 2 * n_edge, field->n_volumes_, VoidP(field->feat_pool_h_), I32P(field->prim_pool_),
                             I32P(field->feat_local_idx_), I32P(field->feat_local_size_), F32P(field->bias_pool_),
                             F32P(field->level_scale_), F32P(fr.pts_all), I32P(fr.vol_all), 1, VoidP(field->mlp_->params_h_),
                             F32P(feat), nullptr, VoidP(field_x)));
    const at::Half* cache = field->prepass_x_.data_ptr<at::Half>();
    const int64_t row = (int64_t) N_LEVELS * N_CHANNELS;
    // survivors: field MLP -> colour path; edge samples (when cached): field MLP only, fp32 rows for the TV loss -- one launch
    F2N_TIMED_CALL("field_shade_fwd", f2n_field_shade_fwd_extra(st, n_kept, n_dev, I32P(fr.src_rows),
                           static_cast<const void*>(cache + row * fr.sample_cache_row), VoidP(field->mlp_->params_h_), F32P(es.dirs),
                           fr.emb ? F32P(app) : nullptr, fr.emb ? I32P(fr.sample_emb_idx) : nullptr, VoidP(shader->mlp_->params_h_),
                           F32P(f0c), static_cast<void*>(field_x.data_ptr<at::Half>() + row * so), VoidP(shade_x), F32P(rgb),
                           edges_ride ? 2 * n_edge : 0, edges_ride ? static_cast<const void*>(cache + row * fr.edge_cache_row) : nullptr,
                           edges_ride ? F32P(feat) : nullptr, edges_ride ? VoidP(field_x) : nullptr));
    field->prepass_x_ = Tensor();
  } else {
    field->ForwardRaw(fr.pts_all, fr.vol_all, 1, fr.src_rows, n_kept, feat, field_x, &f0c);
    field->prepass_x_ = Tensor();
    F2N_TIMED_CALL("shade_fwd", f2n_shade_fwd(st, n_kept, F32P(feat), F32P(es.dirs), fr.emb ? F32P(app) : nullptr,
                           fr.emb ? I32P(fr.sample_emb_idx) : nullptr, VoidP(shader->mlp_->params_h_), F32P(rgb), VoidP(shade_x)));
  }
  if (before_backward_) before_backward_();
  Tensor colors = torch::empty({n_rays, 3}, DevF32());
  Tensor weights = torch::empty({std::max(n_kept, 1)}, DevF32());
  Tensor bg = fr.bg_color.contiguous();
  Tensor dfeat = torch::empty({n, F2N_MLP_OUT_PAD}, DevF32());
  Tensor drgb = torch::empty({std::max(n_kept, 1), 3}, DevF32());  // (WeightVarLoss backward rides inside the compositing backward)
  if (fr.dyn) {
    // ---- compositing forward, loss, compositing backward: one launch (f2n_composite_train); the TV gradient goes straight
    // into the edge rows of dfeat, the loss values are completed by the step's deferred reduction below ----
    F2N_TIMED_CALL("composite_train", f2n_composite_train(st, n_rays, I32P(es.pts_idx_bounds), F32P(f0c), 1, F32P(es.dt), F32P(es.t), F32P(rgb),
                                 F32P(bg), F32P(gt), var_w, disp_w, tv_w, gdp->gradient_scaling_progress_, n_edge, F2N_MLP_OUT_PAD,
                                 n_edge > 0 ? F32P(feat) + F2N_MLP_OUT_PAD * eo : nullptr, n_edge > 0 ? F32P(dfeat) + F2N_MLP_OUT_PAD * eo : nullptr,
                                 F32P(colors), F32P(weights), F32P(drgb), F32P(df0c), 1, F32P(out.losses), /*defer_reduce=*/1));
  } else {
    Tensor disparity = torch::empty({n_rays}, DevF32()), depth = torch::empty({n_rays}, DevF32());
    Tensor var = torch::empty({n_rays}, DevF32());
    F2N_TIMED_CALL("composite_fwd", f2n_composite_fwd(st, n_rays, I32P(es.pts_idx_bounds), F32P(f0c), 1, F32P(es.dt), F32P(es.t), F32P(rgb),
                               F32P(bg), F32P(colors), F32P(disparity), F32P(depth), F32P(weights), F32P(var)));  // (+ WeightVarLoss fwd)

    // ---- loss and its gradients; the TV gradient goes straight into the edge rows of dfeat ----
    Tensor dcolors = torch::empty({n_rays, 3}, DevF32()), ddisp = torch::empty({n_rays}, DevF32()), dvar = torch::empty({n_rays}, DevF32());
    F2N_TIMED_CALL("train_loss", f2n_train_loss(st, n_rays, F32P(colors), F32P(gt), F32P(disparity), F32P(var), n_edge, F2N_MLP_OUT_PAD,
                            F32P(feat) + F2N_MLP_OUT_PAD * eo, var_w, disp_w, tv_w, F32P(out.losses), F32P(dcolors),
                            F32P(ddisp), F32P(dvar), F32P(dfeat) + F2N_MLP_OUT_PAD * eo));

    // ---- backward ----
    F2N_TIMED_CALL("composite_bwd", f2n_composite_bwd(st, n_rays, I32P(es.pts_idx_bounds), F32P(f0c), 1, F32P(es.dt), F32P(es.t), F32P(rgb), F32P(bg),
                               F32P(dcolors), F32P(ddisp), nullptr, nullptr, gdp->gradient_scaling_progress_, F32P(drgb),
                               F32P(df0c), 1, F32P(weights), F32P(dvar)));
  }
  if (digest_taps_) DigestTap(TAP_GRAD_BEFORE, field->grad_h_);  // (must be all zeros: Adam's zero_grad / ZeroGrad)
  F2N_TIMED_CALL("shade_bwd", f2n_shade_bwd_dyn(st, n_kept, n_dev, F32P(drgb), fr.emb ? I32P(fr.sample_emb_idx) : nullptr,
                         VoidP(shader->mlp_->params_h_), VoidP(shade_x), shader->mlp_->loss_scale_, F32P(dfeat) + F2N_MLP_OUT_PAD * so,
                         F32P(shader->mlp_->grad_scaled_), fr.emb ? F32P(app_emb_grad_) : nullptr,
                         fr.emb ? (int) app_emb_grad_.size(0) : 0, F32P(df0c), /*defer_reduce=*/fr.dyn ? 1 : 0));
  if (digest_taps_ && fr.dyn) {  // (the scatter's point / volume rows in FRONT of it: compared with the taps behind it, in the same run)
    const int64_t rows = fr.pts_all.size(0);
    Tensor live = torch::arange(rows, DevI32()).lt(fr.n_kept_dev + (int) so).to(torch::kInt32);
    DigestTap(TAP_PTS_ALL_PRE, fr.pts_all.view(torch::kInt32).sum(1, false, torch::kInt64) * live);
    DigestTap(TAP_VOL_ALL_PRE, fr.vol_all.to(torch::kInt64) * live);
  }
  F2nStepTail tail;
  const bool fused_tail = fr.dyn && step_tail_builder_ && !digest_taps_ && step_tail_builder_(&tail);
  if (fused_tail) {
    // the field backward and the REST of the step in one call: the three deferred reductions, the finiteness flags and the small
    // groups' Adam on the tail stream beside the scatter's producers, the table's Adam inside the scatter's owners
    field->grad_clean_ = false;
    F2N_TIMED_CALL("field_bwd", f2n_field_bwd_step_tail(st, TailStream()->stream(), n, n_dev, 2 * n_edge, field->n_volumes_, I32P(field->prim_pool_),
                           I32P(field->feat_local_idx_), I32P(field->feat_local_size_), F32P(field->bias_pool_),
                           F32P(field->level_scale_), F32P(fr.pts_all), I32P(fr.vol_all), 1, VoidP(field->mlp_->params_h_),
                           VoidP(field_x), F32P(dfeat), field->mlp_->loss_scale_, F32P(field->mlp_->grad_scaled_),
                           VoidP(field->grad_h_), field->pool_size_ / N_LEVELS, &tail, nullptr));
    step_tail_done_ = true;
  } else if (fr.dyn) {
    field->grad_clean_ = false;
    F2N_TIMED_CALL("field_bwd", f2n_field_bwd_dyn(st, n, n_dev, 2 * n_edge, field->n_volumes_, I32P(field->prim_pool_),
                           I32P(field->feat_local_idx_), I32P(field->feat_local_size_), F32P(field->bias_pool_),
                           F32P(field->level_scale_), F32P(fr.pts_all), I32P(fr.vol_all), 1, VoidP(field->mlp_->params_h_),
                           VoidP(field_x), F32P(dfeat), field->mlp_->loss_scale_, F32P(field->mlp_->grad_scaled_),
                           VoidP(field->grad_h_), field->pool_size_ / N_LEVELS, /*defer_reduce=*/1));
    // the three partial-sum reductions (colour-MLP weights, appearance embedding, field-MLP weights) in one launch
    F2N_TIMED_CALL("reduce_partials", f2n_reduce_deferred(st));
  } else {
    field->BackwardRaw(fr.pts_all, fr.vol_all, 1, field_x, dfeat);
  }
  if (digest_taps_ && fr.dyn) {
    // what the scatter has just read, as it stands AFTER the scatter: rows [0, 2E + survivors) of pts_all / vol_all and of the
    // MLP backward's inputs (rows beyond the device-side count are never written: masked out)
    const int64_t rows = fr.pts_all.size(0);
    Tensor live = torch::arange(rows, DevI32()).lt(fr.n_kept_dev + (int) so).to(torch::kInt32);
    DigestTap(TAP_PTS_ALL_AFTER, fr.pts_all.view(torch::kInt32).sum(1, false, torch::kInt64) * live);
    DigestTap(TAP_VOL_ALL_AFTER, fr.vol_all.to(torch::kInt64) * live);
    Tensor live_n = live.narrow(0, 0, n);
    DigestTap(TAP_FIELD_X, field_x.view(torch::kInt32).sum(1, false, torch::kInt64) * live_n);
    DigestTap(TAP_DFEAT, dfeat.view(torch::kInt32).sum(1, false, torch::kInt64) * live_n);
  }
  if (digest_taps_) {
    DigestTap(TAP_COLORS, colors);
    DigestTap(TAP_TABLE_GRAD, field->grad_h_);
    DigestTap(TAP_SMALL_GRADS, small_grads_flat_);
  }
  if ((fr.side_pool_buffers || fr.consumed_deferred) && side_shared_) {  // (see RenderFront: their last readers have been queued only now)
    side_shared_->consumed.record();
    side_shared_->seq++;
  }
  out.colors = colors;
  out.has_samples = true;
  return out;
}

}  // namespace f2n
