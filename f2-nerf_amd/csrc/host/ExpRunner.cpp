// ExpRunner: the training / rendering driver of the hot path (mirrors src/ExpRunner.h/.cpp of the reference:
// loss terms, NaN-skip, LR / fineness / gradient-scaling schedules, checkpoint state order).  Dataset handling
// (image loading, ray generation) is outside the hot path: rays and ground-truth colours are inputs here.
// The optimiser is the fused Adam of csrc/optim.hip driven over the modules' parameter groups; multi-GPU data
// parallelism hooks in through `sync_` (GradSyncPipeline.h: in front of the optimiser, or pipelined under the next step's sampling).
#include "ExpRunner.h"

#include <deque>

namespace f2n {

ExpRunner::ExpRunner(const std::map<std::string, std::string>& flat_config, int n_images) {
  global_data_pool_ = std::make_unique<GlobalDataPool>();
  global_data_pool_->config_.kv = flat_config;
  const auto& c = global_data_pool_->config_;
  pts_batch_size_ = c.Int("train.pts_batch_size");  // ExpRunner.cpp:28-45
  end_iter_ = c.Int("train.end_iter");
  learning_rate_ = c.Float("train.learning_rate");
  learning_rate_alpha_ = c.Float("train.learning_rate_alpha");
  learning_rate_warm_up_end_iter_ = c.Int("train.learning_rate_warm_up_end_iter");
  ray_march_init_fineness_ = c.Float("train.ray_march_init_fineness");
  ray_march_fineness_decay_end_iter_ = c.Int("train.ray_march_fineness_decay_end_iter");
  tv_loss_weight_ = c.Float("train.tv_loss_weight");
  disp_loss_weight_ = c.Float("train.disp_loss_weight");
  var_loss_weight_ = c.Float("train.var_loss_weight");
  var_loss_start_ = c.Int("train.var_loss_start");
  var_loss_end_ = c.Int("train.var_loss_end");
  gradient_scaling_start_ = c.Int("train.gradient_scaling_start");
  gradient_scaling_end_ = c.Int("train.gradient_scaling_end");
  global_data_pool_->n_volumes_ = c.Has("runtime.n_volumes") ? c.Int("runtime.n_volumes") : 1;
  renderer_ = std::make_unique<Renderer>(global_data_pool_.get(), n_images);
  BuildOptimizer();
  FlattenSmallGrads();
  UpdateAdaParams();
  ema_base_value_ = global_data_pool_->meaningful_sampled_pts_per_ray_;
  sync_.apply = [this](bool apply_optimizer, float lr) {  // (a pending step is applied with ITS learning rate)
    const float lr_now = cur_lr_;
    cur_lr_ = lr;
    EnqueueApply(apply_optimizer);
    cur_lr_ = lr_now;
  };
  // The host does not wait for the flags of a pipelined step (it would idle the device between this Adam and the next step's
  // first kernels): they travel to pinned memory and are read after the next sample-count read-back, or by FinishPending().
  sync_.defer_flags = [this]() {
    if (check_nan_) DeferFlags(true);
  };
}

void ExpRunner::BuildOptimizer() {
  groups_ = renderer_->OptimParamGroups();
  exp_avg_.clear();
  exp_avg_sq_.clear();
  for (auto& g : groups_) {
    exp_avg_.push_back(torch::zeros_like(g.param.detach()));
    exp_avg_sq_.push_back(torch::zeros_like(g.param.detach()));
  }
  optim_steps_ = 0;
}

// Re-homes the three small fp32 gradient buffers (field MLP, colour MLP, app_emb) in ONE flat tensor, so that a
// data-parallel run reduces them with a single collective instead of three latency-bound ones.  Returns the flat tensor.
Tensor ExpRunner::FlattenSmallGrads() {
  if (renderer_->small_grads_flat_.defined()) return renderer_->small_grads_flat_;
  auto* field = static_cast<Hash3DAnchored*>(renderer_->scene_field_.get());
  auto* shader = static_cast<SHShader*>(renderer_->shader_.get());
  const int64_t n1 = field->mlp_->grad_scaled_.numel(), n2 = shader->mlp_->grad_scaled_.numel(),
                n3 = renderer_->app_emb_grad_.numel();
  Tensor flat = torch::zeros({n1 + n2 + n3}, DevF32());
  field->mlp_->grad_scaled_ = flat.narrow(0, 0, n1);
  shader->mlp_->grad_scaled_ = flat.narrow(0, n1, n2);
  renderer_->app_emb_grad_ = flat.narrow(0, n1 + n2, n3).view(renderer_->app_emb_grad_.sizes());
  for (auto& g : groups_) {  // the optimiser groups hold handles to the old buffers
    if (g.name == "field_mlp") g.grad = field->mlp_->grad_scaled_;
    if (g.name == "color_mlp") g.grad = shader->mlp_->grad_scaled_;
    if (g.name == "app_emb") g.grad = renderer_->app_emb_grad_;
  }
  renderer_->small_grads_flat_ = flat;
  return flat;
}

void ExpRunner::LoadStates(const std::vector<Tensor>& states) {
  int used = renderer_->LoadStates(states, 0);
  TORCH_CHECK(used == (int) states.size(), "state vector has ", states.size(), " tensors, consumed ", used);
  BuildOptimizer();  // parameter tensors may have been re-created (primes, nodes): re-bind the groups
}

std::vector<Tensor> ExpRunner::AuxStates() {
  auto& oct = *static_cast<PersSampler*>(renderer_->pts_sampler_.get())->pers_octree_;
  auto or_empty = [](const Tensor& t, torch::TensorOptions o) { return t.defined() ? t : torch::empty({0}, o); };
  return {or_empty(oct.edge_pool_gpu_, DevU8()), or_empty(oct.w2c_, DevF32()), or_empty(oct.intri_, DevF32()), or_empty(oct.bound_, DevF32())};
}

void ExpRunner::LoadAuxStates(const std::vector<Tensor>& aux) {
  TORCH_CHECK(aux.size() == 4, "aux states: edge pool, w2c, intri, bounds");
  auto* ps = static_cast<PersSampler*>(renderer_->pts_sampler_.get());
  if (aux[0].numel() > 0) ps->SetEdgePool(aux[0].reshape({-1}));
  if (aux[1].numel() > 0) ps->SetTrainCameras(aux[1].reshape({-1, 3, 4}), aux[2].reshape({-1, 3, 3}), aux[3].reshape({-1, 2}));
}

// ExpRunner.cpp:108-114 (variance-loss ramp) and :221-254 (UpdateAdaParams), spelled as the reference spells them
ExpRunner::ScheduleValues ExpRunner::ScheduleAt(const ScheduleParams& p, int iter) {
  ScheduleValues v;
  if (iter >= p.ray_march_fineness_decay_end_iter) {
    v.fineness = 1.f;
  } else {
    float progress = float(iter) / float(p.ray_march_fineness_decay_end_iter);
    v.fineness = std::exp(std::log(1.f) * progress + std::log(p.ray_march_init_fineness) * (1.f - progress));
  }
  float lr_factor;
  if (iter >= p.learning_rate_warm_up_end_iter) {
    float progress = float(iter - p.learning_rate_warm_up_end_iter) / float(p.end_iter - p.learning_rate_warm_up_end_iter);
    lr_factor = (1.f - p.learning_rate_alpha) * (std::cos(progress * float(M_PI)) * .5f + .5f) + p.learning_rate_alpha;
  } else {
    lr_factor = float(iter) / float(p.learning_rate_warm_up_end_iter);
  }
  v.lr = p.learning_rate * lr_factor;
  float progress = 1.f;
  if (iter < p.gradient_scaling_end) {
    progress = std::max(0.f, (float(iter) - p.gradient_scaling_start) / (p.gradient_scaling_end - p.gradient_scaling_start + 1e-9f));
  }
  v.gradient_scaling_progress = progress;
  v.var_loss_weight = 0.f;
  if (iter > p.var_loss_end) v.var_loss_weight = p.var_loss_weight;
  else if (iter > p.var_loss_start) v.var_loss_weight = float(iter - p.var_loss_start) / float(p.var_loss_end - p.var_loss_start) * p.var_loss_weight;
  return v;
}

ExpRunner::ScheduleParams ExpRunner::Schedule() const {
  return {ray_march_init_fineness_, ray_march_fineness_decay_end_iter_, learning_rate_, learning_rate_alpha_,
          learning_rate_warm_up_end_iter_, end_iter_, gradient_scaling_start_, gradient_scaling_end_, var_loss_weight_,
          var_loss_start_, var_loss_end_};
}

float ExpRunner::FinenessAt(int iter) const { return ScheduleAt(Schedule(), iter).fineness; }

void ExpRunner::UpdateAdaParams() {
  auto* gdp = global_data_pool_.get();
  const ScheduleValues v = ScheduleAt(Schedule(), iter_step_);
  gdp->ray_march_fineness_ = v.fineness;
  cur_lr_ = v.lr;
  gdp->gradient_scaling_progress_ = v.gradient_scaling_progress;
  gdp->iter_step_ = iter_step_;
}

int ExpRunner::CurBatchSize() const {  // ExpRunner.cpp:86
  return int(pts_batch_size_ / global_data_pool_->meaningful_sampled_pts_per_ray_) >> 4 << 4;
}

// The ray count of the batch with sequence number `seq` (ExpRunner.cpp:86 at a FIXED lag): pts_batch_size over the
// meaningful-samples average as it stood after step seq - kBatchSizeLag.  The reference sizes a batch from the average of the
// step before it; a streaming step learns its count one step late (two in a data-parallel run) and Train() draws two batches
// ahead, so the lag a schedule can honour in every mode is 5 -- and using that one lag in every mode (synchronous steps included)
// is what makes the sequence of batch sizes independent of the sampling schedule and of how Train() is chunked.
int ExpRunner::BatchSizeFor(int64_t seq) {
  const int64_t want = seq - kBatchSizeLag;
  float ema = ema_base_value_;
  if (want >= ema_base_seq_) {
    // (a streaming step's count is resolved at the top of the next step; a Train() call that starts right behind one asks earlier)
    if (renderer_->count_pending_ && renderer_->pending_count_seq_ <= want + 1) renderer_->ResolvePendingCount();
    // (a step that recorded nothing -- the first step of a data-parallel run -- falls back on the step before it; the ring holds the
    // last Renderer::kEmaRing steps, and the draw that asks is at most a handful of steps ahead of the newest record)
    int64_t s = want;
    const int64_t floor_seq = std::max(ema_base_seq_, want - Renderer::kEmaRing);
    while (s >= floor_seq && !renderer_->EmaAfter(s, &ema)) s--;
    if (s < floor_seq) ema = s < ema_base_seq_ ? ema_base_value_ : global_data_pool_->meaningful_sampled_pts_per_ray_;
  }
  return int(pts_batch_size_ / ema) >> 4 << 4;
}

// The sequence numbers restart at `seq` (checkpoint load: seq = the iteration, so that a resumed run draws what the uninterrupted
// run would have drawn at that iteration): batches up to seq + kBatchSizeLag are sized from the average as it stands now.
void ExpRunner::ResetStepSequence(int64_t seq) {
  FinishPending();
  renderer_->DropPendingSamples();
  step_seq_ = seq;
  ema_base_seq_ = seq;
  ema_base_value_ = global_data_pool_->meaningful_sampled_pts_per_ray_;
  for (auto& m : renderer_->ema_ring_) m = Renderer::EmaMark();
}

// One optimiser step.  compute_flags (device int32[3], or NULL): the finiteness flags of the two MLP gradients are
// computed inside the small-groups launch and every update of this step is predicated on them (flags[2]).
void ExpRunner::BuildAdamPlan(AdamPlan& plan) {
  auto* field = static_cast<Hash3DAnchored*>(renderer_->scene_field_.get());
  auto* shader = static_cast<SHShader*>(renderer_->shader_.get());
  plan = AdamPlan();
  // the small fp32 groups (field MLP, colour MLP, app_emb), in the order the flag layout names them
  auto add_small = [&](size_t i) {
    auto& g = groups_[i];
    float scale = g.grad_scale;
    if (scale < 0.f) scale = 1.f / (g.name == "color_mlp" ? shader->mlp_->loss_scale_ : field->mlp_->loss_scale_);
    F2nAdamGroup d;
    d.param = F32P(g.param);
    d.grad = F32P(g.grad);
    d.exp_avg = F32P(exp_avg_[i]);
    d.exp_avg_sq = F32P(exp_avg_sq_[i]);
    d.param_h = g.param_h.defined() ? VoidP(g.param_h) : nullptr;
    d.n = (int) (g.active > 0 ? g.active : g.param.numel());
    d.grad_scale = scale;
    d.weight_decay = g.weight_decay;
    d.grad_round_h16 = g.grad_round_h16 ? 1 : 0;
    d.check_finite = 0;
    TORCH_CHECK(plan.n_small < 4, "too many small parameter groups");
    plan.small[plan.n_small++] = d;
  };
  for (size_t i = 0; i < groups_.size(); i++)
    if (groups_[i].name == "field_mlp") add_small(i);
  for (size_t i = 0; i < groups_.size(); i++)
    if (groups_[i].name == "color_mlp") add_small(i);
  // torch::optim::Adam skips parameters whose gradient is undefined: without the appearance embedding in use nothing ever
  // writes app_emb's gradient, and stepping it with weight decay alone (g = wd * p through Adam's normalisation) would walk
  // the embedding to zero at lr per step
  for (size_t i = 0; i < groups_.size(); i++)
    if (!groups_[i].grad_is_h16 && groups_[i].name != "field_mlp" && groups_[i].name != "color_mlp" &&
        (groups_[i].name != "app_emb" || renderer_->use_app_emb_))
      add_small(i);
  for (size_t i = 0; i < groups_.size(); i++) {
    auto& g = groups_[i];
    if (!g.grad_is_h16) continue;
    TORCH_CHECK(plan.n_table == 0, "one h16-gradient table group expected");
    plan.n_table = (int) (g.active > 0 ? g.active : g.param.numel());
    plan.tp = F32P(g.param); plan.tg = VoidP(g.grad); plan.tm = F32P(exp_avg_[i]); plan.tv = F32P(exp_avg_sq_[i]); plan.th = VoidP(g.param_h);
    plan.tscale = g.grad_scale;
    TORCH_CHECK(g.weight_decay == 0.f, "the table group has no weight decay (Hash3DAnchored.cpp:124-150)");
  }
}

void ExpRunner::OptimStep(const int32_t* skip_flag, int32_t* compute_flags) {
  optim_steps_ += 1;
  void* st = CurStream();
  auto* field = static_cast<Hash3DAnchored*>(renderer_->scene_field_.get());
  auto* shader = static_cast<SHShader*>(renderer_->shader_.get());
  torch::NoGradGuard no_grad;
  AdamPlan plan;
  BuildAdamPlan(plan);
  // TCNNWP.cpp:234-240 on the device: flags = {field MLP gradient non-finite, colour MLP gradient non-finite, either}; every
  // update of this step is predicated on flags[2] -- computed by its own small launch so that the ONE launch that steps all
  // groups (f2n_adam_fused) has no block-wide dependency in it
  const int32_t* skip = skip_flag;
  if (compute_flags != nullptr) {
    F2N_TIMED_CALL("adam", f2n_nonfinite_flags_ex(st, field->mlp_->n_params_, F32P(field->mlp_->grad_scaled_), shader->mlp_->n_params_,
                                                  F32P(shader->mlp_->grad_scaled_), compute_flags, NextFlagMirror()));
    skip = compute_flags + 2;
  }
  if (plan.n_table > 0) field->grad_clean_ = true;
  F2N_TIMED_CALL("adam_table", f2n_adam_fused(st, plan.n_small, plan.small, plan.n_table, plan.tp, plan.tg, plan.tscale, plan.tm, plan.tv, plan.th, optim_steps_, cur_lr_,
                                              /*betas: doubles, as AdamOptions holds them*/ 0.9, 0.99, 1e-15f, /*zero_grad=*/1, skip));
  renderer_->small_grads_clean_ = true;  // every group's gradient was consumed and cleared (also on the skipped path)
}

// The arguments of f2n_field_bwd_step_tail for the step that is being queued (called from Renderer::TrainForwardBackward in front of
// the field backward: the previous step's flags have been resolved by then, so the loss scales and the step count are this step's).
bool ExpRunner::BuildStepTail(F2nStepTail* t) {
  auto* field = static_cast<Hash3DAnchored*>(renderer_->scene_field_.get());
  auto* shader = static_cast<SHShader*>(renderer_->shader_.get());
  // data-parallel: only with an exchange that can take the small buffers early (the native RCCL one), and the table's Adam stays ours
  const bool dp = sync_.Installed();
  if (fused_tail_ == 0 || !check_nan_ || (dp && !sync_.small_exchange)) return false;
  if (!nan_flags_.defined()) nan_flags_ = torch::zeros({4}, DevI32());
  BuildAdamPlan(tail_plan_);
  if (tail_plan_.n_table <= 0) return false;
  t->n_flags_a = field->mlp_->n_params_;
  t->flags_grad_a = F32P(field->mlp_->grad_scaled_);
  t->n_flags_b = shader->mlp_->n_params_;
  t->flags_grad_b = F32P(shader->mlp_->grad_scaled_);
  t->flags = I32P(nan_flags_);
  t->flags_mirror = NextFlagMirror();
  t->n_groups = tail_plan_.n_small;
  t->groups = tail_plan_.small;
  t->n_table = tail_plan_.n_table;
  t->table_param = tail_plan_.tp;
  t->table_exp_avg = tail_plan_.tm;
  t->table_exp_avg_sq = tail_plan_.tv;
  t->table_param_h = tail_plan_.th;
  t->table_grad_scale = tail_plan_.tscale;
  t->step = optim_steps_ + 1;
  t->lr = cur_lr_;
  t->beta1 = 0.9;
  t->beta2 = 0.99;
  t->eps = 1e-15f;
  t->after_reduce = nullptr;
  t->after_reduce_user = nullptr;
  t->leave_table_to_caller = 0;
  tail_table_left_ = false;
  if (dp) {
    // The gradients travel before anything is stepped: the small buffers' exchange goes between the reductions and the flags (beside
    // the scatter's producers), the table's Adam stays a launch of this host's -- behind the table's exchange (EnqueueApply).
    t->after_reduce = [](void* user, void* chain_stream) { static_cast<ExpRunner*>(user)->sync_.SmallGradsReady(chain_stream); };
    t->after_reduce_user = this;
    t->leave_table_to_caller = 1;
    tail_table_left_ = true;
  }
  return true;
}

// Loss weights of the current iteration (ExpRunner.cpp:108-114).
float ExpRunner::CurVarLossWeight() const { return ScheduleAt(Schedule(), iter_step_).var_loss_weight; }

// One iteration of ExpRunner::Train (ExpRunner.cpp:82-143) for a given ray batch: untaped forward + loss + backward
// (Renderer::TrainForwardBackward), device-side finiteness flags, Adam predicated on them, ONE flag read-back.
TrainStats ExpRunner::TrainStep(const Tensor& rays_o, const Tensor& rays_d, const Tensor& bounds, const Tensor& gt_colors,
                                const Tensor& emb_idx, bool apply_optimizer, const Tensor& next_rays_o,
                                const Tensor& next_rays_d, const Tensor& next_bounds, const Tensor& next2_rays_o,
                                const Tensor& next2_rays_d) {
  // Network shapes without fused kernels (any tcnn FullyFusedMLP a YAML can ask for: csrc/mlp_generic.hip, unfused field /
  // shader paths) train through the taped iteration: same losses, same optimiser, no streaming.
  if (!renderer_->FusedPathOk()) return TrainStepAutograd(rays_o, rays_d, bounds, gt_colors, emb_idx, apply_optimizer);
  auto* gdp = global_data_pool_.get();
  // this step's batch, and the two behind it, by sequence number: what their random draws are keyed by (KeyedDraws.h)
  const int64_t seq = step_seq_++;
  renderer_->cur_seq_ = seq;
  struct SeqReset {
    Renderer* r;
    ~SeqReset() { r->cur_seq_ = -1; }
  } seq_reset{renderer_.get()};
  gdp->mode_ = RunningMode::TRAIN;
  gdp->backward_nan_ = false;
  const bool prefetch = apply_optimizer && next_rays_o.defined() && next_rays_d.defined();
  // Pipelined: the previous step's gradient all-reduce is still in flight on RCCL's stream.  Ray sampling reads neither the
  // parameters nor the gradients, so it is issued first (unless the previous step already prefetched it) and runs
  // under the collective; only then is the collective awaited and the previous step's (flag-predicated) Adam applied.
  sync_.BeginStep(apply_optimizer, [&]() { renderer_->PreSample(rays_o, rays_d, bounds); });
  renderer_->ZeroGrad();
  renderer_->after_octree_update_ = nullptr;  // (a previous call that threw must not leave its hook / half a prefetch behind)
  renderer_->next_batch_ = renderer_->next2_batch_ = Renderer::NextBatch();
  renderer_->KeepOnlyPending(rays_o, rays_d, prefetch ? next_rays_o : Tensor(), prefetch ? next_rays_d : Tensor(),
                             prefetch ? next2_rays_o : Tensor(), prefetch ? next2_rays_d : Tensor());
  deferred_dropped_ = false;
  if (!renderer_->after_count_readback_) renderer_->after_count_readback_ = [this]() { ResolveDeferredFlags(); };
  if (prefetch) {
    // (Pipelined data-parallel steps used to keep the sampling at the step boundary, underneath the gradient all-reduce; since
    // round 4 they run the same program as a single GPU -- speculative, two-deep sampling on the side streams -- and
    // BeginStep's PreSample finds the batch already in flight.)
    // The NEXT batch's sampling only depends on this step's octree update: its kernels are issued (on a side stream) from
    // inside SampleAndFilter, right behind that update, with the next iteration's fineness -- up to and including the pack;
    // the host comes back for its counts at the top of the next step (Renderer::SampleAndFilter -> PreSampleFinish).
    const float fin = FinenessAt(iter_step_ + 1);
    renderer_->after_octree_update_ = [this, fin, seq, &next_rays_o, &next_rays_d, &next_bounds]() {
      renderer_->PreSampleBegin(next_rays_o, next_rays_d, next_bounds, fin, seq + 1);
    };
    // ... or, outside the ProcOctree iterations, speculatively from the top of this step (Renderer.h: next_batch_); the hook
    // above remains the fallback
    renderer_->next_batch_.rays_o = next_rays_o;
    renderer_->next_batch_.rays_d = next_rays_d;
    renderer_->next_batch_.fineness = fin;
    renderer_->next_batch_.seq = seq + 1;
    renderer_->next_batch_.valid = true;
    if (next2_rays_o.defined() && next2_rays_d.defined()) {  // two-deep pipeline: the batch behind it is walked and marched now
      renderer_->next2_batch_.rays_o = next2_rays_o;
      renderer_->next2_batch_.rays_d = next2_rays_d;
      renderer_->next2_batch_.fineness = FinenessAt(iter_step_ + 2);
      renderer_->next2_batch_.seq = seq + 2;
      renderer_->next2_batch_.valid = true;
    }
  }
  // A streaming step (it was handed the next batch) does not wait for the survivor count either: it stays on the device
  // (f2n_*_dyn entry points), the host queues the whole iteration without a device round trip and learns the count -- for
  // the meaningful-samples EMA and the counters -- at the start of the next step (Renderer::ResolvePendingCount).
  renderer_->async_count_ = (prefetch && async_counts_ == 1) || async_counts_ == 2;
  renderer_->step_tail_done_ = false;
  renderer_->step_tail_builder_ = nullptr;
  renderer_->before_backward_ = nullptr;
  // (tables of 2^21 entries per level and more: the table's Adam is a 0.2-0.45 ms pass, and folding it into the owners wins in a young
  // scene too -- 2.29 -> 2.05 ms per step at 2^22, 1.66 -> 1.58 at 2^21, nothing at 2^20: profiles/r06_fused_tail_ab.txt)
  const bool big_table = static_cast<Hash3DAnchored*>(renderer_->scene_field_.get())->pool_size_ / N_LEVELS >= (1 << 21);
  // (a data-parallel step takes the tail in every regime: what it moves off the main queue there -- the small buffers' exchange, the
  // flags, the small Adam -- sat exposed between the table's exchange and the optimiser)
  const bool tail_ok = apply_optimizer && check_nan_ && (sync_.Installed() ? (fused_tail_ != 0 && (bool) sync_.small_exchange && prefetch)
                                                                            : (fused_tail_ == 1 || (fused_tail_ == 2 && prefetch && (renderer_->TwoDeepRegime() || big_table))));
  if (tail_ok) renderer_->step_tail_builder_ = [this](F2nStepTail* t) { return BuildStepTail(t); };
  sync_.small_first = tail_ok && sync_.Installed();  // (the collective order of this step: the same on every rank, GradSyncPipeline.h)
  // A streaming step learns the PREVIOUS step's finiteness flags late.  With the fused tail those flags are computed half a step
  // before that step ends, so they are read here -- in front of this step's backward -- at no cost, and a dropped step's halved
  // loss scales / taken-back counters hold for this step's backward AND its optimiser call, as in the reference's order
  // (ExpRunner.cpp:131-137).  Without it (data-parallel steps, fused_tail off) they are read behind the backward -- the loss scale the
  // backward used and the one the optimiser divides by then differ for the one step behind a dropped one -- unless
  // exact_flag_order_ asks for the wait (a host stall per step: tests compare the two tails with it).
  if (check_nan_ && (tail_ok || exact_flag_order_)) renderer_->before_backward_ = [this]() { ResolveDeferredFlags(); };
  TrainOutputs out;
  {
    F2N_HOST_SCOPE("step.fwd_bwd");
    out = renderer_->TrainForwardBackward(rays_o, rays_d, bounds, gt_colors, emb_idx, CurVarLossWeight(), disp_loss_weight_,
                                          tv_loss_weight_);
  }
  renderer_->async_count_ = false;
  renderer_->after_octree_update_ = nullptr;
  renderer_->step_tail_builder_ = nullptr;
  renderer_->before_backward_ = nullptr;
  if (prefetch && out.has_samples) renderer_->SpecBeginAtStepEnd();  // (small trees: the batch after next, see Renderer.h)
  renderer_->next_batch_ = renderer_->next2_batch_ = Renderer::NextBatch();
  ResolveDeferredFlags();  // (a batch without samples never reaches the read-back)
  TrainStats stats;
  stats.skipped_nan = deferred_dropped_;  // the PREVIOUS iteration was dropped: reported one step late when prefetching
  stats.n_rays = rays_o.size(0);
  stats.n_samples = renderer_->last_n_all_pts_;
  stats.n_meaningful = renderer_->count_pending_ ? -1 : renderer_->last_n_kept_pts_;  // -1: still on the device (see counters())
  stats.loss = out.losses.slice(0, 0, 1).squeeze(0);
  stats.mse = out.losses.slice(0, 5, 6).squeeze(0);
  bool applied = false;
  // (a data-parallel replica whose batch missed the scene still joins the gradient exchange, with its zero gradients)
  if (out.has_samples || sync_.Installed()) {
    // pipelined: asynchronous all-reduce, awaited in the next step (or Flush); else: all-reduce of the gradient buffers
    // (RCCL), then finiteness flags + predicated Adam with this iteration's learning rate
    applied = sync_.GradientsReady(apply_optimizer, cur_lr_);
  }
  if (apply_optimizer) {
    iter_step_++;
    UpdateAdaParams();
  }
  // The NEXT batch's sampling has been running on the side stream since this step's octree update.  The host does not wait
  // for its sample count here: the next step does, right before it needs it (Renderer::SampleAndFilter), so that whatever
  // the caller does between two steps overlaps the march instead of following it.
  if (digest_table_) {  // diagnostics: an order-free checksum of the f16 table as this step leaves it, kept on the device
    auto* field = static_cast<Hash3DAnchored*>(renderer_->scene_field_.get());
    if (!digest_table_sums_.defined()) digest_table_sums_ = torch::zeros({Renderer::kDigestRing}, torch::TensorOptions().dtype(torch::kInt64).device(torch::kCUDA));
    digest_table_sums_.select(0, seq % Renderer::kDigestRing).copy_(field->feat_pool_h_.view(torch::kInt16).sum(torch::kInt64));
  }
  if (renderer_->digest_taps_ && applied) {  // (parameters as this step's Adam leaves them; Renderer::DigestTap)
    auto* field = static_cast<Hash3DAnchored*>(renderer_->scene_field_.get());
    auto* shader = static_cast<SHShader*>(renderer_->shader_.get());
    renderer_->DigestTap(Renderer::TAP_TABLE, field->feat_pool_h_);
    renderer_->DigestTap(Renderer::TAP_FIELD_MLP, field->mlp_->params_h_);
    renderer_->DigestTap(Renderer::TAP_COLOR_MLP, shader->mlp_->params_h_);
    renderer_->DigestTap(Renderer::TAP_APP_EMB, renderer_->app_emb_);
  }
  if (prefetch && !renderer_->PendingMatches(next_rays_o, next_rays_d))  // (a batch without samples never reached the hook)
    renderer_->PreSampleBegin(next_rays_o, next_rays_d, next_bounds, global_data_pool_->ray_march_fineness_, seq + 1);
  if (applied && check_nan_ && prefetch) {  // streaming: do not stall on this step's flags (see ExpRunner.h)
    DeferFlags(apply_optimizer);
  } else if (applied && ResolveFlags(apply_optimizer)) {
    stats.skipped_nan = true;  // iteration not advanced, like the `continue` at ExpRunner.cpp:133
    if (apply_optimizer) {
      iter_step_ = std::max(0, iter_step_ - 1);
      UpdateAdaParams();
    }
  }
  return stats;
}

bool ExpRunner::ResolveDeferredFlags() {
  if (!flags_deferred_) return false;
  flags_deferred_ = false;
  {
    F2N_HOST_SCOPE("wait.flags");
    nan_flags_ev_.synchronize();
  }
  auto* field = static_cast<Hash3DAnchored*>(renderer_->scene_field_.get());
  auto* shader = static_cast<SHShader*>(renderer_->shader_.get());
  const int32_t f[3] = {flag_words_.Read(4 * deferred_flag_slot_), flag_words_.Read(4 * deferred_flag_slot_ + 1),
                        flag_words_.Read(4 * deferred_flag_slot_ + 2)};
  if (f[0]) field->mlp_->loss_scale_ = std::max(field->mlp_->loss_scale_ / 2.f, 1.f);
  if (f[1]) shader->mlp_->loss_scale_ = std::max(shader->mlp_->loss_scale_ / 2.f, 1.f);
  if (!f[2]) return false;
  global_data_pool_->backward_nan_ = true;
  if (deferred_apply_) {
    optim_steps_ -= 1;
    iter_step_ = std::max(0, iter_step_ - 1);
    UpdateAdaParams();
  }
  deferred_dropped_ = true;
  return true;
}

// The host-visible slot the next flag kernel writes besides the device flags (see ExpRunner.h).
int32_t* ExpRunner::NextFlagMirror() {
  flag_words_.Ensure(8);
  last_flag_slot_ = next_flag_slot_;
  next_flag_slot_ ^= 1;
  return flag_words_.Dev(4 * last_flag_slot_);
}

// Finiteness flags (TCNNWP.cpp:234-240, on the device) and Adam predicated on them: enqueue only.
void ExpRunner::EnqueueApply(bool apply_optimizer) {
  auto* field = static_cast<Hash3DAnchored*>(renderer_->scene_field_.get());
  auto* shader = static_cast<SHShader*>(renderer_->shader_.get());
  if (check_nan_ && !nan_flags_.defined()) nan_flags_ = torch::zeros({4}, DevI32());
  if (apply_optimizer && renderer_->step_tail_done_) {
    // the field backward's call has queued all of it (f2n_field_bwd_step_tail): what is left is the host's own bookkeeping
    renderer_->step_tail_done_ = false;
    flags_on_tail_stream_ = true;
    optim_steps_ += 1;
    if (tail_table_left_) {  // (data-parallel: the table's pass, behind the table's exchange; the flags were computed beside the scatter)
      const AdamPlan& p = tail_plan_;
      F2N_TIMED_CALL("adam_table", f2n_adam_fused(CurStream(), 0, nullptr, p.n_table, p.tp, p.tg, p.tscale, p.tm, p.tv, p.th, optim_steps_, cur_lr_, 0.9, 0.99, 1e-15f,
                                                  /*zero_grad=*/1, I32P(nan_flags_) + 2));
      tail_table_left_ = false;
    }
    field->grad_clean_ = true;
    renderer_->small_grads_clean_ = true;
  } else if (apply_optimizer) {  // the flags are computed by the small-groups launch itself; a no-op on the device when they say so
    flags_on_tail_stream_ = false;
    OptimStep(nullptr, check_nan_ ? I32P(nan_flags_) : nullptr);
  } else if (check_nan_) {
    flags_on_tail_stream_ = false;
    F2N_CALL(f2n_nonfinite_flags_ex(CurStream(), field->mlp_->n_params_, F32P(field->mlp_->grad_scaled_), shader->mlp_->n_params_,
                                    F32P(shader->mlp_->grad_scaled_), I32P(nan_flags_), NextFlagMirror()));
  }
}

// The iteration's only read-back besides the two sample counts.  Returns true when the gradients were not finite (loss
// scales halved, nothing was applied).
bool ExpRunner::ResolveFlags(bool apply_optimizer) {
  if (!check_nan_) return false;
  auto* gdp = global_data_pool_.get();
  auto* field = static_cast<Hash3DAnchored*>(renderer_->scene_field_.get());
  auto* shader = static_cast<SHShader*>(renderer_->shader_.get());
  Tensor h = nan_flags_.cpu();
  const int32_t* f = h.data_ptr<int32_t>();
  if (f[0]) field->mlp_->loss_scale_ = std::max(field->mlp_->loss_scale_ / 2.f, 1.f);
  if (f[1]) shader->mlp_->loss_scale_ = std::max(shader->mlp_->loss_scale_ / 2.f, 1.f);
  if (f[2]) {
    gdp->backward_nan_ = true;
    if (apply_optimizer) optim_steps_ -= 1;
    return true;
  }
  return false;
}

bool ExpRunner::ApplyGradients(bool apply_optimizer) {
  EnqueueApply(apply_optimizer);
  return ResolveFlags(apply_optimizer);
}

// Pipelined data-parallel mode: completes the step whose gradients are still being reduced (awaits the collective,
// applies Adam with that step's learning rate).  A non-finite gradient drops that step after the fact: the iteration
// counter is taken back by one (the sampling of the step in between has already used the advanced schedule -- the one
// deviation from the unpipelined order, and only on that rare path).
void ExpRunner::FinishPending() {
  FinishPendingStep();
  renderer_->ResolvePendingCount();
  ResolveDeferredFlags();
}

void ExpRunner::FinishPendingStep() { sync_.FinishPendingStep(); }

void ExpRunner::DeferFlags(bool apply_optimizer) {
  if (flags_deferred_) ResolveDeferredFlags();  // (one set in flight at a time)
  deferred_flag_slot_ = last_flag_slot_;  // (written by the flag kernel EnqueueApply has just queued)
  // (a fused step tail ran the flag kernel on the tail stream: recorded there, the main queue is spared the packet)
  if (flags_on_tail_stream_) nan_flags_ev_.record(*renderer_->TailStream());
  else nan_flags_ev_.record();
  flags_deferred_ = true;
  deferred_apply_ = apply_optimizer;
}

// The same iteration on the autograd tape, op for op as the reference spells it (ExpRunner.cpp:82-143): Render(),
// ATen loss, loss.backward(), per-MLP host finiteness checks.  Kept as the cross-check of the fused TrainStep
// (tests/test_gpu_e2e.py); ~100 more launches and four host round trips per iteration.
TrainStats ExpRunner::TrainStepAutograd(const Tensor& rays_o, const Tensor& rays_d, const Tensor& bounds, const Tensor& gt_colors,
                                const Tensor& emb_idx, bool apply_optimizer) {
  auto* gdp = global_data_pool_.get();
  gdp->mode_ = RunningMode::TRAIN;
  gdp->backward_nan_ = false;
  const int batch = rays_o.size(0);
  renderer_->cur_seq_ = step_seq_++;
  sync_.ArmBuckets();  // (the taped backward's scatter reports its table buckets to the exchange below)
  sync_.small_first = false;  // (a taped step sends the small buffers behind the table, on every rank)
  auto rr = renderer_->Render(rays_o, rays_d, bounds, emb_idx);
  renderer_->cur_seq_ = -1;
  TrainStats stats;
  stats.n_rays = batch;
  stats.n_samples = renderer_->last_n_all_pts_;
  stats.n_meaningful = renderer_->last_n_kept_pts_;
  Tensor pred_colors = rr.colors.index({Slc(0, batch)});
  Tensor color_loss = torch::sqrt((pred_colors - gt_colors).square() + 1e-4f).mean();
  Tensor loss = color_loss;
  if (rr.weights.defined()) {
    Tensor disparity_loss = rr.disparity.square().mean();
    Tensor tv_loss = (rr.edge_feats.index({Slc(), 0}) - rr.edge_feats.index({Slc(), 1})).square().mean();
    Tensor sampled_var = CustomOps::WeightVar(rr.weights, rr.idx_start_end);
    Tensor var_loss = (sampled_var + 1e-2).sqrt().mean();
    const float var_w = CurVarLossWeight();
    loss = color_loss + var_loss * var_w + disparity_loss * disp_loss_weight_ + tv_loss * tv_loss_weight_;
  }
  stats.loss = loss.detach();
  stats.mse = (pred_colors.detach() - gt_colors).square().mean();
  // (a data-parallel replica whose batch missed the scene still joins the gradient exchange, with its zero gradients: every
  // rank must enter the collective, and the ranks that did see samples average over the whole world)
  const bool has_grad = loss.requires_grad();
  if (has_grad || sync_.Installed()) {
    renderer_->ZeroGrad();
    if (has_grad) loss.backward();
    if (renderer_->app_emb_.grad().defined()) {  // (the unfused shader path delivers this gradient through autograd)
      torch::NoGradGuard ng;
      renderer_->app_emb_grad_.add_(renderer_->app_emb_.grad());
      renderer_->app_emb_.mutable_grad() = Tensor();
    }
    // data-parallel all-reduce of the gradient buffers (RCCL), whichever way it was installed: the taped step has no next
    // step's sampling to hide a pipelined exchange under, so begin / end run back to back
    if (sync_.blocking) {
      sync_.blocking();
    } else if (sync_.begin) {
      sync_.begin();
      if (sync_.end) sync_.end();
    }
    sync_.ResetBuckets();  // (table buckets the taped backward's scatter reported were sent; the rest went with the exchange above)
    if (check_nan_) {  // TCNNWP.cpp:234-240 + ExpRunner.cpp:131-134
      auto* field = static_cast<Hash3DAnchored*>(renderer_->scene_field_.get());
      auto* shader = static_cast<SHShader*>(renderer_->shader_.get());
      bool ok = field->mlp_->CheckGradFinite();
      ok = shader->mlp_->CheckGradFinite() && ok;
      (void) ok;
    }
    if (gdp->backward_nan_) {
      stats.skipped_nan = true;
      return stats;  // iteration not advanced, like the `continue` at ExpRunner.cpp:133
    }
    if (apply_optimizer) OptimStep();
  }
  if (apply_optimizer) {
    iter_step_++;
    UpdateAdaParams();
  }
  return stats;
}

// ExpRunner::RenderWholeImage chunk body (ExpRunner.cpp:268-287), VALIDATE mode.
std::vector<Tensor> ExpRunner::RenderRays(const Tensor& rays_o, const Tensor& rays_d, const Tensor& bounds) {
  FinishPending();
  torch::NoGradGuard no_grad;
  auto prev = global_data_pool_->mode_;
  global_data_pool_->mode_ = RunningMode::VALIDATE;
  auto rr = forward_render_ ? renderer_->RenderForward(rays_o, rays_d, bounds) : renderer_->Render(rays_o, rays_d, bounds, Tensor());
  global_data_pool_->mode_ = prev;
  return {rr.colors, rr.disparity, rr.first_oct_dis, rr.depth};
}

// ExpRunner::RenderWholeImage (ExpRunner.cpp:255-292): all rays of a view in chunks, VALIDATE mode.  The reference bounces
// every 8192-ray chunk through the CPU; here rays and results stay in HBM and a chunk is 65536 rays on
// the forward-only path (one host round trip -- the sample count -- per chunk; rays are independent and VALIDATE noise is a
// constant, so the chunking does not change a bit of the image; sampler scratch: ~5 GB of the 288).
std::vector<Tensor> ExpRunner::RenderWholeImage(const Tensor& rays_o, const Tensor& rays_d, const Tensor& bounds) {
  torch::NoGradGuard no_grad;
  const int n_rays = rays_d.size(0);
  Tensor pred_colors = torch::zeros({n_rays, 3}, DevF32());
  Tensor first_oct_disp = torch::full({n_rays, 1}, 1.f, DevF32());
  Tensor pred_disp = torch::zeros({n_rays, 1}, DevF32());
  const int ray_batch_size = forward_render_ ? render_chunk_rays_ : 8192;
  for (int i = 0; i < n_rays; i += ray_batch_size) {
    const int hi = std::min(i + ray_batch_size, n_rays);
    auto out = RenderRays(rays_o.index({Slc(i, hi)}).contiguous(), rays_d.index({Slc(i, hi)}).contiguous(),
                          bounds.index({Slc(i, hi)}).contiguous());
    pred_colors.index_put_({Slc(i, hi)}, out[0]);
    pred_disp.index_put_({Slc(i, hi)}, out[1].reshape({-1, 1}));
    if (out[2].defined() && out[2].numel() == hi - i) first_oct_disp.index_put_({Slc(i, hi)}, out[2].reshape({-1, 1}));
  }
  pred_disp = pred_disp / pred_disp.max();
  first_oct_disp = first_oct_disp.min() / first_oct_disp;
  return {pred_colors, first_oct_disp, pred_disp};
}

// The per-image body of ExpRunner::TestImages (ExpRunner.cpp:353-369): render camera idx of the data set, quantise the
// prediction to 8 bit exactly as the reference does before it measures ((x.clip(0,1) * 255) -> uint8 -> / 255, :360-362),
// PSNR = 20 log10(1 / sqrt(mse)) against the resident ground truth.  Everything stays on the device.
float ExpRunner::TestImagePSNR(Dataset& dataset, int idx) {
  TORCH_CHECK(dataset.image_tensors_.defined(), "no ground-truth images resident");
  auto rays = dataset.RaysOfCamera(idx);
  auto out = RenderWholeImage(rays.origins, rays.dirs, rays.bounds);
  Tensor pred = (out[0].clip(0.f, 1.f) * 255.f).to(torch::kUInt8).to(torch::kFloat32) / 255.f;
  Tensor gt = dataset.image_tensors_[idx].reshape({-1, 3});
  const float mse = (pred - gt).square().mean().item<float>();
  return 20.f * std::log10(1.f / std::sqrt(mse));
}

// ExpRunner::TestImages (ExpRunner.cpp:343-383) without the PNG / YAML output: per-view PSNR of the test set, mean last.
std::vector<float> ExpRunner::TestImages(Dataset& dataset) {
  FinishPending();
  auto prev = global_data_pool_->mode_;
  global_data_pool_->mode_ = RunningMode::VALIDATE;
  std::vector<float> out;
  float sum = 0.f;
  for (int i : dataset.test_set_) {
    out.push_back(TestImagePSNR(dataset, i));
    sum += out.back();
  }
  out.push_back(out.empty() ? 0.f : sum / float(out.size()));
  global_data_pool_->mode_ = prev;
  return out;
}

// One frame of ExpRunner::RenderPath (ExpRunner.cpp:322-341): the view from `pose` at 1/res_level resolution as the image
// the reference writes -- [H, 3W, 3]: colours | first-octree-hit disparity | disparity, side by side.
Tensor ExpRunner::RenderPathFrame(Dataset& dataset, const Tensor& pose, int res_level) {
  FinishPending();
  auto prev = global_data_pool_->mode_;
  global_data_pool_->mode_ = RunningMode::VALIDATE;
  auto rays = dataset.RaysFromPose(pose, res_level);
  auto out = RenderWholeImage(rays.origins, rays.dirs, rays.bounds);
  const int H = dataset.height_ / res_level, W = dataset.width_ / res_level;
  Tensor img = torch::cat({out[0].reshape({H, W, 3}), out[1].reshape({H, W, 1}).repeat({1, 1, 3}),
                           out[2].reshape({H, W, 1}).repeat({1, 1, 3})}, 1);
  global_data_pool_->mode_ = prev;
  return img;
}

// ExpRunner::VisualizeImage (ExpRunner.cpp:301-321), the body of RenderAllImages (:295-299): training view idx rendered in
// VALIDATE mode as the image the reference writes to images/<iter>_<idx>.png -- [H, 4W, 3]: ground truth | colours |
// first-octree-hit disparity | disparity.
Tensor ExpRunner::VisualizeImage(Dataset& dataset, int idx) {
  TORCH_CHECK(dataset.image_tensors_.defined(), "no ground-truth images resident");
  FinishPending();
  auto prev = global_data_pool_->mode_;
  global_data_pool_->mode_ = RunningMode::VALIDATE;
  auto rays = dataset.RaysOfCamera(idx);
  auto out = RenderWholeImage(rays.origins, rays.dirs, rays.bounds);
  const int H = dataset.height_, W = dataset.width_;
  Tensor img = torch::cat({dataset.image_tensors_[idx].reshape({H, W, 3}), out[0].reshape({H, W, 3}),
                           out[1].reshape({H, W, 1}).repeat({1, 1, 3}), out[2].reshape({H, W, 1}).repeat({1, 1, 3})}, 1);
  global_data_pool_->mode_ = prev;
  return img;
}

// ExpRunner::RenderPath: every pose of `render_poses` [P,3,4]; `sink(i, image)` receives the frames (the reference writes
// novel_images/<iter>_<i>.png -- image encoding is outside the hot path and left to the caller).
void ExpRunner::RenderPath(Dataset& dataset, const Tensor& render_poses, const std::function<void(int, const Tensor&)>& sink,
                           int res_level) {
  Tensor poses = render_poses.to(torch::kCPU).to(torch::kFloat32).contiguous();
  TORCH_CHECK(poses.dim() == 3 && poses.size(1) == 3 && poses.size(2) == 4, "render_poses must be [P,3,4]");
  for (int i = 0; i < poses.size(0); i++) sink(i, RenderPathFrame(dataset, poses[i], res_level));
}

// ExpRunner::SaveCheckpoint / LoadCheckpoint (ExpRunner.cpp:188-219): <dir>/renderer.pt = torch::save of the state vector
// in the reference's order (Renderer::States: sampler [nodes, warps, visit counts, milestones], field [table, primes,
// biases, n_volumes, MLP], shader [MLP], app_emb), <dir>/scalars.pt = float tensor [iter_step].  Same container format
// (LibTorch's pickle archive of a tensor list), so the files are interchangeable with the reference's.
void ExpRunner::SaveCheckpoint(const std::string& dir) {
  FinishPending();  // a pipelined step's Adam / a deferred NaN flag may still change parameters and iter_step_
  std::vector<Tensor> states = States();
  for (auto& t : states) t = t.detach().contiguous();
  torch::save(states, dir + "/renderer.pt");
  // [0] = iter_step, all the reference reads (ExpRunner.cpp:196-199: scalars[0]); [1] = the step sequence number the keyed draws
  // continue from -- it counts the steps that were dropped for non-finite gradients as well, so it can be ahead of [0] (round-5 advisor)
  Tensor scalars = torch::empty({2}, CpuF32());
  scalars.index_put_({0}, float(iter_step_));
  scalars.index_put_({1}, float(step_seq_));
  torch::save(scalars, dir + "/scalars.pt");
}

void ExpRunner::LoadCheckpoint(const std::string& dir) {
  FinishPending();
  int64_t resume_seq = 0;
  {
    Tensor scalars;
    torch::load(scalars, dir + "/scalars.pt");
    iter_step_ = (int) std::round(scalars[0].item<float>());
    UpdateAdaParams();
    // (a checkpoint of the reference, or of an earlier round, holds the iteration only: the sequence restarts there)
    resume_seq = scalars.numel() >= 2 ? (int64_t) std::llround(scalars[1].item<float>()) : (int64_t) iter_step_;
  }
  ResetStepSequence(std::max<int64_t>(resume_seq, iter_step_));
  std::vector<Tensor> states;
  torch::load(states, dir + "/renderer.pt");
  LoadStates(states);
}

// The loop of ExpRunner::Train (ExpRunner.cpp:82-143) without its reporting / checkpoint branches: adaptive ray batch
// (pts_batch_size / meaningful samples per ray, :86), Dataset::RandRaysData on the device, one fused TrainStep per
// iteration with the NEXT batch drawn one iteration ahead so that its ray sampling is prefetched.  Runs until
// iter_step_ reaches `until_iter` (or end_iter_); returns the iterations executed (skipped NaN iterations included).
int ExpRunner::Train(Dataset& dataset, int until_iter, int sets) {
  const int target = until_iter > 0 ? std::min(until_iter, end_iter_) : end_iter_;
  int executed = 0;
  // The draw's one kernel does not sit on the step's main queue (round 6: 7-9 us per step there, in front of the gather): it runs on the
  // device's tail stream, which is idle at the top of a step (Renderer::BeginDraw / EndDraw: ordering and the pool's protection).
  auto draw = [&](int64_t seq) {
    const int n_rays = std::max(16, BatchSizeFor(seq));
    if (!draws_off_main_) return dataset.RandRaysData(n_rays, sets, seq);
    c10::hip::HIPStreamGuardMasqueradingAsCUDA guard(*renderer_->BeginDraw());
    auto batch = dataset.RandRaysData(n_rays, sets, seq);
    renderer_->EndDraw();
    return batch;
  };
  // The batches of the next iteration AND (two-deep sampling pipeline, Renderer::next2_batch_) of the one after it are drawn
  // ahead: one draw per iteration, right before the step, as before -- but the adaptive ray count of a batch now comes from the
  // meaningful-samples average at a fixed lag (BatchSizeFor; the reference: the step before, ExpRunner.cpp:86), because its rays
  // have to exist two steps before it is used.  Always two ahead, whatever the renderer then does with the second batch -- and
  // every batch is keyed by its sequence number (rays, march noise, background, edge samples: KeyedDraws.h), so the batches a
  // Train() call draws at its start are the ones the previous call had drawn ahead and dropped: the sequence of batches depends
  // neither on a scheduling decision nor on how the iterations are split over Train() calls.
  const bool two_deep = true;  // (whether the batch after next is BEGUN two steps ahead is the renderer's decision)
  std::deque<decltype(draw(0))> ahead;
  last_train_meaningful_ = last_train_marched_ = last_train_rays_ = 0;
  FinishPending();
  struct RaysOffMain {  // (also when a step throws)
    Renderer* r;
    RaysOffMain(Renderer* r_, bool on) : r(r_) { r->rays_off_main_ = on; }
    ~RaysOffMain() { r->rays_off_main_ = false; }
  } rays_off_main(renderer_.get(), false);  // (on from this call's second step: whatever the caller queued before this call is behind the first's event)
  const int64_t kept0 = renderer_->total_kept_pts_;
  const int give_up = 4 * (target + 16);  // every iteration non-finite: stop instead of spinning
  while (true) {
  while (iter_step_ < target) {
    {
      F2N_HOST_SCOPE("train.draw");
      while (ahead.size() < (two_deep ? 3u : 2u)) ahead.push_back(draw(step_seq_ + (int64_t) ahead.size()));
    }
    F2N_HOST_SCOPE("train.step");
    auto& cur = ahead[0];
    const BoundedRays& r = std::get<0>(cur);
    const BoundedRays& nr = std::get<0>(ahead[1]);
    Tensor gt = std::get<1>(cur);
    TORCH_CHECK(gt.defined(), "Train needs resident ground-truth images in the Dataset");
    const uint64_t consumed_seq = renderer_->ConsumedSeq();
    TrainStats s = two_deep ? TrainStep(r.origins, r.dirs, r.bounds, gt, std::get<2>(cur), true, nr.origins, nr.dirs, nr.bounds,
                                        std::get<0>(ahead[2]).origins, std::get<0>(ahead[2]).dirs)
                            : TrainStep(r.origins, r.dirs, r.bounds, gt, std::get<2>(cur), true, nr.origins, nr.dirs, nr.bounds);
    // (the batch's tensors come from the tail stream's pool and are released below: every step of this loop leaves a `consumed` behind)
    if (draws_off_main_) renderer_->ConsumedBehindStep(consumed_seq);
    renderer_->rays_off_main_ = draws_off_main_ && spec_start_without_event_;
    ahead.pop_front();
    last_train_marched_ += s.n_samples;
    last_train_rays_ += s.n_rays;
    last_train_stats_ = s;
    executed++;
    if (executed > give_up) break;
  }
  FinishPending();  // may take the last iteration back (its finiteness flags are read one step late)
  if (iter_step_ >= target || executed > give_up) break;
  }
  // (streaming steps resolve their survivor counts one step late: the total is the renderer's running counter, complete
  // after the flush above)
  last_train_meaningful_ = renderer_->total_kept_pts_ - kept0;
  return executed;
}

}  // namespace f2n
