// Renderer host logic (SHShader: SHShader.cpp; the sampling pipeline: RendererPrefetch.cpp; the untaped training iteration:
// RendererTrain.cpp).  Behaviour follows src/Renderer/Renderer.cpp:52-258 of the reference (cited inline).  Where the reference strings ~40 ATen ops and 6 FlexOps launches per Render
// call, this issues: field pre-pass -> early_stop -> scan -> compact -> (mark_visit, update_stats) -> edge
// samples -> fused field -> scatter_idx -> fused shade -> composite, with two host read-backs in total (N, M).
// Render() is the taped (autograd) plugin entry point of the reference; TrainForwardBackward() is the same chain plus
// loss and backward, untaped, for ExpRunner::TrainStep.
#include "Renderer.h"

#include <hip/hip_runtime_api.h>

#include <cstdio>
#include <cstdlib>
#include <mutex>

namespace f2n {

using torch::autograd::AutogradContext;
using torch::autograd::variable_list;

namespace {

// Volume rendering (Renderer.cpp:190-208) as one autograd node.
struct CompositeFunction : public torch::autograd::Function<CompositeFunction> {
  static variable_list forward(AutogradContext* ctx, Tensor feat, Tensor rgb, Tensor dt, Tensor t, Tensor bg, Tensor se,
                               double gs_progress) {
    ctx->set_materialize_grads(false);
    const int n_rays = se.size(0), m = feat.size(0);
    Tensor colors = torch::empty({n_rays, 3}, DevF32()), disparity = torch::empty({n_rays}, DevF32());
    Tensor depth = torch::empty({n_rays}, DevF32()), weights = torch::empty({m}, DevF32());
    F2N_TIMED_CALL("composite_fwd", f2n_composite_fwd(CurStream(), n_rays, I32P(se), F32P(feat), F2N_MLP_OUT_PAD, F32P(dt), F32P(t), F32P(rgb), F32P(bg),
                               F32P(colors), F32P(disparity), F32P(depth), F32P(weights), nullptr));
    ctx->save_for_backward({feat, rgb, dt, t, bg, se});
    ctx->saved_data["gs"] = gs_progress;
    return {colors, disparity, depth, weights};
  }
  static variable_list backward(AutogradContext* ctx, variable_list g) {
    auto s = ctx->get_saved_variables();
    const int n_rays = s[5].size(0), m = s[0].size(0);
    Tensor gc = g[0].defined() ? g[0].contiguous() : Tensor(), gd = g[1].defined() ? g[1].contiguous() : Tensor();
    Tensor gz = g[2].defined() ? g[2].contiguous() : Tensor(), gw = g[3].defined() ? g[3].contiguous() : Tensor();
    Tensor drgb = torch::zeros({m, 3}, DevF32());
    Tensor dfeat = torch::zeros({m, 16}, DevF32());
    F2N_TIMED_CALL("composite_bwd", f2n_composite_bwd(CurStream(), n_rays, I32P(s[5]), F32P(s[0]), F2N_MLP_OUT_PAD, F32P(s[2]), F32P(s[3]), F32P(s[1]), F32P(s[4]),
                               gc.defined() ? F32P(gc) : nullptr, gd.defined() ? F32P(gd) : nullptr,
                               gz.defined() ? F32P(gz) : nullptr, gw.defined() ? F32P(gw) : nullptr,
                               (float) ctx->saved_data["gs"].toDouble(), F32P(drgb), F32P(dfeat), F2N_MLP_OUT_PAD, nullptr, nullptr));
    return {dfeat, drgb, Tensor(), Tensor(), Tensor(), Tensor(), Tensor()};
  }
};

struct WeightVarFunction : public torch::autograd::Function<WeightVarFunction> {
  static variable_list forward(AutogradContext* ctx, Tensor weights, Tensor se) {
    const int n = se.size(0);
    Tensor out = torch::empty({n}, DevF32());
    F2N_TIMED_CALL("weight_var_fwd", f2n_weight_var_fwd(CurStream(), n, F32P(weights), I32P(se), F32P(out)));
    ctx->save_for_backward({weights, se});
    return {out};
  }
  static variable_list backward(AutogradContext* ctx, variable_list g) {
    auto s = ctx->get_saved_variables();
    Tensor dvar = g[0].contiguous();
    Tensor dw = torch::zeros_like(s[0]);
    F2N_TIMED_CALL("weight_var_bwd", f2n_weight_var_bwd(CurStream(), (int) s[1].size(0), F32P(s[0]), I32P(s[1]), F32P(dvar), F32P(dw)));
    return {dw, Tensor()};
  }
};

}  // namespace

Tensor CustomOps::WeightVar(Tensor weights, Tensor idx_start_end) {
  return WeightVarFunction::apply(weights.contiguous(), idx_start_end.contiguous())[0];
}

// ---------------------------------------------------------------------------------------------------------
// Renderer
// ---------------------------------------------------------------------------------------------------------
Renderer::Renderer(GlobalDataPool* gdp, int n_images) {  // Renderer.cpp:22-49
  global_data_pool_ = gdp;
  gdp->renderer_ = this;
  pts_sampler_ = ConstructPtsSampler(gdp);
  RegisterSubPipe(pts_sampler_.get());
  scene_field_ = ConstructField(gdp);
  RegisterSubPipe(scene_field_.get());
  shader_ = ConstructShader(gdp);
  RegisterSubPipe(shader_.get());
  use_app_emb_ = gdp->config_.Bool("renderer.use_app_emb");
  app_emb_ = torch::randn({n_images, 16}, DevF32()) * .1f;
  app_emb_.requires_grad_(true);
  app_emb_grad_ = torch::zeros({n_images, 16}, DevF32());
  const std::string bg = gdp->config_.Str("renderer.bg_color");
  bg_color_type_ = bg == "white" ? BGColorType::white : (bg == "black" ? BGColorType::black : BGColorType::rand_noise);
}

bool Renderer::FusedPathOk() const {
  return static_cast<Hash3DAnchored*>(scene_field_.get())->fused_ok_ && static_cast<SHShader*>(shader_.get())->fused_ok_;
}

void Renderer::ZeroGrad() {
  auto* field = static_cast<Hash3DAnchored*>(scene_field_.get());
  F2N_CALL(f2n_deferred_reset());  // (a step that threw between its deferring launches and f2n_reduce_deferred leaves nothing behind)
  if (small_grads_clean_) {  // the fused Adam step cleared everything it consumed: no fill kernels this iteration
    small_grads_clean_ = false;
    if (!field->grad_clean_) field->grad_h_.zero_();
    field->grad_clean_ = true;
    return;
  }
  if (small_grads_flat_.defined()) {  // field MLP, colour MLP and app_emb gradients are views of one tensor: one fill
    if (!field->grad_clean_) field->grad_h_.zero_();
    field->grad_clean_ = true;
    small_grads_flat_.zero_();
    return;
  }
  field->ZeroGrad();
  static_cast<SHShader*>(shader_.get())->mlp_->ZeroGrad();
  app_emb_grad_.zero_();
}

void Renderer::DigestTap(int tap, const Tensor& t) {
  if (!digest_taps_ || cur_seq_ < 0 || !t.defined() || t.numel() == 0) return;
  if (!digest_tap_sums_.defined())
    digest_tap_sums_ = torch::zeros({kDigestRing, N_TAPS}, torch::TensorOptions().dtype(torch::kInt64).device(torch::kCUDA));
  Tensor flat = t.detach().contiguous().reshape({-1});
  const int64_t bytes = flat.numel() * flat.element_size();
  Tensor bits = bytes % 4 == 0 ? flat.view(torch::kInt32) : (bytes % 2 == 0 ? flat.view(torch::kInt16) : flat.view(torch::kInt8));
  digest_tap_sums_.select(0, cur_seq_ % kDigestRing).select(0, tap).copy_(bits.sum(torch::kInt64));
}

float Renderer::KeptPerRayForEma(int n_kept_local, int n_rays) {
  if (dp_world_ <= 1 || !dp_count_host_.defined()) return float(n_kept_local) / float(n_rays);
  dp_count_ev_.synchronize();  // recorded right behind the occupancy exchange of the same step
  return float(dp_count_host_.data_ptr<int32_t>()[0]) / (float(n_rays) * float(dp_world_));
}

void Renderer::ResolvePendingCount() {
  if (!count_pending_) return;
  count_pending_ = false;
  {
    F2N_HOST_SCOPE("wait.kept");
    kept_wait_ev_->synchronize();  // long since recorded: this is the previous step's count
  }
  const int n_kept = n_kept_words_.Read(1);
  last_n_kept_pts_ = n_kept;
  total_kept_pts_ += n_kept;
  DigestKept(pending_count_seq_, n_kept);
  auto* gdp = global_data_pool_;
  if (dp_world_ > 1) {
    // Data-parallel streaming steps: every rank must size its next batch from the same number, the survivor count summed over
    // the ranks.  The sum rides in the occupancy exchange (DataParallel::OccupancySync) -- which a streaming step issues BEFORE
    // its survivor scan (octree-first, as on one GPU) -- so the exchange of step k carries the count of step k-1, and step k's
    // scan drops the sum into the mapped word next to its own count: no launch, no wait of its own, one more step of lag in an
    // average that only sizes batches.
    if (dp_sum_mirrored_) {
      const float per_ray = float(n_kept_words_.Read(0)) / (float(dp_sum_rays_) * float(dp_world_));
      gdp->meaningful_sampled_pts_per_ray_ = gdp->meaningful_sampled_pts_per_ray_ * 0.9f + per_ray * 0.1f;
      RecordEma(pending_count_seq_ - 1);  // (the sum that has just gone in is the count of the step before the pending one)
    }
    dp_sum_mirrored_ = false;
    return;
  }
  gdp->meaningful_sampled_pts_per_ray_ = gdp->meaningful_sampled_pts_per_ray_ * 0.9f + KeptPerRayForEma(n_kept, pending_count_rays_) * 0.1f;
  RecordEma(pending_count_seq_);
  // (the previous step's finiteness flags are NOT read here: they are written by that step's last kernel, and waiting for
  // them at the top of a step would stop the host from queueing ahead -- ExpRunner::TrainStep reads them once this step's
  // forward and backward are queued)
}

RenderFront Renderer::SampleAndFilter(const Tensor& rays_o, const Tensor& rays_d, const Tensor& bounds, const Tensor& emb_idx,
                                      bool async_count) {
  F2N_HOST_SCOPE("step.sample_and_filter");
  auto* gdp = global_data_pool_;
  ResolvePendingCount();
  if (gdp->mode_ == RunningMode::TRAIN) DebugSkew(2);
  auto* field = static_cast<Hash3DAnchored*>(scene_field_.get());
  const bool train = gdp->mode_ == RunningMode::TRAIN;
  const int n_rays = rays_o.size(0);
  // The side stream of the next batch's speculative sampling (started further down, once this batch's samples are in hand) is
  // ordered behind the point the main stream has reached NOW, not behind the draws and the edge samples that
  // follow: the next batch's intersection then starts as soon as the previous step's Adam has finished and has the otherwise
  // idle device to itself while the main queue hands over the flag copy, the draws and the edge samples (~40 us); under the
  // gather that follows, each of its dependent node reads queues behind the gather's L2 traffic (0.06 ms alone, 0.36 ms
  // underneath it).  Measured: fresh step 1.143-1.156 -> 1.128-1.129 ms.  (The draws keep their place in the generator's
  // sequence -- background / edge samples of this step, then the next batch's march noise -- whichever way the next batch is
  // sampled; only the event moves.)
  spec_start_recorded_ = false;
  spec_start_is_consumed_ = false;
  if (train && async_count && (next_batch_.valid || next2_batch_.valid) && speculative_sampling_ != 0) {
    // (ExpRunner::Train with its draws on the tail stream: the main stream has been handed nothing since the previous step's `consumed`,
    // which every side stream waits for at a begin anyway -- that recording IS this point, and the packet is saved)
    if (rays_off_main_ && side_shared_ && side_shared_->seq > 0) spec_start_is_consumed_ = true;
    else spec_start_ev_.record();
    spec_start_recorded_ = true;
  }
  // A prefetch whose kernels were queued by the previous step: only now does the host wait for its count (everything between
  // the end of that step and this point -- the caller's loop, this step's bookkeeping -- overlaps the march).
  auto* ps = static_cast<PersSampler*>(pts_sampler_.get());
  if (train) {
    // (batches in flight for other rays than this step's, the next step's or the one after: void)
    KeepOnlyPending(rays_o, rays_d, next_batch_.valid ? next_batch_.rays_o : Tensor(), next_batch_.valid ? next_batch_.rays_d : Tensor(),
                    next2_batch_.valid ? next2_batch_.rays_o : Tensor(), next2_batch_.valid ? next2_batch_.rays_d : Tensor());
    const int slot = FindPending(rays_o, rays_d);
    if (slot >= 0) PreSampleFinish(slot);
  } else {
    DropPendingSamples();
  }
  bool wait_for_pack = false;
  int pack_slot = 0;
  auto pack_done = [&]() {
    if (wait_for_pack) presample_done_ev_[pack_slot].block(c10::hip::getCurrentHIPStreamMasqueradingAsCUDA());
    wait_for_pack = false;
  };
  if (train && PresampleMatches(rays_o, rays_d)) {  // PreSample[Async]() already marched these rays
    sample_result_ = std::move(presampled_);
    if (presample_async_) {
      // Produced on a side stream, out of that stream's memory pool: order it before this stream.  The allocator is
      // NOT told (record_stream on the seven tensors cost ~45 us of host time when they are released in the middle of the
      // step, right where the device is waiting for the next launch): instead the device's `consumed` event is recorded on this
      // stream once the last kernel that reads them has been queued, and the side streams wait for it before the next
      // kernels that could be handed this memory again (SideWaitConsumed).
      // (the wait itself is issued further down, right before the first kernel that reads the packed samples)
      wait_for_pack = true;
      pack_slot = presample_slot_;
      consumed_side_samples_ = true;
    }
    presampled_ = SampleResultFlex();
    has_presample_ = false;
    presample_rays_o_ = presample_rays_d_ = Tensor();
  } else {
    presampled_ = SampleResultFlex();  // a presample for other rays (or made for training, in a render) is of no use
    has_presample_ = false;
    presample_rays_o_ = presample_rays_d_ = Tensor();
    static_cast<PersSampler*>(pts_sampler_.get())->extra_sample_rows_ = 2 * n_edge_pts_;
    static_cast<PersSampler*>(pts_sampler_.get())->keyed_seq_ = train ? cur_seq_ : -1;
    if (draw_ev_recorded_) draw_ev_.block(c10::hip::getCurrentHIPStreamMasqueradingAsCUDA());  // (rays drawn on the tail stream)
    sample_result_ = pts_sampler_->GetSamples(rays_o, rays_d, bounds);
  }
  // Random draws of the step (Renderer.cpp:67-81 background, PersSampler.cu:456-457 edge samples): ONE uniform launch for both
  // unless a test pinned either (the edge kernel maps three uniforms to an edge index and two coordinates in [-1,1)).
  // A batch that was prefetched brings them along, drawn and turned into edge samples on its side stream behind its pack
  // (PreGenerateStepDraws): the main queue of the step then starts with the hash gather.
  const int n_edge = train ? n_edge_pts_ : 0;
  const bool pregen = train && async_count && sample_result_.step_draws_ready && sample_result_.extra_rows >= 2 * (int64_t) n_edge;
  const bool draw_bg = !pregen && !forced_bg_.defined() && bg_color_type_ == BGColorType::rand_noise && train;
  const bool draw_edge = !pregen && n_edge > 0 && !ps->forced_edge_idx_.defined() && !ps->forced_edge_coords_.defined();
  Tensor bg_color, edge_u, edge_idx, edge_coord;
  if (pregen) bg_color = sample_result_.bg_color;
  if (draw_bg || draw_edge) {
    const int64_t nb = draw_bg ? (int64_t) n_rays * 3 : 0, ne = draw_edge ? (int64_t) n_edge * 3 : 0;
    Tensor u = DrawStepUniforms(nb + ne, train ? cur_seq_ : -1);
    if (draw_bg) bg_color = u.narrow(0, 0, nb).view({n_rays, 3});
    if (draw_edge) edge_u = u.narrow(0, nb, ne);
  }
  if (n_edge > 0 && !draw_edge && !pregen) {
    auto& oct = *ps->pers_octree_;
    edge_idx = ps->forced_edge_idx_.defined() ? ps->forced_edge_idx_.contiguous()
                                              : torch::randint(0, oct.n_edges_, {n_edge}, DevI32()).contiguous();
    edge_coord = ps->forced_edge_coords_.defined() ? ps->forced_edge_coords_.contiguous()
                                                   : torch::empty({n_edge, 2}, DevF32()).uniform_(-1.f, 1.f);
  }
  if (!draw_bg && !pregen) {  // Renderer.cpp:67-81
    if (forced_bg_.defined()) bg_color = forced_bg_.contiguous();
    else if (bg_color_type_ == BGColorType::white) bg_color = torch::ones({n_rays, 3}, DevF32());
    else if (bg_color_type_ == BGColorType::rand_noise) bg_color = torch::full({n_rays, 3}, .5f, DevF32());  // (not training)
    else bg_color = torch::zeros({n_rays, 3}, DevF32());
  }

  int n_all_pts = sample_result_.pts.size(0);
  last_n_all_pts_ = n_all_pts;
  // The NEXT batch's intersection and march start now, on the side stream, against the octree as it stands (see Renderer.h):
  // they run underneath this step's GATHER -- the pairing is deliberate: the gather is bound by L2 line traffic and leaves the
  // vector ALUs idle, the march is a latency chain of vector instructions.  (Started behind the gather instead, so that it
  // would not share the L2s with it, the sampler lands on the VALU-bound MLP / scatter kernels: measured 1.178 -> 1.245 ms.)
  // Which batches: the next step's, unless it is in flight already (two-deep pipeline: it was begun one step ago), and the
  // one behind it (next2_batch_, spec_depth_ >= 2).  A batch begun now is void if a ProcOctree runs before it is consumed --
  // in this step's update for the next batch, in this or the next step's for the one behind it -- so it is not begun then.
  // WHEN (measured, profiles/r04_pipeline_experiments.txt): a batch begun one step ahead pays while its chain fits underneath
  // the step's gather -- a young scene: the walk runs out of LDS, rays are of one length -- and while no leaf dies (mode 2);
  // a batch begun TWO steps ahead, marched by a small persistent grid, pays once the tree has outgrown the LDS walk (the
  // chain is then as long as the step) and costs on a young scene (it spreads the march from underneath the L2-bound gather
  // over the VALU-bound kernels of the whole step: fresh step 1.13 -> 1.17-1.24 ms).
  const bool quiet = ps->pers_octree_->QuietEpochs() >= kSpecQuietEpochs;
  const bool big_tree = ps->pers_octree_->n_interior_ > ps->LdsWalkMaxInterior();
  const bool two_deep = spec_depth_ >= 3 || (spec_depth_ == 2 && big_tree);
  // (data-parallel replicas speculate like a single GPU: the deaths a repair looks for are stamped by the stat update, which
  // runs behind the occupancy exchange and is therefore the same on every rank)
  const bool spec_base = train && async_count && speculative_sampling_ != 0 && n_all_pts > 0;
  auto spec_begin = [&](const NextBatch& nb, int ahead) {
    if (!nb.valid || FindPending(nb.rays_o, nb.rays_d) >= 0) return;
    const int slot = FreePendingSlot();
    const bool now = ahead >= 1 ? two_deep : (speculative_sampling_ == 1 || quiet || two_deep);
    if (!spec_base || !now || slot < 0 || ps->MaintenanceDue(ahead)) {
      n_spec_fallback_++;
      return;
    }
    ps->persistent_march_ = ahead >= 1;  // (two steps to finish in: a few hundred resident waves do it)
    PreSampleSpecBegin(slot, nb.rays_o, nb.rays_d, nb.fineness, nb.seq);
    ps->persistent_march_ = false;
    n_speculative_++;
  };
  if (train) {
    spec_begin(next_batch_, 0);
    spec_begin(next2_batch_, 1);
  }
  spec_start_recorded_ = false;
  if (train) total_all_pts_ += n_all_pts;
  if (train) DigestBegin(n_rays, n_all_pts);
  if (train) gdp->sampled_pts_per_ray_ = gdp->sampled_pts_per_ray_ * 0.9f + (float(n_all_pts) / float(n_rays)) * 0.1f;

  RenderFront fr;
  fr.bg_color = bg_color;
  if (n_all_pts <= 0) {  // Renderer.cpp:83-97
    pack_done();
    // data-parallel replicas must all take part in the occupancy exchange, also the one whose batch missed the scene
    const bool dp_lagged = train && dp_world_ > 1 && async_count;  // (see ResolvePendingCount)
    if (train && dp_world_ > 1 && !dp_lagged) dp_count_ = torch::zeros({1}, DevI32());
    if (train && static_cast<PersSampler*>(pts_sampler_.get())->occupancy_sync_hook_)
      pts_sampler_->UpdateOctNodes(sample_result_, torch::empty({0}, DevF32()), torch::empty({0}, DevF32()));
    if (dp_lagged) {
      // the exchange has just summed the PREVIOUS step's count over the ranks: a scan over no rays drops it (and this step's
      // count, zero) into the mapped words, exactly as the survivor scan of a step with samples does
      Tensor total = torch::empty({1}, DevI32());
      n_kept_words_.Ensure(2);
      if (dp_count_.defined()) {
        F2N_CALL(f2n_segment_scan_ex(CurStream(), 0, nullptr, nullptr, I32P(total), n_kept_words_.Dev(0), I32P(dp_count_), 1));
        dp_sum_mirrored_ = true;
        dp_sum_rays_ = dp_count_rays_;
      } else {
        F2N_CALL(f2n_segment_scan_ex(CurStream(), 0, nullptr, nullptr, I32P(total), n_kept_words_.Dev(1), nullptr, 0));
      }
      kept_wait_ev_ = &n_kept_ev_;
      n_kept_ev_.record();
      dp_count_ = total;
      dp_count_rays_ = n_rays;
      count_pending_ = true;
      pending_count_rays_ = n_rays;
      pending_count_seq_ = cur_seq_;
    } else if (train && dp_world_ > 1) {
      if (!dp_count_host_.defined()) dp_count_host_ = torch::empty({1}, torch::TensorOptions().dtype(torch::kInt32).pinned_memory(true));
      dp_count_host_.copy_(dp_count_, /*non_blocking=*/true);
      dp_count_ev_.record();
    }
    if (train && !dp_lagged) {
      gdp->meaningful_sampled_pts_per_ray_ = gdp->meaningful_sampled_pts_per_ray_ * 0.9f + KeptPerRayForEma(0, n_rays) * 0.1f;
      RecordEma(cur_seq_);
      DigestKept(cur_seq_, 0);
    }
    last_n_kept_pts_ = 0;
    fr.empty = true;
    consumed_side_samples_ = false;  // (nothing was read from the side stream's buffers)
    octree_ready_ev_.record();
    return fr;
  }
  void* st = CurStream();

  // ---- no-grad pre-pass: density of every sample, early stop (Renderer.cpp:105-137) ----
  SampleResultFlex& es = fr.es;
  Tensor &pts_all = fr.pts_all, &vol_all = fr.vol_all, &src_rows = fr.src_rows;
  int n_kept = 0;
  // Streaming step: the 2E edge samples ride through the density pre-pass IN FRONT of the ray samples (their hash features
  // are then in the cache the grad pass reads; a separate gather + MLP for 2% of the rows cost three launches on the
  // critical path).  They are generated into the spare front rows of the sampler's arrays and into the head of pts_all /
  // vol_all -- by a launch that does not wait for the pack of the ray samples, which is usually still running.
  const int64_t front = 2 * (int64_t) n_edge;
  const bool edges_cached = async_count && n_edge > 0 && sample_result_.extra_rows >= front &&
                            sample_result_.pts.storage_offset() >= 3 * front && sample_result_.anchors.storage_offset() >= 3 * front;
  auto edge_samples_to = [&](float* pts1, int32_t* idx1, int stride1, float* pts2, int32_t* idx2) {
    auto& oct = *ps->pers_octree_;
    F2N_CALL(f2n_edge_samples_ex(st, n_edge, VoidP(oct.edge_pool_gpu_), oct.n_edges_, VoidP(oct.pers_trans_gpu_),
                                 edge_idx.defined() ? I32P(edge_idx) : nullptr, edge_coord.defined() ? F32P(edge_coord) : nullptr,
                                 edge_u.defined() ? F32P(edge_u) : nullptr, pts1, idx1, stride1, pts2, idx2, 1));
  };
  {
    torch::NoGradGuard no_grad;
    Tensor f0_full;
    const float* f0p = nullptr;
    if (edges_cached) {
      const int64_t n_rows = (int64_t) n_all_pts + front;
      Tensor pts_full = sample_result_.pts.as_strided({n_rows, 3}, {3, 1}, sample_result_.pts.storage_offset() - 3 * front);
      Tensor anchors_full = sample_result_.anchors.as_strided({n_rows, 3}, {3, 1}, sample_result_.anchors.storage_offset() - 3 * front);
      if (pregen) {  // (already generated into the front rows and into these two arrays, on the batch's side stream)
        pts_all = sample_result_.pts_all.narrow(0, 0, n_rows);
        vol_all = sample_result_.vol_all.narrow(0, 0, n_rows);
      } else {
        pts_all = torch::empty({n_rows, 3}, DevF32());
        vol_all = torch::empty({n_rows}, DevI32());
        edge_samples_to(F32P(pts_full), I32P(anchors_full), 3, F32P(pts_all), I32P(vol_all));
      }
      pack_done();
      if (digest_taps_ && train) {
        DigestTap(TAP_PTS_PRE, sample_result_.pts);
        DigestTap(TAP_DT_PRE, sample_result_.dt);
        DigestTap(TAP_ANCHORS_PRE, sample_result_.anchors.select(1, 0));
      }
      f0_full = field->QueryDensityPreAct(pts_full, anchors_full, /*keep_features=*/true);
      f0p = F32P(f0_full) + front;
      if (digest_taps_ && train) DigestTap(TAP_EDGE, pts_full.narrow(0, 0, front));
      fr.edge_cache_row = 0;
      fr.sample_cache_row = front;
    } else {
      // the pre-pass keeps the hash features it gathers: the grad pass below reuses them for the surviving samples
      pack_done();
      if (digest_taps_ && train) {
        DigestTap(TAP_PTS_PRE, sample_result_.pts);
        DigestTap(TAP_DT_PRE, sample_result_.dt);
        DigestTap(TAP_ANCHORS_PRE, sample_result_.anchors.select(1, 0));
      }
      f0_full = field->QueryDensityPreAct(sample_result_.pts, sample_result_.anchors, /*keep_features=*/true);
      f0p = F32P(f0_full);
    }
    if (digest_taps_ && train) {
      DigestTap(TAP_PTS, sample_result_.pts);
      DigestTap(TAP_DT, sample_result_.dt);
      DigestTap(TAP_ANCHORS, sample_result_.anchors.select(1, 0));
      DigestTap(TAP_F0, f0_full);
      DigestTap(TAP_BG, bg_color);
    }
    Tensor weights = torch::empty({n_all_pts}, DevF32()), alphas = torch::empty({n_all_pts}, DevF32());
    Tensor mask = torch::empty({n_all_pts}, DevI32()), kept = torch::empty({n_rays}, DevI32());
    // training: the occupancy votes (first half of UpdateOctNodes, Renderer.cpp:140-149) ride in the early-stop launch
    if (train) ps->EarlyStopAndVote(sample_result_, f0p, weights, alphas, mask, kept);
    else F2N_TIMED_CALL("early_stop", f2n_early_stop(st, n_rays, I32P(sample_result_.pts_idx_bounds), f0p, 1, F32P(sample_result_.dt),
                            F32P(weights), F32P(alphas), I32P(mask), I32P(kept)));
    // The occupancy update (Renderer.cpp:140-149) is all the NEXT batch's sampling waits for, and that sampling is the longer
    // of the two chains of a converged step: a streaming step issues the update -- and, behind it, the prefetch on the side
    // stream -- before the survivor scan, whose result nothing waits for.  (Data-parallel: the survivor count rides in the
    // occupancy exchange, so the scan stays first; synchronous steps: the host is about to wait for the count.)
    const bool octree_first = train && async_count;
    auto octree_update_issued = [&]() {
      octree_ready_ev_.record();  // everything the NEXT step's ray sampling depends on has been issued ...
      // ... so the speculatively sampled batch of the next step is repaired and packed now (a batch for the step after it
      // stays as it is: it is repaired behind the NEXT stat update, against every death since it was walked)
      const int nslot = next_batch_.valid ? FindPending(next_batch_.rays_o, next_batch_.rays_d) : -1;
      if (nslot >= 0 && pend_[nslot].s.speculative && !pend_[nslot].s.completed && PreSampleSpecComplete(nslot)) after_octree_update_ = nullptr;
      else if (nslot >= 0 && pend_[nslot].s.completed) after_octree_update_ = nullptr;  // (already prefetched the ordinary way)
      if (after_octree_update_) {  // ... or a prefetching TrainStep starts that sampling now (draw order: bg + edge, noise)
        auto f = std::move(after_octree_update_);
        after_octree_update_ = nullptr;
        f();
      }
    };
    Tensor new_se = torch::empty({n_rays, 2}, DevI32()), total = torch::empty({1}, DevI32());
    n_kept_words_.Ensure(2);  // [0] data-parallel: the previous step's count summed over the ranks; [1] this step's count
    bool scan_issued = false;
    if (octree_first) {
      // the survivor scan rides in the stat update's launch (f2n_oct_update_stats_scan: one dependent launch less on the main queue)
      PersSampler::ScanArgs sa;
      sa.n = n_rays; sa.counts = I32P(kept); sa.start_end = I32P(new_se); sa.total = I32P(total);
      if (train && dp_world_ > 1 && dp_count_.defined()) {  // (see below: the ranks' sum of the previous count, dropped into word 0)
        sa.mirror = n_kept_words_.Dev(0); sa.also = I32P(dp_count_); sa.n_also = 1;
        dp_sum_mirrored_ = true;
        dp_sum_rays_ = dp_count_rays_;
      } else {
        sa.mirror = n_kept_words_.Dev(1);
      }
      ps->FinishOctUpdate(&sa);
      scan_issued = true;
      octree_update_issued();
    }
    // Second (and last) host read-back of a Render call: M, the number of surviving samples.  The scan writes it to mapped
    // host memory as well (FilterIdxBounds, Renderer.cu:20-50); the host reads it behind an event, not a stream drain, so that
    // the work below that does not depend on M (the occupancy update, the edge samples) is already queued and runs while the
    // host wakes up -- and no copy launch sits between the scan and the compaction.
    const bool dp_lagged = train && dp_world_ > 1 && octree_first;
    if (scan_issued) {
      // (issued with the stat update above)
    } else if (dp_lagged && dp_count_.defined()) {
      // (dp_count_ went through this step's occupancy exchange a moment ago: it now holds the ranks' sum of the previous count)
      F2N_CALL(f2n_segment_scan_ex(st, n_rays, I32P(kept), I32P(new_se), I32P(total), n_kept_words_.Dev(0), I32P(dp_count_), 1));
      dp_sum_mirrored_ = true;
      dp_sum_rays_ = dp_count_rays_;
    } else {
      F2N_CALL(f2n_segment_scan_ex(st, n_rays, I32P(kept), I32P(new_se), I32P(total), n_kept_words_.Dev(1), nullptr, 0));
    }
    // (octree_first: octree_ready_ev_ was recorded behind the launch that holds the scan -- the count hides behind that recording)
    kept_wait_ev_ = scan_issued ? &octree_ready_ev_ : &n_kept_ev_;
    if (!scan_issued) n_kept_ev_.record();
    if (digest_taps_ && train) DigestTap(TAP_SURVIVORS, new_se);
    if (dp_lagged) {  // this step's count: summed over the ranks inside the NEXT step's occupancy exchange (in place)
      dp_count_ = total;
      dp_count_rays_ = n_rays;
    } else if (train && dp_world_ > 1) {
      dp_count_ = total.clone();  // synchronous steps: summed inside this step's exchange, read right away
    }
    if (train && !octree_first) ps->FinishOctUpdate();  // Renderer.cpp:140-149 (the votes were cast above)
    if (train && dp_world_ > 1 && !dp_lagged) {
      if (!dp_count_host_.defined()) dp_count_host_ = torch::empty({1}, torch::TensorOptions().dtype(torch::kInt32).pinned_memory(true));
      dp_count_host_.copy_(dp_count_, /*non_blocking=*/true);
      dp_count_ev_.record();
    }
    if (train && !octree_first) octree_update_issued();
    if (async_count) {
      // streaming step: the count stays on the device; n_kept is the capacity every buffer below is sized for
      n_kept = n_all_pts;
      if (train) {  // (bookkeeping of the training counters / EMA: Renderer::ResolvePendingCount)
        count_pending_ = true;
        pending_count_rays_ = n_rays;
        pending_count_seq_ = cur_seq_;
      }
      fr.dyn = true;
      fr.n_kept_dev = total;
    } else {
      kept_wait_ev_->synchronize();
      n_kept = n_kept_words_.Read(1);
      if (train && after_count_readback_) after_count_readback_();  // everything queued before this point has finished
      last_n_kept_pts_ = n_kept;
      if (train) total_kept_pts_ += n_kept;
    }
    const int64_t so = fr.dyn ? 2 * (int64_t) n_edge : 0;   // first survivor row of pts_all / vol_all
    const int64_t eo = fr.dyn ? 0 : n_kept;                 // first edge-sample row
    if (!edges_cached) {
      pts_all = torch::empty({n_kept + 2 * n_edge, 3}, DevF32());
      vol_all = torch::empty({n_kept + 2 * n_edge}, DevI32());
    }
    es.pts = pts_all.slice(0, so, so + n_kept);
    es.dirs = torch::empty({n_kept, 3}, DevF32());
    es.dt = torch::empty({n_kept}, DevF32());
    es.t = torch::empty({n_kept}, DevF32());
    es.anchors = torch::empty({n_kept, 3}, DevI32());
    es.first_oct_dis = sample_result_.first_oct_dis;
    es.pts_idx_bounds = new_se;
    src_rows = torch::empty({std::max(n_kept, 1)}, DevI32());
    // CustomOps::ScatterIdx (Renderer.cpp:185) rides along with the compaction: every survivor gets its ray's image index
    const bool want_emb = train && use_app_emb_ && emb_idx.defined();
    Tensor emb_contig;
    if (want_emb) {
      emb_contig = emb_idx.contiguous();
      CheckDev(emb_contig, torch::kInt32, "emb_idx");
      fr.sample_emb_idx = torch::empty({std::max(n_kept, 1)}, DevI32());
    }
    F2N_TIMED_CALL("compact_samples", f2n_compact_samples_src(st, n_rays, I32P(sample_result_.pts_idx_bounds), I32P(new_se), I32P(mask),
                                 F32P(sample_result_.pts), F32P(sample_result_.dirs), F32P(sample_result_.dt),
                                 F32P(sample_result_.t), I32P(sample_result_.anchors), F32P(pts_all) + 3 * so, F32P(es.dirs),
                                 F32P(es.dt), F32P(es.t), I32P(es.anchors), I32P(src_rows), I32P(vol_all) + so,
                                 want_emb ? I32P(emb_contig) : nullptr, want_emb ? I32P(fr.sample_emb_idx) : nullptr));
    if (train) {
      if (!fr.dyn) {
        gdp->meaningful_sampled_pts_per_ray_ = gdp->meaningful_sampled_pts_per_ray_ * 0.9f + KeptPerRayForEma(n_kept, n_rays) * 0.1f;
        RecordEma(cur_seq_);
        DigestKept(cur_seq_, n_kept);
      }
      // edge samples for the TV loss share the field's point array with the surviving samples (Renderer.cpp:159-166)
      if (!edges_cached && n_edge > 0) edge_samples_to(F32P(pts_all) + 3 * eo, I32P(vol_all) + eo, 1, nullptr, nullptr);
    }
  }

  fr.n_kept = n_kept;
  fr.n_edge = n_edge;
  fr.side_pool_buffers = pregen;
  fr.emb = train && use_app_emb_ && emb_idx.defined();
  if (!fr.emb) fr.sample_emb_idx = torch::empty({0}, DevI32());  // autograd::Function inputs must be defined tensors
  const bool streaming_train = train && fr.dyn;  // (TrainForwardBackward follows and records `consumed` behind its last kernel)
  if (streaming_train && consumed_side_samples_) {
    fr.presamples_keepalive = std::move(sample_result_);
    fr.consumed_deferred = true;
    consumed_side_samples_ = false;
  }
  sample_result_ = SampleResultFlex();  // drop the pre-early-stop buffers
  if (consumed_side_samples_) {         // (see above: every reader of the side stream's sample buffers has been queued)
    side_shared_->consumed.record();
    consumed_side_samples_ = false;
    side_shared_->seq++;
  }
  // everything the NEXT step's ray sampling depends on has been issued (a streaming step recorded this behind its stat update
  // already, and nothing since has touched the tree: no second packet)
  if (!streaming_train) octree_ready_ev_.record();
  return fr;
}

RenderResult Renderer::Render(const Tensor& rays_o, const Tensor& rays_d, const Tensor& bounds, const Tensor& emb_idx) {
  auto* gdp = global_data_pool_;
  auto* field = static_cast<Hash3DAnchored*>(scene_field_.get());
  auto* shader = static_cast<SHShader*>(shader_.get());
  const bool train = gdp->mode_ == RunningMode::TRAIN;
  const int n_rays = rays_o.size(0);
  RenderFront fr = SampleAndFilter(rays_o, rays_d, bounds, emb_idx);
  if (fr.empty)  // Renderer.cpp:83-97
    return {fr.bg_color, torch::zeros({n_rays, 1}, DevF32()), torch::zeros({n_rays}, DevF32()), Tensor(),
            torch::full({n_rays}, 512.f, DevF32()), Tensor(), Tensor()};
  SampleResultFlex& es = fr.es;
  const int n_kept = fr.n_kept, n_edge = fr.n_edge;

  // ---- grad pass (Renderer.cpp:152-208) ----
  Tensor all_feat = field->AnchoredQueryReuse(fr.pts_all, fr.vol_all, fr.src_rows, n_kept);  // [M + 2E, 16]
  Tensor scene_feat = all_feat.slice(0, 0, n_kept);
  Tensor edge_feat;
  if (train) edge_feat = all_feat.slice(0, n_kept, n_kept + 2 * n_edge).reshape({n_edge, 2, -1});
  Tensor sampled_colors = shader->QueryFromField(scene_feat, es.dirs, fr.emb ? app_emb_ : torch::empty({0}, DevF32()),
                                                 fr.sample_emb_idx, fr.emb ? &app_emb_grad_ : nullptr);
  auto out = CompositeFunction::apply(scene_feat, sampled_colors, es.dt, es.t, fr.bg_color, es.pts_idx_bounds,
                                      (double) gdp->gradient_scaling_progress_);
  return {out[0], es.first_oct_dis, out[1], edge_feat, out[2], out[3], es.pts_idx_bounds};
}

// Forward-only rendering (ExpRunner::RenderWholeImage's chunk body, TestImages, RenderPath): the kernels of the streaming
// training step without anything a backward would need -- the survivor count stays on the device (no second host round trip
// per chunk), field MLP on the cached hash features + SH + colour MLP in ONE launch whose only outputs are the density
// pre-activation and the colour of every surviving sample (no `feat`, no saved MLP inputs), then compositing.  No autograd
// nodes, no occupancy update.  Bit-identical to Render() in VALIDATE mode (tests/test_gpu_e2e.py).
RenderResult Renderer::RenderForward(const Tensor& rays_o, const Tensor& rays_d, const Tensor& bounds) {
  auto* gdp = global_data_pool_;
  auto* field = static_cast<Hash3DAnchored*>(scene_field_.get());
  auto* shader = static_cast<SHShader*>(shader_.get());
  TORCH_CHECK(gdp->mode_ != RunningMode::TRAIN, "RenderForward is the inference path");
  if (!FusedPathOk()) return Render(rays_o, rays_d, bounds, Tensor());  // (network shapes without fused kernels: op by op)
  torch::NoGradGuard no_grad;
  const int n_rays = rays_o.size(0);
  RenderFront fr = SampleAndFilter(rays_o, rays_d, bounds, Tensor(), /*async_count=*/true);
  if (fr.empty)
    return {fr.bg_color, torch::zeros({n_rays, 1}, DevF32()), torch::zeros({n_rays}, DevF32()), Tensor(),
            torch::full({n_rays}, 512.f, DevF32()), Tensor(), Tensor()};
  void* st = CurStream();
  SampleResultFlex& es = fr.es;
  const int n_cap = std::max(fr.n_kept, 1);
  TORCH_CHECK(field->prepass_x_.defined(), "no pre-pass feature cache for this query");
  Tensor f0c = torch::empty({n_cap}, DevF32()), rgb = torch::empty({n_cap, 3}, DevF32());
  F2N_TIMED_CALL("field_shade_fwd", f2n_field_shade_fwd_dyn(st, fr.n_kept, I32P(fr.n_kept_dev), I32P(fr.src_rows),
                         static_cast<const void*>(field->prepass_x_.data_ptr<at::Half>() + (int64_t) N_LEVELS * N_CHANNELS * fr.sample_cache_row),
                         VoidP(field->mlp_->params_h_), F32P(es.dirs), nullptr, nullptr, VoidP(shader->mlp_->params_h_), F32P(f0c),
                         nullptr, nullptr, F32P(rgb)));
  field->prepass_x_ = Tensor();
  Tensor colors = torch::empty({n_rays, 3}, DevF32()), disparity = torch::empty({n_rays}, DevF32());
  Tensor depth = torch::empty({n_rays}, DevF32()), weights = torch::empty({n_cap}, DevF32());
  Tensor bg = fr.bg_color.contiguous();
  F2N_TIMED_CALL("composite_fwd", f2n_composite_fwd(st, n_rays, I32P(es.pts_idx_bounds), F32P(f0c), 1, F32P(es.dt), F32P(es.t), F32P(rgb),
                             F32P(bg), F32P(colors), F32P(disparity), F32P(depth), F32P(weights), nullptr));
  return {colors, es.first_oct_dis, disparity, Tensor(), depth, weights, es.pts_idx_bounds};
}

int Renderer::LoadStates(const std::vector<Tensor>& states, int idx) {  // Renderer.cpp:216-224
  for (auto pipe : sub_pipes_) idx = pipe->LoadStates(states, idx);
  torch::NoGradGuard g;
  Tensor e = states[idx++].clone().to(torch::kCUDA).to(torch::kFloat32).contiguous();
  TORCH_CHECK(e.sizes() == app_emb_.sizes(), "app_emb shape mismatch");
  app_emb_.copy_(e);
  return idx;
}

std::vector<Tensor> Renderer::States() {  // Renderer.cpp:226-236
  std::vector<Tensor> ret;
  for (auto pipe : sub_pipes_) {
    auto cur = pipe->States();
    ret.insert(ret.end(), cur.begin(), cur.end());
  }
  ret.push_back(app_emb_.detach());
  return ret;
}

std::vector<ParamGroup> Renderer::OptimParamGroups() {  // Renderer.cpp:238-258
  std::vector<ParamGroup> ret = Pipe::OptimParamGroups();
  ParamGroup g;
  g.name = "app_emb";
  g.param = app_emb_;
  g.grad = app_emb_grad_;
  g.weight_decay = 1e-6f;
  ret.push_back(g);
  return ret;
}

}  // namespace f2n
