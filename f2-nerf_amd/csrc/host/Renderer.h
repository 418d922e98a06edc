// SHShader + Renderer host side (mirror src/Shader/SHShader.h, src/Renderer/Renderer.h).
#pragma once
#include "Hash3DAnchored.h"
#include "PersSampler.h"

namespace f2n {

class SHShader : public Shader {
 public:
  explicit SHShader(GlobalDataPool* global_data_pool);
  Tensor Query(const Tensor& feats, const Tensor& dirs) override;  // feats = the 16 shading features, dirs [n,3]
  // The Renderer's path: feats are the RAW field features [n,16]; the "[1 | feat[1:]] + app_emb[img]" assembly of
  // Renderer.cpp:181-187 happens inside the kernel.  sample_emb_idx/app_emb may be undefined.
  Tensor QueryFromField(const Tensor& field_feats, const Tensor& dirs, const Tensor& app_emb, const Tensor& sample_emb_idx,
                        Tensor* app_emb_grad);
  Tensor SHEncode(const Tensor& dirs);
  std::vector<Tensor> States() override;
  std::vector<ParamGroup> OptimParamGroups() override;
  int LoadStates(const std::vector<Tensor>& states, int idx) override;
  void Reset() override;

  std::unique_ptr<FusedMLP> mlp_;
  int d_hidden_, n_hiddens_, degree_;
  // the fused SH + colour-MLP kernels exist for the shipped shader (SH degree 4, 32 -> 64 -> 64 -> 3); any other
  // shader.degree / d_hidden / n_hiddens (confs/shader/sh_shader.yaml, SHShader.cu:51-102) runs op by op as SHShader.cpp:23-29
  bool fused_ok_ = true;
};

struct RenderResult {  // Renderer.h:18-27 of the reference
  Tensor colors;
  Tensor first_oct_dis;
  Tensor disparity;
  Tensor edge_feats;
  Tensor depth;
  Tensor weights;
  Tensor idx_start_end;
};

// Everything Render() knows after sampling, pre-pass, early stop, compaction and the occupancy update.
struct RenderFront {
  bool empty = false;        // no sample at all (Renderer.cpp:83-97)
  SampleResultFlex es;       // surviving samples
  Tensor pts_all, vol_all;   // [M + 2E] surviving samples followed by the edge (TV) samples
  Tensor src_rows;           // row of every surviving sample in the pre-pass feature cache
  Tensor bg_color, sample_emb_idx;
  int n_kept = 0, n_edge = 0;
  bool emb = false;
  // Streaming training steps do not wait for the survivor count: n_kept is then an UPPER BOUND (the marched count, which
  // sizes every buffer), the true count stays on the device (n_kept_dev, consumed by the f2n_*_dyn entry points) and the edge
  // samples come FIRST in pts_all / vol_all so that every row offset is known without it.
  bool dyn = false;
  Tensor n_kept_dev;
  // >= 0: the edge samples went through the density pre-pass; their hash features are rows [edge_cache_row, +2E) of its cache
  int64_t edge_cache_row = -1;
  int64_t sample_cache_row = 0;  // cache row of ray sample 0 (src_rows count from there)
  // pts_all / vol_all / bg_color came out of a side stream's allocator pool (Renderer::PreGenerateStepDraws) and are read by the
  // step's LAST kernels (compositing, the scatter): the device's `consumed` event is recorded once more behind those
  bool side_pool_buffers = false;
  // Streaming training steps (round 6, one event packet less on the main queue): the pre-early-stop sample buffers -- side-stream
  // pool memory as well -- stay alive in here until the step's last kernel has been queued, and ONE recording of the `consumed`
  // event behind that kernel covers them and the buffers above (an event recorded between two kernels of a stream costs the
  // command processor ~5 us: profiles/r06_event_cost.txt).
  SampleResultFlex presamples_keepalive;
  bool consumed_deferred = false;
};

struct TrainOutputs {
  Tensor losses;  // device [8]: loss, color, var, disp, tv, mse, 0, 0 (f2n_train_loss)
  Tensor colors;
  bool has_samples = false;
};

class Renderer : public Pipe {
  enum BGColorType { white, black, rand_noise };

 public:
  Renderer(GlobalDataPool* global_data_pool, int n_images);
#if F2N_DEBUG_BUILD
  // debug variant only (see RendererPrefetch.cpp): spin kernels in front of every period-th speculative begin / completion / step
  static void SetDebugSideDelay(int begin_us, int complete_us, int main_us, int period, unsigned pollute = 0);
#endif
  static void DebugSkew(int which);  // 0 speculative begin, 1 completion, 2 step; a no-op in the product build
  RenderResult Render(const Tensor& rays_o, const Tensor& rays_d, const Tensor& bounds, const Tensor& emb_idx);
  RenderResult RenderForward(const Tensor& rays_o, const Tensor& rays_d, const Tensor& bounds);  // inference: no tape, no count read-back
  // issues the ray sampling of the next SampleAndFilter / TrainForwardBackward call ahead of time (same rays!)
  void PreSample(const Tensor& rays_o, const Tensor& rays_d, const Tensor& bounds);
  // The same on a SIDE stream that only waits for this step's octree update: the sampler kernels (latency-bound: few
  // waves, long dependent chains) then run underneath the remaining forward/backward kernels of the current step.
  void PreSampleAsync(const Tensor& rays_o, const Tensor& rays_d, const Tensor& bounds);
  // seq: the batch's sequence number in the training run (KeyedDraws.h: keys its march noise and its step's draws); < 0: unkeyed
  void PreSampleBegin(const Tensor& rays_o, const Tensor& rays_d, const Tensor& bounds, float fineness, int64_t seq = -1);
  // Batches whose sampling is in flight on a side stream: at most kPendingSlots of them (round 4: the batch the NEXT step
  // consumes, being repaired / packed, and the one behind it, being walked and marched -- see next2_batch_), each on its own
  // stream.  A batch is identified by the ray tensors it was begun for (held: see PresampleMatches).
  static constexpr int kPendingSlots = 2;
  struct PendingBatch {
    PendingSamples s;
    Tensor rays_o, rays_d;
    int64_t seq = -1;               // the batch's sequence number (keys its noise and the draws of the step that consumes it)
    bool step_draws_ready = false;  // PreGenerateStepDraws ran behind this batch's pack
    Tensor bg_color, pts_all, vol_all;
  };
  // the background colours and edge samples of the step that will consume pend_[slot], on that slot's side stream (behind its pack)
  void PreGenerateStepDraws(int slot);
  // The random draws a training step makes for itself (background colours, edge samples: Renderer.cpp:67-81, PersSampler.cu:456-457)
  // are keyed by the step's sequence number (KeyedDraws.h, as the march noise: PersSampler::BeginSamples): step k gets draw k
  // whether it is made at the top of that step or a step earlier on a side stream, once or -- after a dropped batch -- twice.
  KeyedUniforms step_draws_{0xD1B54A32D192ED03ull};
  Tensor DrawStepUniforms(int64_t n, int64_t seq) { return step_draws_.Draw(n, seq); }
  // The sequence number of the batch the running / next training-mode SampleAndFilter consumes: set by ExpRunner around every
  // step (its count of steps taken); -1: unkeyed (Render() called on its own: each call takes the next draw of every purpose).
  int64_t cur_seq_ = -1;
  PendingBatch pend_[kPendingSlots];
  int64_t n_spec_dropped_ = 0;  // batches begun ahead and thrown away (other rays asked for, tree replaced)
  int FindPending(const Tensor& rays_o, const Tensor& rays_d) const {
    for (int i = 0; i < kPendingSlots; i++)
      if (pend_[i].s.active && pend_[i].rays_o.defined() && pend_[i].rays_d.defined() && rays_o.defined() && rays_d.defined() &&
          rays_o.data_ptr() == pend_[i].rays_o.data_ptr() && rays_d.data_ptr() == pend_[i].rays_d.data_ptr() &&
          rays_o.sizes() == pend_[i].rays_o.sizes())
        return i;
    return -1;
  }
  int FreePendingSlot() const {
    for (int i = 0; i < kPendingSlots; i++)
      if (!pend_[i].s.active) return i;
    return -1;
  }
  void PreSampleFinish(int slot);
  bool PreSampleBegun() const { return pend_[0].s.active || pend_[1].s.active; }
  bool PendingMatches(const Tensor& rays_o, const Tensor& rays_d) const { return FindPending(rays_o, rays_d) >= 0; }
  void DropPendingSlot(int i) {
    if (!pend_[i].s.active) return;
    n_spec_dropped_++;
    pend_[i].s.counts_ready.synchronize();  // its kernels may still be running: keep the buffers until they are done
    if (!pend_[i].s.completed && side_[i]) side_[i]->synchronize();  // (a speculative batch has recorded no count event yet)
    pend_[i] = PendingBatch();
  }
  void DropPendingSamples() {
    for (int i = 0; i < kPendingSlots; i++) DropPendingSlot(i);
  }
  // drops every pending batch that was not begun for one of these ray pairs (undefined tensors match nothing)
  void KeepOnlyPending(const Tensor& o0, const Tensor& d0, const Tensor& o1, const Tensor& d1, const Tensor& o2, const Tensor& d2) {
    for (int i = 0; i < kPendingSlots; i++) {
      if (!pend_[i].s.active) continue;
      const int a = FindPending(o0, d0), b = FindPending(o1, d1), c = FindPending(o2, d2);
      if (i != a && i != b && i != c) DropPendingSlot(i);
    }
  }
  std::function<void()> after_octree_update_;  // one-shot: called in SampleAndFilter right after the occupancy update
  // Speculative sampling of the NEXT batch (streaming single-GPU steps): intersection and march are issued on the side stream
  // as soon as THIS batch's samples are packed -- against the octree as it stands, underneath this step's pre-pass -- and
  // repaired behind this step's stat update (PersSampler::CompleteSpeculative).  The sampler's two latency chains (~0.6 ms on
  // a converged scene) then no longer sit between one step's stat update and the next step's pre-pass.  Not used in the
  // iterations that run ProcOctree (node indices change: the batch is sampled after the update, as before).
  // MEASURED (profiles/r03_speculation_experiments.txt): it pays while no leaf dies -- fresh scene 1.254 -> 1.171 ms per step --
  // and costs where leaves die in most steps -- converged scene 0.917 -> 0.959 ms, 20 000 iterations 17.9 -> 19.7 s: the
  // repair's duration is the re-walk and re-march of the LONGEST invalidated ray (~85 us), and it sits on the cycle the
  // sampler left.  Hence mode 2 (the default): speculate only when no leaf has died for kSpecQuietEpochs stat updates, as far
  // as the host can tell from a pinned word the update kernel writes (a timing decision only: samples are identical).
  int speculative_sampling_ = 2;  // 0 never, 1 always (outside ProcOctree iterations), 2 while the octree is quiet
  static constexpr int kSpecQuietEpochs = 8;
  struct NextBatch {
    Tensor rays_o, rays_d;
    float fineness = 1.f;
    int64_t seq = -1;
    bool valid = false;
  } next_batch_, next2_batch_;  // the batch of the next step, and (two-deep pipeline) of the step after it
  // Two-deep pipeline (round 4).  With ONE batch in flight the sampler chain of batch k+1 -- walk, march, repair, scan, pack:
  // 0.65-0.75 ms of latency on a converged scene -- has exactly step k to finish in, and a converged step is no longer than that:
  // the main queue waits for it.  With next2_batch_ handed over as well, batch k+2 is walked and marched during step k (on the
  // other side stream) and repaired + packed behind step k+1's stat update against every death since: its chain has two steps,
  // and the only sampler work between a stat update and the next step's pre-pass is the tail repair + pack of the batch in front.
  // 1: never; 2: once the octree has outgrown the LDS-resident walk (PersSampler::LdsWalkMaxInterior: the chain is then as long
  // as a step); 3: always.
  int spec_depth_ = 2;
  int64_t n_speculative_ = 0, n_spec_fallback_ = 0;  // batches sampled speculatively / sampled after the update instead
  void PreSampleSpecBegin(int slot, const Tensor& rays_o, const Tensor& rays_d, float fineness, int64_t seq);
  // Small trees (one-step-ahead regime): the batch after next is begun when THIS step's backward has been queued, not at the top
  // of the next step -- its noise draw, prologue and LDS-resident walk then run underneath the step's Adam / reduction tail (an
  // HBM stream: the vector units are idle) instead of colliding with the next step's gather, whose long-lived blocks keep a
  // newly dispatched small kernel waiting for a wave slot (measured: 0.29 ms for the 9 k-element noise draw;
  // profiles/r04_fresh_timeline.txt).  The step's own stat update is in the stream already, so the batch is repaired against the
  // next one only, exactly like a batch begun at the top of the next step.
  void SpecBeginAtStepEnd();
  bool PreSampleSpecComplete(int slot);  // false: could not be repaired (tree re-numbered): dropped
  at::cuda::CUDAEvent spec_start_ev_;
  bool spec_start_recorded_ = false;
  // ExpRunner::Train's batch draws OFF the main queue (round 6): Dataset::RandRaysData's one kernel runs on the device's tail stream
  // -- idle at the top of a step -- out of that stream's pool; a batch's rays are read first by its sampling on a side stream (which
  // waits for draw_ev_ at its begin) and two steps later by the main stream, which is ordered behind the draw through that batch's
  // presample_done_ev_ or, when it samples the batch itself, waits for draw_ev_.  The pool's memory is protected like the sampler's:
  // the tail stream waits for `consumed` in front of every draw, and a step of such a loop records `consumed` whatever its path.
  c10::hip::HIPStreamMasqueradingAsCUDA* BeginDraw();
  void EndDraw();
  void ConsumedBehindStep(uint64_t seq_before);  // records `consumed` if the step that has just been queued did not
  uint64_t ConsumedSeq() { return side_shared_ ? side_shared_->seq : 0; }
  at::cuda::CUDAEvent draw_ev_;
  bool draw_ev_recorded_ = false;
  bool rays_off_main_ = false;          // set by ExpRunner::Train around its loop: the main stream starts a step with nothing queued since `consumed`
  bool spec_start_is_consumed_ = false;  // this step's "start" point for speculative begins is the previous step's `consumed` (no event of its own)
  RenderFront SampleAndFilter(const Tensor& rays_o, const Tensor& rays_d, const Tensor& bounds, const Tensor& emb_idx,
                              bool async_count = false);
  // The survivor count of an async SampleAndFilter arrives in pinned memory; the host-side bookkeeping that depends on it
  // (meaningful-samples EMA, counters) is applied here: at the start of the next step, or by ExpRunner::FinishPending.
  void ResolvePendingCount();
  // data-parallel runs (DataParallel.cpp): the survivor count behind the meaningful-samples EMA is summed over the ranks in
  // the occupancy exchange, so that every replica sizes its next ray batch from the same number
  int dp_world_ = 1;
  Tensor dp_count_, dp_count_host_;
  int dp_count_rays_ = 0, dp_sum_rays_ = 0;  // rays behind dp_count_ / behind the sum a scan has mirrored to the host
  bool dp_sum_mirrored_ = false;
  at::cuda::CUDAEvent dp_count_ev_;
  float KeptPerRayForEma(int n_kept_local, int n_rays);
  // The meaningful-samples average AFTER the count of step `seq` went into it, for the last kEmaRing steps: ExpRunner::Train
  // sizes batch j from the average after step j - kBatchSizeLag in EVERY mode (synchronous steps know their count at once,
  // streaming steps one step late, data-parallel streaming steps two: a fixed lag takes the schedule out of the batch sizes).
  static constexpr int kEmaRing = 32;
  struct EmaMark {
    int64_t seq = -1;
    float value = 0.f;
  } ema_ring_[kEmaRing];
  void RecordEma(int64_t seq) {
    if (seq >= 0) ema_ring_[seq % kEmaRing] = {seq, global_data_pool_->meaningful_sampled_pts_per_ray_};
  }
  bool EmaAfter(int64_t seq, float* value) const {
    if (seq < 0 || ema_ring_[seq % kEmaRing].seq != seq) return false;
    *value = ema_ring_[seq % kEmaRing].value;
    return true;
  }
  int64_t pending_count_seq_ = -1;  // the step the pending survivor count belongs to
  // Per-step digest (diagnostics: WHERE do two trainings part?): rays, marched and surviving samples of the last kDigestRing
  // training steps (a whole 20 000-iteration run), by sequence number; table_sum (when ExpRunner::digest_table_ is on): an order-free integer checksum of the
  // f16 table after the step's Adam, on the device until read.
  static constexpr int kDigestRing = 32768;
  struct StepDigest {
    int64_t seq = -1;
    int iter = 0, n_rays = 0, n_marched = 0, n_kept = -1;
  };
  std::vector<StepDigest> digest_;
  void DigestBegin(int n_rays, int n_marched) {
    if (cur_seq_ < 0) return;
    if (digest_.empty()) digest_.resize(kDigestRing);
    digest_[cur_seq_ % kDigestRing] = {cur_seq_, global_data_pool_->iter_step_, n_rays, n_marched, -1};
  }
  void DigestKept(int64_t seq, int n_kept) {
    if (seq >= 0 && !digest_.empty() && digest_[seq % kDigestRing].seq == seq) digest_[seq % kDigestRing].n_kept = n_kept;
  }
  // Debugging taps (ExpRunner::digest_taps): order-free integer checksums of a step's intermediate arrays -- sampler outputs,
  // pre-pass densities, survivor bounds, background, edge samples, colours, gradient buffers, parameters after Adam -- one
  // int64 per (step, tap), on the device until read (ExpRunner.step_digest).  Two trainings that part are bisected to the
  // first array that differs.  One small reduction launch per tap and step: off unless asked for.
  enum { TAP_PTS, TAP_DT, TAP_ANCHORS, TAP_F0, TAP_SURVIVORS, TAP_BG, TAP_EDGE, TAP_COLORS, TAP_TABLE_GRAD, TAP_SMALL_GRADS, TAP_TABLE,
         TAP_FIELD_MLP, TAP_COLOR_MLP, TAP_APP_EMB, TAP_GRAD_BEFORE, TAP_PTS_ALL_AFTER, TAP_VOL_ALL_AFTER, TAP_FIELD_X, TAP_DFEAT,
         // the same arrays IN FRONT of the kernel that consumes them: a tap pair that differs inside one run is an array that changed while
         // (or after) it was read -- a late write from another stream -- and needs no second run to compare with
         TAP_PTS_PRE, TAP_DT_PRE, TAP_ANCHORS_PRE, TAP_PTS_ALL_PRE, TAP_VOL_ALL_PRE, N_TAPS };
  bool digest_taps_ = false;
  Tensor digest_tap_sums_;  // int64 [kDigestRing, N_TAPS]
  void DigestTap(int tap, const Tensor& t);
  bool async_count_ = false;        // set by ExpRunner::TrainStep for streaming steps
  bool count_pending_ = false;
  int pending_count_rays_ = 0;
  int64_t total_kept_pts_ = 0, total_all_pts_ = 0;  // running totals over training-mode calls (resolved counts only)
  // forward + ExpRunner::Train's loss + backward into the gradient buffers, without the autograd tape
  TrainOutputs TrainForwardBackward(const Tensor& rays_o, const Tensor& rays_d, const Tensor& bounds, const Tensor& gt_colors,
                                    const Tensor& emb_idx, float var_w, float disp_w, float tv_w);

  int LoadStates(const std::vector<Tensor>& states, int idx) override;
  std::vector<Tensor> States() override;
  std::vector<ParamGroup> OptimParamGroups() override;
  void ZeroGrad();
  // both networks have the shapes the fused (untaped, streaming) training step and the forward-only render are built for
  bool FusedPathOk() const;

  GlobalDataPool* global_data_pool_;
  std::unique_ptr<PtsSampler> pts_sampler_;
  std::unique_ptr<Field> scene_field_;
  std::unique_ptr<Shader> shader_;
  bool use_app_emb_;
  Tensor app_emb_;       // [n_images, 16]
  Tensor app_emb_grad_;  // fp32, unscaled
  Tensor small_grads_flat_;  // when defined: the flat home of the three small gradient buffers (ExpRunner::FlattenSmallGrads)
  BGColorType bg_color_type_ = BGColorType::rand_noise;
  SampleResultFlex sample_result_, presampled_;
  bool has_presample_ = false, presample_async_ = false;
  int presample_slot_ = 0;  // the side stream (pending slot) an asynchronous presample was produced on
  Tensor presample_rays_o_, presample_rays_d_;  // the rays the presample belongs to (held: see PresampleMatches)
  bool PresampleMatches(const Tensor& rays_o, const Tensor& rays_d) const;
  at::cuda::CUDAEvent octree_ready_ev_, presample_done_ev_[kPendingSlots], n_kept_ev_;
  // which event the pending survivor count hides behind: n_kept_ev_, or -- streaming steps, whose survivor scan rides in the stat
  // update's launch -- the octree event recorded behind that very launch (one packet instead of two)
  at::cuda::CUDAEvent* kept_wait_ev_ = &n_kept_ev_;
  // The two side streams -- and so their allocator pools -- are per DEVICE (EnsureSideStream), and so is the protocol that stands in
  // for record_stream on the sample buffers that cross from a side stream's pool to the main stream: `consumed` is recorded on the
  // main stream once the last reader of such buffers has been queued and awaited by a side stream before its next kernels, once
  // per recording (seq counts the recordings, waited[slot] what each stream has waited for).  One event per device, not per
  // Renderer: memory one Renderer returns to a shared pool may be handed to another Renderer's side-stream kernels (round-4 advisor).
  struct SideShared {
    std::shared_ptr<c10::hip::HIPStreamMasqueradingAsCUDA> stream[kPendingSlots];
    std::shared_ptr<c10::hip::HIPStreamMasqueradingAsCUDA> tail;  // (created with the first fused step tail)
    at::cuda::CUDAEvent consumed;
    uint64_t seq = 0, waited[kPendingSlots] = {0, 0};
  };
  std::shared_ptr<SideShared> side_shared_;
  bool consumed_side_samples_ = false;
  void SideWaitConsumed(int slot);
  bool small_grads_clean_ = false;  // set by ExpRunner::OptimStep (fused zero_grad), consumed by the next ZeroGrad()
  std::function<void()> after_count_readback_;  // ExpRunner: reads the previous step's finiteness flags here (no extra wait)
  // ExpRunner: the step's tail -- finiteness flags, Adam -- as arguments of the field backward's call (f2n_field_bwd_step_tail);
  // false: this step keeps the separate launches.  step_tail_done_: the call has queued them (ExpRunner::EnqueueApply's cue).
  std::function<bool(F2nStepTail*)> step_tail_builder_;
  // two batches in flight (the walk + march of the batch after next begin at the TOP of a step): see spec_depth_
  bool TwoDeepRegime();
  // ExpRunner: called once the step's forward has been queued and before anything of its backward is (loss scales, the step count
  // and the schedule the backward and the optimiser read are settled there: the previous step's finiteness flags)
  std::function<void()> before_backward_;
  bool step_tail_done_ = false;
  c10::hip::HIPStreamMasqueradingAsCUDA* TailStream();  // the device's third side stream (flags + small Adam beside the scatter)
  MappedWords n_kept_words_;  // [1]: the surviving-sample count, written by the survivor scan itself, read behind n_kept_ev_
  std::shared_ptr<c10::hip::HIPStreamMasqueradingAsCUDA> side_[kPendingSlots];  // (shared by every Renderer of the device)
  void EnsureSideStream(int slot);
  Tensor forced_bg_;  // explicit background colours for parity tests (undefined = as the reference)
  int n_edge_pts_ = 8192;
  int last_n_all_pts_ = 0, last_n_kept_pts_ = 0;
};

namespace CustomOps {
Tensor WeightVar(Tensor weights, Tensor idx_start_end);  // CustomOps.cu:12-66 via f2n_weight_var_*
}

}  // namespace f2n
