// Renderer: the sampling pipeline of a training step (rounds 3-5) -- which batches are begun ahead of the step that consumes them, on
// which side stream, speculatively or behind the stat update, and how they are completed, repaired and handed over.  Nothing here
// changes a result (tests/test_gpu_determinism.py): every schedule yields the batch the reference would have sampled behind
// UpdateOctNodes (src/ExpRunner.cpp:86-93, src/PtsSampler/PersSampler.cu:536-615).  Split out of Renderer.cpp in round 5.
#include "Renderer.h"

#include <hip/hip_runtime_api.h>

#include <cstdio>
#include <cstdlib>
#include <mutex>

namespace f2n {

using torch::autograd::AutogradContext;
using torch::autograd::variable_list;

// A prefetched sampling is identified by the ray tensors it was made for.  The renderer HOLDS those tensors: an address
// alone can be recycled by the allocator for other rays (a test image rendered right after training picked up the
// samples prefetched for the next training batch that way, once in ~10 runs).
bool Renderer::PresampleMatches(const Tensor& rays_o, const Tensor& rays_d) const {
  return has_presample_ && presample_rays_o_.defined() && presample_rays_d_.defined() &&
         rays_o.data_ptr() == presample_rays_o_.data_ptr() && rays_d.data_ptr() == presample_rays_d_.data_ptr() &&
         rays_o.sizes() == presample_rays_o_.sizes();
}

void Renderer::PreSample(const Tensor& rays_o, const Tensor& rays_d, const Tensor& bounds) {
  if (PresampleMatches(rays_o, rays_d)) return;  // already marched (asynchronously) for these rays
  const int slot = FindPending(rays_o, rays_d);
  if (slot >= 0) {          // ... or being marched
    PreSampleFinish(slot);
    if (PresampleMatches(rays_o, rays_d)) return;
  }
  if (draw_ev_recorded_) draw_ev_.block(c10::hip::getCurrentHIPStreamMasqueradingAsCUDA());  // (rays drawn on the tail stream)
  static_cast<PersSampler*>(pts_sampler_.get())->extra_sample_rows_ = 2 * n_edge_pts_;
  static_cast<PersSampler*>(pts_sampler_.get())->keyed_seq_ = cur_seq_;
  presampled_ = pts_sampler_->GetSamples(rays_o, rays_d, bounds);
  has_presample_ = true;
  presample_async_ = false;
  presample_rays_o_ = rays_o;
  presample_rays_d_ = rays_d;
}

#if F2N_DEBUG_BUILD
// F2N_DEBUG_SIDE_DELAY="begin_us:complete_us:main_us:period" (debugging aid, off by default): every period-th speculative begin /
// completion / step is preceded by a spin kernel of that many microseconds on its stream (f2n_debug_spin), which skews the sampler's
// side streams against the main stream.  Results must not depend on it (tools/determinism_probe.py --side-delay).
namespace {
struct SideDelay {
  int begin_us = 0, complete_us = 0, main_us = 0, period = 1;
  unsigned pollute = 0;  // != 0: every begin / completion / step is preceded by f2n_debug_pollute with a value derived from it
  uint64_t calls[3] = {0, 0, 0};
  SideDelay() {
    const char* e = std::getenv("F2N_DEBUG_SIDE_DELAY");
    if (e != nullptr) std::sscanf(e, "%d:%d:%d:%d", &begin_us, &complete_us, &main_us, &period);
    if (period < 1) period = 1;
  }
  void Apply(int which) {
    const int us = which == 0 ? begin_us : which == 1 ? complete_us : main_us;
    const uint64_t call = calls[which]++;
    if (pollute != 0) {
      F2N_CALL(f2n_debug_pollute(CurStream(), pollute * 2654435761u + (unsigned) call * 3u + (unsigned) which));
      if (which == 2) {  // ... and a co-tenant's worth of them beside the step, on a stream nothing is ordered against
        static c10::hip::HIPStreamMasqueradingAsCUDA other = c10::hip::getStreamFromPoolMasqueradingAsCUDA();
        for (unsigned k = 0; k < 6; k++) F2N_CALL(f2n_debug_pollute((void*) other.stream(), pollute * 40503u + (unsigned) call * 7u + k));
      }
    }
    if (us > 0 && (call % (uint64_t) period) == 0) F2N_CALL(f2n_debug_spin(CurStream(), us));
  }
};
SideDelay& DebugSideDelay() {
  static SideDelay d;
  return d;
}
}  // namespace

void Renderer::SetDebugSideDelay(int begin_us, int complete_us, int main_us, int period, unsigned pollute) {
  auto& d = DebugSideDelay();
  d.pollute = pollute;
  d.begin_us = begin_us;
  d.complete_us = complete_us;
  d.main_us = main_us;
  d.period = period < 1 ? 1 : period;
}
void Renderer::DebugSkew(int which) { DebugSideDelay().Apply(which); }
#else
void Renderer::DebugSkew(int) {}  // (the product's step has no debugging hooks: host/Common.h F2N_DEBUG_BUILD)
#endif

// The two side streams are per DEVICE, not per Renderer: a process that builds a second runner (bench.py: the headline runner, then
// the converged leg's) would otherwise hold five streams -- main + 2 + 2 -- and HIP multiplexes streams onto four hardware queues
// by default: the second runner's sampler then shared a queue with its own main stream (measured: 20 000 iterations 17.6 s
// in bench.py against 15.2 s for the same loop in a process of its own; profiles/r04_pipeline_experiments.txt item 9).
void Renderer::EnsureSideStream(int slot) {
  if (side_[slot]) return;
  static std::mutex mu;
  static std::shared_ptr<SideShared> shared[16];
  const int dev = c10::hip::current_device();
  TORCH_CHECK(dev >= 0 && dev < 16, "device index out of range");
  std::lock_guard<std::mutex> lock(mu);
  if (!shared[dev]) shared[dev] = std::make_shared<SideShared>();
  // (the sampler's streams at LOW queue priority -- hipStreamCreateWithPriority -- were measured in round 6: nothing, 0.702-0.708 against
  // 0.699-0.708 ms per converged step, 1.136 against 1.129 ms fresh: profiles/r06_backward_grid_ab.txt)
  if (!shared[dev]->stream[slot])
    shared[dev]->stream[slot] = std::make_shared<c10::hip::HIPStreamMasqueradingAsCUDA>(c10::hip::getStreamFromPoolMasqueradingAsCUDA());
  side_shared_ = shared[dev];
  side_[slot] = shared[dev]->stream[slot];
}

// The stream the step's tail chain (deferred reductions -> finiteness flags -> Adam of the small groups) runs on beside the scatter's
// producers: per device like the sampler's two (and the fourth stream of a single-GPU process: HIP's hardware queues are four).
c10::hip::HIPStreamMasqueradingAsCUDA* Renderer::TailStream() {
  EnsureSideStream(0);  // (side_shared_)
  static std::mutex mu;
  std::lock_guard<std::mutex> lock(mu);
  // (high priority: its three small kernels run beside the scatter's 1024 producer blocks and must not queue behind them -- the
  // owners wait for the flags; measured at default priority: the 9 us reduction took 67 us, profiles/r06_converged_timeline.txt)
  if (!side_shared_->tail)
    side_shared_->tail = std::make_shared<c10::hip::HIPStreamMasqueradingAsCUDA>(c10::hip::getStreamFromPoolMasqueradingAsCUDA(/*isHighPriority=*/true));
  return side_shared_->tail.get();
}

c10::hip::HIPStreamMasqueradingAsCUDA* Renderer::BeginDraw() {
  auto* tail = TailStream();
  if (side_shared_->seq > 0) side_shared_->consumed.block(*tail);  // (the pool's blocks: their last readers on the main stream)
  return tail;
}

void Renderer::EndDraw() {
  draw_ev_.record(*TailStream());
  draw_ev_recorded_ = true;
}

void Renderer::ConsumedBehindStep(uint64_t seq_before) {
  EnsureSideStream(0);
  if (side_shared_->seq != seq_before) return;
  side_shared_->consumed.record();
  side_shared_->seq++;
}

// A side stream's buffers may be handed the memory of samples the main stream is still reading: it waits for the last
// recording of the device's `consumed` event it has not waited for yet.
void Renderer::SideWaitConsumed(int slot) {
  auto& sh = *side_shared_;
  if (sh.waited[slot] == sh.seq) return;
  sh.consumed.block(*side_[slot]);
  sh.waited[slot] = sh.seq;
}

void Renderer::PreSampleAsync(const Tensor& rays_o, const Tensor& rays_d, const Tensor& bounds) {
  PreSampleBegin(rays_o, rays_d, bounds, global_data_pool_->ray_march_fineness_);
  const int slot = FindPending(rays_o, rays_d);
  if (slot >= 0) PreSampleFinish(slot);
}

// First half of the prefetch: everything up to the sample counts, issued on a side stream without blocking the host.
// Called from inside SampleAndFilter as soon as this step's occupancy update (the only thing the next batch's sampling
// depends on) has been issued; the kernels then run underneath this step's forward/backward.
void Renderer::PreSampleBegin(const Tensor& rays_o, const Tensor& rays_d, const Tensor& /*bounds*/, float fineness, int64_t seq) {
  if (FindPending(rays_o, rays_d) >= 0) return;  // (already in flight for these rays)
  int slot = FreePendingSlot();
  if (slot < 0) {  // both slots hold batches for other rays: the one begun last is the furthest ahead, and goes
    slot = kPendingSlots - 1;
    DropPendingSlot(slot);
  }
  EnsureSideStream(slot);
  octree_ready_ev_.block(*side_[slot]);  // the only dependency on this step: its occupancy update / ProcOctree
  if (draw_ev_recorded_) draw_ev_.block(*side_[slot]);  // (rays drawn on the tail stream)
  SideWaitConsumed(slot);
  c10::hip::HIPStreamGuardMasqueradingAsCUDA guard(*side_[slot]);
  auto* ps = static_cast<PersSampler*>(pts_sampler_.get());
  ps->extra_sample_rows_ = 2 * n_edge_pts_;
  pend_[slot].seq = seq;
  ps->BeginSamples(rays_o, rays_d, fineness, pend_[slot].s, /*speculative=*/false, seq);  // ... up to and including the pack
  if (global_data_pool_->mode_ == RunningMode::TRAIN) PreGenerateStepDraws(slot);
  presample_done_ev_[slot].record(*side_[slot]);
  pend_[slot].rays_o = rays_o;
  pend_[slot].rays_d = rays_d;
}

// Speculative variant of PreSampleBegin: intersection + march only, NOT ordered behind this step's stat update.
void Renderer::PreSampleSpecBegin(int slot, const Tensor& rays_o, const Tensor& rays_d, float fineness, int64_t seq) {
  EnsureSideStream(slot);
  // Everything the main stream has been handed so far comes first: the kernels that DRAW the next batch's rays
  // (Dataset::RandRaysData, queued by ExpRunner::Train right before this step) and whatever touched the tree there -- i.e. the
  // speculative sampling starts when this step's own kernels start, not before.  (Waiting only for the previous step's octree
  // update -- as a first version did -- let the side stream read ray buffers that were still to be written whenever the host
  // ran ahead of the device: PSNR fell and octrees blew up at random, worst with a second process on the GPU.)
  if (!spec_start_recorded_) {  // (else: recorded at the top of this step, ahead of its random draws)
    spec_start_ev_.record();
    spec_start_recorded_ = true;  // (a second batch begun in the same step waits for the same point)
    spec_start_is_consumed_ = false;
  }
  if (!(spec_start_recorded_ && spec_start_is_consumed_)) spec_start_ev_.block(*side_[slot]);
  if (draw_ev_recorded_) draw_ev_.block(*side_[slot]);  // (rays drawn on the tail stream)
  SideWaitConsumed(slot);
  c10::hip::HIPStreamGuardMasqueradingAsCUDA guard(*side_[slot]);
  auto* ps = static_cast<PersSampler*>(pts_sampler_.get());
  ps->extra_sample_rows_ = 2 * n_edge_pts_;
  DebugSkew(0);
  ps->BeginSamples(rays_o, rays_d, fineness, pend_[slot].s, /*speculative=*/true, seq);
  pend_[slot].seq = seq;
  pend_[slot].rays_o = rays_o;
  pend_[slot].rays_d = rays_d;
}

bool Renderer::TwoDeepRegime() {
  auto* ps = static_cast<PersSampler*>(pts_sampler_.get());
  const bool big_tree = ps->pers_octree_->n_interior_ > ps->LdsWalkMaxInterior();
  return speculative_sampling_ != 0 && (spec_depth_ >= 3 || (spec_depth_ == 2 && big_tree));
}

void Renderer::SpecBeginAtStepEnd() {
  auto* ps = static_cast<PersSampler*>(pts_sampler_.get());
  auto* gdp = global_data_pool_;
  const NextBatch& nb = next2_batch_;
  if (!nb.valid || gdp->mode_ != RunningMode::TRAIN || speculative_sampling_ == 0 || FindPending(nb.rays_o, nb.rays_d) >= 0) return;
  const bool big_tree = ps->pers_octree_->n_interior_ > ps->LdsWalkMaxInterior();
  if (spec_depth_ >= 3 || (spec_depth_ == 2 && big_tree)) return;  // (two-deep regime: begun at the top of this step already)
  const bool quiet = ps->pers_octree_->QuietEpochs() >= kSpecQuietEpochs;
  const int slot = FreePendingSlot();
  if (!(speculative_sampling_ == 1 || quiet) || slot < 0 || ps->MaintenanceDueAt(gdp->iter_step_ + 1)) return;
  spec_start_recorded_ = false;  // (the side stream starts behind what this step has queued so far)
  PreSampleSpecBegin(slot, nb.rays_o, nb.rays_d, nb.fineness, nb.seq);
  spec_start_recorded_ = false;
  n_speculative_++;
}

// ... and its completion, called with this step's stat update issued (octree_ready_ev_ recorded): repair, scan, count, pack.
bool Renderer::PreSampleSpecComplete(int slot) {
  auto& pb = pend_[slot];
  TORCH_CHECK(pb.s.active && pb.s.speculative && !pb.s.completed, "no speculative sampling in flight");
  octree_ready_ev_.block(*side_[slot]);
  SideWaitConsumed(slot);
  c10::hip::HIPStreamGuardMasqueradingAsCUDA guard(*side_[slot]);
  auto* ps = static_cast<PersSampler*>(pts_sampler_.get());
  ps->extra_sample_rows_ = 2 * n_edge_pts_;
  DebugSkew(1);
  if (!ps->CompleteSpeculative(pb.s)) {
    n_spec_dropped_++;
    pb = PendingBatch();  // (its kernels are ordered on the side stream, whose pool its buffers return to)
    return false;
  }
  if (global_data_pool_->mode_ == RunningMode::TRAIN) PreGenerateStepDraws(slot);
  presample_done_ev_[slot].record(*side_[slot]);
  return true;
}

// The draws of the step that will consume pend_[slot] (random background, 2E edge samples) and the edge-sample launch itself,
// queued on that slot's side stream right behind its pack: the packed arrays (front rows) and worst-case-sized pts_all / vol_all
// are their homes.  Only when nothing is pinned by a test and the background is random (the training configuration).
void Renderer::PreGenerateStepDraws(int slot) {
  auto& pb = pend_[slot];
  auto* ps = static_cast<PersSampler*>(pts_sampler_.get());
  const int n_edge = n_edge_pts_, n_rays = pb.s.n_rays;
  const int64_t front = 2 * (int64_t) n_edge;
  if (n_edge <= 0 || forced_bg_.defined() || bg_color_type_ != BGColorType::rand_noise || ps->forced_edge_idx_.defined() ||
      ps->forced_edge_coords_.defined() || !pb.s.o_pts.defined() || pb.s.extra_rows < front || ps->pers_octree_->n_edges_ <= 0)
    return;
  c10::hip::HIPStreamGuardMasqueradingAsCUDA guard(*side_[slot]);
  const int64_t nb = (int64_t) n_rays * 3, ne = (int64_t) n_edge * 3;
  Tensor u = DrawStepUniforms(nb + ne, pb.seq);
  pb.bg_color = u.narrow(0, 0, nb).view({n_rays, 3});
  const int64_t rows = pb.s.s_dt.numel() + front;  // every ray's slots full: the survivors can never be more
  pb.pts_all = torch::empty({rows, 3}, DevF32());
  pb.vol_all = torch::empty({rows}, DevI32());
  auto& oct = *ps->pers_octree_;
  F2N_CALL(f2n_edge_samples_ex(CurStream(), n_edge, VoidP(oct.edge_pool_gpu_), oct.n_edges_, VoidP(oct.pers_trans_gpu_), nullptr, nullptr,
                               F32P(u) + nb, F32P(pb.s.o_pts), I32P(pb.s.o_anchors), 3, F32P(pb.pts_all), I32P(pb.vol_all), 1));
  pb.step_draws_ready = true;
}

// Second half: wait for the counts (by now the march has usually finished) and take views of the packed rows.  No launch.
void Renderer::PreSampleFinish(int slot) {
  auto& pb = pend_[slot];
  TORCH_CHECK(pb.s.active, "PreSampleFinish without PreSampleBegin");
  // samples marched against a tree that has since been replaced or re-numbered (LoadStates / InstallOctree / ProcOctree between
  // the prefetch and its use) are void, whichever way they were prefetched: the caller samples again
  if (pb.s.generation != static_cast<PersSampler*>(pts_sampler_.get())->pers_octree_->generation_) {
    DropPendingSlot(slot);
    return;
  }
  if (!pb.s.completed && !PreSampleSpecComplete(slot)) return;  // (a speculative batch whose step never reached its update)
  presampled_ = static_cast<PersSampler*>(pts_sampler_.get())->FinishSamples(pb.s);
  presampled_.step_draws_ready = pb.step_draws_ready;
  presampled_.bg_color = pb.bg_color;
  presampled_.pts_all = pb.pts_all;
  presampled_.vol_all = pb.vol_all;
  has_presample_ = true;
  presample_async_ = true;
  presample_slot_ = slot;
  presample_rays_o_ = pb.rays_o;
  presample_rays_d_ = pb.rays_d;
  pb = PendingBatch();
}

}  // namespace f2n
