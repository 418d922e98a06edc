// PersSampler host logic: orchestration of the sampler kernels through the C-ABI, occupancy bookkeeping and
// the octree maintenance (on the device: csrc/octree.hip).  Behaviour follows src/PtsSampler/PersSampler.cu:317-615 and
// src/PtsSampler/PersSampler.cpp:120-330,664-733 of the reference (cited inline); the control flow does not:
// leaf-hit storage is a persistent worst-case workspace, segments are laid out in ray order by a device-side
// scan, and the only host read-back of a GetSamples call is the pair (K, N) at its very end.
#include "PersSampler.h"

#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <cstring>
#include <functional>

namespace f2n {

namespace {

// PersSampler.cpp:106-117 -- child visiting order for each of the 8 ray octants.
Tensor BuildSearchOrder() {
  std::vector<int> search_order;
  for (int st = 0; st < 8; st++) {
    auto cmp = [st](int a, int b) {
      int bt = ((a ^ b) & -(a ^ b));
      return ((a & bt) ^ (st & bt)) != 0;
    };
    for (int i = 0; i < 8; i++) search_order.push_back(i);
    std::sort(search_order.begin() + st * 8, search_order.begin() + (st + 1) * 8, cmp);
  }
  return torch::from_blob(search_order.data(), {64}, CpuI32()).to(torch::kUInt8).to(torch::kCUDA).contiguous();
}

}  // namespace

PersSampler::PersSampler(GlobalDataPool* global_data_pool) {
  global_data_pool_ = global_data_pool;
  global_data_pool_->pts_sampler_ = this;
  const auto& c = global_data_pool->config_;
  sub_div_milestones_ = c.IntList("pts_sampler.sub_div_milestones");
  std::reverse(sub_div_milestones_.begin(), sub_div_milestones_.end());
  compact_freq_ = c.Int("pts_sampler.compact_freq");
  max_oct_intersect_per_ray_ = c.Int("pts_sampler.max_oct_intersect_per_ray");
  global_near_ = c.Float("pts_sampler.near");
  scale_by_dis_ = c.Bool("pts_sampler.scale_by_dis");
  sample_l_ = c.Float("pts_sampler.sample_l");
  pers_octree_ = std::make_unique<PersOctree>();
  pers_octree_->node_search_order_ = BuildSearchOrder();
}

void PersOctree::RebuildChildBlocks() {  // the DFS's one-read-per-node view of the tree (f2n_oct_build_child_blocks)
  const int n = n_nodes_;
  // every caller has just replaced or edited the node array wholesale: speculative samples against the old one are void --
  // and their kernels, on the sampler's side streams, may still be READING the arrays that are released below (the allocator
  // would hand the blocks to this stream's next allocation): the device is drained first.  Rare: construction, state loads,
  // ProcOctree iterations.
  (void) hipDeviceSynchronize();
  generation_++;
  died_at_ = torch::zeros({std::max(n, 1)}, DevI32());
  if (!death_epoch_.defined()) {
    death_epoch_ = torch::zeros({1}, DevI32());
    n_repaired_ = torch::zeros({1}, DevI32());
    void* h = nullptr;
    void* d = nullptr;
    if (hipHostMalloc(&h, sizeof(int32_t), hipHostMallocMapped) == hipSuccess && hipHostGetDevicePointer(&d, h, 0) == hipSuccess) {
      death_epoch_host_ = static_cast<int32_t*>(h);
      death_epoch_host_dev_ = static_cast<int32_t*>(d);
      *death_epoch_host_ = 0;
    } else {  // no hint then: mode "auto" of the speculative sampling never speculates
      if (h != nullptr) (void) hipHostFree(h);
      (void) hipGetLastError();
    }
  }
  child_blocks_gpu_ = torch::empty({int64_t(n) * 8 * 32}, torch::TensorOptions().dtype(torch::kUInt8).device(torch::kCUDA));
  F2N_CALL(f2n_oct_build_child_blocks(CurStream(), n, VoidP(tree_nodes_gpu_), VoidP(child_blocks_gpu_)));
  // which nodes have a child (TreeNode::childs = int32 words 5..12 of the 64-byte node).  (nonzero synchronises: this runs
  // where the tree is replaced or re-numbered -- construction, state loads, ProcOctree -- not in an ordinary iteration.)
  interior_nodes_ = interior_rank_ = Tensor();
  n_interior_ = 0;
  if (n > 0) {
    Tensor words = tree_nodes_gpu_.view(torch::kInt32).view({n, 16});
    Tensor is_interior = words.slice(1, 5, 13).ge(0).any(1);
    interior_nodes_ = torch::nonzero(is_interior).squeeze(1).to(torch::kInt32).contiguous();
    interior_rank_ = (torch::cumsum(is_interior.to(torch::kInt32), 0) - 1).to(torch::kInt32).contiguous();
    n_interior_ = (int) interior_nodes_.size(0);
  }
}

// ---------------------------------------------------------------------------------------------------------
// GetSamples, PersSampler.cu:317-434
// ---------------------------------------------------------------------------------------------------------
SampleResultFlex PersSampler::GetSamples(const Tensor& rays_o_raw, const Tensor& rays_d_raw, const Tensor& /*bounds*/) {
  PendingSamples p;
  BeginSamples(rays_o_raw, rays_d_raw, global_data_pool_->ray_march_fineness_, p);
  return FinishSamples(p);
}

PersOctree::~PersOctree() {
  if (death_epoch_host_ != nullptr) (void) hipHostFree(death_epoch_host_);
}

int PersOctree::QuietEpochs() const {
  if (death_epoch_host_ == nullptr) return 0;
  const int last = *reinterpret_cast<volatile const int32_t*>(death_epoch_host_);  // (may lag behind the device: a hint)
  // no leaf has died since the statistics were (re)armed: quiet from the first update on.  (A fresh scene's statistics start at
  // 1000 and lose at most one per update; waiting for eight updates put the first speculative batches -- and the first-use
  // allocations of their side streams' pools, a few hundred MB of worst-case-sized buffers -- inside a short bench's timed region:
  // the driver's `--steps 20 --warmup 5` read 1.43 ms per step where steps 30+ take 1.13.)
  if (last == 0) return 1 << 20;
  return epoch_ - last;
}

bool PersSampler::MaintenanceDueAt(int it) const {
  return (!sub_div_milestones_.empty() && sub_div_milestones_.back() <= it) || it % compact_freq_ == 0;
}

bool PersSampler::MaintenanceDue(int ahead) const {  // the conditions of FinishOctUpdate below, for the iteration in progress ... + ahead
  for (int d = 0; d <= ahead; d++) {
    const int it = global_data_pool_->iter_step_ + d;
    if ((!sub_div_milestones_.empty() && sub_div_milestones_.back() <= it) || it % compact_freq_ == 0) return true;
  }
  return false;
}

void PersSampler::BeginSamples(const Tensor& rays_o_raw, const Tensor& rays_d_raw, float fineness, PendingSamples& p, bool speculative,
                               int64_t seq) {
  F2N_HOST_SCOPE("sampler.begin");
  Tensor rays_o = rays_o_raw.contiguous();
  Tensor rays_d_in = rays_d_raw.contiguous();
  CheckDev(rays_o, torch::kFloat32, "rays_o");
  CheckDev(rays_d_in, torch::kFloat32, "rays_d");
  const int n_rays = rays_o.size(0);
  auto& oct = *pers_octree_;
  void* st = CurStream();
  Tensor rays_d = torch::empty_like(rays_d_in);  // :319, with a fixed (oracle-restatable) summation order
  Tensor totals = torch::empty({3}, DevI32());   // [K, N, rays a tail repair hands back to the full walk]
  Tensor rays_noise;                             // :372-381
  const int n_noise = F2N_MAX_SAMPLE_PER_RAY + n_rays + 10;
  bool map_noise = false;
  const int64_t keyed = keyed_seq_;
  keyed_seq_ = -1;  // (one-shot, whichever branch runs)
  if (forced_noise_.defined()) {
    rays_noise = forced_noise_.contiguous();
    TORCH_CHECK(rays_noise.numel() >= n_noise, "forced noise too short");
  } else if (global_data_pool_->mode_ == RunningMode::VALIDATE) {
    rays_noise = torch::full({n_noise}, fineness, DevF32());  // ones * fineness (:376-377)
  } else {
    // The march noise of a batch is keyed by the batch's sequence number (KeyedDraws.h): the same batch gets the same noise
    // however far ahead of its step it is sampled (one or two steps, speculatively or behind the stat update: Renderer.h) and
    // however often (a prefetched batch that is dropped and sampled again draws the SAME numbers, not the next ones).
    if (seq < 0) seq = keyed;
    rays_noise = torch::empty({n_noise}, DevF32());  // (drawn by the prologue launch itself: Philox keyed by (seed, purpose, seq))
    map_noise = true;
  }
  // unit directions, zeroed totals and the march noise: one launch
  if (map_noise) {
    const auto key = noise_draws_.KeyFor(seq);
    F2N_CALL(f2n_sampler_prologue_keyed(st, n_rays, F32P(rays_d_in), F32P(rays_d), I32P(totals), 3, n_noise, key.key, key.seq, fineness,
                                        F32P(rays_noise)));
  } else {
    F2N_CALL(f2n_sampler_prologue(st, n_rays, F32P(rays_d_in), F32P(rays_d), I32P(totals), 3, 0, nullptr, fineness, nullptr));
  }
  const float far = 1e8f;  // the `bounds` argument is ignored by the reference too (:322-323)

  Tensor counts = torch::empty({n_rays}, DevI32());
  Tensor oct_se = torch::empty({n_rays, 2}, DevI32());
  // Leaf hits go into fixed-stride per-ray slots of a worst-case workspace (n_rays * max_oct_intersect_per_ray
  // entries of 12 B: ~100 MB at 8192 rays, nothing next to 288 GB of HBM): ONE DFS pass instead of the reference's
  // count pass + host sync + fill pass (PersSampler.cu:342-366).
  // (worst-case buffers are sized for the ray count rounded up to 2048: the adaptive batch of ExpRunner::Train changes its ray
  // count every iteration, and a new size every iteration means a new hipMalloc every iteration -- the caching allocator had
  // reserved 48 GiB by the end of a 20 000-iteration run)
  const int64_t cap_rays = (int64_t(n_rays) + 2047) / 2048 * 2048;
  const int64_t k_cap = cap_rays * max_oct_intersect_per_ray_;
  Tensor oct_idx = torch::empty({k_cap}, DevI32());
  Tensor oct_nf = torch::empty({k_cap, 2}, DevF32());
  Tensor oct_tr = torch::empty({k_cap}, DevI32());  // trans_idx of every listed leaf (the march would re-read the node)
  // (a tree whose interior nodes fit into a CU's LDS is walked out of LDS: the walk is a chain of dependent record reads, and
  // it is prefetched underneath the previous step's hash gather, behind whose L2 traffic each of those reads would queue)
  if (oct.n_interior_ >= 1 && oct.n_interior_ <= f2n_oct_lds_max_interior()) {
    F2N_TIMED_CALL("oct_intersect", f2n_oct_intersect_strided_lds(st, n_rays, max_oct_intersect_per_ray_,
                                    oct.node_search_order_.data_ptr<uint8_t>(), F32P(rays_o), F32P(rays_d), global_near_, far,
                                    VoidP(oct.tree_nodes_gpu_), I32P(oct_se), I32P(oct_idx), F32P(oct_nf), I32P(totals), I32P(oct_tr),
                                    VoidP(oct.child_blocks_gpu_), I32P(oct.interior_nodes_), I32P(oct.interior_rank_), oct.n_interior_));
  } else {
    F2N_TIMED_CALL("oct_intersect", f2n_oct_intersect_strided(st, n_rays, max_oct_intersect_per_ray_,
                                    oct.node_search_order_.data_ptr<uint8_t>(), F32P(rays_o), F32P(rays_d), global_near_, far,
                                    VoidP(oct.tree_nodes_gpu_), I32P(oct_se), I32P(oct_idx), F32P(oct_nf), I32P(totals), I32P(oct_tr),
                                    VoidP(oct.child_blocks_gpu_)));
  }

  // ONE march into fixed-stride per-ray slots (28 B x 1024 per ray of scratch, of which only the filled prefixes are
  // touched) + the per-ray counts; the reference marches twice (count pass, host sync, fill pass: :383-423).
  Tensor pts_se = torch::empty({n_rays, 2}, DevI32());
  const int64_t slots = cap_rays * F2N_MAX_SAMPLE_PER_RAY;
  Tensor s_dt = torch::empty({slots}, DevF32()), s_t = torch::empty({slots}, DevF32());  // warped points: computed by pack
  Tensor s_anchors = torch::empty({slots, 2}, DevI32());
  Tensor first_oct_dis = torch::empty({n_rays, 1}, DevF32());
  const bool tail = speculative && tail_repair_ && max_oct_intersect_per_ray_ <= 2048;
  const int persistent_blocks = march_blocks_;  // (the grid a batch that is begun two steps ahead is marched on)
  if (speculative && persistent_march_ && persistent_blocks > 0 && max_oct_intersect_per_ray_ <= 2048) {  // small persistent grid, rays sorted by leaf count
    Tensor order = torch::empty({n_rays + 1}, DevI32());  // [R] ray order + the group counter
    if (tail) {
      p.leaf_state = torch::empty({k_cap, 2}, DevI32());
      p.reached = torch::empty({n_rays}, DevI32());
    }
    F2N_TIMED_CALL("ray_march", f2n_ray_march_persistent(st, n_rays, max_oct_intersect_per_ray_, persistent_blocks, march_block_waves_, sample_l_, scale_by_dis_,
                                   F32P(rays_o), F32P(rays_d), F32P(rays_noise), I32P(oct_se), I32P(oct_idx), F32P(oct_nf),
                                   VoidP(oct.tree_nodes_gpu_), VoidP(oct.pers_trans_gpu_), I32P(counts), nullptr, F32P(s_dt), F32P(s_t),
                                   I32P(s_anchors), F32P(first_oct_dis), I32P(oct_tr), tail ? VoidP(p.leaf_state) : nullptr,
                                   tail ? I32P(p.reached) : nullptr, I32P(order), I32P(order) + n_rays));
  } else if (tail) {  // ... recording, per leaf-list entry, the state a repair can resume from (f2n_abi.h, "Tail repair")
    p.leaf_state = torch::empty({k_cap, 2}, DevI32());
    p.reached = torch::empty({n_rays}, DevI32());
    F2N_TIMED_CALL("ray_march", f2n_ray_march_strided_rec(st, n_rays, max_oct_intersect_per_ray_, sample_l_, scale_by_dis_, F32P(rays_o),
                                   F32P(rays_d), F32P(rays_noise), I32P(oct_se), I32P(oct_idx), F32P(oct_nf), VoidP(oct.tree_nodes_gpu_),
                                   VoidP(oct.pers_trans_gpu_), I32P(counts), nullptr, F32P(s_dt), F32P(s_t), I32P(s_anchors),
                                   F32P(first_oct_dis), I32P(oct_tr), VoidP(p.leaf_state), I32P(p.reached)));
  } else {
    F2N_TIMED_CALL("ray_march", f2n_ray_march_strided(st, n_rays, sample_l_, scale_by_dis_, F32P(rays_o), F32P(rays_d), F32P(rays_noise),
                                   I32P(oct_se), I32P(oct_idx), F32P(oct_nf), VoidP(oct.tree_nodes_gpu_),
                                   VoidP(oct.pers_trans_gpu_), I32P(counts), nullptr, F32P(s_dt), F32P(s_t), I32P(s_anchors),
                                   F32P(first_oct_dis), I32P(oct_tr)));
  }
  p.tail = tail;
  p.active = true;
  p.n_rays = n_rays;
  p.rays_o = rays_o; p.rays_d = rays_d; p.counts = counts; p.oct_se = oct_se; p.totals = totals;
  p.oct_idx = oct_idx; p.oct_nf = oct_nf; p.oct_tr = oct_tr; p.noise = rays_noise; p.pts_se = pts_se; p.s_dt = s_dt; p.s_t = s_t;
  p.s_anchors = s_anchors; p.first_oct_dis = first_oct_dis;
  p.speculative = speculative;
  p.completed = !speculative;
  p.generation = oct.generation_;
  p.spec_epoch = oct.epoch_ + 1;
  if (speculative) p.repair_flags = torch::empty({n_rays}, DevI32());
  // Scan, count and pack follow the march at once -- also for a speculative batch that is begun one step ahead: its pack then runs in the
  // stretch of the step the march ends in (underneath the hash gather, when the walk ran out of LDS) instead of behind the
  // stat update, where it lands on field_shade_fwd, which is as memory-bound as the pack is (115 us against 59 alone:
  // profiles/r03_speculation_experiments.txt).  CompleteSpeculative scans again and packs again only if a leaf died since.
  // (not for a batch that is begun two steps ahead: where that pays, leaves die in most steps and the pack would run twice)
  if (speculative && persistent_march_) return;
  IssueScanAndPack(p);
  p.packed_once = speculative;
}

// The rays a stat update invalidated are walked and marched again (f2n_oct_intersect_repair / f2n_ray_march_repair: both
// return at once on the device when no leaf died, the common case), then the tail of a GetSamples call.
bool PersSampler::CompleteSpeculative(PendingSamples& p) {
  F2N_HOST_SCOPE("sampler.complete");
  TORCH_CHECK(p.active && p.speculative && !p.completed, "CompleteSpeculative: nothing to complete");
  auto& oct = *pers_octree_;
  if (p.generation != oct.generation_) return false;
  void* st = CurStream();
  const float far = 1e8f;
  if (p.tail) {
    // no second walk: dead entries leave the lists in place, the few lists that were cut at the cap are walked again, and the
    // march resumes behind the first dead leaf of every ray it invalidated
    F2N_TIMED_CALL("oct_repair", f2n_oct_list_repair(st, p.n_rays, max_oct_intersect_per_ray_, I32P(p.oct_se), I32P(p.oct_idx), F32P(p.oct_nf),
                                   I32P(p.oct_tr), I32P(p.totals), I32P(oct.died_at_), p.spec_epoch, I32P(oct.death_epoch_), I32P(p.reached),
                                   I32P(p.repair_flags), I32P(oct.n_repaired_), I32P(p.totals) + 2));
    F2N_CALL(f2n_oct_intersect_repair_flagged(st, p.n_rays, max_oct_intersect_per_ray_, oct.node_search_order_.data_ptr<uint8_t>(),
                                   F32P(p.rays_o), F32P(p.rays_d), global_near_, far, VoidP(oct.tree_nodes_gpu_), I32P(p.oct_se),
                                   I32P(p.oct_idx), F32P(p.oct_nf), I32P(p.totals), I32P(p.oct_tr), VoidP(oct.child_blocks_gpu_),
                                   I32P(oct.death_epoch_), p.spec_epoch, I32P(p.repair_flags), I32P(p.totals) + 2));
    F2N_TIMED_CALL("march_repair", f2n_ray_march_repair_tail(st, p.n_rays, max_oct_intersect_per_ray_, sample_l_, scale_by_dis_, F32P(p.rays_o),
                                   F32P(p.rays_d), F32P(p.noise), I32P(p.oct_se), I32P(p.oct_idx), F32P(p.oct_nf), VoidP(oct.tree_nodes_gpu_),
                                   VoidP(oct.pers_trans_gpu_), I32P(p.counts), nullptr, F32P(p.s_dt), F32P(p.s_t), I32P(p.s_anchors),
                                   F32P(p.first_oct_dis), I32P(p.oct_tr), VoidP(p.leaf_state), I32P(p.reached), I32P(p.repair_flags),
                                   I32P(oct.death_epoch_), p.spec_epoch));
  } else {
  F2N_TIMED_CALL("oct_repair", f2n_oct_intersect_repair(st, p.n_rays, max_oct_intersect_per_ray_, oct.node_search_order_.data_ptr<uint8_t>(),
                                 F32P(p.rays_o), F32P(p.rays_d), global_near_, far, VoidP(oct.tree_nodes_gpu_), I32P(p.oct_se),
                                 I32P(p.oct_idx), F32P(p.oct_nf), I32P(p.totals), I32P(p.oct_tr), VoidP(oct.child_blocks_gpu_),
                                 I32P(oct.died_at_), p.spec_epoch, I32P(oct.death_epoch_), I32P(p.repair_flags), I32P(oct.n_repaired_)));
  F2N_TIMED_CALL("march_repair", f2n_ray_march_repair(st, p.n_rays, sample_l_, scale_by_dis_, F32P(p.rays_o), F32P(p.rays_d), F32P(p.noise),
                                 I32P(p.oct_se), I32P(p.oct_idx), F32P(p.oct_nf), VoidP(oct.tree_nodes_gpu_), VoidP(oct.pers_trans_gpu_),
                                 I32P(p.counts), nullptr, F32P(p.s_dt), F32P(p.s_t), I32P(p.s_anchors), F32P(p.first_oct_dis),
                                 I32P(p.oct_tr), I32P(p.repair_flags), I32P(oct.death_epoch_), p.spec_epoch));
  }
  p.completed = true;
  if (!p.packed_once) {
    IssueScanAndPack(p);
    return true;
  }
  // packed optimistically right behind the march: the counts are taken again (the repaired rays' may have changed; the host
  // reads THIS copy), and the pack runs again on the device only if a leaf died since -- into the same worst-case-sized arrays
  IssueScan(p);
  F2N_TIMED_CALL("pack_repair", f2n_pack_samples_repair(st, p.n_rays, I32P(p.pts_se), F32P(p.rays_o), F32P(p.rays_d), VoidP(oct.pers_trans_gpu_),
                                  nullptr, F32P(p.s_dt), F32P(p.s_t), I32P(p.s_anchors), F32P(p.o_pts) + 3 * (int64_t) p.extra_rows,
                                  F32P(p.o_dirs), F32P(p.o_dt), F32P(p.o_t), I32P(p.o_anchors) + 3 * (int64_t) p.extra_rows,
                                  I32P(oct.death_epoch_), p.spec_epoch));
  return true;
}

void PersSampler::IssueScan(PendingSamples& p) {
  void* st = CurStream();
  // the single host read-back of a GetSamples call: [K, N] go to mapped host memory from the scan kernel itself, read after an
  // event (no stream drain, and no copy launch between the scan and the pack); a batch that is scanned a second time
  // (CompleteSpeculative) keeps its slot -- the host reads it only behind the last scan's event
  totals_words_.Ensure(16);
  if (p.totals_slot < 0) {
    p.totals_slot = next_totals_slot_;
    next_totals_slot_ = (next_totals_slot_ + 1) & 7;
  }
  F2N_CALL(f2n_segment_scan_ex(st, p.n_rays, I32P(p.counts), I32P(p.pts_se), I32P(p.totals) + 1, totals_words_.Dev(2 * p.totals_slot),
                               I32P(p.totals), 1));
  p.counts_ready.record();
}

void PersSampler::IssueScanAndPack(PendingSamples& p) {
  auto& oct = *pers_octree_;
  void* st = CurStream();
  const int n_rays = p.n_rays;
  const int64_t slots = p.s_dt.numel();
  IssueScan(p);
  // The pack does not wait for the host to learn N: its outputs are sized for the worst case (every ray's slots full; pages
  // beyond the N rows actually written are never touched) and it is queued right behind the scan.  With the host in the
  // loop (count -> allocate -> launch) the pack of a prefetched batch started ~30 us after the count landed and the density
  // pre-pass of the step behind it ~100 us later than it could: both sat on the step's critical path once the sampler
  // chain was as long as the step it runs under (converged scenes).
  // (extra_rows: spare rows IN FRONT of the packed samples, see SampleResultFlex)
  const int extra = global_data_pool_->mode_ == RunningMode::TRAIN ? std::max(extra_sample_rows_, 0) : 0;
  const int64_t cap = slots + extra;
  p.extra_rows = extra;
  p.o_pts = torch::empty({cap, 3}, DevF32());
  p.o_dirs = torch::empty({slots, 3}, DevF32());
  p.o_dt = torch::empty({slots}, DevF32());
  p.o_t = torch::empty({slots}, DevF32());
  p.o_anchors = torch::empty({cap, 3}, DevI32());
  F2N_TIMED_CALL("pack_samples", f2n_pack_samples(st, n_rays, I32P(p.pts_se), F32P(p.rays_o), F32P(p.rays_d), VoidP(oct.pers_trans_gpu_), nullptr,
                                    F32P(p.s_dt), F32P(p.s_t), I32P(p.s_anchors), F32P(p.o_pts) + 3 * (int64_t) extra, F32P(p.o_dirs),
                                    F32P(p.o_dt), F32P(p.o_t), I32P(p.o_anchors) + 3 * (int64_t) extra));
}

SampleResultFlex PersSampler::FinishSamples(PendingSamples& p) {
  TORCH_CHECK(p.active && p.completed, "FinishSamples without (completed) BeginSamples");
  const int n_rays = p.n_rays;
  {
    F2N_HOST_SCOPE("wait.sample_counts");
    p.counts_ready.synchronize();
  }
  const int n_all_oct = totals_words_.Read(2 * p.totals_slot);
  const int n_all_pts = totals_words_.Read(2 * p.totals_slot + 1);
  if (global_data_pool_->mode_ == RunningMode::TRAIN) {
    float per_ray = float(n_all_oct) / float(n_rays);
    global_data_pool_->sampled_oct_per_ray_ = global_data_pool_->sampled_oct_per_ray_ * .9f + per_ray * .1f;
  }
  SampleResultFlex res;  // views of the first N rows of the arrays the pack is filling (or has filled)
  res.first_oct_dis = p.first_oct_dis;
  res.extra_rows = p.extra_rows;
  res.pts = p.o_pts.narrow(0, p.extra_rows, n_all_pts);
  res.dirs = p.o_dirs.narrow(0, 0, n_all_pts);
  res.dt = p.o_dt.narrow(0, 0, n_all_pts);
  res.t = p.o_t.narrow(0, 0, n_all_pts);
  res.anchors = p.o_anchors.narrow(0, p.extra_rows, n_all_pts);
  res.pts_idx_bounds = p.pts_se;
  // the scratch goes back to the allocator of the stream it was allocated on (and the pack runs on): whoever is handed
  // it next is ordered behind the pack
  p = PendingSamples();
  return res;
}

// PersSampler.cu:454-473
std::tuple<Tensor, Tensor> PersSampler::GetEdgeSamples(int n_pts) {
  const int n_edges = pers_octree_->n_edges_;
  TORCH_CHECK(n_edges > 0, "edge pool is empty: call SetEdgePool");
  Tensor edge_idx = forced_edge_idx_.defined() ? forced_edge_idx_.contiguous()
                                                : torch::randint(0, n_edges, {n_pts}, DevI32()).contiguous();
  Tensor edge_coord = forced_edge_coords_.defined() ? forced_edge_coords_.contiguous()
                                                    : (torch::rand({n_pts, 2}, DevF32()) * 2.f - 1.f).contiguous();
  Tensor out_pts = torch::empty({n_pts, 2, 3}, DevF32());
  Tensor out_idx = torch::empty({n_pts, 2}, DevI32());
  F2N_CALL(f2n_edge_samples(CurStream(), n_pts, VoidP(pers_octree_->edge_pool_gpu_), VoidP(pers_octree_->pers_trans_gpu_),
                            I32P(edge_idx), F32P(edge_coord), F32P(out_pts), I32P(out_idx)));
  return {out_pts, out_idx};
}

// one [4, n_nodes] buffer: weight votes, alpha votes (init -1, :555-556), visited marks (0), running visit counts -- a
// data-parallel run max-combines it across ranks with ONE collective.
// The buffer persists between iterations: the visit counts stay where they are and the stat update re-arms the vote rows,
// so an iteration issues no fill / copy here.  It is rebuilt whenever the node array or the visit counts were replaced
// behind its back (ProcOctree, LoadStates, InstallOctree).
Tensor& PersSampler::VoteBuffer() {
  auto& oct = *pers_octree_;
  const int n_nodes = oct.n_nodes_;
  Tensor& occ = oct.occ_;
  if (!occ.defined() || occ.size(1) != n_nodes || oct.tree_visit_cnt_.data_ptr() != (void*) (occ.data_ptr<int32_t>() + 3 * (int64_t) n_nodes)) {
    occ = torch::empty({4, n_nodes}, DevI32());
    occ.slice(0, 0, 2).fill_(-1);
    occ.select(0, 2).zero_();
    occ.select(0, 3).copy_(oct.tree_visit_cnt_);
    oct.tree_visit_cnt_ = occ.select(0, 3);
  }
  return occ;
}

// PersSampler.cu:536-615
void PersSampler::UpdateOctNodes(const SampleResultFlex& sample_result, const Tensor& sampled_weight,
                                 const Tensor& sampled_alpha) {
  const int n_nodes = pers_octree_->n_nodes_;
  const int n_rays = sample_result.pts_idx_bounds.size(0);
  CheckDev(sampled_weight, torch::kFloat32, "sampled_weight");
  CheckDev(sampled_alpha, torch::kFloat32, "sampled_alpha");
  Tensor& occ = VoteBuffer();
  F2N_TIMED_CALL("oct_mark_visit", f2n_oct_mark_visit(CurStream(), n_rays, n_nodes, I32P(sample_result.pts_idx_bounds), I32P(sample_result.anchors), 3,
                              F32P(sampled_weight), F32P(sampled_alpha), I32P(occ), I32P(occ) + n_nodes,
                              I32P(occ) + 2 * (int64_t) n_nodes, I32P(occ) + 3 * (int64_t) n_nodes));
  FinishOctUpdate();
}

// The early stop of the density pre-pass (Renderer.cpp:115-126) and the votes of UpdateOctNodes in one launch (training steps:
// the votes are cast over the weights / alphas the early stop produces); FinishOctUpdate() must follow.
void PersSampler::EarlyStopAndVote(const SampleResultFlex& sample_result, const float* f0, Tensor& weights, Tensor& alphas, Tensor& mask,
                                   Tensor& kept) {
  const int n_nodes = pers_octree_->n_nodes_;
  const int n_rays = sample_result.pts_idx_bounds.size(0);
  Tensor& occ = VoteBuffer();
  F2N_TIMED_CALL("early_stop", f2n_early_stop_votes(CurStream(), n_rays, I32P(sample_result.pts_idx_bounds), f0, 1, F32P(sample_result.dt),
                              F32P(weights), F32P(alphas), I32P(mask), I32P(kept), n_nodes, I32P(sample_result.anchors), 3, I32P(occ),
                              I32P(occ) + n_nodes, I32P(occ) + 2 * (int64_t) n_nodes, I32P(occ) + 3 * (int64_t) n_nodes));
}

// Everything of UpdateOctNodes behind the votes: the data-parallel exchange, the stat update (:579-593, MarkInvalidNodes
// :528-534) and the octree maintenance that is due (:605-614).
void PersSampler::FinishOctUpdate(const ScanArgs* scan) {
  auto& oct = *pers_octree_;
  const int n_nodes = oct.n_nodes_;
  Tensor& occ = oct.occ_;
  void* st = CurStream();
  if (occupancy_sync_hook_) occupancy_sync_hook_(occ);
  Tensor adders = occ.slice(0, 0, 2), visit_mark = occ.select(0, 2);
  oct.epoch_++;  // (deaths of this update are stamped with it: speculative samplers repair against them)
  if (scan != nullptr) {
    F2N_TIMED_CALL("oct_update_stats", f2n_oct_update_stats_scan(st, n_nodes, I32P(adders), I32P(adders) + n_nodes, I32P(visit_mark),
                                  I32P(oct.tree_weight_stats_), I32P(oct.tree_alpha_stats_), VoidP(oct.tree_nodes_gpu_),
                                  VoidP(oct.child_blocks_gpu_), /*reset_votes=*/1, I32P(oct.died_at_), oct.epoch_, I32P(oct.death_epoch_),
                                  oct.death_epoch_host_dev_, scan->n, scan->counts, scan->start_end, scan->total, scan->mirror, scan->also,
                                  scan->n_also));
  } else {
    F2N_TIMED_CALL("oct_update_stats", f2n_oct_update_stats_ex(st, n_nodes, I32P(adders), I32P(adders) + n_nodes, I32P(visit_mark),
                                  I32P(oct.tree_weight_stats_), I32P(oct.tree_alpha_stats_), VoidP(oct.tree_nodes_gpu_),
                                  VoidP(oct.child_blocks_gpu_), /*reset_votes=*/1, I32P(oct.died_at_), oct.epoch_, I32P(oct.death_epoch_),
                                  oct.death_epoch_host_dev_));
  }

  while (!sub_div_milestones_.empty() && sub_div_milestones_.back() <= global_data_pool_->iter_step_) {  // :605-610
    oct.ProcOctree(true, true, sub_div_milestones_.back() <= 0);
    oct.MarkInvisibleNodes();
    oct.ProcOctree(true, false, false);
    sub_div_milestones_.pop_back();
  }
  if (global_data_pool_->iter_step_ % compact_freq_ == 0) {  // :612-614
    oct.ProcOctree(true, false, false);
  }
}

void PersOctree::MarkInvisibleNodes() {  // PersSampler.cu:663-680
  TORCH_CHECK(w2c_.defined(), "training cameras not set: call SetTrainCameras");
  F2N_CALL(f2n_oct_mark_invisible(CurStream(), n_nodes_, (int) intri_.size(0), VoidP(tree_nodes_gpu_),
                                  F32P(intri_), F32P(w2c_), F32P(bound_)));
  RebuildChildBlocks();  // trans_idx of invisible leaves changed on the device
}

// ---------------------------------------------------------------------------------------------------------
// ProcOctree, PersSampler.cpp:120-330: pruning of dead leaves, path compression, renumbering, subdivision of visited
// leaves -- on the device (csrc/octree.hip), where the reference copies the node array to the host and back.  The only
// host involvement left is reading the two new node counts (to size the new arrays).
// ---------------------------------------------------------------------------------------------------------
void PersOctree::ProcOctree(bool compact, bool subdivide, bool brute_force) {
  TORCH_CHECK(compact, "ProcOctree without compaction is not used by the sampler (PersSampler.cu:605-614)");
  torch::NoGradGuard no_grad;
  void* st = CurStream();
  const int n = n_nodes_;
  auto u8 = [](int64_t bytes) { return torch::empty({bytes}, DevU8()); };
  auto i32 = [](int64_t k) { return torch::empty({k}, DevI32()); };
  Tensor w = tree_weight_stats_.contiguous(), a = tree_alpha_stats_.contiguous(), v = tree_visit_cnt_.contiguous();
  Tensor work = u8(int64_t(n) * sizeof(TreeNode)), edited = u8(int64_t(n) * sizeof(TreeNode));
  Tensor alive = i32(n), n_child = i32(n), keep = i32(n), new_pos = torch::empty({n, 2}, DevI32()), total = i32(1);
  F2N_CALL(f2n_oct_prune_compress(st, n, VoidP(tree_nodes_gpu_), VoidP(work), VoidP(edited), I32P(alive), I32P(n_child), I32P(keep)));
  F2N_CALL(f2n_segment_scan(st, n, I32P(keep), I32P(new_pos), I32P(total)));
  const int m = total.item<int>();
  TORCH_CHECK(m >= 1, "octree root was removed");
  Tensor k_nodes = u8(int64_t(m) * sizeof(TreeNode)), k_w = i32(m), k_a = i32(m), k_v = i32(m);
  F2N_CALL(f2n_oct_gather_kept(st, n, VoidP(edited), I32P(keep), I32P(new_pos), I32P(w), I32P(a), I32P(v), VoidP(k_nodes), I32P(k_w),
                               I32P(k_a), I32P(k_v)));
  if (!subdivide) {
    tree_nodes_gpu_ = k_nodes;
    tree_weight_stats_ = k_w;
    tree_alpha_stats_ = k_a;
    n_nodes_ = m;
  } else {
    Tensor depth = i32(m), size = i32(m), new_idx = i32(m);
    F2N_CALL(f2n_oct_subtree_sizes(st, m, VoidP(k_nodes), I32P(k_v), brute_force ? 1 : 0, I32P(depth), I32P(size)));
    const int m2 = size.slice(0, 0, 1).item<int>();
    Tensor d_nodes = u8(int64_t(m2) * sizeof(TreeNode)), d_w = i32(m2), d_a = i32(m2);
    F2N_CALL(f2n_oct_subdivide(st, m, VoidP(k_nodes), I32P(k_v), brute_force ? 1 : 0, I32P(size), I32P(k_w), I32P(k_a), I32P(new_idx),
                               VoidP(d_nodes), I32P(d_w), I32P(d_a)));
    tree_nodes_gpu_ = d_nodes;
    tree_weight_stats_ = d_w;
    tree_alpha_stats_ = d_a;
    n_nodes_ = m2;
  }
  tree_visit_cnt_ = torch::zeros({n_nodes_}, DevI32());
  RebuildChildBlocks();
}

// ---------------------------------------------------------------------------------------------------------
// States / LoadStates, PersSampler.cpp:692-733 (checkpoint order: nodes, warps, visit counts, milestones)
// ---------------------------------------------------------------------------------------------------------
std::vector<Tensor> PersSampler::States() {
  std::vector<Tensor> ret;
  ret.push_back(pers_octree_->tree_nodes_gpu_);
  ret.push_back(pers_octree_->pers_trans_gpu_);
  ret.push_back(pers_octree_->tree_visit_cnt_.clone());  // (a row of the persistent vote buffer)
  ret.push_back(torch::from_blob(sub_div_milestones_.data(), {(int64_t) sub_div_milestones_.size()}, CpuI32()).to(torch::kCUDA));
  return ret;
}

int PersSampler::LoadStates(const std::vector<Tensor>& states, int idx) {
  auto& oct = *pers_octree_;
  oct.tree_nodes_gpu_ = states[idx++].clone().to(torch::kCUDA).to(torch::kUInt8).contiguous();
  oct.pers_trans_gpu_ = states[idx++].clone().to(torch::kCUDA).to(torch::kUInt8).contiguous();
  oct.tree_visit_cnt_ = states[idx++].clone().to(torch::kCUDA).to(torch::kInt32).contiguous();
  Tensor milestones = states[idx++].clone().to(torch::kCPU).to(torch::kInt32).contiguous();
  TORCH_CHECK(oct.tree_nodes_gpu_.numel() % sizeof(TreeNode) == 0 && oct.pers_trans_gpu_.numel() % sizeof(TransInfo) == 0,
              "state blobs do not match the TreeNode/TransInfo layout");
  oct.n_nodes_ = int(oct.tree_nodes_gpu_.numel() / sizeof(TreeNode));
  sub_div_milestones_.resize(milestones.numel());
  std::memcpy(sub_div_milestones_.data(), milestones.data_ptr(), milestones.numel() * sizeof(int));
  const int64_t n = oct.n_nodes_;
  TORCH_CHECK(oct.tree_visit_cnt_.numel() == n, "visit_cnt size mismatch");
  oct.tree_weight_stats_ = torch::full({n}, INIT_NODE_STAT, DevI32());  // stats are NOT checkpointed (:721-722)
  oct.tree_alpha_stats_ = torch::full({n}, INIT_NODE_STAT, DevI32());
  global_data_pool_->n_volumes_ = int(oct.pers_trans_gpu_.numel() / sizeof(TransInfo));
  oct.RebuildChildBlocks();
  return idx;
}

// Installs a freshly constructed octree (BuildPersOctree): what the reference's PersSampler constructor leaves behind
// (PersSampler.cpp:84-101, :688): nodes, warps, zeroed visit counts, initial occupancy stats, edge pool.
void PersSampler::InstallOctree(const Tensor& tree_nodes_bytes, const Tensor& pers_trans_bytes, const Tensor& edge_pool_bytes) {
  auto& oct = *pers_octree_;
  TORCH_CHECK(tree_nodes_bytes.numel() % sizeof(TreeNode) == 0 && pers_trans_bytes.numel() % sizeof(TransInfo) == 0,
              "blobs do not match the TreeNode/TransInfo layout");
  oct.tree_nodes_gpu_ = tree_nodes_bytes.clone().to(torch::kCUDA).to(torch::kUInt8).contiguous();
  oct.n_nodes_ = int(oct.tree_nodes_gpu_.numel() / sizeof(TreeNode));
  oct.pers_trans_gpu_ = pers_trans_bytes.clone().to(torch::kCUDA).to(torch::kUInt8).contiguous();
  const int64_t n = oct.n_nodes_;
  oct.tree_visit_cnt_ = torch::zeros({n}, DevI32());
  oct.tree_weight_stats_ = torch::full({n}, INIT_NODE_STAT, DevI32());
  oct.tree_alpha_stats_ = torch::full({n}, INIT_NODE_STAT, DevI32());
  oct.RebuildChildBlocks();
  TORCH_CHECK(global_data_pool_->n_volumes_ == int(oct.pers_trans_gpu_.numel() / sizeof(TransInfo)),
              "the octree has ", oct.pers_trans_gpu_.numel() / sizeof(TransInfo), " warps but the field was built for ",
              global_data_pool_->n_volumes_, " (runtime.n_volumes)");
  SetEdgePool(edge_pool_bytes);
}

void PersSampler::SetEdgePool(const Tensor& edge_pool_bytes) {
  TORCH_CHECK(edge_pool_bytes.numel() % sizeof(EdgePool) == 0, "edge pool blob does not match the EdgePool layout");
  pers_octree_->edge_pool_gpu_ = edge_pool_bytes.clone().to(torch::kCUDA).to(torch::kUInt8).contiguous();
  pers_octree_->n_edges_ = int(edge_pool_bytes.numel() / sizeof(EdgePool));
}

void PersSampler::SetTrainCameras(const Tensor& w2c, const Tensor& intri, const Tensor& bounds) {
  pers_octree_->w2c_ = w2c.to(torch::kCUDA).to(torch::kFloat32).contiguous();
  pers_octree_->intri_ = intri.to(torch::kCUDA).to(torch::kFloat32).contiguous();
  pers_octree_->bound_ = bounds.to(torch::kCUDA).to(torch::kFloat32).contiguous();
}

std::unique_ptr<PtsSampler> ConstructPtsSampler(GlobalDataPool* global_data_pool) {  // PtsSamplerFactory.cpp:7-13
  const std::string type = global_data_pool->config_.Str("pts_sampler.type");
  TORCH_CHECK(type == "PersSampler", "unknown pts_sampler.type: ", type);
  return std::make_unique<PersSampler>(global_data_pool);
}

}  // namespace f2n
