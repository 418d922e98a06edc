// PersSampler / PersOctree host side (mirrors src/PtsSampler/PersSampler.h).  Kernels: csrc/sampler.hip and
// csrc/octree.hip via the C-ABI.  The octree is created from a serialised state (the reference's checkpoint byte layout)
// or built from the cameras (OctreeBuilder.cpp); the per-milestone maintenance (ProcOctree) runs on the device.
#pragma once
#include <functional>

#include "KeyedDraws.h"
#include "MappedHost.h"
#include "PtsSampler.h"

namespace f2n {

#define INIT_NODE_STAT 1000
#define N_PROS 12

struct alignas(32) TransInfo {   // 544 B, PersSampler.h:15-20 of the reference
  float w2xz[N_PROS][2][4];
  float weight[3][N_PROS];
  float center[3];
  float dis_summary;
};
struct alignas(32) TreeNode {    // 64 B, PersSampler.h:22-29 of the reference
  float center[3];
  float side_len;
  int parent;
  int childs[8];
  bool is_leaf_node;
  int trans_idx;
};
struct alignas(32) EdgePool {    // 64 B
  int t_idx_a, t_idx_b;
  float center[3], dir_0[3], dir_1[3];
};
static_assert(sizeof(TransInfo) == 544 && sizeof(TreeNode) == 64 && sizeof(EdgePool) == 64, "checkpoint layout");

// Construction from the training cameras (OctreeBuilder.cpp)
struct OctreeBuildResult {
  std::vector<TreeNode> nodes;
  std::vector<TransInfo> trans;
  std::vector<EdgePool> edges;
};
TransInfo ConstructTrans(const Tensor& rand_pts, const Tensor& c2w_vis, const Tensor& intri0, const Tensor& center, int first_cam);
std::vector<EdgePool> ConstructEdgePool(const std::vector<TreeNode>& nodes);
OctreeBuildResult BuildPersOctree(const Tensor& c2w, const Tensor& intri, const Tensor& bounds, int max_depth, float bbox_side_len,
                                  float split_dist_thres, int n_rand_pts = 32 * 32 * 32);

class PersOctree {
 public:
  void ProcOctree(bool compact, bool subdivide, bool brute_force);
  void MarkInvisibleNodes();
  void RebuildChildBlocks();

  Tensor w2c_, intri_, bound_;  // training cameras, for MarkInvisibleNodes
  int n_nodes_ = 0;
  Tensor tree_nodes_gpu_;  // TreeNode[n_nodes_], the reference's checkpoint bytes; lives (and is maintained) on the device
  Tensor child_blocks_gpu_;  // [n_nodes][8] x 32 B, derived from tree_nodes_gpu_ (f2n_oct_build_child_blocks)
  // The nodes a walk can expand (those with a child), for the walk that keeps their records in LDS
  // (f2n_oct_intersect_strided_lds): interior_nodes_[r] = node index, interior_rank_[node] = r.  Rebuilt with the child blocks.
  Tensor interior_nodes_, interior_rank_;
  int n_interior_ = 0;
  Tensor tree_weight_stats_, tree_alpha_stats_, tree_visit_cnt_;
  Tensor occ_;  // [4, n_nodes] weight votes, alpha votes, visited marks, visit counts (tree_visit_cnt_ is its last row)
  // Speculative sampling (f2n_abi.h, "Speculative sampling"): every stat update is an epoch; a leaf that dies in it is stamped
  // (died_at_[node] = epoch, death_epoch_[0] = epoch).  generation_ counts the changes that invalidate node indices or
  // revive / re-number leaves (ProcOctree, MarkInvisibleNodes, LoadStates, InstallOctree): samples marched speculatively
  // against an older generation cannot be repaired and are dropped.
  Tensor died_at_, death_epoch_, n_repaired_;
  // host word (hipHostMalloc, mapped) + its device address: the last epoch in which a leaf died, written by the stat update
  // kernel itself (a torch pinned tensor's host address is NOT a device address here: the first version faulted on the first death)
  int32_t* death_epoch_host_ = nullptr;
  int32_t* death_epoch_host_dev_ = nullptr;
  ~PersOctree();
  int QuietEpochs() const;   // stat updates since a leaf last died, as far as the host can tell without synchronising
  int epoch_ = 0;
  int64_t generation_ = 0;
  Tensor node_search_order_;
  Tensor pers_trans_gpu_;
  Tensor edge_pool_gpu_;
  int n_edges_ = 0;
};

// A GetSamples call cut at its one host read-back: BeginSamples issues everything up to the sample counts (intersection,
// march, scan, the counts on their way to pinned memory) without blocking the host; FinishSamples waits for the counts,
// allocates the outputs and issues the pack.  A training step begins the NEXT batch's sampling as soon as its own octree
// update is issued and finishes it after its own backward has been queued, so the host never sits in the sampler's
// read-back while the step's kernels are still to be issued.
struct PendingSamples {
  bool active = false;
  int n_rays = 0;
  Tensor rays_o, rays_d, counts, oct_se, totals, oct_idx, oct_nf, oct_tr, noise, pts_se, s_dt, s_t, s_anchors,
      first_oct_dis;
  Tensor o_pts, o_dirs, o_dt, o_t, o_anchors;  // packed outputs, sized for the worst case (+ extra_rows), already being filled
  int extra_rows = 0;
  int totals_slot = -1;  // which pair of PersSampler::totals_words_ the scan writes [K, N] to (read after counts_ready)
  at::cuda::CUDAEvent counts_ready;
  // speculative: intersection and march were issued BEFORE the stat update they would normally wait for; scan / count / pack
  // are issued by CompleteSpeculative once that update is in the stream, behind the repair of the rays it invalidated
  bool speculative = false, completed = true;
  bool packed_once = false;    // speculative and already scanned + packed behind its march (one-step-ahead batches: PersSampler::BeginSamples)
  int spec_epoch = 0;          // first stat-update epoch whose deaths the speculative walk may have missed
  int64_t generation = 0;      // PersOctree::generation_ the samples were marched against
  Tensor repair_flags;         // (tail repair: repair_from, F2N_REPAIR_* or the list entry the march resumes from)
  Tensor leaf_state, reached;  // tail repair: the march's resumable states per leaf-list entry / last entry looked at
  bool tail = false;           // sampled with f2n_ray_march_strided_rec: repaired without a second walk
};

class PersSampler : public PtsSampler {
 public:
  explicit PersSampler(GlobalDataPool* global_data_pool);
  SampleResultFlex GetSamples(const Tensor& rays_o, const Tensor& rays_d, const Tensor& bounds) override;
  // speculative = true: stop after the march (no scan / count / pack); CompleteSpeculative issues the rest
  // seq: the batch's sequence number in the training run (its march noise is draw `seq` of the noise purpose, KeyedDraws.h);
  // < 0: keyed_seq_ if the caller set it (GetSamples has the reference's signature), else the next draw of the sampler's own sequence
  void BeginSamples(const Tensor& rays_o, const Tensor& rays_d, float fineness, PendingSamples& p, bool speculative = false,
                    int64_t seq = -1);
  int64_t keyed_seq_ = -1;  // one-shot: the sequence number of the batch the NEXT BeginSamples / GetSamples call samples
  // Call with the stat update(s) since BeginSamples already in the current stream's order.  False: the tree was re-numbered
  // since (generation mismatch): nothing was issued, the caller must sample again.
  bool CompleteSpeculative(PendingSamples& p);
  void IssueScanAndPack(PendingSamples& p);
  void IssueScan(PendingSamples& p);  // segment scan + count read-back
  // speculative batches record resumable march states and are repaired by list compaction + a march of the tail behind the
  // first dead leaf (f2n_oct_list_repair / f2n_ray_march_repair_tail) instead of a second walk and march from the origin
  bool tail_repair_ = true;
  // > 0: speculative batches are marched by this many persistent one-wave blocks, rays sorted by leaf count
  // (f2n_ray_march_persistent): a small footprint underneath the main queue's kernels, for batches that have two steps to finish
  int march_blocks_ = 512;
  int march_block_waves_ = 4;  // (measured: profiles/r06_march_block_waves.txt) the persistent march's waves come in workgroups of this many (one CU each): see f2n_ray_march_persistent
  bool persistent_march_ = false;  // set by the Renderer around the BeginSamples of a batch that is begun two steps ahead
  int LdsWalkMaxInterior() const { return f2n_oct_lds_max_interior(); }  // small trees are walked out of LDS (same bits either way)
  // a FinishOctUpdate of the iteration in progress or of one of the `ahead` iterations behind it runs ProcOctree
  // (milestone / compact_freq, PersSampler.cu:605-614)
  bool MaintenanceDue(int ahead = 0) const;
  bool MaintenanceDueAt(int iter) const;  // ... of iteration `iter`
  SampleResultFlex FinishSamples(PendingSamples& p);
  std::tuple<Tensor, Tensor> GetEdgeSamples(int n_pts) override;
  void UpdateOctNodes(const SampleResultFlex& sample_result, const Tensor& sampled_weights,
                      const Tensor& sampled_alpha) override;
  // UpdateOctNodes split for the training step: early stop + votes in one launch, then the rest (exchange, stats, maintenance)
  void EarlyStopAndVote(const SampleResultFlex& sample_result, const float* f0, Tensor& weights, Tensor& alphas, Tensor& mask,
                        Tensor& kept);
  // scan (optional): the survivor scan of the same step rides in the stat update's launch (f2n_oct_update_stats_scan)
  struct ScanArgs {
    int n = 0;
    const int32_t* counts = nullptr;
    int32_t *start_end = nullptr, *total = nullptr, *mirror = nullptr;
    const int32_t* also = nullptr;
    int n_also = 0;
  };
  void FinishOctUpdate(const ScanArgs* scan = nullptr);
  Tensor& VoteBuffer();
  // [K, N] of the sampling calls in flight, written by the scan kernel itself (MappedHost.h); eight rotating pairs: up to
  // Renderer::kPendingSlots prefetched batches and one synchronous GetSamples hold a pair each, a dropped batch's kernels are
  // waited for before its slot is released (Renderer::DropPendingSlot), so a pair comes round again only after five newer
  // calls have taken theirs
  MappedWords totals_words_;
  int next_totals_slot_ = 0;
  std::vector<Tensor> States() override;
  int LoadStates(const std::vector<Tensor>& states, int idx) override;

  // Not part of the reference checkpoint (LoadStates there keeps the constructor's edge pool / cameras).
  void InstallOctree(const Tensor& tree_nodes_bytes, const Tensor& pers_trans_bytes, const Tensor& edge_pool_bytes);
  void SetEdgePool(const Tensor& edge_pool_bytes);
  void SetTrainCameras(const Tensor& w2c, const Tensor& intri, const Tensor& bounds);

  std::unique_ptr<PersOctree> pers_octree_;
  std::vector<int> sub_div_milestones_;
  int compact_freq_;
  int max_oct_intersect_per_ray_;
  float global_near_;
  float sample_l_;
  bool scale_by_dis_;
  // data-parallel hook: called between MarkVisit and the stats update with (adders [2,n], mark [n], visit_cnt [n])
  std::function<void(Tensor)> occupancy_sync_hook_;  // gets the [4, n_nodes] vote / mark / visit-count buffer
  // explicit random draws for parity tests (empty = draw from torch's generator like the reference)
  Tensor forced_noise_, forced_edge_idx_, forced_edge_coords_;
  KeyedUniforms noise_draws_{0x9E3779B97F4A7C15ull};  // the march noise: draw k belongs to batch k (BeginSamples)
  int extra_sample_rows_ = 0;  // SampleResultFlex::extra_rows of the training samples (set by the Renderer: 2 * n_edge_pts)
};

}  // namespace f2n
