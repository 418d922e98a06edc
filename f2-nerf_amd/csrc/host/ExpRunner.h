// ExpRunner (mirrors src/ExpRunner.h): owns GlobalDataPool, Renderer and the optimiser state.
#pragma once
#include <functional>

#include "Dataset.h"
#include "GradSyncPipeline.h"
#include "Renderer.h"

namespace f2n {

struct TrainStats {
  Tensor loss, mse;  // device scalars (read them only when reporting: no per-iteration host sync)
  int n_rays = 0, n_samples = 0, n_meaningful = 0;
  bool skipped_nan = false;
};

class ExpRunner {
 public:
  ExpRunner(const std::map<std::string, std::string>& flat_config, int n_images);
  void LoadStates(const std::vector<Tensor>& states);  // checkpoint order, see SURVEY.md section 5
  std::vector<Tensor> States() {
    FinishPending();
    return renderer_->States();
  }
  // What a replica needs besides the checkpoint vector to be a copy of another one: the edge pool (built once from the
  // construction-time tree, PersSampler.cpp:615-659; its t_idx_a/b index the warps) and the training cameras MarkInvisibleNodes
  // projects into.  The reference's checkpoint does not hold them (its constructor rebuilds them from the data set).
  std::vector<Tensor> AuxStates();
  void LoadAuxStates(const std::vector<Tensor>& aux);
  bool ApplyGradients(bool apply_optimizer);
  // Flush: completes whatever a streaming TrainStep left open -- the pipelined data-parallel step whose all-reduce is
  // still in flight, and the finiteness flags a prefetching TrainStep reads one step late.
  void FinishPending();
  void FinishPendingStep();     // the pipelined data-parallel part only
  void DeferFlags(bool apply_optimizer);  // start the asynchronous read-back of nan_flags_
  bool ResolveDeferredFlags();  // true: the step they belong to was dropped (loss scales halved, counters taken back)
  // next_*: optionally the NEXT iteration's rays (already resident): their sampling is prefetched on a side stream;
  // next2_*: and those of the iteration after it (two-deep sampling pipeline, Renderer::next2_batch_)
  TrainStats TrainStep(const Tensor& rays_o, const Tensor& rays_d, const Tensor& bounds, const Tensor& gt_colors,
                       const Tensor& emb_idx, bool apply_optimizer = true, const Tensor& next_rays_o = Tensor(),
                       const Tensor& next_rays_d = Tensor(), const Tensor& next_bounds = Tensor(),
                       const Tensor& next2_rays_o = Tensor(), const Tensor& next2_rays_d = Tensor());
  void EnqueueApply(bool apply_optimizer);
  int32_t* NextFlagMirror();
  bool ResolveFlags(bool apply_optimizer);
  TrainStats TrainStepAutograd(const Tensor& rays_o, const Tensor& rays_d, const Tensor& bounds, const Tensor& gt_colors,
                               const Tensor& emb_idx, bool apply_optimizer = true);
  float CurVarLossWeight() const;
  // The iteration-dependent scalars of ExpRunner::Train / UpdateAdaParams (ExpRunner.cpp:108-114, 221-254) as a pure function
  // of the configuration: what the members below feed, and what tests pin against the reference's own code without a GPU.
  struct ScheduleParams {
    float ray_march_init_fineness;
    int ray_march_fineness_decay_end_iter;
    float learning_rate, learning_rate_alpha, learning_rate_warm_up_end_iter;
    int end_iter;
    float gradient_scaling_start, gradient_scaling_end;
    float var_loss_weight;
    int var_loss_start, var_loss_end;
  };
  struct ScheduleValues {
    float fineness, lr, gradient_scaling_progress, var_loss_weight;
  };
  static ScheduleValues ScheduleAt(const ScheduleParams& p, int iter);
  ScheduleParams Schedule() const;
  bool forward_render_ = true;      // inference through Renderer::RenderForward (false: the taped Render(), as a comparator)
  int render_chunk_rays_ = 65536;   // rays per chunk of RenderWholeImage on that path
  std::vector<Tensor> RenderRays(const Tensor& rays_o, const Tensor& rays_d, const Tensor& bounds);
  std::vector<Tensor> RenderWholeImage(const Tensor& rays_o, const Tensor& rays_d, const Tensor& bounds);
  float TestImagePSNR(Dataset& dataset, int idx);       // 8-bit quantised prediction, as ExpRunner.cpp:360-369
  std::vector<float> TestImages(Dataset& dataset);      // per-view PSNR of the test set, then the mean
  Tensor RenderPathFrame(Dataset& dataset, const Tensor& pose, int res_level = 1);
  Tensor VisualizeImage(Dataset& dataset, int idx);     // [H, 4W, 3]: gt | colours | first-hit disparity | disparity (ExpRunner.cpp:301-321)
  void RenderPath(Dataset& dataset, const Tensor& render_poses, const std::function<void(int, const Tensor&)>& sink,
                  int res_level = 1);
  void SaveCheckpoint(const std::string& dir);          // <dir>/renderer.pt + <dir>/scalars.pt (ExpRunner.cpp:205-219)
  void LoadCheckpoint(const std::string& dir);          // ExpRunner.cpp:188-203
  int Train(Dataset& dataset, int until_iter = -1, int sets = DATA_TRAIN_SET);
  int64_t last_train_meaningful_ = 0, last_train_marched_ = 0, last_train_rays_ = 0;  // totals of the last Train call
  TrainStats last_train_stats_;
  void UpdateAdaParams();
  float FinenessAt(int iter) const;
  // skip_flag: device int, != 0 drops the update; compute_flags: device int32[3], finiteness flags computed (and obeyed) in the step
  void OptimStep(const int32_t* skip_flag = nullptr, int32_t* compute_flags = nullptr);
  // One optimiser step's arguments as the C-ABI takes them (f2n_adam_fused / F2nStepTail): the small fp32 groups in the order the
  // flag layout names them, the h16-gradient table group, the scalars of step `step`.
  struct AdamPlan {
    F2nAdamGroup small[4];
    int n_small = 0, n_table = 0;
    float *tp = nullptr, *tm = nullptr, *tv = nullptr;
    void *tg = nullptr, *th = nullptr;
    float tscale = 1.f;
  };
  void BuildAdamPlan(AdamPlan& plan);
  // The step's tail inside the field backward's call (round 6; f2n_field_bwd_step_tail): finiteness flags and the small groups'
  // Adam on a second stream behind the field-MLP backward, the table's Adam in the scatter's owner blocks.  Single-GPU streaming
  // steps that apply the optimiser take it; a data-parallel step (the gradients travel first), a step that only inspects
  // gradients, the diagnostics taps and check_nan == false keep the separate launches.  Same parameters, bit for bit
  // (tests/test_gpu_e2e.py::test_fused_step_tail_equals_separate_launches).
  // 0: never; 1: whenever legal; 2 (default): in the two-deep sampling regime, and at any time for tables of 2^21 entries per level and
  // more (where the table's Adam is the longer half of the tail).  A young scene samples ONE batch ahead and
  // issues the head of that batch's chain (noise, prologue, walk) at the END of the step precisely so that it runs under the Adam /
  // reduction tail -- bandwidth-bound, vector units idle; with the tail folded away the walk lands on the next gather instead:
  // measured 1.129-1.141 against 1.112-1.115 ms on the fresh fox scene, 2.47 against 2.29 ms on the llff rig
  // (profiles/r06_fused_tail_ab.txt).  In the two-deep regime the next batches' chains start at the TOP of a step whatever its
  // tail looks like, and the fold is worth 2-3 % there.
  int fused_tail_ = 2;
  bool draws_off_main_ = true;  // Train(): the batch draws on the tail stream instead of the main queue (A/B knob)
  bool spec_start_without_event_ = true;  // ... and then no spec_start event at the top of a step (A/B knob)
  bool exact_flag_order_ = false;  // read the previous step's flags in front of this step's backward also without the fused tail (see TrainStep)
  bool BuildStepTail(F2nStepTail* tail);
  AdamPlan tail_plan_;
  bool tail_table_left_ = false;  // the step tail in flight left the table's Adam to EnqueueApply (F2nStepTail::leave_table_to_caller)
  bool flags_on_tail_stream_ = false;  // where the last finiteness-flag kernel was queued (DeferFlags records its event there)
  void BuildOptimizer();
  Tensor FlattenSmallGrads();
  int CurBatchSize() const;
  // Sequence numbers (KeyedDraws.h): step_seq_ counts the training steps taken (dropped non-finite ones included); the batch of
  // step k, its march noise and its background / edge draws are draw k of their purposes.
  int64_t step_seq_ = 0;
  static constexpr int kBatchSizeLag = 5;
  int64_t ema_base_seq_ = 0;       // steps before this one left no record: batches are sized from ema_base_value_ instead
  float ema_base_value_ = 512.f;   // (GlobalDataPool's initial meaningful-samples average)
  int BatchSizeFor(int64_t seq);
  bool digest_table_ = false;   // per-step table checksums into the digest (Renderer::StepDigest; two small launches per step)
  Tensor digest_table_sums_;
  void ResetStepSequence(int64_t seq);

  int iter_step_ = 0, end_iter_;
  int pts_batch_size_;
  float learning_rate_, learning_rate_alpha_, learning_rate_warm_up_end_iter_;
  float ray_march_init_fineness_;
  int ray_march_fineness_decay_end_iter_;
  float tv_loss_weight_, disp_loss_weight_, var_loss_weight_;
  int var_loss_start_, var_loss_end_;
  float gradient_scaling_start_, gradient_scaling_end_;
  float cur_lr_ = 0.f;
  bool check_nan_ = true;
  int async_counts_ = 1;  // 1: streaming steps keep the survivor count on the device; 0: always read it back (as Render does); 2: never read it back in TrainStep (tests)
  int optim_steps_ = 0;
  // The data-parallel gradient exchange and its place in the step (GradSyncPipeline.h): sync_.blocking = all-reduce in front
  // of the optimiser; sync_.begin / sync_.end + sync_.pipelined = launch the asynchronous all-reduce right after backward,
  // make the compute stream wait for it in the NEXT TrainStep after ray sampling has been issued (or in FinishPending).
  GradSyncPipeline sync_;
  // what a DataParallel object holds weakly: its destructor unhooks itself from a runner that is still there, and leaves a dead one alone
  std::shared_ptr<int> alive_ = std::make_shared<int>(0);
  Tensor nan_flags_;  // device int32 [4]: field MLP, colour MLP, either
  // A TrainStep that is handed the next batch (streaming use) does not wait for its own flags: the kernel that computes
  // them also writes them to mapped host memory (MappedHost.h; two slots of four words, alternating -- the next step's kernel
  // is queued before this step's slot is read), and the host reads them in the NEXT step, behind an event.  The update
  // itself is predicated on the device either way; only the host-side reaction (halved loss scale, iteration counter)
  // lags by one step, on the rare non-finite path.
  MappedWords flag_words_;
  int next_flag_slot_ = 0, last_flag_slot_ = 0, deferred_flag_slot_ = 0;
  at::cuda::CUDAEvent nan_flags_ev_;
  bool flags_deferred_ = false, deferred_apply_ = false, deferred_dropped_ = false;

  std::unique_ptr<GlobalDataPool> global_data_pool_;
  std::unique_ptr<Renderer> renderer_;
  std::vector<ParamGroup> groups_;
  std::vector<Tensor> exp_avg_, exp_avg_sq_;
  std::shared_ptr<void> data_parallel_;  // DataParallel (DataParallel.h), when attached: released before the renderer
};

}  // namespace f2n
