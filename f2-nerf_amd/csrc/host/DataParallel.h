// Ray-data-parallel replicas over RCCL, driven from the C++ host (no Python in the training step).
//
// The reference is single-GPU.  Rays are independent given a replicated model + octree: each rank renders
// its own ray batch and the replicas exchange, per iteration,
//   (1) gradients  -- ONE ncclGroup: all-reduce(AVG) of the active prefix of the f16 (x128) hash-gradient table
//                     (17 * 2^log2 halves) + all-reduce(AVG) of the flat fp32 buffer holding field-MLP, colour-MLP and
//                     app_emb gradients (ExpRunner::FlattenSmallGrads).  Issued on the communicator's own stream right
//                     after backward; the compute stream only waits for it in the NEXT step, after that step's ray
//                     sampling has been queued (ExpRunner's pipelined hooks), so the 17 MiB reduction over xGMI runs
//                     underneath ~0.3 ms of sampler kernels;
//   (2) occupancy  -- all-reduce(MAX) of the [4, n_nodes] vote / mark / visit-count buffer between MarkVisit and the
//                     stat update (PersSampler.cu:555-603) on the compute stream, together with all-reduce(SUM) of the
//                     survivor count (the meaningful-samples EMA that sizes the next ray batch, ExpRunner.cpp:86, must be
//                     the same on every rank or the replicas draw different batch sizes).
// Gradients are identical on all ranks after (1), so the device-side finiteness flags and the predicated Adam agree
// everywhere.  The communicator is created with ncclCommInitRank from a unique id that rank 0 generates and the launcher
// distributes (f2-nerf_amd/parallel.py: through the torch.distributed store; a file or an environment variable work too).
#pragma once
#include "ExpRunner.h"

struct ncclComm;

namespace f2n {

class DataParallel {
 public:
  ~DataParallel();
  static std::vector<uint8_t> NewUniqueId();  // rank 0
  // Creates the communicator (collective: every rank must call), replicates rank 0's state on every rank and wires the
  // exchanges into the runner.  overlap = pipelined gradient exchange (see above); false: in front of the optimiser.
  // hooks_for_one_rank: a one-rank world has nothing to exchange and installs no hooks unless asked to (tests and overhead
  // measurements drive the RCCL calls with one rank)
  void Attach(ExpRunner* runner, int rank, int world, const std::vector<uint8_t>& unique_id, bool overlap = true,
              bool hooks_for_one_rank = false);
  void BroadcastStates();           // rank 0's checkpoint vector -> every rank (collective)
  int rank() const { return rank_; }
  int world() const { return world_; }
  // table all-reduce buckets of the NEXT Attach (default kTableBuckets; 1 = the whole prefix in one all-reduce: A/B, bench.py --dp-buckets)
  static int table_buckets;
  int CommRanks() const;
  // Diagnostics of the first measured multi-GPU run (round-5 verdict, next 6): with timing on, every step's gradient exchange is
  // bracketed by two timing events on the communicator's stream (first table bucket started -> flat buffer reduced: what the
  // exchange took, its waits for the scatter's later buckets included) and the compute stream's wait for it at the top of the next
  // step by two on the compute stream (what of it was NOT hidden).  CollectTiming synchronises and returns
  // {steps, exchange ms total, exposed wait ms total} since the last call.  ~4 event packets per step: off by default.
  void EnableTiming(bool on);
  std::vector<double> CollectTiming();
  int64_t SmallExchangesEarly() const { return n_small_early_; }  // steps whose small buffers travelled beside the scatter (SmallGradsExchange)
  int64_t BucketCallbacks() const { return n_bucket_callbacks_; }  // table ranges the scatter reported while it ran            // what RCCL itself reports for the communicator (ncclCommCount)

 private:
  // bucketed table exchange (GradSyncPipeline.h): range b of n of the active table prefix, in halves
  static constexpr int kTableBuckets = 4;
  int n_buckets_ = 1;
  int64_t n_bucket_callbacks_ = 0;  // table ranges the scatter reported while it ran (the rest went with GradSyncBegin)
  int64_t n_small_early_ = 0;
  std::pair<int64_t, int64_t> BucketRange(int b) const;
  void SendBucket(int b);
  void SmallGradsExchange(void* chain_stream);  // GradSyncPipeline::small_exchange: the flat buffer's all-reduce, early
  void GradSyncBegin();
  void GradSyncEnd();
  void OccupancySync(Tensor occ);
  ExpRunner* runner_ = nullptr;
  std::weak_ptr<int> runner_alive_;
  bool hooks_installed_ = false;
  int device_ = -1;  // the device current at Attach: where the scatter hook lives and where the destructor removes it
  ncclComm* comm_ = nullptr;
  int rank_ = 0, world_ = 1;
  Tensor table_prefix_, flat_;
  std::unique_ptr<c10::hip::HIPStreamMasqueradingAsCUDA> comm_stream_;
  at::cuda::CUDAEvent grads_ready_ev_, reduced_ev_, small_ready_ev_, small_done_ev_;
  std::vector<at::cuda::CUDAEvent> bucket_ev_;
  bool timing_ = false, span_open_ = false;
  struct TimedSpan {
    at::cuda::CUDAEvent a{0u}, b{0u};  // (flags 0 = hipEventDefault: timing enabled)
    bool closed = false;
  };
  std::vector<std::unique_ptr<TimedSpan>> exchange_spans_, wait_spans_;
};

}  // namespace f2n
