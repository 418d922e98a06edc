// The ORDER in which a training step and the data-parallel gradient exchange interleave, kept apart from everything that
// touches the device so that it can be driven -- and asserted -- without one (tests/test_pipeline_cpu.py drives it through
// pybind with recording callbacks; ExpRunner::TrainStep drives it with the real ones).
//
//   blocking exchange:   ... backward | exchange | flags + Adam | next step ...
//   pipelined exchange:  ... backward | begin (asynchronous, own stream) ] [ next step: ray sampling | end (the compute stream
//                        waits) | flags + Adam of the PREVIOUS step with ITS learning rate | forward ...
// The reference is single-GPU; the exchanges themselves live in DataParallel.cpp / parallel.py.
#pragma once
#include <functional>
#include <stdexcept>

namespace f2n {

class GradSyncPipeline {
 public:
  // the exchange, installed by DataParallel::Attach or the Python hooks (either `blocking` or the `begin` / `end` pair)
  std::function<void()> blocking, begin, end;
  bool pipelined = false;
  // Bucketed table exchange (round 5): while the backward's last kernels run, the scatter reports the ranges of the gradient table
  // that are final (f2n_set_scatter_buckets); `bucket` starts range b's all-reduce at once.  Buckets arrive in order, at most once
  // each, between BeginStep and GradientsReady; `begin` / `blocking` then send whatever was not sent (all of it when the scatter
  // reported nothing: small batches) -- every rank issues the same sequence of collectives whatever path its scatter took.
  // A notification only counts INSIDE a step whose gradients will be exchanged (armed by BeginStep / ArmBuckets, disarmed by
  // GradientsReady / ResetBuckets): a scatter that runs on this device for anybody else -- a test's f2n_hash_bwd, a second
  // runner, a taped backward outside TrainStepAutograd -- must not start all-reduces the other ranks never issue (round-5 advisor).
  std::function<void(int bucket, int n_buckets)> bucket;
  // The small gradient buffers (the two MLPs, the appearance embedding) are complete long before the table is -- behind the field-MLP
  // backward, in front of the whole scatter --: a runner that queues the step's tail through f2n_field_bwd_step_tail hands their
  // exchange this callback (round 6).  It is called with the stream the deferred reductions were queued on and must leave that stream
  // ordered behind the exchange; `begin` / `blocking` then send the table only.  Same rule as for the buckets: only inside an armed step.
  std::function<void(void* chain_stream)> small_exchange;
  // The ORDER every rank must keep: a step that is eligible for the early exchange sends the small buffers FIRST, then the table's
  // buckets -- also on a rank whose batch missed the scene and never reached its backward (`begin` / `blocking` send the small buffers in
  // front of the buckets then); every other step sends them last, as before.  Set by the runner per step, from arguments that are the same
  // on every rank.
  bool small_first = false;
  bool small_sent() const { return small_sent_; }
  void SmallGradsReady(void* chain_stream) {
    if (!small_exchange || !armed_ || small_sent_) return;
    small_exchange(chain_stream);
    small_sent_ = true;
  }
  int buckets_sent() const { return buckets_sent_; }
  bool armed() const { return armed_; }
  void ArmBuckets() {
    buckets_sent_ = 0;
    small_sent_ = false;
    armed_ = Installed();
  }
  void BucketReady(int b, int n) {
    if (!bucket || !armed_) return;
    if (b != buckets_sent_ || b >= n) throw std::logic_error("GradSyncPipeline: gradient bucket out of order");
    bucket(b, n);
    buckets_sent_ = b + 1;
  }
  // what the runner does around it
  std::function<void(bool apply_optimizer, float lr)> apply;  // finiteness flags + predicated Adam: enqueue only
  std::function<void()> defer_flags;                          // start the asynchronous read-back of the flags

  bool Installed() const { return (bool) blocking || (bool) begin; }
  bool Pending() const { return pending_; }

  // Top of a step.  `presample` (may be empty) issues this step's ray sampling, which reads neither parameters nor
  // gradients: in pipelined mode it goes first, so that it runs underneath the exchange that is still in flight.
  void ResetBuckets() {
    buckets_sent_ = 0;
    small_sent_ = false;
    armed_ = false;
  }
  void BeginStep(bool apply_optimizer, const std::function<void()>& presample) {
    ArmBuckets();  // (also: a step that threw behind a bucket must not leave its count behind)
    if (pipelined && apply_optimizer && presample) presample();
    FinishPendingStep();
  }

  // This step's gradients are in their buffers.  True: exchanged (if there is an exchange) and applied now.  False: the
  // exchange was started and the step is completed by the next BeginStep / FinishPendingStep.
  bool GradientsReady(bool apply_optimizer, float lr) {
    struct Reset {
      GradSyncPipeline& p;
      ~Reset() { p.ResetBuckets(); }
    } reset{*this};  // (begin / blocking read buckets_sent(): what is left to send)
    if (pipelined && apply_optimizer) {
      if (begin) begin();
      pending_ = true;
      pending_lr_ = lr;
      return false;
    }
    // A step that does not apply the optimiser (gradient inspection) still exchanges: with the pipelined pair back to back --
    // the table buckets its scatter reported are on their way, and a rank that skipped the rest would leave the others' collectives
    // without a partner (round-5 advisor, medium).
    if (blocking) blocking();
    else if (begin) {
      begin();
      if (end) end();
    }
    if (apply) apply(apply_optimizer, lr);
    return true;
  }

  // Completes the step whose exchange is in flight: the compute stream waits for it (the host does not), Adam runs with THAT
  // step's learning rate, the flags start their way to the host.
  void FinishPendingStep() {
    if (!pending_) return;
    pending_ = false;
    if (end) end();
    if (apply) apply(true, pending_lr_);
    if (defer_flags) defer_flags();
  }

 private:
  bool pending_ = false;
  float pending_lr_ = 0.f;
  int buckets_sent_ = 0;
  bool armed_ = false, small_sent_ = false;
};

}  // namespace f2n
