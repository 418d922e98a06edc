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

namespace f2n {

class GradSyncPipeline {
 public:
  // the exchange, installed by DataParallel::Attach or the Python hooks (either `blocking` or the `begin` / `end` pair)
  std::function<void()> blocking, begin, end;
  bool pipelined = false;
  // what the runner does around it
  std::function<void(bool apply_optimizer, float lr)> apply;  // finiteness flags + predicated Adam: enqueue only
  std::function<void()> defer_flags;                          // start the asynchronous read-back of the flags

  bool Installed() const { return (bool) blocking || (bool) begin; }
  bool Pending() const { return pending_; }

  // Top of a step.  `presample` (may be empty) issues this step's ray sampling, which reads neither parameters nor
  // gradients: in pipelined mode it goes first, so that it runs underneath the exchange that is still in flight.
  void BeginStep(bool apply_optimizer, const std::function<void()>& presample) {
    if (pipelined && apply_optimizer && presample) presample();
    FinishPendingStep();
  }

  // This step's gradients are in their buffers.  True: exchanged (if there is an exchange) and applied now.  False: the
  // exchange was started and the step is completed by the next BeginStep / FinishPendingStep.
  bool GradientsReady(bool apply_optimizer, float lr) {
    if (pipelined && apply_optimizer) {
      if (begin) begin();
      pending_ = true;
      pending_lr_ = lr;
      return false;
    }
    if (blocking) blocking();
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
};

}  // namespace f2n
