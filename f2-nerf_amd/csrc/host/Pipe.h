// State / optimiser-group tree (mirrors src/Utils/Pipe.h:8-19 and Pipe.cpp).
#pragma once
#include "Common.h"

namespace f2n {

// One Adam parameter group handled by the fused optimiser (f2n_adam_step*).  `grad` is owned by the module
// that produces it; `grad_scale` turns it into the true gradient (1/loss_scale for the loss-scaled MLP and
// hash-table gradients).
struct ParamGroup {
  std::string name;
  Tensor param;         // fp32 master
  Tensor grad;          // fp32, or fp16 for the hash table
  Tensor param_h;       // fp16 working copy refreshed by the step (may be undefined)
  float grad_scale = 1.f;
  float weight_decay = 0.f;
  bool grad_is_h16 = false;
  bool grad_round_h16 = false;
  int64_t active = -1;  // number of leading elements that can ever receive gradient (-1: all)
};

class Pipe {
 public:
  virtual ~Pipe() = default;
  virtual int LoadStates(const std::vector<Tensor>& states, int idx) {
    for (auto pipe : sub_pipes_) idx = pipe->LoadStates(states, idx);
    return idx;
  }
  virtual std::vector<Tensor> States() {
    std::vector<Tensor> ret;
    for (auto pipe : sub_pipes_) {
      auto cur = pipe->States();
      ret.insert(ret.end(), cur.begin(), cur.end());
    }
    return ret;
  }
  virtual std::vector<ParamGroup> OptimParamGroups() {
    std::vector<ParamGroup> ret;
    for (auto pipe : sub_pipes_) {
      auto cur = pipe->OptimParamGroups();
      ret.insert(ret.end(), cur.begin(), cur.end());
    }
    return ret;
  }
  virtual void Reset() {
    for (auto pipe : sub_pipes_) pipe->Reset();
  }
  void RegisterSubPipe(Pipe* sub_pipe) { sub_pipes_.push_back(sub_pipe); }
  std::vector<Pipe*> sub_pipes_;
};

}  // namespace f2n
