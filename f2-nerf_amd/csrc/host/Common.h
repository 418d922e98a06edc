// Host-side common definitions (mirrors the role of src/Common.h of the reference).
#pragma once
#include <torch/torch.h>
#include <ATen/hip/HIPEvent.h>
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>
#include <c10/hip/HIPStream.h>

#include <chrono>
#include <cmath>
#include <map>
#include <memory>
#include <sstream>
#include <string>
#include <vector>

#include "f2n_abi.h"
#ifndef F2N_DEBUG_BUILD
#define F2N_DEBUG_BUILD 0  // 1: the debug variant of the host layer (build.py variant "debug"): stream-skew hooks + measurement knobs
#endif
#if F2N_DEBUG_BUILD
#include "f2n_debug.h"
#endif

#define None torch::indexing::None
#define Slc torch::indexing::Slice

namespace f2n {

using Tensor = torch::Tensor;

inline torch::TensorOptions DevF32() { return torch::TensorOptions().dtype(torch::kFloat32).device(torch::kCUDA); }
inline torch::TensorOptions DevF16() { return torch::TensorOptions().dtype(torch::kFloat16).device(torch::kCUDA); }
inline torch::TensorOptions DevI32() { return torch::TensorOptions().dtype(torch::kInt32).device(torch::kCUDA); }
inline torch::TensorOptions DevU8() { return torch::TensorOptions().dtype(torch::kUInt8).device(torch::kCUDA); }
inline torch::TensorOptions CpuF32() { return torch::TensorOptions().dtype(torch::kFloat32).device(torch::kCPU); }
inline torch::TensorOptions CpuI32() { return torch::TensorOptions().dtype(torch::kInt32).device(torch::kCPU); }
inline torch::TensorOptions CpuU8() { return torch::TensorOptions().dtype(torch::kUInt8).device(torch::kCPU); }

// The HIP stream torch is currently enqueuing on (also correct on the autograd engine thread).
inline void* CurStream() { return (void*) c10::hip::getCurrentHIPStream().stream(); }

// Every C-ABI call is checked; there is no fallback path: a missing device or a failed launch throws.
#define F2N_CALL(expr)                                                           \
  do {                                                                           \
    int f2n_rc_ = (expr);                                                        \
    TORCH_CHECK(f2n_rc_ == 0, #expr, " failed with status ", f2n_rc_);           \
  } while (0)

inline void CheckDev(const Tensor& t, c10::ScalarType dt, const char* what) {
  TORCH_CHECK(t.defined(), what, ": undefined tensor");
  TORCH_CHECK(t.is_cuda(), what, ": must live on the HIP device (there is no CPU path)");
  TORCH_CHECK(t.is_contiguous(), what, ": must be contiguous");
  TORCH_CHECK(t.scalar_type() == dt, what, ": wrong dtype");
}
// Optional HIP-event timing of individual kernel launches on the stream they are enqueued on (what bench.py's
// `roofline` object is computed from).  Disabled by default: two event records per launch when enabled.
class KernelTimers {
 public:
  static KernelTimers& Get() {
    static KernelTimers inst;
    return inst;
  }
  void Enable(const std::vector<std::string>& names) {
    enabled_.clear();
    for (auto& n : names) enabled_[n] = true;
    all_ = names.size() == 1 && names[0] == "*";
  }
  void Disable() { enabled_.clear(); all_ = false; }
  bool On(const char* name) const { return all_ || (!enabled_.empty() && enabled_.count(name) > 0); }
  void Begin(const char* name) {
    if (!On(name)) return;
    auto e = std::make_unique<at::cuda::CUDAEvent>(hipEventDefault);
    e->record();
    pending_[name] = std::move(e);
  }
  void End(const char* name) {
    if (!On(name)) return;
    auto e = std::make_unique<at::cuda::CUDAEvent>(hipEventDefault);
    e->record();
    spans_[name].emplace_back(std::move(pending_[name]), std::move(e));
  }
  // Synchronises, returns {name: (launches, total milliseconds)} and clears the recorded spans.
  std::map<std::string, std::pair<int, double>> Collect() {
    std::map<std::string, std::pair<int, double>> out;
    for (auto& kv : spans_) {
      double ms = 0;
      for (auto& se : kv.second) {
        se.second->synchronize();
        ms += se.first->elapsed_time(*se.second);
      }
      out[kv.first] = {(int) kv.second.size(), ms};
    }
    spans_.clear();
    return out;
  }

 private:
  std::map<std::string, bool> enabled_;
  bool all_ = false;
  std::map<std::string, std::unique_ptr<at::cuda::CUDAEvent>> pending_;
  std::map<std::string, std::vector<std::pair<std::unique_ptr<at::cuda::CUDAEvent>, std::unique_ptr<at::cuda::CUDAEvent>>>> spans_;
};

// Host-side wall-clock accumulators per named region (where does the HOST thread spend an iteration: queueing, drawing batches,
// waiting for which event?).  A few nanoseconds per region when disabled; measurement aid (ExpRunner.host_profile binding).
class HostProf {
 public:
  static HostProf& Get() {
    static HostProf inst;
    return inst;
  }
  bool on = false;
  std::map<std::string, std::pair<int64_t, double>> acc;  // name -> (entries, seconds)
  struct Scope {
    const char* name;
    std::chrono::steady_clock::time_point t0;
    bool live;
    explicit Scope(const char* n) : name(n), live(HostProf::Get().on) {
      if (live) t0 = std::chrono::steady_clock::now();
    }
    ~Scope() {
      if (!live) return;
      auto& a = HostProf::Get().acc[name];
      a.first += 1;
      a.second += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
  };
};
#define F2N_HOST_SCOPE(name) HostProf::Scope f2n_host_scope_##__LINE__(name)

#define F2N_TIMED_CALL(name, expr)      \
  do {                                  \
    KernelTimers::Get().Begin(name);    \
    F2N_CALL(expr);                     \
    KernelTimers::Get().End(name);      \
  } while (0)

inline float* F32P(const Tensor& t) { return t.data_ptr<float>(); }
inline int32_t* I32P(const Tensor& t) { return t.data_ptr<int32_t>(); }
inline void* VoidP(const Tensor& t) { return t.data_ptr(); }

}  // namespace f2n
