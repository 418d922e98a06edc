// Test infrastructure -- NOT part of the product.  Forced in front of every file of f2-nerf_amd/csrc/host/ (-include) when the host
// layer is built for the emulated wavefront (tests/wave_emul/wemu_build.py build_host): the C++/LibTorch plugin classes -- Renderer,
// PersSampler, Hash3DAnchored, SHShader, ExpRunner, the octree builder, the dataset -- compile from their own source text, with the
// handful of GPU-runtime names they use pointed at CPU stand-ins: tensors live on the CPU (torch::kCUDA reads torch::kCPU), streams
// are one (launches of the emulated kernel library are synchronous), events are always reached, "mapped host memory" is host
// memory, a CUDA generator is a CPU generator.  What the host DOES -- which kernels it calls with what, in which order, what it
// prefetches, repairs, drops and resolves when -- is its own code, unchanged.
#pragma once
#include <torch/torch.h>
#include <torch/extension.h>
#include <ATen/hip/HIPEvent.h>
#include <ATen/hip/HIPGeneratorImpl.h>
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>
#include <ATen/CPUGeneratorImpl.h>
#include <c10/hip/HIPStream.h>
#include <hip/hip_runtime.h>

#include <cstdlib>

extern "C" {
long wemu_hb_record(int stream);
void wemu_hb_wait(int stream, long token);
void wemu_hb_host_sync(long token);
}
namespace c10 { namespace hip {
inline int& emu_cur_stream_id() { static thread_local int id = 0; return id; }
}  // namespace hip
}  // namespace c10

namespace at { namespace cuda {
// The stand-ins keep the ORDER the host asks for, even though nothing here runs concurrently: every record / wait / host-side
// synchronisation is reported to the emulated kernel library (wemu_hb_*: vector clocks per stream), which checks the accesses of every
// launch against it (tests/wave_emul/stream_races.py).  And they count waits on events that were never recorded: on the GPU such a wait
// returns at once -- an ordering the code believes it has and has not (wemu_host_unrecorded_waits).
inline long g_wemu_unrecorded_waits = 0;
struct EmuEvent {
  EmuEvent(unsigned = 0) {}
  bool recorded = false;
  long token = -1;
  void record() { recorded = true; token = wemu_hb_record(c10::hip::emu_cur_stream_id()); }
  template <typename S> void record(const S& s) { recorded = true; token = wemu_hb_record(s.id); }
  template <typename S> void block(const S& s) {
    if (!recorded) g_wemu_unrecorded_waits++;
    wemu_hb_wait(s.id, token);
  }
  bool query() const { wemu_hb_host_sync(token); return true; }  // (always reached: the host goes on as if it had waited)
  void synchronize() const { wemu_hb_host_sync(token); }
  float elapsed_time(const EmuEvent&) const { return 1.f; }  // (a millisecond: whoever divides by a kernel's time can)
};
namespace detail {
// A CUDA generator is a Philox stream addressed by (seed, offset); host/KeyedDraws.h sets the offset to pick draw number `seq` of a
// purpose.  The CPU stand-in keeps that addressing -- the same (seed, offset) gives the same uniforms, another offset others -- by
// reseeding its mt19937 from the pair (the VALUES differ from the GPU's, as they do between any two generators: nothing compares
// them across the two; the keyed draws the kernels make themselves are Philox on both sides).
struct EmuGeneratorImpl : public at::CPUGeneratorImpl {
  EmuGeneratorImpl() : at::CPUGeneratorImpl(0) { device_ = c10::Device(c10::DeviceType::CPU, 0); }
  void set_current_seed(uint64_t s) override {
    seed_ = s;
    at::CPUGeneratorImpl::set_current_seed(s);
  }
  uint64_t current_seed() const override { return seed_; }
  void set_offset(uint64_t o) override {
    offset_ = o;
    at::CPUGeneratorImpl::set_current_seed(seed_ * 0x9E3779B97F4A7C15ull + o * 0xD1B54A32D192ED03ull + 1);
  }
  uint64_t get_offset() const override { return offset_; }
  EmuGeneratorImpl* clone_impl() const override {
    auto* g = new EmuGeneratorImpl();
    g->set_current_seed(seed_);
    g->set_offset(offset_);
    return g;
  }
  uint64_t seed_ = 0, offset_ = 0;
};
inline const at::Generator& getDefaultEmuGenerator(int = -1) { return at::detail::getDefaultCPUGenerator(); }
inline at::Generator createEmuGenerator(int = -1) { return at::make_generator<EmuGeneratorImpl>(); }
}  // namespace detail
}  // namespace cuda
}  // namespace at

namespace c10 { namespace hip {
struct EmuStream {
  int id = 0;  // 0: the default ("main") stream; 1..7: streams from the pool
  hipStream_t stream() const { return reinterpret_cast<hipStream_t>(static_cast<uintptr_t>(id)); }
  void synchronize() const { wemu_hb_host_sync(-1); }
  bool operator==(const EmuStream& o) const { return id == o.id; }
  bool operator!=(const EmuStream& o) const { return id != o.id; }
  int device_index() const { return 0; }
};
inline EmuStream getStreamFromPoolEmu(bool = false, int = -1) {
  static int next = 0;
  EmuStream s;
  s.id = 1 + (next++ % 7);
  return s;
}
inline EmuStream getCurrentEmuStream(int = -1) {
  EmuStream s;
  s.id = emu_cur_stream_id();
  return s;
}
struct EmuStreamGuard {
  int prev;
  explicit EmuStreamGuard(const EmuStream& s) : prev(emu_cur_stream_id()) { emu_cur_stream_id() = s.id; }
  ~EmuStreamGuard() { emu_cur_stream_id() = prev; }
  EmuStreamGuard(const EmuStreamGuard&) = delete;
};
inline int emu_current_device() { return 0; }
}  // namespace hip
}  // namespace c10

extern "C" __attribute__((used, visibility("default"))) inline long wemu_host_unrecorded_waits() { return at::cuda::g_wemu_unrecorded_waits; }

static inline hipError_t emu_host_malloc(void** p, size_t n) { *p = calloc(1, n ? n : 4); return *p ? hipSuccess : hipErrorOutOfMemory; }
static inline hipError_t emu_host_device_pointer(void** d, void* h) { *d = h; return hipSuccess; }
static inline hipError_t emu_host_free(void* p) { free(p); return hipSuccess; }

// `torch::from_blob(host array).to(torch::kCUDA)` is a COPY to the device in the product; with the CPU as the device `.to` would keep the
// alias of an array that is about to go out of scope (Hash3DAnchored's level scales, the sampler's search order): every use of from_blob
// in the host layer wraps a read-only source, so the stand-in copies at once.
namespace torch {
inline at::Tensor from_blob_copy(void* data, at::IntArrayRef sizes, const at::TensorOptions& options = at::TensorOptions()) {
  return at::from_blob(data, sizes, options).clone();
}
}  // namespace torch
#define from_blob from_blob_copy
#define kCUDA kCPU
#define is_cuda is_cpu
#define CUDAEvent EmuEvent
#define getDefaultCUDAGenerator getDefaultEmuGenerator
#define createCUDAGenerator createEmuGenerator
#define HIPStreamMasqueradingAsCUDA EmuStream
#define getStreamFromPoolMasqueradingAsCUDA getStreamFromPoolEmu
#define getCurrentHIPStreamMasqueradingAsCUDA getCurrentEmuStream
#define getCurrentHIPStream getCurrentEmuStream
#define HIPStreamGuardMasqueradingAsCUDA EmuStreamGuard
#define current_device emu_current_device
#define hipHostMalloc(p, n, flags) emu_host_malloc((void**) (p), (n))
#define hipHostGetDevicePointer(d, h, flags) emu_host_device_pointer((void**) (d), (h))
#define hipHostFree(p) emu_host_free(p)
#define hipDeviceSynchronize() (wemu_hb_host_sync(-1), hipSuccess)
#define hipGetLastError() hipSuccess
