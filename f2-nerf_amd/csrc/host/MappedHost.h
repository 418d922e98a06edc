// A few int32 words of host memory that device kernels write directly (hipHostMallocMapped): the host's copy of a count or a
// flag lands where the kernel that produces it runs, without a device-to-host copy launch behind that kernel -- on an in-order
// queue every such launch is a dependent boundary of ~5 us.  The host reads a word after synchronising with an event recorded
// behind the producing kernel (kernel completion makes its stores to coherent host memory visible), never while it may run.
#pragma once
#include <hip/hip_runtime.h>
#include <torch/torch.h>

#include <cstdint>

namespace f2n {

class MappedWords {
 public:
  MappedWords() = default;
  MappedWords(const MappedWords&) = delete;
  MappedWords& operator=(const MappedWords&) = delete;
  ~MappedWords() {
    if (host_ != nullptr) (void) hipHostFree(host_);
  }
  // Allocates on first use (a device must be current).  Throws when the platform cannot map host memory.
  void Ensure(int n_words) {
    if (host_ != nullptr) {
      TORCH_CHECK(n_words <= n_, "MappedWords: ", n_words, " words asked of a buffer of ", n_);
      return;
    }
    void *h = nullptr, *d = nullptr;
    TORCH_CHECK(hipHostMalloc(&h, sizeof(int32_t) * n_words, hipHostMallocMapped) == hipSuccess, "hipHostMalloc(mapped) failed");
    if (hipHostGetDevicePointer(&d, h, 0) != hipSuccess) {
      (void) hipHostFree(h);
      TORCH_CHECK(false, "hipHostGetDevicePointer failed");
    }
    host_ = static_cast<int32_t*>(h);
    dev_ = static_cast<int32_t*>(d);
    n_ = n_words;
    for (int i = 0; i < n_words; i++) host_[i] = 0;
  }
  int32_t* Dev(int word = 0) const { return dev_ + word; }
  int32_t Read(int word) const { return *reinterpret_cast<volatile const int32_t*>(host_ + word); }
  bool Allocated() const { return host_ != nullptr; }

 private:
  int32_t* host_ = nullptr;
  int32_t* dev_ = nullptr;
  int n_ = 0;
};

}  // namespace f2n
