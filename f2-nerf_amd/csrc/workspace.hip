// Internal scratch workspaces of the C-ABI library (never visible to callers): per-block partial sums of the
// backward kernels.  Grown on demand with hipMalloc, one buffer per (device, slot), never freed.
//
// Why partial sums instead of atomics: thousands of fp32 atomics onto the SAME few thousand addresses (MLP
// parameters, appearance embeddings) serialise at the memory-side atomic unit (~1 us per dependent RMW measured
// on MI355X: 1e3-deep chains cost ~1 ms).  Each block therefore writes its partial vector with plain coalesced
// stores and a tiny second kernel reduces them.
#include "f2n_dev.h"

#include <mutex>

namespace {
struct Slot {
  void* ptr = nullptr;
  size_t bytes = 0;
};
Slot g_slots[16][8];
std::mutex g_mu;
}  // namespace

void* f2n_ws_get(int slot, size_t bytes) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16 || slot < 0 || slot >= 8) return nullptr;
  std::lock_guard<std::mutex> lock(g_mu);
  Slot& s = g_slots[dev][slot];
  if (s.bytes < bytes) {
    if (s.ptr) (void) hipFree(s.ptr);
    s.ptr = nullptr;
    s.bytes = 0;
    size_t want = bytes + bytes / 4;
    if (hipMalloc(&s.ptr, want) != hipSuccess) return nullptr;
    s.bytes = want;
  }
  return s.ptr;
}

// out[i] += sum_b partials[b * n + i]
__global__ void f2n_reduce_partials_kernel(int n, int n_blocks, const float* __restrict__ partials, float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float s = 0.f;
  for (int b = 0; b < n_blocks; b++) s += partials[(size_t) b * n + i];
  out[i] += s;
}

int f2n_reduce_partials(void* stream, int n, int n_blocks, const float* partials, float* out) {
  hipLaunchKernelGGL(f2n_reduce_partials_kernel, dim3(f2n_div_up(n, 256)), dim3(256), 0, (hipStream_t) stream, n, n_blocks, partials, out);
  return f2n_launch_status();
}
