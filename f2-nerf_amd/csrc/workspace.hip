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
    // Growth is rare (sizes settle after a few iterations) but the host runs up to an iteration ahead of the device:
    // kernels that were handed the old buffer may still be queued.  Drain the device before the buffer goes away
    // (hipFree alone was observed not to wait: an intermittent memory-access fault in long trainings, once the
    // edge-sample forward and the field backward of one iteration shared F2N_WS_FIELD_PLANES).
    if (s.ptr) {
      (void) hipDeviceSynchronize();
      (void) hipFree(s.ptr);
    }
    s.ptr = nullptr;
    s.bytes = 0;
    size_t want = bytes + bytes / 4;
    if (hipMalloc(&s.ptr, want) != hipSuccess) return nullptr;
    s.bytes = want;
  }
  return s.ptr;
}

// out[i] += sum_b partials[b * n + i].  A 1024-thread block handles 64 parameters x 16 groups of source blocks; every
// thread keeps 8 independent loads in flight (the sequential one-thread-per-parameter loop was a 256-deep chain of
// dependent ~0.25 us reads: 60 us per call, three calls per training step; 4 groups: 14 us; 16 groups: two load rounds
// for 256 source blocks).  The summation order is fixed: the result does not depend on scheduling.
#define F2N_RED_GROUPS 16
__global__ __launch_bounds__(64 * F2N_RED_GROUPS) void f2n_reduce_partials_kernel(int n, int n_blocks, const float* __restrict__ partials,
                                                                                   float* __restrict__ out) {
  __shared__ float s_part[F2N_RED_GROUPS][64];
  const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + lane;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (i < n) {
    int b = grp;
    for (; b + 7 * F2N_RED_GROUPS < n_blocks; b += 8 * F2N_RED_GROUPS) {
#pragma unroll
      for (int u = 0; u < 8; u++) acc[u] += partials[(size_t) (b + F2N_RED_GROUPS * u) * n + i];
    }
    for (; b < n_blocks; b += F2N_RED_GROUPS) acc[0] += partials[(size_t) b * n + i];
  }
  s_part[grp][lane] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
  __syncthreads();
  if (grp == 0 && i < n) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < F2N_RED_GROUPS; k += 4) t += (s_part[k][lane] + s_part[k + 1][lane]) + (s_part[k + 2][lane] + s_part[k + 3][lane]);
    out[i] += t;
  }
}

int f2n_reduce_partials(void* stream, int n, int n_blocks, const float* partials, float* out) {
  hipLaunchKernelGGL(f2n_reduce_partials_kernel, dim3(f2n_div_up(n, 64)), dim3(64 * F2N_RED_GROUPS), 0, (hipStream_t) stream, n, n_blocks, partials, out);
  return f2n_launch_status();
}
