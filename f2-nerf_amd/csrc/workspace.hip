// Internal scratch workspaces of the C-ABI library (never visible to callers): per-block partial sums of the
// backward kernels.  Grown on demand with hipMalloc, one buffer per (device, slot), never freed.
//
// Why partial sums instead of atomics: thousands of fp32 atomics onto the SAME few thousand addresses (MLP
// parameters, appearance embeddings) serialise at the memory-side atomic unit (~1 us per dependent RMW measured
// on MI355X: 1e3-deep chains cost ~1 ms).  Each block therefore writes its partial vector with plain coalesced
// stores and a tiny second kernel reduces them.
#include "f2n_dev.h"

#include <mutex>

#define F2N_RED_GROUPS 16  // source-block groups per reduction block (see f2n_reduce_partials_kernel)

namespace {
struct Slot {
  void* ptr = nullptr;
  size_t bytes = 0;
};
Slot g_slots[16][F2N_WS_SLOTS];
std::mutex g_mu;
}  // namespace

void* f2n_ws_get(int slot, size_t bytes) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16 || slot < 0 || slot >= F2N_WS_SLOTS) return nullptr;
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
    // a workspace starts out zeroed (arrival counters of last-block reductions live in some of them and return to zero by
    // themselves afterwards); the callers' streams do not synchronise with the null stream, hence the drain
    if (hipMemset(s.ptr, 0, want) != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
      (void) hipFree(s.ptr);
      s.ptr = nullptr;
      return nullptr;
    }
    s.bytes = want;
  }
  return s.ptr;
}

// out[i] += sum_b partials[b * n + i].  A 1024-thread block handles 64 parameters x 16 groups of source blocks; every
// thread keeps 8 independent loads in flight (the sequential one-thread-per-parameter loop was a 256-deep chain of
// dependent ~0.25 us reads: 60 us per call, three calls per training step; 4 groups: 14 us; 16 groups: two load rounds
// for 256 source blocks).  The summation order is fixed: the result does not depend on scheduling.
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

// Deferred reductions: the backward entry points called with defer_reduce leave their per-block partials in the workspace
// and register them here; f2n_reduce_deferred folds all of them into their destinations with ONE launch (three dependent
// launches of a few microseconds each -- colour-MLP weights, appearance embedding, field-MLP weights -- sat on the step's
// critical path, each with the dispatcher's ~4 us hand-over).  Same per-element summation order as the single reduction.
struct Deferred {
  const float* partials;
  float* out;
  int n, n_blocks;
};
static Deferred g_deferred[16][4];
static int g_n_deferred[16];

int f2n_defer_reduction(int n, int n_blocks, const float* partials, float* out) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return F2N_ERR_INVALID_ARG;
  std::lock_guard<std::mutex> lock(g_mu);
  if (g_n_deferred[dev] >= 4) return F2N_ERR_UNSUPPORTED;
  g_deferred[dev][g_n_deferred[dev]++] = Deferred{partials, out, n, n_blocks};
  return F2N_OK;
}

struct F2nDeferredArgs {
  Deferred d[4];
  int first_block[5];
};

__global__ __launch_bounds__(64 * F2N_RED_GROUPS) void f2n_reduce_deferred_kernel(F2nDeferredArgs a, int n_seg) {
  __shared__ float s_part[F2N_RED_GROUPS][64];
  int seg = 0;
  while (seg + 1 < n_seg && (int) blockIdx.x >= a.first_block[seg + 1]) seg++;
  const Deferred d = a.d[seg];
  const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int i = (blockIdx.x - a.first_block[seg]) * 64 + lane;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (i < d.n) {
    int b = grp;
    for (; b + 7 * F2N_RED_GROUPS < d.n_blocks; b += 8 * F2N_RED_GROUPS) {
#pragma unroll
      for (int u = 0; u < 8; u++) acc[u] += d.partials[(size_t) (b + F2N_RED_GROUPS * u) * d.n + i];
    }
    for (; b < d.n_blocks; b += F2N_RED_GROUPS) acc[0] += d.partials[(size_t) b * d.n + i];
  }
  s_part[grp][lane] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
  __syncthreads();
  if (grp == 0 && i < d.n) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < F2N_RED_GROUPS; k += 4) t += (s_part[k][lane] + s_part[k + 1][lane]) + (s_part[k + 2][lane] + s_part[k + 3][lane]);
    d.out[i] += t;
  }
}

// Forgets whatever was registered and not yet folded (a caller whose backward threw between its deferring launches and
// f2n_reduce_deferred must not have those stale entries folded into the NEXT step's gradients).
extern "C" int f2n_deferred_reset(void) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return F2N_ERR_INVALID_ARG;
  std::lock_guard<std::mutex> lock(g_mu);
  g_n_deferred[dev] = 0;
  return F2N_OK;
}

extern "C" int f2n_reduce_deferred(void* stream) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return F2N_ERR_INVALID_ARG;
  F2nDeferredArgs a = {};
  int n_seg = 0;
  {
    std::lock_guard<std::mutex> lock(g_mu);
    n_seg = g_n_deferred[dev];
    int block = 0;
    for (int k = 0; k < n_seg; k++) {
      a.d[k] = g_deferred[dev][k];
      a.first_block[k] = block;
      block += (int) f2n_div_up(a.d[k].n, 64);
    }
    a.first_block[n_seg] = block;
    g_n_deferred[dev] = 0;
  }
  if (n_seg == 0 || a.first_block[n_seg] == 0) return F2N_OK;
  hipLaunchKernelGGL(f2n_reduce_deferred_kernel, dim3(a.first_block[n_seg]), dim3(64 * F2N_RED_GROUPS), 0, (hipStream_t) stream, a, n_seg);
  return f2n_launch_status();
}

int f2n_reduce_partials(void* stream, int n, int n_blocks, const float* partials, float* out) {
  hipLaunchKernelGGL(f2n_reduce_partials_kernel, dim3(f2n_div_up(n, 64)), dim3(64 * F2N_RED_GROUPS), 0, (hipStream_t) stream, n, n_blocks, partials, out);
  return f2n_launch_status();
}
