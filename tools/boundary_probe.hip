// Micro-probe (not part of the product): what a dependent kernel boundary costs on this box, EAGER (the host queues far ahead of
// the device, as ExpRunner::TrainStep does) against the same chain replayed from a hipGraph.  A chain of `n` dependent kernels,
// each a grid of `blocks` x 256 threads that reads and writes `bytes_per_kernel` of a buffer (so that the predecessor leaves
// dirty lines behind, as the step's real kernels do) -- device time of the whole chain by HIP events.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

__global__ void touch_kernel(float* buf, size_t n_floats, float add) {
  size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t) gridDim.x * blockDim.x;
  for (; i < n_floats; i += stride) buf[i] = buf[i] + add;
}

static float run_chain(hipStream_t st, float* buf, size_t n_floats, int n, int blocks) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipEventRecord(e0, st);
  for (int k = 0; k < n; k++) hipLaunchKernelGGL(touch_kernel, dim3(blocks), dim3(256), 0, st, buf, n_floats, 1.f);
  hipEventRecord(e1, st);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  hipEventDestroy(e0);
  hipEventDestroy(e1);
  return ms;
}

extern "C" int boundary_probe(size_t bytes_per_kernel, int n, int blocks, float* out_ms /*[4]: eager, graph, eager 1 kernel x n work, reserved*/) {
  float* buf = nullptr;
  const size_t n_floats = bytes_per_kernel / 8;  // read + write = 8 bytes per float touched
  if (hipMalloc(&buf, (n_floats ? n_floats : 1) * sizeof(float)) != hipSuccess) return 1;
  hipMemset(buf, 0, (n_floats ? n_floats : 1) * sizeof(float));
  hipStream_t st;
  hipStreamCreate(&st);
  run_chain(st, buf, n_floats, n, blocks);  // warm-up
  float best_e = 1e30f, best_g = 1e30f;
  for (int rep = 0; rep < 5; rep++) {
    const float ms = run_chain(st, buf, n_floats, n, blocks);
    if (ms < best_e) best_e = ms;
  }
  // the same chain captured once and replayed
  hipGraph_t graph;
  hipGraphExec_t exec;
  hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
  for (int k = 0; k < n; k++) hipLaunchKernelGGL(touch_kernel, dim3(blocks), dim3(256), 0, st, buf, n_floats, 1.f);
  if (hipStreamEndCapture(st, &graph) != hipSuccess) return 2;
  if (hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) != hipSuccess) return 3;
  hipGraphLaunch(exec, st);
  hipStreamSynchronize(st);
  for (int rep = 0; rep < 5; rep++) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipEventRecord(e0, st);
    hipGraphLaunch(exec, st);
    hipEventRecord(e1, st);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    if (ms < best_g) best_g = ms;
    hipEventDestroy(e0);
    hipEventDestroy(e1);
  }
  // one kernel doing the n-fold work in one launch: the boundary-free reference point
  float best_1 = 1e30f;
  for (int rep = 0; rep < 5; rep++) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipEventRecord(e0, st);
    hipLaunchKernelGGL(touch_kernel, dim3(blocks), dim3(256), 0, st, buf, n_floats, 1.f);
    hipEventRecord(e1, st);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    if (ms < best_1) best_1 = ms;
    hipEventDestroy(e0);
    hipEventDestroy(e1);
  }
  out_ms[0] = best_e;
  out_ms[1] = best_g;
  out_ms[2] = best_1;
  out_ms[3] = 0.f;
  hipGraphExecDestroy(exec);
  hipGraphDestroy(graph);
  hipStreamDestroy(st);
  hipFree(buf);
  return 0;
}
