// Micro-probe (not part of the product): does XCD-aware partitioning of a hash table make random 4-byte gathers /
// packed-f16 atomics faster on MI355X?  Each lane does `iters` random accesses into a table region.
//   mode 0: every block uses the whole table (n_regions * region_entries entries)
//   mode 1: block b only touches region (b % 8)   (observed placement: block b runs on XCD b % 8)
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t rnd(uint32_t& s) { s = s * 1664525u + 1013904223u; uint32_t x = s; x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; return x; }

__global__ void probe_kernel(half2_t* table, uint32_t region_entries, int n_regions, int mode, int do_atomic, int iters, float* sink) {
  uint32_t s = (blockIdx.x * blockDim.x + threadIdx.x) * 9781u + 12345u;
  const uint32_t region = mode ? (blockIdx.x % 8u) % n_regions : 0u;
  const uint32_t span = mode ? region_entries : region_entries * n_regions;
  half2_t* base = table + (mode ? region * (size_t) region_entries : 0);
  float acc = 0.f;
  for (int i = 0; i < iters; i += 8) {
    uint32_t idx[8];
#pragma unroll
    for (int k = 0; k < 8; k++) idx[k] = rnd(s) % span;
    if (do_atomic) {
#pragma unroll
      for (int k = 0; k < 8; k++) {
        half2_t v = {(_Float16) 0.001f, (_Float16) 0.002f};
        __builtin_amdgcn_global_atomic_fadd_v2f16((__attribute__((address_space(1))) half2_t*) (base + idx[k]), v);
      }
    } else {
      half2_t v[8];
#pragma unroll
      for (int k = 0; k < 8; k++) v[k] = base[idx[k]];
#pragma unroll
      for (int k = 0; k < 8; k++) acc += (float) v[k][0] + (float) v[k][1];
    }
  }
  if (acc == 123.456f) sink[0] = acc;
}

extern "C" int probe_launch(void* stream, void* table, uint32_t region_entries, int n_regions, int mode, int do_atomic, int iters,
                            int blocks, void* sink) {
  hipLaunchKernelGGL(probe_kernel, dim3(blocks), dim3(256), 0, (hipStream_t) stream, (half2_t*) table, region_entries, n_regions, mode,
                     do_atomic, iters, (float*) sink);
  return (int) hipGetLastError();
}
