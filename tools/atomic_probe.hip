// Micro-probe (not part of the product): throughput of different global atomic flavours on random addresses.
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t rnd(uint32_t& s) { s = s * 1664525u + 1013904223u; uint32_t x = s; x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; return x; }

// kind: 0 pk_f16, 1 f32, 2 i32, 3 u64, 4 f32 x2 (two adjacent floats), 5 pk_f16 with wave-uniform duplicate addresses
__global__ void atomic_probe_kernel(uint32_t* table, uint32_t span_dwords, int kind, int part_mode, int iters) {
  uint32_t s = (blockIdx.x * blockDim.x + threadIdx.x) * 9781u + 12345u;
  if (part_mode == 2 && (blockIdx.x % 8u) != 0u) return;  // one XCD only
  const uint32_t region = part_mode ? (blockIdx.x % 8u) : 0u;
  const uint32_t span = part_mode ? span_dwords / 8u : span_dwords;
  uint32_t* base = table + (part_mode ? region * (size_t) span : 0);
  for (int i = 0; i < iters; i++) {
    uint32_t idx = rnd(s) % span;
    if (kind == 5) idx = (idx & ~15u) | ((threadIdx.x >> 2) & 15u);  // 4 lanes share an address
    if (kind == 0 || kind == 5) {
      half2_t v = {(_Float16) 0.001f, (_Float16) 0.002f};
      __builtin_amdgcn_global_atomic_fadd_v2f16((__attribute__((address_space(1))) half2_t*) (base + idx), v);
    } else if (kind == 1) {
      atomicAdd((float*) (base + idx), 0.001f);
    } else if (kind == 2) {
      atomicAdd((int*) (base + idx), 3);
    } else if (kind == 3) {
      atomicAdd((unsigned long long*) (base + (idx & ~1u)), 0x0000000100000001ull);
    } else if (kind == 6) {
      __hip_atomic_fetch_add((float*) (base + idx), 0.001f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    } else if (kind == 7) {
      float old = atomicAdd((float*) (base + idx), 0.001f);
      if (old == 123.456f) base[0] = 1;  // keep the returned value alive
    } else if (kind == 8) {  // plain (racy) read-modify-write: the no-atomic upper bound
      float* q = (float*) (base + idx);
      *q = *q + 0.001f;
    } else if (kind == 4) {
      atomicAdd((float*) (base + (idx & ~1u)), 0.001f);
      atomicAdd((float*) (base + (idx | 1u)), 0.002f);
    }
  }
}
extern "C" int atomic_probe_launch(void* stream, void* table, uint32_t span_dwords, int kind, int part_mode, int iters, int blocks) {
  hipLaunchKernelGGL(atomic_probe_kernel, dim3(blocks), dim3(256), 0, (hipStream_t) stream, (uint32_t*) table, span_dwords, kind, part_mode, iters);
  return (int) hipGetLastError();
}
