// Micro-probe (not part of the product): throughput of LDS atomic flavours on random addresses inside a 64 KB image.
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t rnd(uint32_t& s) { s = s * 1664525u + 1013904223u; uint32_t x = s; x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; return x; }

// kind: 0 ds_add_f32, 1 ds_add_u32, 2 ds_add_u64, 3 ds_pk_add_f16, 4 ds_add_f64, 5 plain racy RMW, 6 2x ds_add_f32 (adjacent), 7 address generation only
__global__ __launch_bounds__(256) void lds_probe_kernel(int kind, int iters, uint32_t* sink) {
  __shared__ uint32_t s[16384];
  for (int i = threadIdx.x; i < 16384; i += 256) s[i] = 0;
  __syncthreads();
  uint32_t st = (blockIdx.x * 256 + threadIdx.x) * 9781u + 12345u;
  uint32_t acc = 0;
  for (int i = 0; i < iters; i++) {
    const uint32_t idx = rnd(st) & 16383u;
    if (kind == 0) atomicAdd((float*) &s[idx], 0.001f);
    else if (kind == 1) atomicAdd(&s[idx], 3u);
    else if (kind == 2) atomicAdd((unsigned long long*) &s[idx & ~1u], 0x0000000100000001ull);
    else if (kind == 3) { half2_t v = {(_Float16) 0.001f, (_Float16) 0.002f}; __builtin_amdgcn_ds_atomic_fadd_v2f16((__attribute__((address_space(3))) half2_t*) &s[idx], v); }
    else if (kind == 4) atomicAdd((double*) &s[idx & ~1u], 0.001);
    else if (kind == 5) s[idx] = s[idx] + 3u;
    else if (kind == 6) { atomicAdd((float*) &s[idx & ~1u], 0.001f); atomicAdd((float*) &s[idx | 1u], 0.002f); }
    else acc += idx;
  }
  __syncthreads();
  if (threadIdx.x == 0) sink[blockIdx.x] = s[acc & 16383u] + acc;
}
extern "C" int lds_probe_launch(void* stream, int kind, int iters, int blocks, void* sink) {
  hipLaunchKernelGGL(lds_probe_kernel, dim3(blocks), dim3(256), 0, (hipStream_t) stream, kind, iters, (uint32_t*) sink);
  return (int) hipGetLastError();
}
