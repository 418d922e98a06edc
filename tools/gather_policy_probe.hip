// Micro-probe (not part of the product): random 4-byte gathers from an L2-resident table slice (block b -> slice b % 8,
// 2^20 entries = 4 MiB... region_entries given) under the cache policies a gfx950 load can carry.
//   policy 0: global_load_dword           1: buffer_load_dword          2: buffer_load sc0        3: buffer_load nt
//   policy 4: buffer_load sc1             5: buffer_load sc0 sc1        6: buffer_load sc0 nt     7: global_load nt
//   policy 8: two adjacent entries as one 8-byte load (what a locality-preserving hash would allow)
#include <hip/hip_runtime.h>
#include <stdint.h>
__device__ __forceinline__ uint32_t rnd(uint32_t& s) { s = s * 1664525u + 1013904223u; uint32_t x = s; x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; return x; }

template <int POLICY>
__global__ __launch_bounds__(256) void gather_probe_kernel(const uint32_t* table, uint32_t region_entries, int iters, uint32_t* sink) {
  uint32_t s = (blockIdx.x * blockDim.x + threadIdx.x) * 9781u + 12345u;
  const uint32_t* base = table + (size_t) (blockIdx.x % 8u) * region_entries;
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*) base, 0, region_entries * 4u, 0x00020000);
  uint32_t acc = 0;
  for (int i = 0; i < iters; i += 8) {
    uint32_t idx[8], v[8];
#pragma unroll
    for (int k = 0; k < 8; k++) idx[k] = rnd(s) & (region_entries - 1u);
#pragma unroll
    for (int k = 0; k < 8; k++) {
      if (POLICY == 0) v[k] = base[idx[k]];
      else if (POLICY == 1) v[k] = __builtin_amdgcn_raw_buffer_load_b32(r, idx[k] * 4u, 0, 0);
      else if (POLICY == 2) v[k] = __builtin_amdgcn_raw_buffer_load_b32(r, idx[k] * 4u, 0, 1);
      else if (POLICY == 3) v[k] = __builtin_amdgcn_raw_buffer_load_b32(r, idx[k] * 4u, 0, 2);
      else if (POLICY == 4) v[k] = __builtin_amdgcn_raw_buffer_load_b32(r, idx[k] * 4u, 0, 16);
      else if (POLICY == 5) v[k] = __builtin_amdgcn_raw_buffer_load_b32(r, idx[k] * 4u, 0, 17);
      else if (POLICY == 6) v[k] = __builtin_amdgcn_raw_buffer_load_b32(r, idx[k] * 4u, 0, 3);
      else if (POLICY == 7) v[k] = __builtin_nontemporal_load(base + idx[k]);
      else { const uint2 t = *(const uint2*) (base + (idx[k] & ~1u)); v[k] = t.x + t.y; }
    }
#pragma unroll
    for (int k = 0; k < 8; k++) acc += v[k];
  }
  if (acc == 0x12345678u) sink[0] = acc;
}

extern "C" int gather_probe_launch(void* stream, int policy, void* table, uint32_t region_entries, int iters, int blocks, void* sink) {
#define L(P) case P: hipLaunchKernelGGL(gather_probe_kernel<P>, dim3(blocks), dim3(256), 0, (hipStream_t) stream, (const uint32_t*) table, region_entries, iters, (uint32_t*) sink); break;
  switch (policy) { L(0) L(1) L(2) L(3) L(4) L(5) L(6) L(7) L(8) default: return -1; }
  return (int) hipGetLastError();
}
