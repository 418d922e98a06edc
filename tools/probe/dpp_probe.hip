// Probe: what do the wave-level DPP controls return on gfx950?  (tools only)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* out) {
  const int lane = threadIdx.x;
  out[lane] = __builtin_amdgcn_update_dpp(-1, lane, 0x138, 0xF, 0xF, false);        // wave_shr:1
  out[64 + lane] = __builtin_amdgcn_update_dpp(-1, lane, 0x111, 0xF, 0xF, false);   // row_shr:1
  out[128 + lane] = __builtin_amdgcn_update_dpp(-1, lane, 0x142, 0xA, 0xF, false);  // row_bcast:15 row_mask 0xA
  out[192 + lane] = __shfl_up(lane, 1);
}
int main() {
  int* d; hipMalloc(&d, 256 * 4);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  int h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  const char* names[4] = {"wave_shr:1", "row_shr:1", "row_bcast15", "shfl_up"};
  for (int r = 0; r < 4; r++) { printf("%-12s", names[r]); for (int i = 0; i < 64; i++) printf(" %d", h[64 * r + i]); printf("\n"); }
  return 0;
}
