// Test infrastructure -- NOT part of the product.  Known-answer kernels for the wavefront emulation itself: every cross-lane
// operation the product's kernels use, under full and partial EXEC masks, written as a kernel would write it (plain HIP source);
// the expected values in tests/test_wave_emul_cpu.py are the CDNA ISA's.
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef _Float16 st_half4 __attribute__((ext_vector_type(4)));
typedef _Float16 st_half8 __attribute__((ext_vector_type(8)));
typedef float st_float4 __attribute__((ext_vector_type(4)));

// out[case][64]; one wave
__global__ void st_permutes_kernel(int* __restrict__ out) {
  const int l = threadIdx.x;
  int c = 0;
  out[64 * c++ + l] = __builtin_amdgcn_mov_dpp(l, 0xB1, 0xF, 0xF, true);            // quad_perm [1,0,3,2]
  out[64 * c++ + l] = __builtin_amdgcn_mov_dpp(l, 0x4E, 0xF, 0xF, true);            // quad_perm [2,3,0,1]
  out[64 * c++ + l] = __builtin_amdgcn_update_dpp(-7, l, 0x111, 0xF, 0xF, false);   // row_shr:1, old kept at the row's head
  out[64 * c++ + l] = __builtin_amdgcn_update_dpp(-7, l, 0x113, 0xF, 0xF, true);    // row_shr:3, bound_ctrl: 0 at the row's head
  out[64 * c++ + l] = __builtin_amdgcn_update_dpp(-7, l, 0x102, 0xF, 0xF, false);   // row_shl:2
  out[64 * c++ + l] = __builtin_amdgcn_mov_dpp(l, 0x140, 0xF, 0xF, true);           // row_mirror
  out[64 * c++ + l] = __builtin_amdgcn_mov_dpp(l, 0x141, 0xF, 0xF, true);           // row_half_mirror
  out[64 * c++ + l] = __builtin_amdgcn_update_dpp(-7, l, 0xB1, 0x5, 0xF, false);    // row_mask 0b0101: rows 1 and 3 keep old
  out[64 * c++ + l] = __builtin_amdgcn_update_dpp(-7, l, 0xB1, 0xF, 0x3, false);    // bank_mask 0b0011: lanes 8..15 of a row keep old
  out[64 * c++ + l] = __builtin_amdgcn_update_dpp(-7, l, 0x121, 0xF, 0xF, false);   // row_ror:1
  out[64 * c++ + l] = __shfl(l, 15, 16);
  out[64 * c++ + l] = __shfl(l, 70);                                                // source lane taken modulo the width
  out[64 * c++ + l] = __shfl_xor(l, 32);
  out[64 * c++ + l] = __shfl_xor(l, 3, 16);
  out[64 * c++ + l] = __shfl_up(l, 1, 16);
  out[64 * c++ + l] = __shfl_down(l, 2, 8);
  out[64 * c++ + l] = (int) __shfl((float) l * 0.5f, 7, 8);
  const long long big = (1ll << 40) + l;
  out[64 * c++ + l] = (int) (__shfl_xor(big, 1) - (1ll << 40));                      // 8-byte value
}

// out[case][64] of 64-bit words; one wave
__global__ void st_masks_kernel(unsigned long long* __restrict__ out) {
  const int l = threadIdx.x;
  for (int c = 0; c < 8; c++) out[64 * c + l] = 0xDEADull;
  out[64 * 0 + l] = __ballot(l % 3 == 0);
  if (l & 1) out[64 * 1 + l] = __ballot(l < 40);          // only odd lanes vote
  else out[64 * 2 + l] = __ballot(l >= 60);               // the other branch: only even lanes
  if (l >= 13) out[64 * 3 + l] = (unsigned long long) __builtin_amdgcn_readfirstlane(l);
  if (l != 5) out[64 * 4 + l] = (unsigned long long) (unsigned) __shfl(l + 100, 5);       // the source lane is inactive: ds_bpermute returns 0
  if (l >= 2) out[64 * 5 + l] = (unsigned long long) (unsigned) __builtin_amdgcn_update_dpp(-7, l, 0x111, 0xF, 0xF, false);  // lane 2's source is inactive: old
  int seen = 0;                                           // lanes leave the loop at different times: each trip's vote counts who is left
  for (int i = 0; i < (l >> 3); i++) seen += __popcll(__ballot(1));
  out[64 * 6 + l] = (unsigned long long) seen;
  if (l >= 50) return;
  out[64 * 7 + l] = __ballot(1);                          // lanes 50..63 have left the kernel
}

// 256 work-items: static and dynamic LDS, barriers, the barrier's vote
__global__ void st_block_kernel(int* __restrict__ out, int n) {
  __shared__ int s_rev[256];
  extern __shared__ int s_dyn[];
  const int t = threadIdx.x;
  s_rev[t] = blockIdx.x * 1000 + t;
  s_dyn[t] = t * t;
  __syncthreads();
  const int a = s_rev[255 - t], b = s_dyn[255 - t];
  const int any_big = __syncthreads_or(t == 200 && blockIdx.x == 1);
  if (t < 64) {  // (wave 0 alone goes on; the others have left)
    int s = a;
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off);
    out[blockIdx.x * 256 + t] = s;
  } else {
    out[blockIdx.x * 256 + t] = a + b + 1000000 * any_big;
  }
  if (t == 0) atomicAdd(out + n, 1);
}

// A persistent wave: groups of work off a counter; the four rows of the wave work for different lengths inside a trip, and the wave
// must be whole again at the top of the next one (the lanes that finish early wait at the end of the body, not at the loop's head)
__global__ void st_persistent_kernel(int* __restrict__ out, int* __restrict__ counter, int n_groups) {
  for (;;) {
    int g = 0;
    if ((threadIdx.x & 63) == 0) g = atomicAdd(counter, 1);
    g = __builtin_amdgcn_readfirstlane(g);
    if (g >= n_groups) break;
    const int row = threadIdx.x >> 4;
    const int trips = (g * 7 + row * 5) % 11;
    int acc = 0;
    if (trips != 3)  // (one row in some trips skips the work altogether)
      for (int i = 0; i < trips; i++) acc += __shfl_xor(i + (int) threadIdx.x, 1, 16);
    out[g * 64 + threadIdx.x] += acc + 1;
  }
}

// D = A B + C with the fragment layouts of v_mfma_f32_16x16x16_f16 / v_mfma_f32_16x16x32_f16; A, B row-major f16 [16][K], [K][16]
__global__ void st_mfma_kernel(const _Float16* __restrict__ A, const _Float16* __restrict__ B, const float* __restrict__ C,
                               float* __restrict__ D16, float* __restrict__ D32) {
  const int l = threadIdx.x, i = l & 15, q = l >> 4;
  st_half4 a4, b4;
  st_half8 a8, b8;
  st_float4 c;
  for (int k = 0; k < 4; k++) {
    a4[k] = A[i * 32 + 4 * q + k];
    b4[k] = B[(4 * q + k) * 16 + i];
  }
  for (int k = 0; k < 8; k++) {
    a8[k] = A[i * 32 + 8 * q + k];
    b8[k] = B[(8 * q + k) * 16 + i];
  }
  for (int r = 0; r < 4; r++) c[r] = C[(4 * q + r) * 16 + i];
  const st_float4 d16 = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, c, 0, 0, 0);   // K = 16: the first 16 columns of A / rows of B
  const st_float4 d32 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, c, 0, 0, 0);  // K = 32
  for (int r = 0; r < 4; r++) {
    D16[(4 * q + r) * 16 + i] = d16[r];
    D32[(4 * q + r) * 16 + i] = d32[r];
  }
}

extern "C" {
int st_permutes(int* out) {
  hipLaunchKernelGGL(st_permutes_kernel, dim3(1), dim3(64), 0, (hipStream_t) 0, out);
  return (int) hipGetLastError();
}
int st_masks(unsigned long long* out) {
  hipLaunchKernelGGL(st_masks_kernel, dim3(1), dim3(64), 0, (hipStream_t) 0, out);
  return (int) hipGetLastError();
}
int st_block(int* out, int n_blocks) {
  hipLaunchKernelGGL(st_block_kernel, dim3(n_blocks), dim3(256), 256 * sizeof(int), (hipStream_t) 0, out, n_blocks * 256);
  return (int) hipGetLastError();
}
int st_persistent(int* out, int* counter, int n_groups, int n_blocks) {
  hipLaunchKernelGGL(st_persistent_kernel, dim3(n_blocks), dim3(64), 0, (hipStream_t) 0, out, counter, n_groups);
  return (int) hipGetLastError();
}
int st_mfma(const void* A, const void* B, const float* C, float* D16, float* D32) {
  hipLaunchKernelGGL(st_mfma_kernel, dim3(1), dim3(64), 0, (hipStream_t) 0, (const _Float16*) A, (const _Float16*) B, C, D16, D32);
  return (int) hipGetLastError();
}
}
