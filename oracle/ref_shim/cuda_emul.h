// TEST INFRASTRUCTURE ONLY -- not part of the product path.
//
// Minimal CUDA-on-CPU emulation layer used by oracle/build_ref.py to compile the reference's own
// first-party __global__ kernels (read in place from /root/reference/src/**.cu, never copied into
// this repo) into oracle/_ref/libf2n_ref.so.  Threads are executed serially: every "launch" is a
// plain nested loop over (blockIdx, threadIdx), so atomics degenerate to ordinary read-modify-write
// and segment allocation via atomicAdd happens in ray order.
//
// What this shim deliberately mirrors from CUDA device semantics:
//   * Eigen is compiled with EIGEN_DONT_VECTORIZE, as it is inside nvcc device code, so fixed-size
//     products/reductions use Eigen's scalar (tree-shaped) unrollers -- the op ORDER of the reference.
//   * __half / __half2 arithmetic = IEEE binary16 with round-to-nearest-even (Eigen::half).
// What it cannot mirror: nvcc's FMA contraction and CUDA libm ulps (exp2f etc.); the TU is built with
// -ffp-contract=off and glibc libm.  See DESIGN.md "oracle".
#pragma once
#include <cstdint>
#include <cstring>
#include <cmath>
#include <algorithm>
#include <limits>
#include <vector>
#include <functional>

#define EIGEN_DONT_VECTORIZE 1
#define EIGEN_DISABLE_UNALIGNED_ARRAY_ASSERT 1
#define EIGEN_MAX_STATIC_ALIGN_BYTES 0
#include <Eigen/Eigen>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline

struct emu_dim3 {
  unsigned x = 1, y = 1, z = 1;
};
static thread_local emu_dim3 threadIdx, blockIdx, blockDim, gridDim;

template <class F>
static inline void emu_launch(unsigned gx, unsigned gy, unsigned bx, F&& body) {
  gridDim.x = gx; gridDim.y = gy; gridDim.z = 1;
  blockDim.x = bx; blockDim.y = 1; blockDim.z = 1;
  for (unsigned by_ = 0; by_ < gy; by_++) {
    for (unsigned bx_ = 0; bx_ < gx; bx_++) {
      for (unsigned t = 0; t < bx; t++) {
        blockIdx.x = bx_; blockIdx.y = by_; blockIdx.z = 0;
        threadIdx.x = t; threadIdx.y = 0; threadIdx.z = 0;
        body();
      }
    }
  }
}

static inline int atomicAdd(int* p, int v) { int o = *p; *p = o + v; return o; }
static inline float atomicAdd(float* p, float v) { float o = *p; *p = o + v; return o; }
static inline int atomicMax(int* p, int v) { int o = *p; *p = (v > o) ? v : o; return o; }

using __half = Eigen::half;
struct __half2 {
  __half x, y;
};
static inline __half2 atomicAdd(__half2* p, __half2 v) {
  __half2 o = *p;
  p->x = __half(float(o.x) + float(v.x));
  p->y = __half(float(o.y) + float(v.y));
  return o;
}

// Device float->unsigned conversion.  The hash kernels do static_cast<unsigned>(floorf(q)) with q
// possibly negative (Hash3DAnchored.cu:44-46,115-117).  CUDA's cvt.rzi.u32.f32 (and AMD's
// v_cvt_u32_f32) SATURATE (negative/NaN -> 0, >= 2^32 -> 0xffffffff); the same cast on x86 is
// undefined behaviour and wraps in practice.  floorf() is therefore routed through a thin wrapper whose
// conversion to unsigned saturates like the device; every other use sees a plain float.
struct emu_floor_t {
  float v;
  operator float() const { return v; }
  explicit operator unsigned() const {
    if (!(v > 0.f)) return 0u;
    if (v >= 4294967296.f) return 0xffffffffu;
    return (unsigned) v;
  }
};
static inline emu_floor_t emu_floorf(float x) { return emu_floor_t{std::floor(x)}; }
#define floorf(x) emu_floorf(x)
