// Test infrastructure -- NOT part of the product.  A CPU stand-in for <hip/hip_runtime.h> under which the kernels of
// f2-nerf_amd/csrc/*.hip compile for x86-64 (ROCm's clang++ as a plain host compiler) and run with the execution model they are
// written for: 64-lane wavefronts, cross-lane operations, LDS, workgroup barriers.  tests/wave_emul/wemu_build.py builds
// libf2n_emul.so from the product's own source text; tests/test_wave_emul_cpu.py runs the bodies of the -m gpu parity tests
// against it, so that the cooperative part of every kernel (DPP chains, ballots, shuffles, LDS stacks, MFMA tiles) is held against
// the oracle on every CPU run, not only on the GPU box.  Nothing here is measured or shipped; the product library is built by
// hipcc for gfx950 from the same files and has no CPU path.
//
// Execution model (wemu_rt.cpp): every work-item of a workgroup is a fibre (own stack, hand-written context switch) on ONE OS
// thread; workgroups run one after another.  A fibre runs until it reaches a cross-lane operation or a barrier and parks there
// with its operand.  A wave's cross-lane operation is carried out when every live lane of the wave is parked: the lanes parked at
// the same call site form the EXEC mask of that operation (lanes that returned from the kernel, or that sit in the other branch
// of a divergent `if`, are inactive -- exactly what the hardware's mask would hold); a barrier opens when every live work-item of
// the group is parked at it.  Sources outside the mask read as the ISA says: 0 from ds_bpermute, `old` / 0 (bound_ctrl) from DPP.
// `__shared__` variables are function-local statics (one workgroup is alive at a time), dynamic LDS is one buffer refilled with
// 0xCD bytes before every workgroup (a kernel that reads LDS it did not write shows).  Atomics are plain operations in the
// order the fibres run.
#pragma once
#ifndef __HIPCC__
#define __HIPCC__ 1
#endif
#define F2N_WAVE_EMUL 1

#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <type_traits>

// ------------------------------------------------------------------------------------------------- qualifiers
#define __global__
#define __device__
#define __host__
#define __constant__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static
#define HIP_SYMBOL(x) (&(x))

struct dim3 {
  unsigned x, y, z;
  constexpr dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct uint2 { unsigned x, y; };
struct uint4 { unsigned x, y, z, w; };
struct int2 { int x, y; };
struct int4 { int x, y, z, w; };
struct float2 { float x, y; };
struct float3 { float x, y, z; };
struct alignas(16) float4 { float x, y, z, w; };
static inline uint2 make_uint2(unsigned x, unsigned y) { return {x, y}; }
static inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return {x, y, z, w}; }
static inline int2 make_int2(int x, int y) { return {x, y}; }
static inline int4 make_int4(int x, int y, int z, int w) { return {x, y, z, w}; }
static inline float2 make_float2(float x, float y) { return {x, y}; }
static inline float3 make_float3(float x, float y, float z) { return {x, y, z}; }
static inline float4 make_float4(float x, float y, float z, float w) { return {x, y, z, w}; }

// ------------------------------------------------------------------------------------------------- runtime (wemu_rt.cpp)
namespace wemu {
struct Idx { unsigned x, y, z; };
extern Idx g_tid, g_bid;      // of the fibre that is running
extern dim3 g_bdim, g_gdim;
extern int g_lane;            // lane of the running fibre in its wave
enum Kind { PERMUTE = 1, BALLOT, FIRST, MFMA_16x16x16_F16, MFMA_16x16x32_F16, BARRIER, BARRIER_OR };
struct Op {
  int kind;
  const void* site;    // return address of the call: lanes parked at one site form one EXEC mask
  uint32_t val;        // PERMUTE / FIRST: the lane's operand;  BALLOT / BARRIER_OR: its predicate
  int src;             // PERMUTE: the lane it reads, -1 = none (out of range)
  uint32_t fallback;   // PERMUTE: what it gets when src is -1 or inactive
  const void *a, *b, *c;  // MFMA fragments of this lane
  void* d;
  uint64_t result;     // filled in by the scheduler
};
uint64_t park(Op& op);  // parks the running fibre at `op`, returns op.result once the operation has been carried out
void* dyn_lds();        // the dynamic LDS of the running workgroup
typedef void (*Body)(void* closure);
void launch(const char* kernel, const void* stream, dim3 grid, dim3 block, size_t dyn_lds_bytes, Body body, void* closure);
extern "C" void wemu_set_schedule(int mode);  // the order workgroups and waves run in (a race detector: results must not depend on it)
extern "C" long wemu_counter(int which);
// the memory traffic of every launch since wemu_traffic_reset(), from the compiler's load / store instrumentation (only the
// "traffic" build of wemu_build.py has it): out[0..7] = global bytes loaded, stored, distinct 128-byte lines loaded, stored,
// LDS (dynamic buffer + the library's own statics) bytes loaded, stored, workgroups, work-items per workgroup
extern "C" void wemu_hb_enable(int on);          // happens-before bookkeeping + race check (the "traffic" build sees the accesses)
extern "C" long wemu_hb_record(int stream);       // an event recorded on `stream` -> token
extern "C" void wemu_hb_wait(int stream, long token);
extern "C" void wemu_hb_host_sync(long token);    // the host waited for that event (token < 0: for the whole device)
extern "C" int wemu_hb_races(void);
extern "C" long wemu_hb_race(int i, char* first, char* second, char* kind, int size, unsigned long long* addr);
extern "C" int wemu_traffic_launches(void);
extern "C" int wemu_traffic_get(int i, char* name, int name_size, unsigned long long* out8);
extern "C" void wemu_traffic_reset(void);  // 0: launches, 1: cross-lane operations, 2: operations that found a wave in more than one
                                          // group (divergent), 3: barriers, 4: work-items run, 5: activations / loops that did not fit a fibre's record (must stay 0)
}  // namespace wemu

#define threadIdx (wemu::g_tid)
#define blockIdx (wemu::g_bid)
#define blockDim (wemu::g_bdim)
#define gridDim (wemu::g_gdim)
#define warpSize 64

template <typename F>
static inline void wemu_launch_(const char* kernel, const void* stream, dim3 grid, dim3 block, size_t lds, F&& f) {
  wemu::launch(kernel, stream, grid, block, lds, [](void* c) { (*static_cast<std::remove_reference_t<F>*>(c))(); }, &f);
}
// (launches are synchronous; the stream argument only names the stream for the race detector of wemu_rt.cpp)
#define hipLaunchKernelGGL(kernel, grid, block, lds, stream, ...) \
  wemu_launch_(#kernel, (const void*) (stream), dim3(grid), dim3(block), (size_t) (lds), [&]() { kernel(__VA_ARGS__); })

// ------------------------------------------------------------------------------------------------- host API subset
typedef int hipError_t;
typedef void* hipStream_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2 };
enum hipMemcpyKind { hipMemcpyHostToHost, hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
static inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
// events: the emulation's launches are synchronous, so an event has always been reached (what the step tail's fork / join needs)
typedef void* hipEvent_t;
enum { hipEventDisableTiming = 2 };
static inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = (void*) 1; return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t, hipStream_t = nullptr) { return hipSuccess; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned = 0) { return hipSuccess; }
static inline hipError_t hipMalloc(void** p, size_t n) { return posix_memalign(p, 256, n ? n : 256) == 0 ? hipSuccess : hipErrorOutOfMemory; }
template <typename T> static inline hipError_t hipMalloc(T** p, size_t n) { return hipMalloc((void**) p, n); }
static inline hipError_t hipFree(void* p) { free(p); return hipSuccess; }
static inline hipError_t hipMemset(void* p, int v, size_t n) { memset(p, v, n); return hipSuccess; }
static inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t = nullptr) { memset(p, v, n); return hipSuccess; }
static inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { memmove(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t = nullptr) { memmove(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpyToSymbol(void* sym, const void* s, size_t n, size_t off = 0, hipMemcpyKind = hipMemcpyHostToDevice) {
  memcpy((char*) sym + off, s, n); return hipSuccess;
}
static inline hipError_t hipMemcpyFromSymbol(void* d, const void* sym, size_t n, size_t off = 0, hipMemcpyKind = hipMemcpyDeviceToHost) {
  memcpy(d, (const char*) sym + off, n); return hipSuccess;
}
template <typename K> static inline hipError_t hipFuncSetAttribute(K, hipFuncAttribute, int) { return hipSuccess; }

// ------------------------------------------------------------------------------------------------- lane-local intrinsics
static inline int __float_as_int(float f) { int i; memcpy(&i, &f, 4); return i; }
static inline unsigned __float_as_uint(float f) { unsigned i; memcpy(&i, &f, 4); return i; }
static inline float __int_as_float(int i) { float f; memcpy(&f, &i, 4); return f; }
static inline float __uint_as_float(unsigned i) { float f; memcpy(&f, &i, 4); return f; }
static inline int __popc(unsigned v) { return __builtin_popcount(v); }
static inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
static inline int __ffs(int v) { return __builtin_ffs(v); }
static inline int __ffsll(long long v) { return __builtin_ffsll(v); }
static inline int __clz(int v) { return v == 0 ? 32 : __builtin_clz((unsigned) v); }
static inline int __clzll(long long v) { return v == 0 ? 64 : __builtin_clzll((unsigned long long) v); }
static inline long long __double2ll_rn(double d) { return llrint(d); }  // (round to nearest even: the default mode)
static inline unsigned __umulhi(unsigned a, unsigned b) { return (unsigned) (((uint64_t) a * b) >> 32); }
#define __expf(x) expf(x)  // (glibc declares the double-underscore names itself)
#define __logf(x) logf(x)
static inline float __fdividef(float a, float b) { return a / b; }
static inline float __frcp_rn(float a) { return 1.f / a; }
static inline float rsqrtf(float a) { return 1.f / sqrtf(a); }
static inline float __saturatef(float a) { return a < 0.f ? 0.f : a > 1.f ? 1.f : a; }
static inline long long wall_clock64() { return 0; }
static inline long long clock64() { return 0; }
static inline void __threadfence() {}
static inline void __threadfence_block() {}
#define __builtin_amdgcn_s_sleep(n) ((void) 0)
#define __builtin_amdgcn_sched_barrier(n) ((void) 0)
#define __builtin_amdgcn_s_setprio(n) ((void) 0)
#define __HIP_MEMORY_SCOPE_AGENT 0
#define __HIP_MEMORY_SCOPE_WORKGROUP 0
#define __HIP_MEMORY_SCOPE_SYSTEM 0
#define __hip_atomic_load(p, order, scope) (*(p))
#define __hip_atomic_store(p, v, order, scope) ((void) (*(p) = (v)))

using std::max;
using std::min;
static inline int min(int a, unsigned b) { return a < (int) b ? a : (int) b; }
static inline int min(unsigned a, int b) { return (int) a < b ? (int) a : b; }
static inline int max(int a, unsigned b) { return a > (int) b ? a : (int) b; }
static inline int max(unsigned a, int b) { return (int) a > b ? (int) a : b; }
static inline long min(long a, int b) { return a < b ? a : b; }
static inline long min(int a, long b) { return a < b ? a : b; }
static inline long max(long a, int b) { return a > b ? a : b; }
static inline long max(int a, long b) { return a > b ? a : b; }
static inline float min(float a, float b) { return fminf(a, b); }
static inline float max(float a, float b) { return fmaxf(a, b); }

// ------------------------------------------------------------------------------------------------- atomics (one OS thread)
template <typename T> static inline T atomicAdd(T* p, T v) { T o = *p; *p = o + v; return o; }
static inline float atomicAdd(float* p, double v) { float o = *p; *p = o + (float) v; return o; }
static inline unsigned long long atomicAdd(unsigned long long* p, long long v) { auto o = *p; *p = o + (unsigned long long) v; return o; }
template <typename T> static inline T atomicSub(T* p, T v) { T o = *p; *p = o - v; return o; }
template <typename T> static inline T atomicMax(T* p, T v) { T o = *p; *p = o > v ? o : v; return o; }
template <typename T> static inline T atomicMin(T* p, T v) { T o = *p; *p = o < v ? o : v; return o; }
template <typename T> static inline T atomicOr(T* p, T v) { T o = *p; *p = o | v; return o; }
template <typename T> static inline T atomicAnd(T* p, T v) { T o = *p; *p = o & v; return o; }
template <typename T> static inline T atomicExch(T* p, T v) { T o = *p; *p = v; return o; }
template <typename T> static inline T atomicCAS(T* p, T cmp, T v) { T o = *p; if (o == cmp) *p = v; return o; }
// global_atomic_pk_add_f16: two independent binary16 additions, each rounded once
typedef _Float16 wemu_half2 __attribute__((ext_vector_type(2)));
static inline wemu_half2 wemu_atomic_fadd_v2f16(void* p, wemu_half2 v) {
  wemu_half2* q = (wemu_half2*) p;
  const wemu_half2 o = *q;
  wemu_half2 n;
  n[0] = (_Float16) (o[0] + v[0]);
  n[1] = (_Float16) (o[1] + v[1]);
  *q = n;
  return o;
}
#define __builtin_amdgcn_global_atomic_fadd_v2f16(p, v) wemu_atomic_fadd_v2f16((void*) (p), (v))
static inline unsigned atomicInc(unsigned* p, unsigned lim) { unsigned o = *p; *p = o >= lim ? 0 : o + 1; return o; }

// ------------------------------------------------------------------------------------------------- cross-lane operations
// Every operation carries its call site as DATA: WEMU_HERE() is the return address of a marker call that is unique per expansion
// (and per inlined copy of a device helper).  The kernels are compiled at -O0 -- an optimiser that merges the operations of two
// branches into one call, or duplicates one operation into two paths, changes who votes with whom (the GPU target forbids both
// through `convergent`; tests/wave_emul/selftest.hip holds the emulation to the answers the hardware gives) -- and the operations
// themselves live in wemu_rt.cpp, compiled with optimisation.
#define WEMU_HERE() ([]() __attribute__((noinline)) -> const void* { return __builtin_return_address(0); }())
namespace wemu {
enum { SHFL_IDX, SHFL_UP, SHFL_DOWN, SHFL_XOR };
uint32_t xl_shfl(const void* site, uint32_t v, int mode, int arg, int width);
int xl_dpp(const void* site, int old, int src, int ctrl, int row_mask, int bank_mask, bool bound_ctrl);
unsigned long long xl_ballot(const void* site, int pred);
int xl_first(const void* site, int v);
int xl_barrier(int kind, int pred);
void xl_mfma(const void* site, int kind, const void* a, const void* b, const void* c, void* d);
// any trivially copyable value of 4 or 8 bytes, 32 bits at a time (as HIP's own __shfl overloads do)
template <typename T>
static inline T shfl_any(const void* site, T v, int mode, int arg, int width) {
  static_assert(sizeof(T) == 4 || sizeof(T) == 8, "shuffle of a 4- or 8-byte value");
  uint32_t w[sizeof(T) / 4];
  memcpy(w, &v, sizeof(T));
  for (unsigned i = 0; i < sizeof(T) / 4; i++) w[i] = xl_shfl((const char*) site + i, w[i], mode, arg, width);
  T r;
  memcpy(&r, w, sizeof(T));
  return r;
}
}  // namespace wemu
template <typename T> static inline T wemu_shfl(const void* site, T v, int src_lane, int width = 64) { return wemu::shfl_any(site, v, wemu::SHFL_IDX, src_lane, width); }
template <typename T> static inline T wemu_shfl_up(const void* site, T v, unsigned delta, int width = 64) { return wemu::shfl_any(site, v, wemu::SHFL_UP, (int) delta, width); }
template <typename T> static inline T wemu_shfl_down(const void* site, T v, unsigned delta, int width = 64) { return wemu::shfl_any(site, v, wemu::SHFL_DOWN, (int) delta, width); }
template <typename T> static inline T wemu_shfl_xor(const void* site, T v, int mask, int width = 64) { return wemu::shfl_any(site, v, wemu::SHFL_XOR, mask, width); }
#define __shfl(...) wemu_shfl(WEMU_HERE(), __VA_ARGS__)
#define __shfl_up(...) wemu_shfl_up(WEMU_HERE(), __VA_ARGS__)
#define __shfl_down(...) wemu_shfl_down(WEMU_HERE(), __VA_ARGS__)
#define __shfl_xor(...) wemu_shfl_xor(WEMU_HERE(), __VA_ARGS__)
#define __ballot(pred) wemu::xl_ballot(WEMU_HERE(), (pred))
#define __any(pred) (wemu::xl_ballot(WEMU_HERE(), (pred)) != 0)
#define __all(pred) (wemu::xl_ballot(WEMU_HERE(), !(pred)) == 0)  // (all <=> nobody in the mask has !pred)
#define __builtin_amdgcn_readfirstlane(v) wemu::xl_first(WEMU_HERE(), (v))
#define __builtin_amdgcn_update_dpp(old, src, ctrl, rm, bm, bc) wemu::xl_dpp(WEMU_HERE(), (old), (src), (ctrl), (rm), (bm), (bc))
#define __builtin_amdgcn_mov_dpp(src, ctrl, rm, bm, bc) wemu::xl_dpp(WEMU_HERE(), 0, (src), (ctrl), (rm), (bm), (bc))
#define __syncthreads() ((void) wemu::xl_barrier(wemu::BARRIER, 0))
#define __syncthreads_or(pred) wemu::xl_barrier(wemu::BARRIER_OR, (pred))

// ------------------------------------------------------------------------------------------------- matrix cores
// v_mfma_f32_16x16x16_f16 / v_mfma_f32_16x16x32_f16 (CDNA3 / CDNA4 ISA): D[i][j] = C[i][j] + sum_k A[i][k] B[k][j], one wave.
// Lane l holds A[i = l % 16][k = K/4 * (l / 16) ..], B[k = K/4 * (l / 16) ..][j = l % 16], and C / D[i = 4 * (l / 16) .. + 3][j = l % 16].
// Products of two binary16 values are exact in fp32; the hardware's accumulation order inside one instruction is not documented,
// the emulation adds in k order in fp32 (the oracle's reading: oracle_set_mlp_accumulator's default) -- MLP outputs are held to the
// parity tolerance, not to bits.
typedef _Float16 wemu_half4 __attribute__((ext_vector_type(4)));
typedef _Float16 wemu_half8 __attribute__((ext_vector_type(8)));
typedef float wemu_float4 __attribute__((ext_vector_type(4)));
static inline wemu_float4 wemu_mfma_16x16x16(const void* site, wemu_half4 a, wemu_half4 b, wemu_float4 c) {
  wemu_float4 d;
  wemu::xl_mfma(site, wemu::MFMA_16x16x16_F16, &a, &b, &c, &d);
  return d;
}
static inline wemu_float4 wemu_mfma_16x16x32(const void* site, wemu_half8 a, wemu_half8 b, wemu_float4 c) {
  wemu_float4 d;
  wemu::xl_mfma(site, wemu::MFMA_16x16x32_F16, &a, &b, &c, &d);
  return d;
}
#define __builtin_amdgcn_mfma_f32_16x16x16f16(a, b, c, cbsz, abid, blgp) wemu_mfma_16x16x16(WEMU_HERE(), (a), (b), (c))
#define __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, cbsz, abid, blgp) wemu_mfma_16x16x32(WEMU_HERE(), (a), (b), (c))
