// Host-side common definitions (mirrors the role of src/Common.h of the reference).
#pragma once
#include <torch/torch.h>
#include <c10/hip/HIPStream.h>

#include <cmath>
#include <map>
#include <memory>
#include <sstream>
#include <string>
#include <vector>

#include "f2n_abi.h"

#define None torch::indexing::None
#define Slc torch::indexing::Slice

namespace f2n {

using Tensor = torch::Tensor;

inline torch::TensorOptions DevF32() { return torch::TensorOptions().dtype(torch::kFloat32).device(torch::kCUDA); }
inline torch::TensorOptions DevF16() { return torch::TensorOptions().dtype(torch::kFloat16).device(torch::kCUDA); }
inline torch::TensorOptions DevI32() { return torch::TensorOptions().dtype(torch::kInt32).device(torch::kCUDA); }
inline torch::TensorOptions DevU8() { return torch::TensorOptions().dtype(torch::kUInt8).device(torch::kCUDA); }
inline torch::TensorOptions CpuF32() { return torch::TensorOptions().dtype(torch::kFloat32).device(torch::kCPU); }
inline torch::TensorOptions CpuI32() { return torch::TensorOptions().dtype(torch::kInt32).device(torch::kCPU); }
inline torch::TensorOptions CpuU8() { return torch::TensorOptions().dtype(torch::kUInt8).device(torch::kCPU); }

// The HIP stream torch is currently enqueuing on (also correct on the autograd engine thread).
inline void* CurStream() { return (void*) c10::hip::getCurrentHIPStream().stream(); }

// Every C-ABI call is checked; there is no fallback path: a missing device or a failed launch throws.
#define F2N_CALL(expr)                                                           \
  do {                                                                           \
    int f2n_rc_ = (expr);                                                        \
    TORCH_CHECK(f2n_rc_ == 0, #expr, " failed with status ", f2n_rc_);           \
  } while (0)

inline void CheckDev(const Tensor& t, c10::ScalarType dt, const char* what) {
  TORCH_CHECK(t.defined(), what, ": undefined tensor");
  TORCH_CHECK(t.is_cuda(), what, ": must live on the HIP device (there is no CPU path)");
  TORCH_CHECK(t.is_contiguous(), what, ": must be contiguous");
  TORCH_CHECK(t.scalar_type() == dt, what, ": wrong dtype");
}
inline float* F32P(const Tensor& t) { return t.data_ptr<float>(); }
inline int32_t* I32P(const Tensor& t) { return t.data_ptr<int32_t>(); }
inline void* VoidP(const Tensor& t) { return t.data_ptr(); }

}  // namespace f2n
