"""TEST INFRASTRUCTURE: the product's LANE-LOCAL device functions, compiled for the host from their source text.

The kernels of f2-nerf_amd/csrc/*.hip cannot run without a GPU, but the arithmetic that decides a sample index, a hash cell or a
Philox draw lives in small `__device__` functions that use nothing of the GPU (no cross-lane builtin, no LDS): the slab test, the
perspective warp and its Jacobian, the hash cell, the SH basis, the undistortion, Philox4x32-10, the saturating float -> u32
conversion, the Adam update.  This module copies exactly those functions -- by name, from the files the library is built from, no
edited copy anywhere -- into one translation unit behind `__device__` / `__forceinline__` defined away, compiles it with g++ under
the product's floating-point flags (-ffp-contract=off -fno-fast-math) and exposes them through ctypes, so that
tests/test_lane_code_cpu.py can hold the PRODUCT's source against the oracle bit for bit on every CPU run.  x86 SSE and gfx950
both round +, -, *, /, sqrt and floor correctly in binary32, so equal operation orders mean equal bits.
"""
import ctypes
import os
import re
import subprocess
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "f2-nerf_amd", "csrc")

# (file, name): functions and structs taken verbatim, in dependency order
PIECES = [("f2n_dev.h", "struct F2nTransInfo"), ("f2n_dev.h", "f2n_philox4x32"), ("f2n_dev.h", "f2n_u01"), ("f2n_dev.h", "f2n_sum3"),
          ("f2n_dev.h", "f2n_sum4"), ("f2n_dev.h", "f2n_sum12"), ("f2n_dev.h", "f2n_norm3"), ("f2n_dev.h", "f2n_f2u_sat"),
          ("f2n_dev.h", "f2n_proj"), ("f2n_dev.h", "f2n_warp"), ("f2n_dev.h", "f2n_warp_jac"), ("sampler.hip", "f2n_slab"),
          ("field.hip", "struct F2nCell"), ("field.hip", "f2n_hash_cell"), ("shade.hip", "f2n_sh16"), ("shade.hip", "f2n_sh_high"),
          ("dataset.hip", "f2n_distort"), ("dataset.hip", "f2n_undistort"), ("adam_dev.h", "struct F2nAdamCoef"),
          ("adam_dev.h", "f2n_adam_update")]

PRELUDE = r"""
#include <cmath>
#include <cstdint>
#include <cstddef>
#define __device__
#define __forceinline__ inline
static inline uint32_t __umulhi(uint32_t a, uint32_t b) { return (uint32_t) (((uint64_t) a * (uint64_t) b) >> 32); }
"""

WRAPPERS = r"""
extern "C" {
void lane_philox(int n, const uint32_t* ctr, uint32_t k0, uint32_t k1, uint32_t* out) {
  for (int i = 0; i < n; i++) f2n_philox4x32(ctr[4 * i], ctr[4 * i + 1], ctr[4 * i + 2], ctr[4 * i + 3], k0, k1, out + 4 * i);
}
void lane_u01(int n, const uint32_t* x, float* out) { for (int i = 0; i < n; i++) out[i] = f2n_u01(x[i]); }
void lane_f2u_sat(int n, const float* f, uint32_t* out) { for (int i = 0; i < n; i++) out[i] = f2n_f2u_sat(f[i]); }
void lane_norm3(int n, const float* v, float* out) { for (int i = 0; i < n; i++) out[i] = f2n_norm3(v[3 * i], v[3 * i + 1], v[3 * i + 2]); }
void lane_slab(int n, const float* o, const float* d, const float* c, const float* side, float* near_far) {
  for (int i = 0; i < n; i++) f2n_slab(o + 3 * i, d + 3 * i, c + 3 * i, side[i], near_far[2 * i], near_far[2 * i + 1]);
}
void lane_warp(int n, const uint8_t* trans, const int32_t* trans_idx, const float* p, float* out, float* jac) {
  const F2nTransInfo* tr = (const F2nTransInfo*) trans;
  for (int i = 0; i < n; i++) {
    f2n_warp(tr + trans_idx[i], p + 3 * i, out + 3 * i);
    f2n_warp_jac(tr + trans_idx[i], p + 3 * i, (float (*)[3]) (jac + 9 * i));
  }
}
void lane_hash_cell(int n, const float* pt01, const float* mul, const int32_t* prim, const float* bias, const uint32_t* lsize,
                    uint32_t* pos, float* w, uint32_t* cell) {
  for (int i = 0; i < n; i++) {
    F2nCell c;
    f2n_hash_cell(pt01 + 3 * i, mul[i], prim + 3 * i, bias + 3 * i, lsize[i], c);
    for (int k = 0; k < 8; k++) { pos[8 * i + k] = c.pos[k]; w[8 * i + k] = c.w[k]; }
    for (int k = 0; k < 3; k++) cell[3 * i + k] = c.p[k];
  }
}
void lane_sh(int n, int degree, const float* dirs, float* out) {
  for (int i = 0; i < n; i++) {
    float sh[64];
    f2n_sh16(dirs[3 * i], dirs[3 * i + 1], dirs[3 * i + 2], sh);
    if (degree > 4) f2n_sh_high(degree, dirs[3 * i], dirs[3 * i + 1], dirs[3 * i + 2], sh);
    for (int k = 0; k < degree * degree; k++) out[(size_t) i * degree * degree + k] = sh[k];
  }
}
void lane_undistort(int n, const float* k4, float* uv) { for (int i = 0; i < n; i++) f2n_undistort(k4 + 4 * i, uv[2 * i], uv[2 * i + 1]); }
void lane_adam(int n, float* p, const float* g, float* m, float* v, const float* k9 /* f2n_adam_coefficients */) {
  F2nAdamCoef k;
  k.lr_over_bc1 = k9[0]; k.sqrt_bc2 = k9[1]; k.beta1 = k9[2]; k.beta2 = k9[3]; k.one_m_beta1 = k9[4];
  k.one_m_beta2 = k9[5]; k.eps = k9[6]; k.weight_decay = k9[7]; k.grad_scale = k9[8];
  for (int i = 0; i < n; i++) p[i] = f2n_adam_update(p[i], g[i], m[i], v[i], k);
}
}
"""


def _balanced(text, start):
    """end (exclusive) of the {...} block that opens at or after `start`"""
    i = text.index("{", start)
    depth = 0
    while True:
        ch = text[i]
        if ch == "{":
            depth += 1
        elif ch == "}":
            depth -= 1
            if depth == 0:
                return i + 1
        i += 1


def extract(fname, name):
    """The definition of `name` (a `struct X` or a __device__ function) exactly as it stands in csrc/<fname>."""
    text = open(os.path.join(CSRC, fname)).read()
    if name.startswith("struct "):
        m = re.search(r"^struct\s+(alignas\(\d+\)\s+)?%s\s*\{" % re.escape(name[7:]), text, re.M)
        assert m, (fname, name)
        end = _balanced(text, m.start())
        return text[m.start():text.index(";", end) + 1]
    m = re.search(r"^__device__ __forceinline__ [^;{]*?\b%s\s*\(" % re.escape(name), text, re.M)
    assert m, (fname, name)
    return text[m.start():_balanced(text, m.start())]


def source():
    return PRELUDE + "\n\n".join("// ---- csrc/%s: %s ----\n%s" % (f, n, extract(f, n)) for f, n in PIECES) + "\n" + WRAPPERS


_lib = None
_tmp = None  # (kept for the life of the process: the loaded library lives in it)


def lib():
    global _lib, _tmp
    if _lib is None:
        _tmp = tempfile.TemporaryDirectory(prefix="f2n_lane_")
        tmp = _tmp.name
        cpp, so = os.path.join(tmp, "lane_code.cpp"), os.path.join(tmp, "liblane_code.so")
        with open(cpp, "w") as f:
            f.write(source())
        # the product's floating-point contract (f2-nerf_amd/build.py: HIPCC_FLAGS); no -march: SSE2 scalar arithmetic, no FMA
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-shared", "-fPIC", "-Wall",
                               "-Wno-unknown-pragmas", "-Wno-unused-function", cpp, "-o", so])
        _lib = ctypes.CDLL(so)
    return _lib


def ptr(a):
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(ctypes.c_void_p)
