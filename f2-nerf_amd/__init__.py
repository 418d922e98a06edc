"""f2-nerf_amd -- MI355X-native (gfx950 / HIP) implementation of F2-NeRF's per-ray hot path.

Layout
  csrc/*.hip        hand-written CDNA4 kernels + the C-ABI (include/f2n_abi.h) -> libf2n_hip.so
  csrc/host/*.cpp   C++/LibTorch host layer mirroring the reference's PtsSampler / Field / Shader / Renderer
                    plugin classes -> _f2n_host*.so (pybind11), used by bench.py and the end-to-end tests
  capi.py           ctypes binding of the C-ABI for torch tensors (what the parity tests call)
  build.py          in-tree build of both

There is NO CPU fallback: every entry point requires the native library and a HIP device, and fails loudly
otherwise.  The CPU oracle lives in /oracle and is never imported from here.
"""
from . import build  # noqa: F401


def lib_path():
    return build.LIB
