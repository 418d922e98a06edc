"""The GPU parity tests' own bodies, run on the CPU against the product's kernel sources under an emulated wavefront.

tests/wave_emul/ compiles f2-nerf_amd/csrc/*.hip -- the text libf2n_hip.so is built from -- for x86-64 against a stand-in for
<hip/hip_runtime.h> that executes workgroups as 64-lane waves of fibres (cross-lane operations with the EXEC mask the hardware
would hold, LDS, barriers, MFMA tiles; tests/wave_emul/include/hip/hip_runtime.h).  The result, libf2n_emul.so, exports the C-ABI
of include/f2n_abi.h with host pointers.  Each test below IS a function of tests/test_gpu_parity.py (same inputs, same oracle, same
asserts, same bit-exactness bars) called with the ctypes binding pointed at that library and its tensors on the CPU; what differs
from the `-m gpu` run is the instruction set the source was compiled for, not the source.  So the cooperative code of every kernel
-- DPP row chains, ballots, shuffles, the LDS work stacks of the octree walk, the owner-binned scatter, the MFMA tile chains -- is
held against the oracle on every CPU run.  What this cannot see: anything the gfx950 compiler or the hardware does differently
from the source's meaning (scheduling, memory ordering between workgroups, rounding of the hardware's exp / MFMA accumulation);
that is what the `-m gpu` run is for.

Test infrastructure only: the product binding (f2-nerf_amd/capi.py) refuses CPU tensors and has no CPU path; the patches below
live in this process's test fixtures."""
import ctypes
import functools
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "wave_emul"))

import tests.test_gpu_parity as gp  # noqa: E402


@pytest.fixture(scope="session")
def emul_lib():
    import build as wave_emul_build
    lib, _ = wave_emul_build.build()
    L = ctypes.CDLL(lib)
    L.f2n_build_info.restype = ctypes.c_char_p
    L.wemu_counter.restype = ctypes.c_long
    return L


@pytest.fixture
def hip(emul_lib, monkeypatch):
    """tests/test_gpu_parity.py's `hip` fixture, with the binding pointed at the emulated library for the length of one test."""
    import f2_nerf_amd  # noqa: F401
    from f2_nerf_amd import capi
    assert emul_lib.f2n_abi_version() == capi.ABI_VERSION

    def host_pointer(t, kind=None, allow_none=False):
        if t is None:
            if allow_none:
                return ctypes.c_void_p(0)
            raise capi.F2nError("required tensor is None")
        if t.device.type != "cpu":
            raise capi.F2nError("the emulated library takes host pointers")
        if not t.is_contiguous():
            raise capi.F2nError("tensor must be contiguous")
        if kind is not None and t.dtype != capi._DT[kind]:
            raise capi.F2nError("expected dtype %s, got %s" % (kind, t.dtype))
        return ctypes.c_void_p(t.data_ptr())

    monkeypatch.setattr(capi, "_lib", emul_lib)
    monkeypatch.setattr(capi, "_p", host_pointer)
    monkeypatch.setattr(capi, "_stream", lambda: ctypes.c_void_p(0))
    monkeypatch.setattr(torch.cuda, "synchronize", lambda *a, **k: None)
    monkeypatch.setattr(gp, "DEV", "cpu")
    # (on the GPU T() and N() copy by crossing the bus; here they must copy explicitly, or a kernel would update the oracle's inputs)
    monkeypatch.setattr(gp, "T", lambda a: torch.from_numpy(np.array(a, copy=True, order="C")))
    monkeypatch.setattr(gp, "N", lambda t: t.detach().numpy().copy())
    return capi


def _on_the_emulator(name, params=None):
    """The GPU test `name` as a test of this module; params = (argnames, values) replaces its own parametrisation where the GPU
    sizes would take the emulator minutes."""
    fn = getattr(gp, name)

    @functools.wraps(fn)
    def test(*a, **k):
        return fn(*a, **k)
    if params is not None:
        own = [m for m in getattr(fn, "pytestmark", []) if m.name == "parametrize"]
        replaced = {n.strip() for m in own for n in (m.args[0].split(",") if isinstance(m.args[0], str) else m.args[0])}
        assert {n.strip() for n in params[0].split(",")} == replaced, (name, replaced)
        test.pytestmark = [m for m in getattr(fn, "pytestmark", []) if m.name != "parametrize"] + [pytest.mark.parametrize(*params).mark]
    return test


# every test of tests/test_gpu_parity.py by name; None = with the GPU run's own parameters
_TESTS = {
    "test_sampler_golden": None,
    "test_normalize_dirs": None,
    "test_segment_scan": None,
    "test_segmented_ops_bit_exact": None,
    "test_early_stop_and_compaction": None,
    "test_adam": None,
}
for _name, _params in _TESTS.items():
    globals()[_name] = _on_the_emulator(_name, _params)


def test_the_emulated_library_is_the_products_source_text(emul_lib):
    """Every file the emulated library is compiled from is the tree's file up to the two GPU-only spellings build.py names."""
    import build as wave_emul_build
    for name in wave_emul_build.HEADERS + wave_emul_build.SOURCES:
        with open(os.path.join(wave_emul_build.CSRC, name)) as f:
            want, _ = wave_emul_build._rewrite(f.read())
        with open(os.path.join(wave_emul_build.OUT, "csrc", name)) as f:
            assert f.read() == want, name
        # the rewrites touch nothing but the lines they are written for
        with open(os.path.join(wave_emul_build.CSRC, name)) as f:
            src = f.read().splitlines()
        changed = [a for a, b in zip(src, want.splitlines()) if a != b]
        assert len(src) == len(want.splitlines()) and all("extern __shared__" in a or 'asm volatile("" : "+v"' in a for a in changed), name
    assert emul_lib.f2n_build_info() is not None
