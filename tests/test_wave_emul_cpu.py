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
    monkeypatch.setattr(capi, "_mapped", lambda t: host_pointer(t, "i32", allow_none=True))  # (a "mapped" mirror is a host tensor)
    monkeypatch.setattr(torch.cuda, "synchronize", lambda *a, **k: None)
    monkeypatch.setattr(torch.Tensor, "pin_memory", lambda self, *a, **k: self)  # (mapped host memory is host memory)
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


# every test of tests/test_gpu_parity.py by name (but test_errors_are_loud, which is about the product binding refusing CPU tensors --
# the very thing the fixture above replaces --, test_sampler_full_size_properties -- 8192 rays twice: three minutes here -- and
# test_workspace_growth_with_queued_kernels -- about launches that are still queued, which the emulator's never are);
# None = with the GPU run's own parameters, otherwise the same test at sizes the emulator finishes in seconds (a cross-lane operation
# costs it ~3 us: the 2^20 / 2^22 tables and the 700- / 1500-ray batches stay with the GPU run)
_TESTS = {
    "test_sampler_golden": None,
    "test_sampler_vs_oracle": ("seed,fineness,scale_by_dis,max_hits,n", [(1, 8.0, True, 1024, 150), (2, 2.0, False, 1024, 60), (3, 16.0, True, 5, 513)]),
    "test_sampler_sample_cap": None,
    "test_normalize_dirs": None,
    "test_segment_scan": None,
    "test_segment_scan_and_flags_write_their_host_mirror": None,
    "test_edge_samples_and_occupancy": None,
    "test_early_stop_and_votes_in_one_launch": None,
    "test_edge_samples_from_uniforms_and_sampler_prologue": None,
    "test_hash_forward_bit_exact": None,
    "test_hash_forward_golden_and_linearity": None,
    "test_hash_backward": None,
    "test_hash_backward_owner_binned": ("log2", [14]),
    "test_mlp_forward": None,
    "test_mlp_backward": None,
    "test_partitioned_gather_equals_fused_forward": None,
    "test_binned_gather_equals_partitioned_gather": ("log2_t,n,p0,clump", [(21, 1, 0, False), (21, 1537, 1, True), (21, 0, 0, False), (20, 66000, 1, True)]),
    "test_field_fused_forward_backward": None,
    "test_field_forward_from_prepass_cache": None,
    "test_field_and_shade_forward_in_one_launch": None,
    "test_shade_fused_forward_backward": None,
    "test_sh_encode": None,
    "test_segmented_ops_bit_exact": None,
    "test_early_stop_and_compaction": None,
    "test_composite_forward_backward": None,
    "test_composite_train_equals_three_launches": None,
    "test_adam": None,
    "test_train_loss_and_gradients": None,
    "test_nonfinite_flags_and_skipped_adam": None,
    "test_adam_small_groups_equals_separate_launches": None,
    "test_adam_fused_equals_separate_launches": None,
    "test_img2world_rays_and_pixel_gather": None,
    "test_empty_and_ragged_inputs": None,
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
