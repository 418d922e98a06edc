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

sys.path.insert(0, os.path.join(ROOT, "tests"))

import test_gpu_converged as gconv  # noqa: E402
import test_gpu_mlp_shapes as gmlp  # noqa: E402
import test_gpu_parity as gp  # noqa: E402
import test_gpu_scale as gscale  # noqa: E402

_GPU_MODULES = (gp, gscale, gconv, gmlp)


@pytest.fixture(scope="session")
def emul_lib():
    import wemu_build as wave_emul_build
    lib, _ = wave_emul_build.build()
    L = ctypes.CDLL(lib)
    L.f2n_build_info.restype = ctypes.c_char_p
    L.wemu_counter.restype = ctypes.c_long
    L.wemu_set_schedule(int(os.environ.get("WEMU_SCHEDULE", "0")))
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
    for mod in _GPU_MODULES:
        for name, value in (("DEV", "cpu"),
                            # (on the GPU T() and N() copy by crossing the bus; here they must copy explicitly, or a kernel would
                            # update the oracle's inputs in place)
                            ("T", lambda a: torch.from_numpy(np.array(a, copy=True, order="C"))),
                            ("N", lambda t: t.detach().numpy().copy())):
            if hasattr(mod, name):
                monkeypatch.setattr(mod, name, value)
    monkeypatch.setattr(gscale, "np", _NumpyWithASmallConvergedBatch())
    return capi


class _NumpyWithASmallConvergedBatch:
    """numpy, except that the converged octree's 13 056-ray training batch (tools/data/converged_sampler.npz: what the kernel-level
    tests of test_gpu_scale.py sample) comes back as every 97th ray: the same tree, the same kind of rays, 135 of them."""

    def __getattr__(self, name):
        return getattr(np, name)

    @staticmethod
    def load(path, *a, **k):
        z = np.load(path, *a, **k)
        if os.path.basename(str(path)) != "converged_sampler.npz":
            return z
        z = dict(z)
        n = z["rays_o"].shape[0]
        return {key: (v[::97].copy() if getattr(v, "shape", None) and v.shape[0] == n else v) for key, v in z.items()}


def _on_the_emulator(name, params=None, module=gp):
    """The GPU test `name` as a test of this module; params = (argnames, values) replaces its own parametrisation where the GPU
    sizes would take the emulator minutes."""
    fn = getattr(module, name)

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
    "test_sampler_vs_oracle": ("seed,fineness,scale_by_dis,max_hits,n", [(1, 8.0, True, 1024, 80), (2, 2.0, False, 1024, 24), (3, 16.0, True, 5, 200)]),
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
    "test_hash_backward_overflow_lists_keep_the_sums_order_free": ("log2", [20]),
    "test_mlp_forward": None,
    "test_mlp_backward": None,
    "test_partitioned_gather_equals_fused_forward": None,
    "test_binned_gather_equals_partitioned_gather": ("log2_t,n,p0,clump", [(21, 1537, 1, True), (20, 66000, 1, True), (20, 0, 0, False)]),
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
    "test_step_tail_equals_separate_launches": ("n_use,bad,side,dirty,leave,amp", [(33000, False, False, True, False, 1e-3), (900, True, False, False, False, 1e-3),
                                                                                   (33000, False, False, True, True, 1e-3), (33000, False, False, True, False, 1e-6)]),
    "test_img2world_rays_and_pixel_gather": None,
    "test_empty_and_ragged_inputs": None,
}
for _name, _params in _TESTS.items():
    globals()[_name] = _on_the_emulator(_name, _params)

# kernel-level tests of the other GPU modules (the ones that go through the C-ABI alone; the host classes need a device)
_MORE = [
    (gscale, "test_sampler_on_converged_octree", None),
    (gscale, "test_deep_tree_fills_the_dfs_stack", None),
    (gscale, "test_speculative_sampling_repair", None),
    (gscale, "test_speculative_tail_repair", None),
    (gscale, "test_speculative_walk_that_saw_a_later_tree", None),
    (gscale, "test_persistent_march", ("n_blocks,record,block_waves", [(64, True, 1), (64, True, 4), (30, False, 16)])),
    (gmlp, "test_general_mlp_forward_and_backward", ("d_in,d_hidden,n_hidden,n", [sh + (31,) for sh in gmlp.SHAPES] + [(17, 64, 2, 4099), (48, 16, 5, 4099)])),
    (gmlp, "test_unsupported_shapes_still_say_so", None),
    # the run-combining / cost-balanced gather (every training iteration from fineness ~4 down takes it)
    (gconv, "test_balanced_gather_edge_sizes", None),
    (gconv, "test_balanced_gather_unstaged_with_many_warps", ("fineness", [4.0])),
]
for _mod, _name, _params in _MORE:
    globals()[_name] = _on_the_emulator(_name, _params, _mod)


@pytest.fixture(scope="session")
def converged_march():
    """test_gpu_converged.py's fixture of the same name -- the converged octree's training batch marched at fineness 1 / 2 / 4 -- from
    the ORACLE's sampler instead of the device's (the device sampler equals it bit for bit on this very scene:
    test_gpu_scale.py::test_sampler_on_converged_octree; on the emulator: the tests above), and from every 8th ray: input for the gather
    tests, 6.9e4 samples at fineness 1."""
    z = dict(np.load(os.path.join(ROOT, "tools", "data", "converged_sampler.npz")))
    ro, rd = np.ascontiguousarray(z["rays_o"][::8]), gp.oc.normalize_dirs(np.ascontiguousarray(z["rays_d"][::8]))
    n = len(ro)
    hits = gp.oc.oct_intersect(z["search_order"], ro, rd, 0.01, 1e8, z["tree_nodes"], 1024)
    out = {}
    for fin in (1.0, 4.0):
        rng = np.random.default_rng(int(fin))
        noise = (((rng.random(1024 + n + 10, dtype=np.float32) - np.float32(.5)) + np.float32(1.)) * np.float32(fin)).astype(np.float32)
        m = gp.oc.ray_march(ro, rd, noise, 1. / 256., True, *hits, z["tree_nodes"], z["pers_trans"])
        out[fin] = (np.ascontiguousarray(m["pts"]), np.ascontiguousarray(m["anchors"]))
    return out


def test_the_emulated_library_is_the_products_source_text(emul_lib):
    """Every file the emulated library is compiled from is the tree's file up to the two GPU-only spellings wemu_build.py names."""
    import wemu_build as wave_emul_build
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


# ---------------------------------------------------------------------------------------------------------------------------------
# the emulation itself: known answers of every cross-lane operation, under full and partial EXEC masks (tests/wave_emul/selftest.hip)
# ---------------------------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="session")
def selftest_lib():
    import wemu_build as wave_emul_build
    return ctypes.CDLL(wave_emul_build.build_selftest())


def _host_run(lib):
    """known_answers.py's runner for a library whose launches take host pointers (the emulation)."""
    def run(name, *args):
        conv = [x.ctypes.data_as(ctypes.c_void_p) if isinstance(x, np.ndarray) else x for x in args]
        assert getattr(lib, name)(*conv) == 0, name
    return run


@pytest.mark.parametrize("case", ["permutes", "masks", "block", "mfma", "persistent"])
def test_emulated_wave_knows_the_isa(selftest_lib, case):
    """Every cross-lane operation under full and partial EXEC masks, divergent loop exits, LDS / barriers / the barrier's vote, the
    two MFMA tiles against exact integer products, and a persistent wave whose rows work for different lengths (lanes that finish
    a trip early wait at the end of the loop's body: a scheduler that went by addresses alone would let them run ahead to the
    loop's head and vote alone).  Expected values: tests/wave_emul/known_answers.py -- the same file
    tools/wave_selftest_on_gpu.py holds the HARDWARE to."""
    import known_answers
    getattr(known_answers, "check_" + case)(_host_run(selftest_lib))


# ---------------------------------------------------------------------------------------------------------------------------------
# the emulated suite has teeth: a one-token change in the COOPERATIVE part of a kernel (nothing a lane-local check could see) fails
# the parity test of that kernel
# ---------------------------------------------------------------------------------------------------------------------------------
_MUTANTS = [
    # the segmented row walks shift by two lanes instead of one (FlexOps, compositing)
    ("rows_dev.h", "__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x111, 0xF, 0xF, false)",
     "__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x112, 0xF, 0xF, false)", "test_segmented_ops_bit_exact", {}),
    # the octree walk parks the hits behind the first interior child one LDS stack slot too high
    ("sampler.hip", "const int pos = sp + 1 + __popc(rest >> (k + 1));", "const int pos = sp + 1 + __popc(rest >> k);", "test_sampler_golden", {}),
    # the MFMA of the fused MLPs takes its fragments the other way round
    ("mlp_dev.h", "return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);", "return __builtin_amdgcn_mfma_f32_16x16x32_f16(b, a, acc, 0, 0, 0);",
     "test_mlp_forward", {"n_hidden": 2, "n": 1000}),
]


@pytest.mark.parametrize("mutant", range(len(_MUTANTS)))
def test_a_broken_cooperative_step_fails_its_parity_test(hip, monkeypatch, fox_state, fox_golden, mutant):
    import wemu_build as wave_emul_build
    name, old, new, test, kw = _MUTANTS[mutant]
    lib, _ = wave_emul_build.build(mutate=[(name, old, new)], tag="mutant%d" % mutant)
    monkeypatch.setattr(hip, "_lib", ctypes.CDLL(lib))
    fn = getattr(gp, test)
    import inspect
    args = {"hip": hip, "fox_state": fox_state, "fox_golden": fox_golden}
    call = {k: (kw[k] if k in kw else args[k]) for k in inspect.signature(fn).parameters}
    with pytest.raises(AssertionError):
        fn(**call)


# ---------------------------------------------------------------------------------------------------------------------------------
# a race detector: the order in which the waves of a workgroup (and the workgroups of a launch) get to run is not the kernel's to
# rely on.  The whole module passes with WEMU_SCHEDULE=1 / 2 in the environment (recorded in DESIGN.md); here the multi-wave kernels
# with LDS traffic between their waves run under both other orders on every CPU run.
# ---------------------------------------------------------------------------------------------------------------------------------
_ORDER_FREE = [
    ("test_sampler_golden", {}), ("test_segment_scan", {}), ("test_early_stop_and_compaction", {}),
    ("test_early_stop_and_votes_in_one_launch", {"n_nodes": 897}), ("test_edge_samples_and_occupancy", {}),
    ("test_hash_backward", {}), ("test_field_fused_forward_backward", {"n_use": 1500}),
    ("test_shade_fused_forward_backward", {"n_emb": 240}), ("test_composite_train_equals_three_launches", {"gs": 0.3, "var_w": 0.0}),
    ("test_train_loss_and_gradients", {}), ("test_adam_fused_equals_separate_launches", {}),
]


@pytest.mark.parametrize("schedule", [1, 2] if os.environ.get("WEMU_FULL", "0") not in ("", "0") else [2])
def test_results_do_not_depend_on_the_order_waves_and_workgroups_run_in(hip, emul_lib, fox_state, fox_golden, schedule):
    import inspect
    args = {"hip": hip, "fox_state": fox_state, "fox_golden": fox_golden}
    emul_lib.wemu_set_schedule(schedule)
    try:
        for name, kw in _ORDER_FREE:
            fn = getattr(gp, name)
            fn(**{k: (kw[k] if k in kw else args[k]) for k in inspect.signature(fn).parameters})
    finally:
        emul_lib.wemu_set_schedule(int(os.environ.get("WEMU_SCHEDULE", "0")))


# ---------------------------------------------------------------------------------------------------------------------------------
# octree maintenance on the device: the GPU test drives it through the host class (PersOctree::ProcOctree,
# csrc/host/PersSampler.cpp); here the same call sequence is issued from Python against the emulated kernels of csrc/octree.hip
# ---------------------------------------------------------------------------------------------------------------------------------
def _proc_octree_on_the_emulator(L, nodes, w, a, visit, subdivide, brute):
    """PersOctree::ProcOctree(compact = true, subdivide, brute) of csrc/host/PersSampler.cpp, call for call."""
    octc = gscale.octc
    n = len(nodes)
    vp = lambda x: x.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
    i32 = lambda *shape: np.zeros(shape, np.int32)  # noqa: E731
    src = np.ascontiguousarray(nodes).view(np.uint8).reshape(-1).copy()
    work, edited = np.zeros_like(src), np.zeros_like(src)
    alive, n_child, keep, new_pos, total = i32(n), i32(n), i32(n), i32(n, 2), i32(1)
    assert L.f2n_oct_prune_compress(None, n, vp(src), vp(work), vp(edited), vp(alive), vp(n_child), vp(keep)) == 0
    assert L.f2n_segment_scan(None, n, vp(keep), vp(new_pos), vp(total)) == 0
    m = int(total[0])
    assert m >= 1
    k_nodes, k_w, k_a, k_v = np.zeros(m * 64, np.uint8), i32(m), i32(m), i32(m)
    assert L.f2n_oct_gather_kept(None, n, vp(edited), vp(keep), vp(new_pos), vp(w), vp(a), vp(visit), vp(k_nodes), vp(k_w), vp(k_a), vp(k_v)) == 0
    if not subdivide:
        return k_nodes.view(octc.NODE_DT), k_w, k_a
    depth, size, new_idx = i32(m), i32(m), i32(m)
    assert L.f2n_oct_subtree_sizes(None, m, vp(k_nodes), vp(k_v), int(brute), vp(depth), vp(size)) == 0
    m2 = int(size[0])
    d_nodes, d_w, d_a = np.zeros(m2 * 64, np.uint8), i32(m2), i32(m2)
    assert L.f2n_oct_subdivide(None, m, vp(k_nodes), vp(k_v), int(brute), vp(size), vp(k_w), vp(k_a), vp(new_idx), vp(d_nodes), vp(d_w), vp(d_a)) == 0
    return d_nodes.view(octc.NODE_DT), d_w, d_a


@pytest.mark.parametrize("scene", ["fox", "converged"])
def test_device_proc_octree_chain_on_the_emulator(emul_lib, fox_state, scene):
    """tests/test_gpu_scale.py::test_device_proc_octree_chain, kernel level: chains of prune / compress / subdivide rounds with
    random leaf deaths and visit counts; node arrays field by field and both statistics equal to the reference's sequential
    algorithm (oracle/_ref where it is built, else its pinned restatement)."""
    rng = np.random.default_rng(4)
    if scene == "converged":  # the 148 k-node tree of a finished training
        z = np.load(os.path.join(ROOT, "tools", "data", "converged_sampler.npz"))
        nodes = z["tree_nodes"].view(gscale.octc.NODE_DT).copy()
        rounds = [(False, False, 0.3), (True, False, 0.2), (False, False, 0.7)]
    else:                     # the 897-node construction tree
        nodes = fox_state["tree_nodes"].view(gscale.octc.NODE_DT).copy()
        rounds = [(True, False, 0.3), (False, False, 0.2), (True, True, 0.0), (False, False, 0.6), (True, False, 0.5), (False, False, 0.95)]
    sizes = [len(nodes)]
    for rnd, (sub, brute, kill) in enumerate(rounds):
        n = len(nodes)
        valid = np.nonzero(nodes["trans_idx"] >= 0)[0]
        nodes["trans_idx"][rng.choice(valid, int(len(valid) * kill), replace=False)] = -1
        visit = rng.integers(0, 10, n).astype(np.int32)
        w = rng.integers(-5, 2000, n).astype(np.int32)
        a = rng.integers(-5, 2000, n).astype(np.int32)
        want_nodes, want_w, want_a = gscale._ref_proc(nodes, w, a, visit, True, sub, brute)
        got, gw, ga = _proc_octree_on_the_emulator(emul_lib, nodes, w, a, visit, sub, brute)
        assert len(got) == len(want_nodes), (rnd, len(got), len(want_nodes))
        for f in ("center", "side_len", "parent", "childs", "is_leaf_node", "trans_idx"):
            assert (got[f] == want_nodes[f]).all(), (rnd, f)
        assert (gw == want_w).all() and (ga == want_a).all(), rnd
        nodes = got.copy()
        sizes.append(len(nodes))
    assert max(sizes) > sizes[0] and min(sizes) < sizes[0], sizes  # (the chain both grew and shrank the tree)


# ---------------------------------------------------------------------------------------------------------------------------------
# entry points the GPU suite reaches through the host classes only: direct calls on the emulated library
# ---------------------------------------------------------------------------------------------------------------------------------
def _vp(x):
    return x.ctypes.data_as(ctypes.c_void_p)


def test_keyed_draws_of_the_kernels_on_the_emulator(emul_lib, fox_state):
    """The march noise a prologue launch draws for itself and the three uniforms of every ray of a keyed batch draw are
    Philox4x32-10 of (key, sequence number, element) bit for bit (tests/test_gpu_determinism.py::test_in_kernel_draws_are_philox_of_
    their_key, and f2n_draw_ray_batch_keyed == f2n_draw_ray_batch on the uniforms the numpy restatement gives)."""
    from philox_ref import philox4x32_10
    F32 = np.float32
    L, st = emul_lib, fox_state
    key, seq, fin = 0x1234567890ABCDEF, (1 << 40) + 7, F32(2.5)
    n_rays, n_noise = 256, 1024 + 256 + 10
    rng = np.random.default_rng(1)
    dirs = rng.standard_normal((n_rays, 3)).astype(F32)
    out, zero, noise = np.zeros_like(dirs), np.ones(3, np.int32), np.zeros(n_noise, F32)
    assert L.f2n_sampler_prologue_keyed(None, n_rays, _vp(dirs), _vp(out), _vp(zero), 3, n_noise, ctypes.c_uint64(key), ctypes.c_uint64(seq),
                                        ctypes.c_float(fin), _vp(noise)) == 0
    i = np.arange(n_noise)
    ctr = np.stack([i >> 2, np.zeros_like(i), np.full_like(i, seq & 0xFFFFFFFF), np.full_like(i, seq >> 32)], 1).astype(np.uint32)
    x = philox4x32_10(ctr, (key & 0xFFFFFFFF, key >> 32))[i, i & 3]
    u = (x >> np.uint32(8)).astype(F32) * F32(1.0 / 16777216.0)
    want = ((u - F32(.5)) + F32(1.)) * fin
    assert (noise.view(np.uint32) == want.view(np.uint32)).all()
    assert (zero == 0).all() and (out.view(np.uint32) == gp.oc.normalize_dirs(dirs).view(np.uint32)).all()
    # a keyed ray batch: ray r draws Philox(counter = (r, 0, seq), key) and takes its first three words
    ts = np.ascontiguousarray(st["train_set"].astype(np.int32))
    C = len(st["poses"])
    H, W, R = 24, 32, 1000
    images = rng.random((C, H, W, 3)).astype(F32)
    poses, intri = np.ascontiguousarray(st["poses"][:, :3, :4].astype(F32)), np.ascontiguousarray(st["intri"].astype(F32))
    dist = (rng.standard_normal((C, 4)) * 0.01).astype(F32)
    bounds = np.ascontiguousarray(st["bounds"].astype(F32))
    r = np.arange(R)
    ctr = np.stack([r, np.zeros_like(r), np.full_like(r, seq & 0xFFFFFFFF), np.full_like(r, seq >> 32)], 1).astype(np.uint32)
    u01 = np.ascontiguousarray(((philox4x32_10(ctr, (key & 0xFFFFFFFF, key >> 32))[:, :3] >> np.uint32(8)).astype(F32) * F32(1.0 / 16777216.0)))

    def outputs():
        return [np.zeros(R, np.int32), np.zeros((R, 2), np.int32), np.zeros((R, 3), F32), np.zeros((R, 3), F32), np.zeros((R, 3), F32), np.zeros((R, 2), F32)]
    a, b = outputs(), outputs()
    common = (_vp(ts), len(ts), H, W, _vp(poses), _vp(intri), _vp(dist), _vp(images), _vp(bounds))
    assert L.f2n_draw_ray_batch_keyed(None, R, ctypes.c_uint64(key), ctypes.c_uint64(seq), *common, *[_vp(t) for t in a]) == 0
    assert L.f2n_draw_ray_batch(None, R, _vp(u01), *common, *[_vp(t) for t in b]) == 0
    for x_, y_ in zip(a, b):
        assert (x_.view(np.uint32) == y_.view(np.uint32)).all()
    assert len(np.unique(a[0])) > 10 and a[1][:, 0].max() < H and a[1][:, 1].max() < W


def test_stat_update_and_scan_in_one_launch_on_the_emulator(emul_lib, fox_state):
    """f2n_oct_update_stats_scan (round 5: one launch) == f2n_oct_update_stats_ex + f2n_segment_scan_ex: statistics, node records,
    child blocks, death stamps, scan, totals and the host mirror, on the fox octree with random votes."""
    L, st = emul_lib, fox_state
    rng = np.random.default_rng(8)
    n_nodes = st["tree_nodes"].size // 64
    n = 4097
    counts = rng.integers(0, 300, n).astype(np.int32)
    also = np.array([41, 42], np.int32)
    cb0 = np.zeros(n_nodes * 8 * 32, np.uint8)
    tn0 = st["tree_nodes"].copy()
    assert L.f2n_oct_build_child_blocks(None, n_nodes, _vp(tn0), _vp(cb0)) == 0

    def inputs():
        g = np.random.default_rng(9)
        return dict(wa=g.integers(-1, 3, n_nodes).astype(np.int32), aa=g.integers(-1, 3, n_nodes).astype(np.int32),
                    mk=g.integers(0, 2, n_nodes).astype(np.int32), ws=g.integers(-2, 5, n_nodes).astype(np.int32),
                    as_=g.integers(-2, 5, n_nodes).astype(np.int32), tn=tn0.copy(), cb=cb0.copy(), died=np.full(n_nodes, -1, np.int32),
                    de=np.zeros(1, np.int32), deh=np.zeros(1, np.int32), se=np.zeros((n, 2), np.int32), tot=np.zeros(1, np.int32),
                    mirror=np.full(3, -1, np.int32))
    a, b = inputs(), inputs()
    epoch = 5
    assert L.f2n_oct_update_stats_ex(None, n_nodes, _vp(a["wa"]), _vp(a["aa"]), _vp(a["mk"]), _vp(a["ws"]), _vp(a["as_"]), _vp(a["tn"]), _vp(a["cb"]), 1,
                                     _vp(a["died"]), epoch, _vp(a["de"]), _vp(a["deh"])) == 0
    assert L.f2n_segment_scan_ex(None, n, _vp(counts), _vp(a["se"]), _vp(a["tot"]), _vp(a["mirror"]), _vp(also), 2) == 0
    assert L.f2n_oct_update_stats_scan(None, n_nodes, _vp(b["wa"]), _vp(b["aa"]), _vp(b["mk"]), _vp(b["ws"]), _vp(b["as_"]), _vp(b["tn"]), _vp(b["cb"]), 1,
                                       _vp(b["died"]), epoch, _vp(b["de"]), _vp(b["deh"]), n, _vp(counts), _vp(b["se"]), _vp(b["tot"]),
                                       _vp(b["mirror"]), _vp(also), 2) == 0
    for k in a:
        assert (a[k] == b[k]).all(), k
    assert int(a["tot"][0]) == int(counts.sum()) and list(a["mirror"]) == [41, 42, int(counts.sum())]
    assert (a["died"] == epoch).any() and int(a["de"][0]) == epoch  # (leaves did die in this update)
    ew, ea, enodes = gp.oc.update_node_stats(inputs()["wa"], inputs()["aa"], inputs()["mk"], inputs()["ws"], inputs()["as_"], tn0)
    assert (a["ws"] == ew).all() and (a["as_"] == ea).all() and (a["tn"] == enodes).all()  # ... and the oracle agrees


def test_reference_numerics_build_on_the_emulator(hip, monkeypatch, fox_state, fox_golden):
    """The kernel-level half of tests/test_gpu_refnum.py: the same sources with -DF2N_REFERENCE_NUMERICS=1 (libf2n_hip_refnum.so's
    build) -- the MLP forward with a binary16 accumulator fragment is the oracle's accumulator reading 1, the hash backward adds
    f16-rounded addends into f16 running sums (Hash3DAnchored.cu:145-153)."""
    import wemu_build as wave_emul_build
    lib, _ = wave_emul_build.build(tag="refnum", defines=("F2N_REFERENCE_NUMERICS=1",))
    L = ctypes.CDLL(lib)
    L.f2n_build_info.restype = ctypes.c_char_p
    monkeypatch.setattr(hip, "_lib", L)
    assert L.f2n_numerics_mode() == 1 and b"REFERENCE-NUMERICS" in L.f2n_build_info()
    oc, op, T, N, F32 = gp.oc, gp.op, gp.T, gp.N, np.float32
    rng = np.random.default_rng(3)
    for n_hidden in (1, 2):
        params = (rng.standard_normal(oc.mlp_n_params(32, 64, n_hidden)) * 0.25).astype(F32)
        x = rng.standard_normal((1024, 32)).astype(F32)
        out = torch.zeros((1024, 16), dtype=torch.float16)
        hip.mlp_fwd(1024, 32, 64, n_hidden, T(oc.f2h(params).view(np.float16)), T(x), out)
        got = N(out).astype(F32)
        ref0 = oc.h2f(oc.mlp_fwd(params, x, 64, n_hidden))
        with oc.mlp_accumulator(1):
            ref1 = oc.h2f(oc.mlp_fwd(params, x, 64, n_hidden))
        e0, e1 = np.abs(got - ref0), np.abs(got - ref1)
        assert e1.max() <= 4 * 2.0 ** -11 * np.abs(ref1).max() and e1.mean() < 0.25 * e0.mean(), (e1.mean(), e0.mean())
        assert (got == ref1).mean() > 0.97, (got == ref1).mean()
    st, g, log2 = fox_state, fox_golden, 14
    grid = op.HashGrid(np.zeros((16 << log2, 2), F32), st["prim_pool"], st["bias_pool"], int(st["n_volumes"]), log2)
    reps = 3
    pts = np.tile(g["march_pts"], (reps, 1))
    vol = np.tile(np.ascontiguousarray(g["march_anchors"][:, 0]), reps)
    n = len(pts)
    gin_h = oc.f2h((rng.standard_normal((n, 32)) * 0.05).astype(F32))
    gtab = torch.zeros(grid.table_f32.size, dtype=torch.float16)
    hip.hash_bwd(n, grid.n_volumes, T(grid.prim_pool), T(grid.local_idx), T(grid.local_size), T(grid.bias_pool), T(grid.scales), T(pts), True,
                 T(vol), 1, T(gin_h.view(np.float16)), gtab, 1 << log2)
    q01 = ((pts + F32(1.)) * F32(.5)).astype(F32)
    args = (grid.table_f32.size, grid.prim_pool, grid.local_idx, grid.local_size, grid.bias_pool, q01, vol, grid.n_volumes, gin_h, grid.scales)
    ref32 = oc.hash_bwd(*args, fp32_accumulate=True)
    ref16 = oc.h2f(oc.hash_bwd(*args, fp32_accumulate=False))
    got = N(gtab).astype(F32)
    d32, o16 = np.abs(got - ref32), np.abs(ref16 - ref32)
    assert ((got != 0) == (ref16 != 0)).mean() > 0.999
    # f16 running sums in SOME order: as far from the exact sums as the oracle's f16 accumulation in ITS order, not closer
    assert 0.3 * o16.mean() <= d32.mean() <= 3.0 * o16.mean(), (d32.mean(), o16.mean())
    assert d32.max() <= 0.05 * np.abs(ref32).max()


# ---------------------------------------------------------------------------------------------------------------------------------
# the drop-in boundary as a whole: ExpRunner::Train's iteration composed from the C-ABI's seam entry points, one call where the
# reference has one, on the emulated kernels -- against the oracle's iteration (tests/test_gpu_e2e.py's helper)
# ---------------------------------------------------------------------------------------------------------------------------------
def test_one_training_iteration_through_the_seams_on_the_emulator(hip, fox_state):
    """BASELINE config 1 (256 rays, 2^14 x 16 table, 2048 edge samples): sampler (count / scan / fill) -> density pre-pass -> early
    stop + FilterIdxBounds + compaction -> edge samples -> field (hash grid + MLP) -> ScatterIdx + SH + colour MLP -> compositing (+
    WeightVar) -> loss -> compositing backward -> colour backward -> field backward (MLP + hash scatter) -- what a maintainer who binds
    the reference's seams to include/f2n_abi.h one by one would run.  Bars: tests/test_gpu_e2e.py::test_config1_end_to_end_parity's."""
    import test_gpu_e2e as e2e
    from f2_nerf_amd import config
    st, oc, op, T, N, F32 = fox_state, gp.oc, gp.op, gp.T, gp.N, np.float32
    cfg = config.preset("wanjinyou", ["field.log2_table_size=14"])
    rng = np.random.default_rng(42)
    R, NE, log2, it = 256, e2e.N_EDGE, 14, 1
    n_img = len(st["poses"])
    table = (rng.standard_normal((16 << log2, 2)) * 0.3).astype(F32)
    p_field, p_color = gp.rand_params(rng, 1), gp.rand_params(rng, 2)
    app_emb = (rng.standard_normal((n_img, 16)) * 0.1).astype(F32)
    arrays = (st["tree_nodes"], st["pers_trans"], None, None, table, st["prim_pool"], st["bias_pool"], np.array([int(st["n_volumes"])]), p_field, p_color, app_emb)
    ro, rd, bounds, cam = e2e.fox_batch(st, rng, R)
    gt = rng.random((R, 3), dtype=F32)
    fineness = 16.0
    noise = (((rng.random(1024 + R + 10, dtype=F32) - F32(.5)) + F32(1.)) * F32(fineness)).astype(F32)
    bg = rng.random((R, 3), dtype=F32)
    eidx = rng.integers(0, st["edge_pool"].size // 64, NE).astype(np.int32)
    ecoord = (rng.random((NE, 2), dtype=F32) * F32(2.) - F32(1.)).astype(F32)
    ref = e2e.oracle_train_iteration(st, cfg, arrays, ro, rd, cam, gt, noise, bg, eidx, ecoord, iter_step=it)

    z = lambda *shape, dtype=torch.float32: torch.zeros(shape, dtype=dtype)  # noqa: E731
    i32 = torch.int32
    # ---- PersSampler::GetSamples ----
    rdn = z(R, 3)
    hip.normalize_dirs(R, T(rd), rdn)
    ps = cfg["pts_sampler"]
    hits, smp = gp.gpu_sample(hip, st, ro, N(rdn), noise, float(ps["sample_l"]), bool(ps["scale_by_dis"]), near=float(ps["near"]),
                              max_hits=int(ps["max_oct_intersect_per_ray"]))
    for k in ("pts_idx_bounds", "anchors", "t", "dt", "pts", "dirs"):
        a, b = smp[k], ref["smp"][k]
        assert a.shape == b.shape and (a.view(np.uint32) == b.view(np.uint32)).all(), k
    n = len(smp["t"])
    grid = op.HashGrid(table, st["prim_pool"], st["bias_pool"], int(st["n_volumes"]), log2)
    gd = gp.grid_dev(grid)
    ph_f, ph_c = T(oc.f2h(p_field).view(np.float16)), T(oc.f2h(p_color).view(np.float16))
    gargs = (grid.n_volumes, gd["table_h"], gd["prim"], gd["lidx"], gd["lsize"], gd["bias"], gd["scale"])
    # ---- density pre-pass, early stop (Renderer.cpp:101-135), FilterIdxBounds, compaction ----
    se, pts, dirs, dt, t, anc = T(smp["pts_idx_bounds"]), T(smp["pts"]), T(smp["dirs"]), T(smp["dt"]), T(smp["t"]), T(smp["anchors"])
    f0_all = z(n)
    hip.field_fwd(n, *gargs, pts, anc, 3, ph_f, None, f0_all, None)
    w_pre, a_pre, mask, kept = z(n), z(n), z(n, dtype=i32), z(R, dtype=i32)
    hip.early_stop(R, se, f0_all, 1, dt, w_pre, a_pre, mask, kept)
    new_se, tot = z(R, 2, dtype=i32), z(1, dtype=i32)
    hip.segment_scan(R, kept, new_se, tot)
    m = int(tot.item())
    assert abs(m - ref["n_kept"]) <= 2  # (early-stop threshold against 1-ulp expf differences)
    k_pts, k_dirs, k_dt, k_t, k_anc = z(m, 3), z(m, 3), z(m), z(m), z(m, 3, dtype=i32)
    hip.compact_samples(R, se, new_se, mask, pts, dirs, dt, t, anc, k_pts, k_dirs, k_dt, k_t, k_anc)
    same_kept = m == ref["n_kept"] and (N(new_se) == ref["new_se"]).all()
    # ---- edge samples, field over kept + edge points ----
    e_pts, e_idx = z(NE, 2, 3), z(NE, 2, dtype=i32)
    hip.edge_samples(NE, T(st["edge_pool"]), T(st["pers_trans"]), T(eidx), T(ecoord), e_pts, e_idx)
    q_pts = torch.cat([k_pts, e_pts.reshape(-1, 3)], 0).contiguous()
    q_vol = torch.cat([k_anc[:, 0], e_idx.reshape(-1)], 0).contiguous()
    nq = m + 2 * NE
    feat, f0, sx_f = z(nq, 16), z(nq), z(nq, 32, dtype=torch.float16)
    hip.field_fwd(nq, *gargs, q_pts, q_vol, 1, ph_f, feat, f0, sx_f)
    scene_feat, edge_feat = feat[:m].contiguous(), feat[m:].reshape(NE, 2, 16).contiguous()
    # ---- ScatterIdx + SHShader::Query ----
    sidx = z(m, dtype=i32)
    hip.scatter_idx(R, new_se, T(cam), sidx)
    rgb, sx_c = z(m, 3), z(m, 32, dtype=torch.float16)
    hip.shade_fwd(m, scene_feat, k_dirs, T(app_emb), sidx, ph_c, rgb, sx_c)
    # ---- compositing (+ WeightVar), loss ----
    col, disp, dep, wts, var = z(R, 3), z(R), z(R), z(m), z(R)
    hip.composite_fwd(R, new_se, scene_feat, k_dt, k_t, rgb, T(bg), col, disp, dep, wts, out_vars=var)
    colors = N(col)
    assert np.abs(colors - ref["colors"]).max() <= 1e-3, np.abs(colors - ref["colors"]).max()  # north-star RGB tolerance
    mse_g, mse_r = float(((colors - gt) ** 2).mean()), float(((ref["colors"] - gt) ** 2).mean())
    assert abs(10 * np.log10(1 / mse_g) - 10 * np.log10(1 / mse_r)) <= 1e-3  # PSNR within 1e-3 dB
    assert np.abs(N(disp) - ref["disparity"]).max() <= 1e-3 * max(1.0, np.abs(ref["disparity"]).max())
    if same_kept:
        assert np.abs(N(wts) - ref["weights"]).max() <= 1e-3
        assert np.abs(N(edge_feat) - ref["edge_feat"]).max() <= 4 * 2.0 ** -11 * max(1.0, np.abs(ref["edge_feat"]).max())
    tc = cfg["train"]
    var_w = 0.0 if it <= tc["var_loss_start"] else tc["var_loss_weight"] * min(1.0, (it - tc["var_loss_start"]) / (tc["var_loss_end"] - tc["var_loss_start"]))
    gs0, gs1 = float(tc["gradient_scaling_start"]), float(tc["gradient_scaling_end"])
    gs = 1.0 if it >= gs1 else max(0.0, (it - gs0) / (gs1 - gs0 + 1e-9))
    losses, dc, dd, dv, de = z(8), z(R, 3), z(R), z(R), z(NE, 2, 16)
    hip.train_loss(R, col, T(gt), disp, var, NE, 16, edge_feat, float(var_w), float(tc["disp_loss_weight"]), float(tc["tv_loss_weight"]), losses, dc, dd, dv, de)
    assert abs(float(losses[0]) - ref["loss"]) <= 1e-3 * max(1.0, abs(ref["loss"]))
    # ---- backward: compositing, colour path, field ----
    drgb, df0 = z(m, 3), z(m)
    hip.composite_bwd(R, new_se, f0[:m].contiguous(), k_dt, k_t, rgb, T(bg), dc, dd, None, None, gs, drgb, df0, f0_stride=1, df0_stride=1,
                      var_weights=wts, dvars=dv)
    dfeat = z(nq, 16)
    dfeat_scene = z(m, 16)
    dp_c, demb = z(p_color.size), z(n_img, 16)
    hip.shade_bwd(m, drgb, sidx, ph_c, sx_c, 128.0, dfeat_scene, dp_c, demb, df0=df0)
    dfeat[:m] = dfeat_scene
    dfeat[m:] = de.reshape(-1, 16)
    dp_f, gtab = z(p_field.size), z(grid.table_f32.size, dtype=torch.float16)
    hip.field_bwd(nq, grid.n_volumes, gd["prim"], gd["lidx"], gd["lsize"], gd["bias"], gd["scale"], q_pts, q_vol, 1, ph_f, sx_f, dfeat, 128.0, dp_f, gtab,
                  1 << log2)
    if same_kept:  # (the gradients of another sample set are another sum)
        rg = ref["grads"]
        rel = e2e.rel_err
        assert rel(N(dp_c) / F32(128.), rg["color_mlp"]) <= 3e-2, rel(N(dp_c) / F32(128.), rg["color_mlp"])
        assert rel(N(dp_f) / F32(128.), rg["field_mlp"]) <= 3e-2, rel(N(dp_f) / F32(128.), rg["field_mlp"])
        assert rel(N(demb), rg["app_emb"]) <= 3e-2, rel(N(demb), rg["app_emb"])
        gt_tab, rt_tab = N(gtab).astype(F32).reshape(-1) / F32(128.), rg["feat_pool"].reshape(-1)
        cos = float((gt_tab * rt_tab).sum() / (np.linalg.norm(gt_tab) * np.linalg.norm(rt_tab)))
        assert cos > 0.999 and rel(gt_tab, rt_tab) <= 5e-2, (cos, rel(gt_tab, rt_tab))
    assert same_kept  # (this seed's batch has no sample at the early-stop threshold: everything above was compared)


def test_the_emulation_never_lost_track_of_a_fibre(emul_lib):
    """(the last test of this module) no kernel nested activations or loops deeper than a fibre's record holds: whenever the lanes
    of a wave had parted, who runs next was decided from their places in the program."""
    assert emul_lib.wemu_counter(0) > 0 and emul_lib.wemu_counter(5) == 0


# ---------------------------------------------------------------------------------------------------------------------------------
# the HOST layer on the emulated wavefront: csrc/host/*.cpp -- ExpRunner, Renderer, PersSampler, Hash3DAnchored, SHShader, the
# octree builder, the dataset -- compiled from its own text against CPU stand-ins for the GPU-runtime names it uses
# (tests/wave_emul/host_shim/host_shim.h) and linked against libf2n_emul.so.  The GPU suite's host-level tests then run here:
# which kernels the host calls with what, in which order, what it prefetches, repairs, drops and resolves when, is its own code.
# ---------------------------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="session")
def emul_host(emul_lib):
    import importlib.util
    import wemu_build as wave_emul_build
    path = wave_emul_build.build_host()
    ctypes.CDLL(wave_emul_build.LIB, mode=ctypes.RTLD_GLOBAL)
    spec = importlib.util.spec_from_file_location("_f2n_host_emul", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.fixture
def rt(emul_host, hip, monkeypatch):
    """The `rt` fixture of the GPU modules (f2_nerf_amd.runtime) with the emulated host module behind runtime.host() and the CPU
    as "the device" (the `hip` fixture has pointed the ctypes binding at the emulated kernel library already)."""
    import test_gpu_e2e as e2e
    from f2_nerf_amd import runtime
    monkeypatch.setattr(runtime, "_host", emul_host)
    monkeypatch.setattr(torch.cuda, "is_available", lambda: True)
    monkeypatch.setattr(torch.cuda, "set_device", lambda *a, **k: None)
    monkeypatch.setattr(torch.Tensor, "cuda", lambda self, *a, **k: self)
    monkeypatch.setattr(torch.cuda, "current_stream", lambda *a, **k: type("TheOneStream", (), {"cuda_stream": 0})())
    for name in ("rand", "randn", "zeros", "ones", "empty", "full", "tensor"):  # (a literal device="cuda" in a test body means "the device")
        def factory(*a, _orig=getattr(torch, name), **k):
            if str(k.get("device", "")).startswith("cuda"):
                k["device"] = "cpu"
            return _orig(*a, **k)
        monkeypatch.setattr(torch, name, factory)
    to_dev = runtime.to_dev
    monkeypatch.setattr(runtime, "to_dev", lambda *arrays, device="cpu": to_dev(*arrays, device="cpu"))
    for mod in (e2e,) + _GPU_MODULES:
        if hasattr(mod, "DEV"):
            monkeypatch.setattr(mod, "DEV", "cpu")
    return runtime


# WEMU_FULL=1 in the environment: the host-level tests that take the emulator a minute or more each as well (recorded once per
# round: profiles/r05_emulated_host_suite.txt); the default run keeps the CPU suite at a few minutes
_FULL = os.environ.get("WEMU_FULL", "0") not in ("", "0")
_HOST_TESTS = [
    ("test_gpu_e2e", "test_config1_end_to_end_parity", None),
    ("test_gpu_e2e", "test_proc_octree_matches_restatement", None),
] + ([
    ("test_gpu_e2e", "test_streaming_step_equals_synchronous_step", None),
    ("test_gpu_e2e", "test_prefetched_sampling_is_not_used_by_a_render", None),
    ("test_gpu_e2e", "test_deferred_finiteness_flags_when_prefetching", None),
    ("test_gpu_e2e", "test_aux_states_and_deferred_reset", None),
    ("test_gpu_e2e", "test_prefetched_samples_do_not_survive_a_state_load", None),
    ("test_gpu_scale", "test_device_proc_octree_chain", ("scene", ["fox"])),
    # 36 iterations of the host on the emulated kernels against 36 iterations of the ORACLE (nodes, visit counts, sample counts exact
    # after every iteration): seven minutes here
    ("test_gpu_scale", "test_training_trajectory_across_milestone_and_compaction", ("realign", [True])),
] if _FULL else [])
for _modname, _name, _params in _HOST_TESTS:
    globals()[_name] = _on_the_emulator(_name, _params, __import__(_modname))


# ---------------------------------------------------------------------------------------------------------------------------------
# tests/test_gpu_determinism.py at the emulator's size: ExpRunner::Train on the fox photographs under several sampling schedules
# and chunkings of the loop -- every step's digest, every parameter checksum, the node array and the statistics identical.  (The
# GPU run trains 2400 iterations per schedule at the shipped batch size; here 128-ray batches, a 2^12 table, a compaction every 4
# iterations and a subdivision + MarkInvisibleNodes at iteration 6, so that batches begun ahead are dropped and begun again.)
# ---------------------------------------------------------------------------------------------------------------------------------
import test_gpu_determinism as gdet  # noqa: E402

_SMALL_RUN = ["train.end_iter=20000", "field.log2_table_size=12", "train.pts_batch_size=4096", "pts_sampler.compact_freq=4",
              "pts_sampler.sub_div_milestones=[6]"]


@pytest.fixture(scope="session")
def fox_scene_small():
    from f2_nerf_amd import fox_data
    st = fox_data.load_state()
    sc, images = fox_data.scene(8)
    return st, sc, images


def _train_small(rt, fox_scene, spec, depth, tail, chunk, iters):
    st, sc, images = fox_scene
    ds = rt.make_dataset(sc, images)
    runner, cfg, _ = rt.make_runner(st, "wanjinyou", _SMALL_RUN, seed=2022)
    runner.speculative_sampling, runner.speculation_depth, runner.tail_repair, runner.digest_table = spec, depth, tail, True
    torch.manual_seed(2022)
    it = 0
    while it < iters:
        it = min(iters, it + chunk)
        runner.train(ds, it, 1)
    s, c = runner.states(), runner.counters()
    return dict(table=gdet._csum(s[4]), field_mlp=gdet._csum(s[8]), color_mlp=gdet._csum(s[9]), app_emb=gdet._csum(s[-1]), nodes=s[0].numpy().copy(),
                visit=s[2].numpy().copy(), stats=[t.numpy().copy() for t in runner.occupancy_buffers()[:2]], marched=c["total_marched"],
                meaningful=c["total_meaningful"], iter_step=runner.iter_step, step_seq=runner.step_seq,
                digest=[tuple(int(v) for v in row) for row in runner.step_digest()], spec=dict(runner.speculation_counters()))


@pytest.fixture
def fox_scene(rt, fox_scene_small):
    """(tests/test_gpu_determinism.py's fixture: the fox scene with its photographs on "the device"; `rt` has put the emulated host
    behind runtime.host())"""
    return fox_scene_small


# the keyed draws of the host (Dataset::RandRaysData through f2n_draw_ray_batch_keyed; the per-rank streams of data-parallel replicas)
test_keyed_draws_do_not_depend_on_when_or_how_often_they_are_made = _on_the_emulator(
    "test_keyed_draws_do_not_depend_on_when_or_how_often_they_are_made", None, gdet)
test_replicas_of_a_data_parallel_run_draw_their_own_streams = _on_the_emulator(
    "test_replicas_of_a_data_parallel_run_draw_their_own_streams", None, gdet)

_SCHEDULES = {"after_the_update": (0, 1, True, 1000), "always_two_ahead_chunks_of_3": (1, 3, True, 3), "default": (2, 2, True, 1000),
              "always_one_ahead_full_repair": (1, 1, False, 1000)}


def _assert_same_training(ref, r, m):
    tail = ref["digest"][-min(len(ref["digest"]), len(r["digest"])):]
    tail_r = r["digest"][-len(tail):]
    first = next((i for i, (a, b) in enumerate(zip(tail, tail_r)) if a != b), None)
    assert first is None, "%s parts from sampling-after-the-update at step %s: %s vs %s" % (m, tail[first][0], tail[first], tail_r[first])
    for k in ("table", "field_mlp", "color_mlp", "app_emb", "marched", "meaningful", "iter_step", "step_seq"):
        assert r[k] == ref[k], (m, k, r[k], ref[k])
    assert r["nodes"].shape == ref["nodes"].shape and (r["nodes"] == ref["nodes"]).all(), m
    assert (r["visit"] == ref["visit"]).all() and all((a == b).all() for a, b in zip(r["stats"], ref["stats"])), m


@pytest.mark.parametrize("other", [k for k in _SCHEDULES if k != "after_the_update"][:None if _FULL else 1])
def test_training_is_independent_of_the_sampling_schedule_on_the_emulator(rt, fox_scene_small, other):
    iters = 9
    ref = _train_small(rt, fox_scene_small, *_SCHEDULES["after_the_update"], iters)
    assert ref["iter_step"] == iters and ref["spec"]["speculative"] == 0 and ref["nodes"].size // 64 != 897  # (the subdivision ran)
    r = _train_small(rt, fox_scene_small, *_SCHEDULES[other], iters)
    assert r["spec"]["speculative"] > 0, r["spec"]  # (it did sample ahead)
    _assert_same_training(ref, r, other)


@pytest.mark.parametrize("tail,depth", [(True, 3), (True, 2), (True, 1), (False, 1)] if _FULL else [(True, 3)])
def test_speculative_training_equals_sampling_after_the_update_on_the_emulator(rt, fox_state, tail, depth):
    """tests/test_gpu_scale.py::test_speculative_training_equals_sampling_after_the_update at the emulator's size (64 rays and 9 steps;
    192 and 12 with WEMU_FULL=1; no matrix products to keep a device busy -- there is no queue to fall behind here): ExpRunner::TrainStep with the next batches
    sampled AHEAD of the stat update that kills leaves under them (statistics re-armed at 0 before every step: every visited leaf
    without a positive vote dies at once), repaired by list compaction + tail march or by a second walk + march, against the same
    steps with the sampling behind the update -- per-step sample counts, node array and statistics identical; a compaction and a
    subdivision fall inside the run."""
    import test_gpu_e2e as e2e
    st, F32, N = fox_state, np.float32, (lambda t: t.detach().numpy())
    overrides = ["field.log2_table_size=12", "train.learning_rate=0.0", "pts_sampler.sub_div_milestones=[7]", "pts_sampler.compact_freq=5"]
    R, NE, ITERS = (192, 128, 12) if _FULL else (64, 64, 9)
    rng0 = np.random.default_rng(31)
    batches = []
    for _ in range(ITERS + 2):
        ro, rd, bounds, cam = e2e.fox_batch(st, rng0, R)
        batches.append([torch.from_numpy(np.ascontiguousarray(a)) for a in (ro, rd, bounds, rng0.random((R, 3), dtype=F32), cam)])
    logs = {}
    for spec in (True, False):
        runner, cfg, _ = rt.make_runner(st, "wanjinyou", overrides, seed=5, table_init=0.3)
        states = [t.clone() for t in runner.states()]
        states[8][-16 * 64:-15 * 64] *= 16.0  # (opaque and empty stretches instead of a uniform fog: see the GPU test)
        runner.load_states(states)
        runner.n_edge_pts = NE
        runner.speculative_sampling, runner.tail_repair, runner.speculation_depth, runner.march_blocks = spec, tail, depth, 96
        torch.manual_seed(11)
        log = []
        for it in range(ITERS):
            for t in runner.occupancy_buffers()[:2]:
                t.fill_(0)
            b, nb, nb2 = batches[it], batches[it + 1], batches[it + 2]
            if depth >= 2:
                s = runner.train_step(b[0], b[1], b[2], b[3], b[4], True, nb[0], nb[1], nb[2], nb2[0], nb2[1])
            else:
                s = runner.train_step(b[0], b[1], b[2], b[3], b[4], True, nb[0], nb[1], nb[2])
            runner.flush()
            w, a, v = [N(t).copy() for t in runner.occupancy_buffers()]
            log.append(dict(n_samples=s["n_samples"], kept=runner.counters()["total_meaningful"], nodes=N(runner.tree_nodes()).copy(), w=w, a=a, v=v,
                            loss=float(s["loss"])))
        logs[spec] = (log, dict(runner.speculation_counters()))
    (la, ca), (lb, cb) = logs[True], logs[False]
    assert ca["speculative"] >= ITERS - 6 and ca["rays_repaired"] > 0, ca  # most steps sampled ahead, and deaths did invalidate rays
    assert cb["speculative"] == 0 and cb["rays_repaired"] == 0, cb
    n_nodes = set()
    for it in range(ITERS):
        x, y = la[it], lb[it]
        assert x["n_samples"] == y["n_samples"] and x["kept"] == y["kept"], (it, x["n_samples"], y["n_samples"], x["kept"], y["kept"])
        assert x["nodes"].shape == y["nodes"].shape and (x["nodes"] == y["nodes"]).all(), it
        assert (x["w"] == y["w"]).all() and (x["a"] == y["a"]).all() and (x["v"] == y["v"]).all(), it
        assert abs(x["loss"] - y["loss"]) <= 1e-6 * max(1.0, abs(y["loss"])), (it, x["loss"], y["loss"])
        n_nodes.add(x["nodes"].size // 64)
    assert len(n_nodes) >= 3, n_nodes  # compaction and subdivision happened inside the run


# ---------------------------------------------------------------------------------------------------------------------------------
# what the driver runs at the end of a round, on the emulated stack: __graft_entry__.smoke() as it stands, and bench.py's main path
# (two warm-up and three timed steps of 64-ray batches on a 2^14 table; no converged leg, no CPU baseline, no other configs) -- the
# JSON line's fields and arithmetic, the two-deep pipeline's arguments, the kernel timers' bookkeeping.  Not a measurement of anything.
# ---------------------------------------------------------------------------------------------------------------------------------
def test_smoke_on_the_emulator(rt, monkeypatch, capsys):
    import __graft_entry__ as entry
    monkeypatch.setattr(torch.cuda, "get_device_name", lambda *a, **k: "emulated wavefront")
    entry.smoke()
    assert "smoke ok" in capsys.readouterr().out


def test_bench_main_path_on_the_emulator(rt, monkeypatch, capsys):
    import json
    import bench
    monkeypatch.setattr(torch.cuda, "max_memory_allocated", lambda *a, **k: 0)
    monkeypatch.setattr(torch.cuda, "max_memory_reserved", lambda *a, **k: 0)
    monkeypatch.setattr(torch.cuda, "empty_cache", lambda *a, **k: None)
    # (the dominant call -- the gather of the large-batch path, what `roofline` is about -- is issued from 32768 samples per step on:
    # ~400 rays of the fresh fox scene, half a minute per step here: only with WEMU_FULL=1)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "1", "--steps", "3", "--warmup", "2", "--rays", "400" if _FULL else "64", "--log2", "14", "--no-converged",
                                      "--no-cpu-baseline", "--no-steady", "--other-configs", "0"])
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    bench.main()
    line = json.loads([x for x in capsys.readouterr().out.splitlines() if x.startswith("{")][-1])
    assert line["metric"].startswith("training ray-samples/s") and line["unit"] == "ray-samples/s" and line["n_gpus"] == 1
    assert line["steps"] == 3 and line["warmup"] == 2 and line["higher_is_better"] is True and line["scaling"] == "weak" and line["dtype"] == "f16"
    assert line["vs_baseline"] is None and line["data"] == "synthetic" and "workload" in line["config"] and "model" not in line["config"]
    assert line["value"] > 0 and abs(line["value"] - line["config"]["meaningful_samples_per_step"] * 3 / (line["ms_per_step"] * 3e-3)) < 1e-6 * line["value"]
    rf = line["roofline"]
    assert (rf is not None) == _FULL
    if rf is not None:
        assert rf["bound"] == "hbm" and rf["peak"] == 8000.0 and rf["unit"] == "GB/s" and rf["launches"] == 3 and rf["bytes_per_sample"] == 592
        assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-4 and rf["traffic"] is None  # (no counter file for this table size)


# ---------------------------------------------------------------------------------------------------------------------------------
# SURVEY 8(e) without hardware: a data-parallel training of TWO ranks -- two processes, each the emulated host on the emulated
# kernels, attached through DataParallel::Attach (csrc/host/DataParallel.cpp as it stands) over a shared-memory <rccl/rccl.h> -- the
# program bench.py --gpus 2 runs: replicas built from different seeds, rank 0's state broadcast at the attach, per-rank ray batches,
# two-deep sampling, the pipelined gradient exchange (table buckets + one flat buffer), the MAX of the occupancy votes and the SUM of
# the survivor counts inside every step, compactions and a subdivision on every rank's own copy of the octree.
# ---------------------------------------------------------------------------------------------------------------------------------
def _run_ranks(world, steps, rays, overlap, uid_hex):
    import json
    import subprocess
    env = dict(os.environ, OMP_NUM_THREADS="1")
    worker = os.path.join(ROOT, "tests", "wave_emul", "dp_worker.py")
    procs = [subprocess.Popen([sys.executable, worker, str(r), str(world), uid_hex, str(steps), str(rays), str(overlap)], stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE, text=True, env=env) for r in range(world)]
    outs = []
    for p in procs:
        try:
            so, se = p.communicate(timeout=900)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise AssertionError("a rank hung: the ranks did not issue the same sequence of collectives")
        assert p.returncode == 0, se[-3000:]
        outs.append(json.loads([x for x in so.splitlines() if x.startswith("{")][-1]))
    return outs


@pytest.mark.parametrize("overlap", [1, 0] if _FULL else [1])
def test_two_rank_data_parallel_training_on_the_emulator(emul_host, overlap):
    uid = emul_host.dp_new_unique_id().hex()
    steps, rays = 7, 64
    a, b = _run_ranks(2, steps, rays, overlap, uid)
    assert a["rank"] == 0 and b["rank"] == 1 and a["comm_ranks"] == 2 and b["comm_ranks"] == 2
    assert a["table_before_attach"] != b["table_before_attach"]  # (they were built differently: the attach had something to do)
    for k in ("table", "field_mlp", "color_mlp", "app_emb", "nodes", "n_nodes", "meaningful_per_ray"):
        assert a[k] == b[k], (k, a[k], b[k])  # bit-identical replicas after the run: parameters, octree, the batch-sizing average
    assert a["n_nodes"] != 897  # (the compaction / subdivision ran on both)
    assert a["empty_steps"] == 0 and b["empty_steps"] == 1  # (one of rank 1's batches missed the scene: its collectives still paired with rank 0's)
    assert a["losses"] != b["losses"] and all(np.isfinite(a["losses"] + b["losses"]))  # (each rank trained on its own rays)
    # one rank alone, same seed and rays as rank 0: another training (the exchange did change what rank 0 learnt)
    solo = _run_ranks(1, steps, rays, -1, uid)[0]
    assert solo["table_before_attach"] == a["table_before_attach"] and solo["table"] != a["table"]


def test_bench_with_two_ranks_on_the_emulator(emul_host):
    """`bench.py --gpus 2` as the driver launches it -- one process per rank, RANK / WORLD_SIZE / MASTER_* in the environment -- on the
    emulated stack (tests/wave_emul/bench_dp_worker.py): bench.main() as it stands, f2_nerf_amd.parallel's native attach, the barrier
    bracket, the MAX of the ranks' times and the SUM of their samples, the replica checksums gathered on rank 0, ONE JSON line from
    rank 0 and none from rank 1."""
    import json
    import socket
    import subprocess
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    worker = os.path.join(ROOT, "tests", "wave_emul", "bench_dp_worker.py")
    args = ["--gpus", "2", "--steps", "3", "--warmup", "2", "--rays", "48", "--log2", "12", "--no-steady"]
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, worker] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env))
    outs = []
    for p in procs:
        try:
            so, se = p.communicate(timeout=900)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise AssertionError("a rank of bench.py --gpus 2 hung")
        assert p.returncode == 0, se[-3000:]
        outs.append([x for x in so.splitlines() if x.startswith("{")])
    assert len(outs[0]) == 1 and len(outs[1]) == 0  # rank 0 prints the line, rank 1 nothing
    line = json.loads(outs[0][0])
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["warmup"] == 2 and line["scaling"] == "weak" and line["config"]["parallelism"] == "ray-dp2"
    rep = line["replicas"]
    assert rep["identical"] is True and rep["rccl_comm_ranks"] == 2 and rep["torch_distributed_world"] == 2 and len(rep["checksums_table_fieldmlp_colormlp_nodes_nnodes"]) == 2
    assert line["value"] > 0 and line["cpu_baseline"] is None and line["converged"] is None and line["other_configs"] is None
    # the per-rank diagnostics of the exchange (round-5 verdict, next 6): both fields, one entry per rank, every timed step's exchange
    diag = line["data_parallel"]
    assert len(diag["dp_exchange_ms"]) == 2 and len(diag["dp_wait_ms"]) == 2 and all(v > 0 for v in diag["dp_exchange_ms"])
    assert all(n >= 3 for n in diag["exchanges_timed"])
    assert diag["small_buffers_exchanged_beside_the_scatter_rank0"] >= 3  # (round 6: the flat buffer's all-reduce leaves from the step's tail chain)
    # whole-job aggregate: the samples of BOTH ranks over the slower rank's time
    assert abs(line["value"] - line["config"]["meaningful_samples_per_step"] * 3 / (line["ms_per_step"] * 3e-3)) < 1e-6 * line["value"]
    assert line["config"]["rays_per_s"] > 0 and abs(line["config"]["rays_per_s"] - 48 * 2 * 3 / (line["ms_per_step"] * 3e-3)) < 1e-6 * line["config"]["rays_per_s"]


@pytest.mark.parametrize("world", [1, 2])
def _launcher_on_the_emulator(emul_host, tmp_path, world):
    """`python -m f2_nerf_amd.run --config-name=llff ... mode=train` on the emulated stack (tests/wave_emul/launcher_worker.py: run.main()
    as it stands) on a data directory in the reference's layout: the octree is built from the cameras on "the device", six iterations of
    ExpRunner::Train with a subdivision and a compaction, checkpoints in the reference's container, the test views' PSNR.  world = 2: the
    launcher's data-parallel training as torch.distributed.run would start it -- both ranks finish, rank 0 alone writes."""
    import socket
    import subprocess
    from PIL import Image
    from f2_nerf_amd import rigs
    rng = np.random.default_rng(2)
    meta, hw = rigs.forward_facing(rng, n_side=(4, 3), hw=(24, 32), focal=28.0)
    meta[:, 12:14] *= 4.0; meta[:, 14] *= 4.0; meta[:, 16:18] *= 4.0  # (intrinsics on disk refer to the factor-1 images, Dataset.cpp:49)
    data = tmp_path / "data" / "synth" / "rig"
    (data / "images_4").mkdir(parents=True)
    np.save(data / "cams_meta.npy", meta)
    for i in range(len(meta)):
        Image.fromarray(rng.integers(0, 255, (24, 32, 3), dtype=np.uint8)).save(data / "images_4" / ("%03d.png" % i))
    args = ["--config-name=llff", "dataset_name=synth", "case_name=rig", "exp_name=t", "+work_dir=%s" % tmp_path, "field.log2_table_size=12",
            "train.end_iter=6", "train.save_freq=3", "train.report_freq=2", "train.learning_rate_warm_up_end_iter=3", "pts_sampler.sub_div_milestones=[3]",
            "pts_sampler.compact_freq=4", "train.pts_batch_size=2048", "mode=train"]
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    procs = []
    for r in range(world):
        env = dict(os.environ, OMP_NUM_THREADS="1")
        if world > 1:
            env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "wave_emul", "launcher_worker.py")] + args, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True, env=env))
    outs = []
    for p in procs:
        so, se = p.communicate(timeout=1200)
        assert p.returncode == 0, se[-3000:]
        outs.append(so)
    exp = tmp_path / "exp" / "rig" / "t"
    assert (exp / "checkpoints" / "00000003" / "renderer.pt").exists() and (exp / "checkpoints" / "00000006" / "scalars.pt").exists()
    assert (exp / "train_info.txt").exists() and (exp / "test_images" / "info.yaml").exists()
    assert "Mean psnr" in outs[0] and all("Mean psnr" not in o and "Iter:" not in o for o in outs[1:])  # one writer


if _FULL:  # (a minute per case on the emulator: recorded in profiles/r05_emulated_host_suite.txt)
    test_launcher_on_the_emulator = _launcher_on_the_emulator


def test_the_host_never_waited_on_an_event_that_was_not_recorded(emul_host):
    """(after the host-level tests of this module) One ordering bug a stand-in without asynchrony can still see: a stream made to wait
    on an event nobody has recorded -- on the GPU that wait returns at once, an ordering the code believes it has and has not.  The
    stand-in for at::cuda::CUDAEvent counts them (tests/wave_emul/host_shim/host_shim.h)."""
    import wemu_build as wave_emul_build
    L = ctypes.CDLL(wave_emul_build.build_host())
    L.wemu_host_unrecorded_waits.restype = ctypes.c_long
    assert L.wemu_host_unrecorded_waits() == 0
