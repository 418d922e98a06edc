"""CPU: the PRODUCT's lane-local device functions (copied by name out of f2-nerf_amd/csrc/*.hip / f2n_dev.h and compiled for the
host: tests/lane_code.py) against the oracle, bit for bit -- the arithmetic that decides a leaf hit, a sample position, a hash
cell or a random draw, checked on every CPU run from the very text the gfx950 library is built from.  The kernels around these
functions (lanes cooperating through DPP / ballots / LDS) are what the -m gpu tests cover."""
import ctypes
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import lane_code as lc  # noqa: E402
from oracle import capi as oc  # noqa: E402
from oracle import pipeline as op  # noqa: E402

F32 = np.float32


def bits(a):
    return np.ascontiguousarray(a, F32).view(np.uint32)


def test_extraction_takes_the_functions_verbatim():
    """The harness compiles the text that is in the source files -- no edited copy: every extracted piece is a substring of its file."""
    for fname, name in lc.PIECES:
        piece = lc.extract(fname, name)
        assert piece in open(os.path.join(lc.CSRC, fname)).read() and len(piece) > 40, (fname, name)
    src = lc.source()
    assert "__builtin_amdgcn" not in src and "__shfl" not in src and "threadIdx" not in src  # lane-local: nothing of the GPU in them


def test_philox_of_the_kernels_matches_the_known_answers_and_the_numpy_checker():
    from philox_ref import philox4x32_10
    lib = lc.lib()
    kats = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
            ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
            ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0), (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for c, k, want in kats:  # Random123 kat_vectors, Philox4x32-10
        ctr = np.array([c], np.uint32)
        out = np.empty((1, 4), np.uint32)
        lib.lane_philox(1, lc.ptr(ctr), ctypes.c_uint32(k[0]), ctypes.c_uint32(k[1]), lc.ptr(out))
        assert tuple(int(v) for v in out[0]) == want
    rng = np.random.default_rng(5)
    ctr = rng.integers(0, 2 ** 32, (4096, 4), dtype=np.uint64).astype(np.uint32)
    key = (0x9E3779B9 ^ 2022, 0x7F4A7C15)
    out = np.empty_like(ctr)
    lib.lane_philox(len(ctr), lc.ptr(ctr), ctypes.c_uint32(key[0]), ctypes.c_uint32(key[1]), lc.ptr(out))
    assert (out == philox4x32_10(ctr, key)).all()
    # 24-bit uniforms in [0, 1): what the keyed draws of a batch's rays and march noise are made of
    x = np.array([0, 1, 255, 256, 0xffffffff, 0x80000000, 0x12345678], np.uint32)
    u = np.empty(len(x), F32)
    lib.lane_u01(len(x), lc.ptr(x), lc.ptr(u))
    assert (u == (x >> 8).astype(F32) * F32(1.0 / 16777216.0)).all() and u.max() < 1.0 and u.min() == 0.0


def test_saturating_conversion_and_norm():
    lib = lc.lib()
    f = np.array([0.0, -0.0, -1.5, 0.99, 1.0, 3.7, 4294967040.0, 4294967296.0, 1e20, np.inf, -np.inf, np.nan, 2147483648.0], F32)
    out = np.empty(len(f), np.uint32)
    lib.lane_f2u_sat(len(f), lc.ptr(f), lc.ptr(out))
    want = [0, 0, 0, 0, 1, 3, 4294967040, 0xffffffff, 0xffffffff, 0xffffffff, 0, 0, 2147483648]
    assert [int(v) for v in out] == want  # v_cvt_u32_f32 / cvt.rzi.u32.f32: truncating, saturating, NaN -> 0
    rng = np.random.default_rng(1)
    v = (rng.standard_normal((5000, 3)) * 10).astype(F32)
    n = np.empty(len(v), F32)
    lib.lane_norm3(len(v), lc.ptr(v), lc.ptr(n))
    x2, y2, z2 = (v[:, 0] * v[:, 0]).astype(F32), (v[:, 1] * v[:, 1]).astype(F32), (v[:, 2] * v[:, 2]).astype(F32)
    assert (bits(n) == bits(np.sqrt((x2 + (y2 + z2).astype(F32)).astype(F32)))).all()  # Eigen's three-term order a + (b + c)


def test_slab_test_of_the_octree_walk(fox_state):
    """f2n_slab (straight-line, both quotients formed and selected) == the reference's three-way branch per axis, also where a
    direction component is inside the parallel-axis guard, zero, or the origin sits on a face."""
    rng = np.random.default_rng(2)
    n = 20000
    o = (rng.standard_normal((n, 3)) * 2).astype(F32)
    d = rng.standard_normal((n, 3)).astype(F32)
    d[::7, 0] = F32(3e-7); d[::11, 1] = F32(-5e-7); d[::13, 2] = 0.0; d[::17] = [0.0, 1.0, 0.0]; d[5::29, 0] = F32(1e-6); d[6::29, 1] = F32(-1e-6)
    c = (rng.standard_normal((n, 3))).astype(F32)
    side = (2.0 ** rng.integers(-6, 3, n)).astype(F32)
    o[::19] = c[::19] + side[::19, None] * F32(.5)  # on a corner of the box
    o[3::23] = c[3::23]                             # at its centre
    got = np.ascontiguousarray(np.stack([np.full(n, 0.05, F32), np.full(n, 1e8, F32)], -1))
    lc.lib().lane_slab(n, lc.ptr(o), lc.ptr(d), lc.ptr(c), lc.ptr(side), lc.ptr(got))
    want = oc.slab(o, d, c, side, 0.05, 1e8)
    assert (bits(got) == bits(want)).all()
    assert (got[:, 0] < got[:, 1]).sum() > n // 50  # (there are hits among them)


def test_perspective_warp_and_jacobian_on_the_fox_warps(fox_state):
    """f2n_warp / f2n_warp_jac on every warp of the fox octree (12 projections, Eigen's 12-term reduction tree) == the oracle."""
    trans = np.ascontiguousarray(fox_state["pers_trans"]).view(np.uint8).reshape(-1, 544)
    n_tr = len(trans)
    rng = np.random.default_rng(3)
    n = 30000
    idx = rng.integers(0, n_tr, n).astype(np.int32)
    centers = trans.view(F32).reshape(n_tr, 136)[:, 132:135]
    p = (centers[idx] + rng.standard_normal((n, 3)).astype(F32) * F32(0.05)).astype(F32)
    out = np.empty((n, 3), F32)
    jac = np.empty((n, 3, 3), F32)
    lc.lib().lane_warp(n, lc.ptr(trans), lc.ptr(idx), lc.ptr(p), lc.ptr(out), lc.ptr(jac))
    want, want_j = oc.warp(trans, idx, p)
    assert (bits(out) == bits(want)).all() and (bits(jac) == bits(want_j)).all()
    assert np.isfinite(out).mean() > 0.99


@pytest.mark.parametrize("lsize", [1 << 19, 1 << 22, 1 << 14, 524287, 1000003])
def test_hash_cell_of_the_gather_and_the_scatter(lsize):
    """f2n_hash_cell factors the corner hashes (x * pa and (x + 1) * pa formed once, XOR per corner, a mask where the level size is a
    power of two) -- the same eight entries and trilinear weights as the reference's per-corner form, for in-range points, points
    below 0 (saturating conversion), huge coordinates, negative primes (int32 -> uint32) and level sizes that are not powers of two."""
    rng = np.random.default_rng(4 + lsize % 97)
    n = 20000
    pt = rng.random((n, 3), dtype=F32)
    pt[::9] -= F32(1.5); pt[::13] *= F32(1e7); pt[5::31] = 0.0; pt[6::31] = 1.0
    mul = np.exp2(rng.integers(3, 11, n)).astype(F32) * (1 + rng.random(n, dtype=F32) * F32(.3))
    prim = rng.integers(-2 ** 31, 2 ** 31, (n, 3), dtype=np.int64).astype(np.int32)
    prim[::5] = [1, 19349663, 83492791]
    bias = (rng.random((n, 3), dtype=F32) * 100).astype(F32)
    ls = np.full(n, lsize, np.uint32)
    pos, w, cell = np.empty((n, 8), np.uint32), np.empty((n, 8), F32), np.empty((n, 3), np.uint32)
    lc.lib().lane_hash_cell(n, lc.ptr(pt), lc.ptr(mul), lc.ptr(prim), lc.ptr(bias), lc.ptr(ls), lc.ptr(pos), lc.ptr(w), lc.ptr(cell))
    want_pos, want_w = oc.hash_cell(pt, mul, prim, bias, ls)
    assert (pos == want_pos).all() and (bits(w) == bits(want_w)).all() and (pos < lsize).all()


@pytest.mark.parametrize("degree", [1, 2, 3, 4, 5, 6, 7, 8])
def test_sh_basis(degree):
    rng = np.random.default_rng(6)
    d = rng.standard_normal((4000, 3)).astype(F32)
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(F32)
    out = np.empty((len(d), degree * degree), F32)
    lc.lib().lane_sh(len(d), degree, lc.ptr(d), lc.ptr(out))
    assert (bits(out) == bits(oc.sh_encode(d, degree))).all()


def test_newton_undistortion_of_the_ray_generator():
    rng = np.random.default_rng(7)
    n = 5000
    k4 = (rng.standard_normal((n, 4)) * np.array([0.1, 0.02, 1e-3, 1e-3])).astype(F32)
    k4[::10] = 0
    uv = (rng.standard_normal((n, 2)) * 0.6).astype(F32)
    got = uv.copy()
    lc.lib().lane_undistort(n, lc.ptr(k4), lc.ptr(got))
    want = oc.undistort(k4, uv)
    assert (bits(got) == bits(want)).all()
    assert (got[::10] == uv[::10]).all()  # no distortion: the Newton iteration leaves the pixel where it is


def test_adam_scalars_are_the_ones_libtorch_forms_and_the_update_equals_the_oracle():
    """f2n_adam_coefficients (what every Adam kernel of the library is launched with; ABI v12: betas are doubles as in
    torch::optim::AdamOptions) against the scalars of torch's Adam::step -- 1 - beta and 1 - beta^step formed in double, narrowed to
    float where they meet the float tensors -- and the kernels' update function with those scalars against the oracle's Adam."""
    import f2_nerf_amd  # noqa: F401
    from f2_nerf_amd import build, capi
    build.build_hip()  # (a no-op when the library is up to date; the entry point is a host function: no GPU)
    rng = np.random.default_rng(8)
    n = 50000
    for step, lr, wd in ((1, 1e-2, 0.0), (2, 3.3e-3, 1e-6), (10, 1e-2, 1e-6), (137, 7.7e-4, 0.0), (20000, 1e-3, 1e-6)):
        lr32 = float(F32(lr))  # (ExpRunner hands the schedule's float on: ExpRunner.cpp:241-244 of the reference)
        k = capi.adam_coefficients(step, lr32, 0.9, 0.99, 1e-15, wd, 1.0 / 128)
        bc1, bc2 = 1.0 - 0.9 ** step, 1.0 - 0.99 ** step
        want = [F32(lr32 / bc1), F32(np.sqrt(bc2)), F32(0.9), F32(0.99), F32(1.0 - 0.9), F32(1.0 - 0.99), F32(1e-15), F32(wd), F32(1.0 / 128)]
        assert [F32(v) for v in k] == want, (step, k, want)
        assert F32(k[4]) == F32(0.1) and F32(k[5]) == F32(0.01)  # (a float beta gave 0.100000024 and 0.00999999046)
        p = rng.standard_normal(n).astype(F32); g = (rng.standard_normal(n) * 1e-3).astype(F32); g[::3] = 0
        m = (rng.standard_normal(n) * 1e-3).astype(F32); v = (rng.random(n) * 1e-6).astype(F32)
        gp, gm, gv = p.copy(), m.copy(), v.copy()
        lc.lib().lane_adam(n, lc.ptr(gp), lc.ptr(g), lc.ptr(gm), lc.ptr(gv), lc.ptr(np.array(k, F32)))
        rp, rm, rv = op.adam_step(p, g, m, v, step, lr32, 0.9, 0.99, 1e-15, wd)
        assert (bits(gm) == bits(rm)).all() and (bits(gv) == bits(rv)).all() and (bits(gp) == bits(rp)).all(), step
