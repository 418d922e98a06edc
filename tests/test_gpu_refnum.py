"""GPU (run with `-m gpu`): the REFERENCE-NUMERICS build of the kernel library (libf2n_hip_refnum.so, include/f2n_abi.h
f2n_numerics_mode) -- the comparator bench.py's `psnr_numerics_ab` trains with.  A process has one numerics, chosen by the
environment before the package is imported, so the checks run in a child process."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import os, sys
import numpy as np, torch
sys.path.insert(0, os.environ["F2N_ROOT"]); sys.path.insert(0, os.path.join(os.environ["F2N_ROOT"], "tests"))
import f2_nerf_amd
from f2_nerf_amd import capi, runtime
from oracle import capi as oc, pipeline as op
F32 = np.float32
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
N = lambda t: t.detach().cpu().numpy()
assert capi.lib().f2n_numerics_mode() == 1 and "REFERENCE-NUMERICS" in capi.build_info()
assert "refnum" in runtime.host().__file__
rng = np.random.default_rng(3)
# ---- MLP forward: the f16 accumulator fragment is the oracle's accumulator mode 1 (or_mlp_dot) ----
for n_hidden in (1, 2):
    nparam = oc.mlp_n_params(32, 64, n_hidden)
    params = (rng.standard_normal(nparam) * 0.25).astype(F32)
    x = rng.standard_normal((4096, 32)).astype(F32)
    out = torch.zeros((4096, 16), dtype=torch.float16, device="cuda")
    capi.mlp_fwd(4096, 32, 64, n_hidden, T(oc.f2h(params).view(np.float16)), T(x), out)
    got = N(out).astype(F32)
    ref0 = oc.h2f(oc.mlp_fwd(params, x, 64, n_hidden))
    with oc.mlp_accumulator(1):
        ref1 = oc.h2f(oc.mlp_fwd(params, x, 64, n_hidden))
    e0, e1 = np.abs(got - ref0), np.abs(got - ref1)
    scale = np.abs(ref1).max()
    # (the 16 products of a k-block are summed by the matrix core, not in k order: the last bit of a block sum may differ)
    print("REFNUM_MLP n_hidden %d: |gpu - mode1| mean %.3e max %.3e   |gpu - mode0| mean %.3e max %.3e   (scale %.2f)" % (
        n_hidden, e1.mean(), e1.max(), e0.mean(), e0.max(), scale))
    assert e1.max() <= 4 * 2.0 ** -11 * scale and e1.mean() < 0.25 * e0.mean(), (e1.mean(), e0.mean())
    assert (got == ref1).mean() > 0.97, (got == ref1).mean()
# ---- hash backward: per-addend f16 atomics (every entry a sum of f16-rounded addends with f16 running sums) ----
st = dict(np.load(os.path.join(os.environ["F2N_ROOT"], "tests", "golden", "fox_state.npz")))
g = dict(np.load(os.path.join(os.environ["F2N_ROOT"], "tests", "golden", "fox_golden.npz")))
log2 = 14
grid = op.HashGrid(np.zeros((16 << log2, 2), F32), st["prim_pool"], st["bias_pool"], int(st["n_volumes"]), log2)
reps = 12
pts = np.tile(g["march_pts"], (reps, 1)); vol = np.tile(np.ascontiguousarray(g["march_anchors"][:, 0]), reps)
n = len(pts)
assert n > 40000  # (the product build would take the owner-binned path here)
gin = (rng.standard_normal((n, 32)) * 0.05).astype(F32)
gin_h = oc.f2h(gin)
gtab = torch.zeros(grid.table_f32.size, dtype=torch.float16, device="cuda")
capi.hash_bwd(n, grid.n_volumes, T(grid.prim_pool), T(grid.local_idx), T(grid.local_size), T(grid.bias_pool), T(grid.scales), T(pts), True,
              T(vol), 1, T(gin_h.view(np.float16)), gtab, 1 << log2)
q01 = ((pts + F32(1.)) * F32(.5)).astype(F32)
ref32 = oc.hash_bwd(grid.table_f32.size, grid.prim_pool, grid.local_idx, grid.local_size, grid.bias_pool, q01, vol, grid.n_volumes, gin_h,
                    grid.scales, fp32_accumulate=True)
ref16 = oc.h2f(oc.hash_bwd(grid.table_f32.size, grid.prim_pool, grid.local_idx, grid.local_size, grid.bias_pool, q01, vol, grid.n_volumes,
                           gin_h, grid.scales, fp32_accumulate=False))
got = N(gtab).astype(F32)
d32, d16, o16 = np.abs(got - ref32), np.abs(got - ref16), np.abs(ref16 - ref32)
print("REFNUM_HASH_BWD n %d: |gpu - fp32 sums| mean %.3e max %.3e   |oracle f16 sums - fp32 sums| mean %.3e max %.3e   max |ref| %.3f" % (
    n, d32.mean(), d32.max(), o16.mean(), o16.max(), np.abs(ref32).max()))
assert ((got != 0) == (ref16 != 0)).mean() > 0.999
# f16 running sums in SOME order: as far from the exact sums as the oracle's f16 accumulation in ITS order, not closer
assert 0.3 * o16.mean() <= d32.mean() <= 3.0 * o16.mean(), (d32.mean(), o16.mean())
assert d32.max() <= 0.05 * np.abs(ref32).max()
# ---- a short training run with this build ----
runner, cfg, arrays = runtime.make_runner(st, "wanjinyou", ["field.log2_table_size=15", "train.learning_rate_warm_up_end_iter=20"], seed=1, table_init=0.3)
runner.n_edge_pts = 512
R = 2048
import test_gpu_e2e as e2e
ro, rd, bounds, cam = e2e.fox_batch(st, rng, R)
gt = np.tile(np.array([[0.7, 0.4, 0.1]], F32), (R, 1))
d = runtime.to_dev(ro, rd, bounds, gt, cam)
mse = []
for it in range(40):
    s = runner.train_step(d[0], d[1], d[2], d[3], d[4], True)
    assert not s["skipped_nan"]
    mse.append(float(s["mse"]))
assert np.isfinite(mse).all() and min(mse[-5:]) < 0.8 * mse[0], (mse[0], mse[-5:])
print("REFNUM_TRAIN mse %.4f -> %.4f" % (mse[0], mse[-1]))
print("REFNUM_OK")
'''


def test_reference_numerics_build_on_the_device():
    """MLP forward == the oracle's f16-accumulator reading (mode 1), hash backward == f16 running sums of f16 addends (as far from
    the exact sums as the oracle's f16 accumulation, every entry touched), and the build trains."""
    if not torch.cuda.is_available():
        pytest.fail("no HIP device visible")
    env = dict(os.environ, F2N_REFERENCE_NUMERICS="1", F2N_ROOT=ROOT)
    r = subprocess.run([sys.executable, "-c", CHILD], capture_output=True, text=True, env=env, timeout=900)
    print(r.stdout[-3000:])
    assert r.returncode == 0 and "REFNUM_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])


OWNER_CHILD = r'''
import os, sys
import numpy as np, torch
sys.path.insert(0, os.environ["F2N_ROOT"]); sys.path.insert(0, os.path.join(os.environ["F2N_ROOT"], "tests"))
import f2_nerf_amd
from f2_nerf_amd import capi
from oracle import pipeline as op
F32 = np.float32
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
st = dict(np.load(os.path.join(os.environ["F2N_ROOT"], "tests", "golden", "fox_state.npz")))
g = dict(np.load(os.path.join(os.environ["F2N_ROOT"], "tests", "golden", "fox_golden.npz")))
rng = np.random.default_rng(5)
out = {}
for log2, reps, amp in ((14, 12, 2e-4), (19, 12, 0.05), (19, 12, 3.0)):
    grid = op.HashGrid(np.zeros((16 << log2, 2), F32), st["prim_pool"], st["bias_pool"], int(st["n_volumes"]), log2)
    pts = np.tile(g["march_pts"], (reps, 1)); vol = np.tile(np.ascontiguousarray(g["march_anchors"][:, 0]), reps)
    n = len(pts)
    gin = (rng.standard_normal((n, 32)) * amp).astype(np.float16)
    gin[rng.random(n) < 0.3] = 0
    gtab = torch.zeros(grid.table_f32.size, dtype=torch.float16, device="cuda")
    capi.debug_counters(reset=True)
    for _ in range(2):   # the second call adds to what the first left (old != 0)
        capi.hash_bwd(n, grid.n_volumes, T(grid.prim_pool), T(grid.local_idx), T(grid.local_size), T(grid.bias_pool), T(grid.scales), T(pts), True,
                      T(vol), 1, T(gin), gtab, 1 << log2)
    torch.cuda.synchronize()
    c = capi.debug_counters()
    out["tab_%d_%g" % (log2, amp)] = gtab.cpu().numpy().view(np.uint16)
    out["cnt_%d_%g" % (log2, amp)] = np.array([c[0], c[1]])
np.savez(os.environ["F2N_OUT"], **out)
'''


def test_owner_fixed_point_route_equals_fp64_route(tmp_path):
    """The scatter's owners (round 6) sum a slice in a packed fixed-point image -- one 64-bit integer add per record -- and fall back
    to fp64 sums, one channel at a time, when the slice's addends could leave the fields' range.  Both are exact sums of the same
    f16 addends: the debug variant with every slice forced onto the fp64 route (F2N_OWNER_F64=1) must leave the same bits as the
    default; small gradients stay on the integer route, large ones leave it (f2n_debug_counters()[1]), by themselves."""
    import numpy as np
    res = {}
    for name, env_extra in (("default", {}), ("f64", {"F2N_OWNER_F64": "1"})):
        out = str(tmp_path / (name + ".npz"))
        env = dict(os.environ, F2N_DEBUG_BUILD="1", F2N_ROOT=ROOT, F2N_OUT=out, **env_extra)
        env.pop("F2N_REFERENCE_NUMERICS", None)
        r = subprocess.run([sys.executable, "-c", OWNER_CHILD], env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
        res[name] = dict(np.load(out))
    keys = [k for k in res["default"] if k.startswith("tab_")]
    assert len(keys) == 3
    for k in keys:
        assert (res["default"][k] != 0).any()
        assert (res["default"][k] == res["f64"][k]).all(), k
    d, f = res["default"], res["f64"]
    assert (d["cnt_14_0.0002"] == 0).all() and d["cnt_19_0.05"][0] == 0    # no atomic fallback; tiny gradients: integer route only
    assert d["cnt_19_3"][1] > 0                                             # large gradients: slices leave the integer route
    assert f["cnt_14_0.0002"][1] > 0 and f["cnt_19_0.05"][1] > 0            # forced: every slice with records
