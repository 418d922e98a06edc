#!/usr/bin/env python3
"""Times f2n_hash_gather_planes (+ checks its planes against the fused per-sample gather f2n_hash_fwd, bit for bit) on two
sample sets: the converged fox octree's training batch (tools/data/converged_sampler.npz, fineness ~1: long runs of samples
per coarse cell) and a fresh-scene batch (8192 synthetic rays, fineness 16).  Both the plain one-pair-per-XCD
split and the run-combined, cost-balanced one (f2n_hash_gather_planes_balanced) are timed: measurement aid."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import f2_nerf_amd  # noqa: F401
from f2_nerf_amd import capi, runtime

dev = "cuda"
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
st = dict(np.load(os.path.join(ROOT, "tests", "golden", "fox_state.npz")))


def sample(tree_nodes, pers_trans, so, ro, rd_raw, fineness, seed=0):
    n = ro.shape[0]
    tn, tr, so = T(tree_nodes), T(pers_trans), T(so)
    ro, rd_raw = T(ro), T(rd_raw)
    rd = torch.empty_like(rd_raw); capi.normalize_dirs(n, rd_raw, rd)
    n_nodes = tn.numel() // 64
    cb = torch.zeros(n_nodes * 8 * 32, dtype=torch.uint8, device=dev); capi.oct_build_child_blocks(n_nodes, tn, cb)
    MH = 1024
    se = torch.zeros((n, 2), dtype=torch.int32, device=dev); oi = torch.zeros(n * MH, dtype=torch.int32, device=dev)
    nf = torch.zeros((n * MH, 2), device=dev); otr = torch.zeros(n * MH, dtype=torch.int32, device=dev)
    tot = torch.zeros(1, dtype=torch.int32, device=dev)
    capi.oct_intersect_strided(n, MH, so, ro, rd, 0.01, 1e8, tn, se, oi, nf, tot, otr, cb)
    torch.manual_seed(seed)
    noise = ((torch.rand(1024 + n + 10, device=dev) - .5) + 1.) * float(fineness)
    cnt = torch.zeros(n, dtype=torch.int32, device=dev)
    s_dt = torch.zeros(n * 1024, device=dev); s_t = torch.zeros(n * 1024, device=dev)
    s_an = torch.zeros((n * 1024, 2), dtype=torch.int32, device=dev); fod = torch.zeros(n, device=dev)
    capi.ray_march_strided(n, 1. / 256., True, ro, rd, noise, se, oi, nf, tn, tr, cnt, None, s_dt, s_t, s_an, fod, otr)
    pse = torch.zeros((n, 2), dtype=torch.int32, device=dev)
    capi.segment_scan(n, cnt, pse, tot)
    m = int(tot.item())
    out = dict(pts=torch.zeros((m, 3), device=dev), dirs=torch.zeros((m, 3), device=dev), dt=torch.zeros(m, device=dev),
               t=torch.zeros(m, device=dev), anchors=torch.zeros((m, 3), dtype=torch.int32, device=dev))
    capi.pack_samples(n, pse, ro, rd, tr, None, s_dt, s_t, s_an, out["pts"], out["dirs"], out["dt"], out["t"], out["anchors"])
    return out


def timeit(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


rng = np.random.default_rng(1)
log2 = 19
nvol = int(st["n_volumes"])
table = T((rng.standard_normal((16 << log2, 2)) * 0.1).astype(np.float16))
prim, bias = T(st["prim_pool"]), T(st["bias_pool"])
lidx = T((np.arange(16) * (1 << log2)).astype(np.int32)); lsize = T(np.full(16, 1 << log2, np.int32))
scale = T(np.exp2(7.0 * np.arange(16) / 15.0 + 3.0).astype(np.float32))
z = np.load(os.path.join(ROOT, "tools", "data", "converged_sampler.npz"))
sets = {"converged (13056 rays, fineness %.2f)" % float(z["fineness"]): sample(z["tree_nodes"], z["pers_trans"], z["search_order"], z["rays_o"], z["rays_d"], float(z["fineness"]))}
ro, rd, _, _, _ = runtime.synthetic_ray_batch(st, 8192, rng)
sets["fresh (8192 rays, fineness 16)"] = sample(st["tree_nodes"], st["pers_trans"], st["search_order"], ro, rd, 16.0)
scales_host = np.exp2(7.0 * np.arange(16) / 15.0 + 3.0).astype(np.float32)
for name, smp in sets.items():
    n = smp["pts"].shape[0]
    fin = float(z["fineness"]) if name.startswith("converged") else 16.0
    step01 = (1. / 256.) * 1.25 * fin * 0.5
    ref = torch.zeros((n, 32), dtype=torch.float16, device=dev)
    capi.hash_fwd(n, nvol, table, prim, lidx, lsize, bias, scale, smp["pts"], True, smp["anchors"], 3, ref)
    for mode, st01 in (("one pair per XCD", 0.0), ("run-combined, cost-balanced", step01)):
        planes = torch.zeros((8, n, 4), dtype=torch.float16, device=dev)
        f = lambda: capi.hash_gather_planes_balanced(n, nvol, table, prim, lidx, lsize, bias, scale, smp["pts"], True, smp["anchors"], 3,
                                                     planes, st01, scales_host)
        ms = timeit(f)
        rows = planes.view(8, n, 2, 2).permute(1, 0, 2, 3).reshape(n, 32)  # plane p = levels (2p, 2p+1) -> row-major [n][32]
        same = bool((rows.contiguous().view(torch.int16) == ref.view(torch.int16)).all())
        print("%-40s %-28s n %7d  %.4f ms  %6.1f G lane-gathers/s  bit-identical to f2n_hash_fwd: %s" % (name, mode, n, ms, n * 128 / ms / 1e6, same))
