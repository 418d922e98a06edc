"""What early_stop_votes' time is made of: the longest ray, the votes' atomics, or the sample count?  (measurement aid)"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import f2_nerf_amd  # noqa: F401
from f2_nerf_amd import capi as hip
rng = np.random.default_rng(1)
R = 14224
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
def bench(f, reps=30):
    for _ in range(3): f()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
def case(name, L, run_len=4, n_nodes=140000):
    L = L.astype(np.int64); start = np.concatenate([[0], np.cumsum(L)[:-1]]); end = start + L; n = int(L.sum())
    f0 = T((rng.standard_normal(n) * 1.5 + 1.0).astype(np.float32)); dt = T(np.full(n, 1 / 256., np.float32))
    anchors = T(np.stack([rng.integers(0, 300, n), np.repeat(rng.integers(0, n_nodes, n // run_len + 1), run_len)[:n], np.zeros(n, np.int64)], 1).astype(np.int32))
    w = torch.zeros(n, device="cuda"); a = torch.zeros(n, device="cuda"); mask = torch.zeros(n, dtype=torch.int32, device="cuda"); kept = torch.zeros(R, dtype=torch.int32, device="cuda")
    ad = [torch.full((n_nodes,), -1, dtype=torch.int32, device="cuda") for _ in range(2)] + [torch.zeros(n_nodes, dtype=torch.int32, device="cuda") for _ in range(2)]
    se = T(np.stack([start, end], 1).astype(np.int32))
    t1 = bench(lambda: hip.early_stop_votes(R, se, f0, 1, dt, w, a, mask, kept, anchors, 3, ad[0], ad[1], ad[2], ad[3]))
    t0 = bench(lambda: hip.early_stop(R, se, f0, 1, dt, w, a, mask, kept))
    t2 = bench(lambda: hip.oct_mark_visit(R, se, anchors, 3, w, a, ad[0], ad[1], ad[2], ad[3]))
    tt = T(np.sort(rng.random(n)).astype(np.float32)); rgb = T(rng.random((n, 3), dtype=np.float32)); bg = T(rng.random((R, 3), dtype=np.float32)); gt = T(rng.random((R, 3), dtype=np.float32))
    colors = torch.zeros((R, 3), device="cuda"); drgb = torch.zeros((n, 3), device="cuda"); df0 = torch.zeros(n, device="cuda"); losses = torch.zeros(8, device="cuda")
    t3 = bench(lambda: hip.composite_train(R, se, f0, 1, dt, tt, rgb, bg, gt, 0.01, 0.0, 0.1, 1.0, 0, 16, None, None, colors, w, drgb, df0, 1, losses))
    print("%-46s samples %7d  early_stop_votes %.1f us   early_stop alone %.1f   mark_visit alone %.1f   composite_train %.1f" % (name, n, t1, t0, t2, t3))
Lc = np.clip(np.round(np.exp(rng.normal(3.18, 0.9, R))), 1, 396); Lc[7] = 396
case("converged-like (max 396)", Lc)
case("same, clipped at 128", np.minimum(Lc, 128))
case("same, clipped at 48", np.minimum(Lc, 48))
case("uniform 36", np.full(R, 36))
case("uniform 16", np.full(R, 16))
Lk = np.clip(np.round(np.exp(rng.normal(2.5, 0.9, R))), 1, 396); Lk[7] = 300
case("kept-like (mean ~18, max 300)", Lk)
case("kept-like, clipped at 64", np.minimum(Lk, 64))
case("converged-like, leaf runs of 16 samples", Lc, run_len=16)
case("converged-like, 900 nodes (votes in LDS)", Lc, n_nodes=900)
