"""Does the ORDER in which rays are handed to the row-per-ray kernels matter (early_stop_votes, composite_train)?  Same samples, same rays,
the (start, end) rows permuted: in ray order (lengths arrive at random), longest first, shortest first.  Lengths: log-normal matched to the
converged batch (p50 24, p99 ~200, max 396, mean ~39).  Measurement aid."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import f2_nerf_amd  # noqa: F401
from f2_nerf_amd import capi as hip
rng = np.random.default_rng(1)
R = 14224
L = np.clip(np.round(np.exp(rng.normal(3.18, 0.9, R))), 1, 396).astype(np.int64)
L[rng.integers(0, R)] = 396
start = np.concatenate([[0], np.cumsum(L)[:-1]]); end = start + L
n = int(L.sum())
print("rays %d samples %d mean %.1f p50 %d p99 %d max %d" % (R, n, L.mean(), np.percentile(L, 50), np.percentile(L, 99), L.max()))
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
f0 = T((rng.standard_normal(n) * 1.5 + 1.0).astype(np.float32)); dt = T(np.full(n, 1 / 256., np.float32)); t = T(np.sort(rng.random(n)).astype(np.float32))
anchors = T(np.stack([rng.integers(0, 300, n), np.repeat(rng.integers(0, 100000, n // 4 + 1), 4)[:n], np.zeros(n, np.int64)], 1).astype(np.int32))
n_nodes = 140000
rgb = T(rng.random((n, 3), dtype=np.float32)); bg = T(rng.random((R, 3), dtype=np.float32)); gt = T(rng.random((R, 3), dtype=np.float32))
w = torch.zeros(n, device="cuda"); a = torch.zeros(n, device="cuda"); mask = torch.zeros(n, dtype=torch.int32, device="cuda"); kept = torch.zeros(R, dtype=torch.int32, device="cuda")
adders = [torch.full((n_nodes,), -1, dtype=torch.int32, device="cuda") for _ in range(2)] + [torch.zeros(n_nodes, dtype=torch.int32, device="cuda") for _ in range(2)]
colors = torch.zeros((R, 3), device="cuda"); drgb = torch.zeros((n, 3), device="cuda"); df0 = torch.zeros(n, device="cuda"); losses = torch.zeros(8, device="cuda")
def bench(f, reps=30):
    for _ in range(3): f()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for name, perm in (("ray order", np.arange(R)), ("longest first", np.argsort(-L, kind="stable")), ("shortest first", np.argsort(L, kind="stable"))):
    se = T(np.stack([start[perm], end[perm]], 1).astype(np.int32))
    t1 = bench(lambda: hip.early_stop_votes(R, se, f0, 1, dt, w, a, mask, kept, anchors, 3, adders[0], adders[1], adders[2], adders[3]))
    t2 = bench(lambda: hip.composite_train(R, se, f0, 1, dt, t, rgb, bg, gt, 0.0, 0.0, 0.1, 1.0, 0, 16, None, None, colors, w, drgb, df0, 1, losses))
    print("%-15s early_stop_votes %.1f us   composite_train %.1f us" % (name, t1, t2))
