#!/usr/bin/env python3
"""Times the sampler kernels alone on a dumped converged scene (tools/train_fox.py --dump-sampler): measurement aid."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import f2_nerf_amd  # noqa: F401
from f2_nerf_amd import capi

z = np.load(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "tools", "data", "converged_sampler.npz"))
dev = "cuda"
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
tn, tr, so = T(z["tree_nodes"]), T(z["pers_trans"]), T(z["search_order"])
ro, rd_raw = T(z["rays_o"]), T(z["rays_d"])
n = ro.shape[0]
rd = torch.empty_like(rd_raw); capi.normalize_dirs(n, rd_raw, rd)
n_nodes = tn.numel() // 64
cb = torch.zeros(n_nodes * 8 * 32, dtype=torch.uint8, device=dev); capi.oct_build_child_blocks(n_nodes, tn, cb)
MH = 1024
se = torch.zeros((n, 2), dtype=torch.int32, device=dev); oi = torch.zeros(n * MH, dtype=torch.int32, device=dev)
nf = torch.zeros((n * MH, 2), device=dev); otr = torch.zeros(n * MH, dtype=torch.int32, device=dev)
tot = torch.zeros(1, dtype=torch.int32, device=dev)
torch.manual_seed(0)
noise = ((torch.rand(1024 + n + 10, device=dev) - .5) + 1.) * float(z["fineness"])
cnt = torch.zeros(n, dtype=torch.int32, device=dev)
s_pts = torch.zeros((n * 1024, 3), device=dev); s_dt = torch.zeros(n * 1024, device=dev); s_t = torch.zeros(n * 1024, device=dev)
s_an = torch.zeros((n * 1024, 2), dtype=torch.int32, device=dev); fod = torch.zeros(n, device=dev)

def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

def isect():
    tot.zero_()
    capi.oct_intersect_strided(n, MH, so, ro, rd, 0.01, 1e8, tn, se, oi, nf, tot, otr, cb)
def march():
    capi.ray_march_strided(n, 1. / 256., True, ro, rd, noise, se, oi, nf, tn, tr, cnt, None, s_dt, s_t, s_an, fod, otr)
def march_count():
    capi.ray_march_count(n, 1. / 256., True, ro, rd, noise, se, oi, nf, tn, tr, cnt)
flush_buf = torch.empty(192 * 1024 * 1024, device=dev)  # 768 MB: evicts the L2s and the 256 MB Infinity Cache
def cold(fn, reps=5):
    tot_ms = 0.0
    for r in range(reps):
        flush_buf.fill_(float(r)); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        tot_ms += e0.elapsed_time(e1)
    return tot_ms / reps
print("rays %d nodes %d" % (n, n_nodes))
print("oct_intersect_strided cold caches %.3f ms" % cold(isect))
print("oct_intersect_strided %.3f ms" % timeit(isect))
print("ray_march_strided     %.3f ms" % timeit(march))
print("ray_march_count       %.3f ms" % timeit(march_count))
print("ray_march_strided cold caches %.3f ms" % cold(march))
c = cnt.cpu().numpy(); h = (se[:, 1] - se[:, 0]).cpu().numpy()
print("samples/ray mean %.1f max %d ; leaf hits/ray mean %.1f max %d" % (c.mean(), c.max(), h.mean(), h.max()))
# how often does the transform change along a ray?
otr_n, se_n = otr.cpu().numpy(), se.cpu().numpy()
ch = [int((np.diff(otr_n[a:b]) != 0).sum()) for a, b in se_n[:2000]]
print("transform changes per ray (first 2000 rays): mean %.1f max %d" % (np.mean(ch), np.max(ch)))
