"""Times the pieces of the field kernels separately on a realistic bench batch (GPU box only; measurement aid)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import f2_nerf_amd
from f2_nerf_amd import runtime, capi

st = dict(np.load(os.path.join(ROOT, "tests", "golden", "fox_state.npz")))
runner, cfg, arrays = runtime.make_runner(st, "wanjinyou", seed=2022)
rng = np.random.default_rng(1000)
b = runtime.to_dev(*runtime.synthetic_ray_batch(st, 8192, rng))
s = runner.get_samples(b[0], b[1], b[2])
pts, anchors = s["pts"], s["anchors"]
n = pts.shape[0]
print("samples", n)
dev = "cuda"
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
log2 = 19; local = 1 << log2
table_h = torch.from_numpy(arrays[4]).to(dev).to(torch.float16)
prim, bias = T(arrays[5]), T(arrays[6]); nv = int(arrays[7][0])
lidx = T((np.arange(16) * local).astype(np.int32)); lsize = T(np.full(16, local, np.int32)); scale = T(np.exp2(3.0 + 7.0 * np.arange(16) / 15.0).astype(np.float32))  # Hash3DAnchored.cu:28
ph = T(arrays[8]).to(torch.float16)

def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

xh = torch.zeros((n, 32), dtype=torch.float16, device=dev)
feat = torch.zeros((n, 16), device=dev); f0 = torch.zeros(n, device=dev)
print("hash_fwd (all levels per wave)      %.3f ms" % timeit(lambda: capi.hash_fwd(n, nv, table_h, prim, lidx, lsize, bias, scale, pts, True, anchors, 3, xh)))
print("field_fwd density (partitioned)      %.3f ms" % timeit(lambda: capi.field_fwd(n, nv, table_h, prim, lidx, lsize, bias, scale, pts, anchors, 3, ph, None, f0, None)))
print("field_fwd train (partitioned)        %.3f ms" % timeit(lambda: capi.field_fwd(n, nv, table_h, prim, lidx, lsize, bias, scale, pts, anchors, 3, ph, feat, None, xh)))
x32 = xh.float()
out_h = torch.zeros((n, 16), dtype=torch.float16, device=dev)
print("mlp_fwd NH=1 alone                   %.3f ms" % timeit(lambda: capi.mlp_fwd(n, 32, 64, 1, ph, x32, out_h)))
dy = torch.randn((n, 16), device=dev) * 1e-3
dp = torch.zeros(3072, device=dev); dx = torch.zeros((n, 32), device=dev)
print("mlp_bwd NH=1 alone (with dx store)   %.3f ms" % timeit(lambda: capi.mlp_bwd(n, 32, 64, 1, 128.0, ph, x32, dy, dp, dx)))
print("mlp_bwd NH=1 alone (no dx)           %.3f ms" % timeit(lambda: capi.mlp_bwd(n, 32, 64, 1, 128.0, ph, x32, dy, dp, None)))
p2 = T(arrays[9]).to(torch.float16); dp2 = torch.zeros(7168, device=dev)
print("mlp_bwd NH=2 alone (no dx)           %.3f ms" % timeit(lambda: capi.mlp_bwd(n, 32, 64, 2, 128.0, p2, x32, dy, dp2, None)))
gin = (torch.randn((n, 32), device=dev) * 0.05).to(torch.float16)
gtab = torch.zeros(16 * local * 2, dtype=torch.float16, device=dev)
sparse20 = gin.clone(); sparse20[torch.rand(n, device=dev) > 0.2] = 0  # ~ the bench's share of non-zero gradients
variants = {"dense random grads": gin, "20% of samples non-zero": sparse20}
for lo, hi in ((0, 4), (4, 8), (8, 12), (12, 16)):
    g = torch.zeros_like(gin); g[:, 2 * lo:2 * hi] = gin[:, 2 * lo:2 * hi]
    variants["levels %d-%d only" % (lo, hi - 1)] = g
for name, g in variants.items():
    for le, tag in ((0, "atomics"), (local, "binned ")):
        print("hash_bwd %s (%-24s) %.3f ms" % (tag, name, timeit(lambda: capi.hash_bwd(n, nv, prim, lidx, lsize, bias, scale, pts, True, anchors, 3, g, gtab, le))))
print("field_bwd fused (binned)              %.3f ms" % timeit(lambda: capi.field_bwd(n, nv, prim, lidx, lsize, bias, scale, pts, anchors, 3, ph, xh, dy, 128.0, dp, gtab, local)))
