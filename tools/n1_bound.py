#!/usr/bin/env python3
"""Judge row N1 ("compositing fused into the colour-MLP epilogue"), closed by a number: what can an order-free sum of w * c in the
colour network's epilogue take off the step AT MOST?  The sum's only consumer is composite_train (compositing forward + loss +
compositing backward, one launch), whose forward walk would then skip the samples' colour loads and three of its eight scan
chains -- everything else of that launch (densities, weights, disparity, depth, WeightVar statistics, the loss, the whole backward
walk, which reads the colours anyway) stays.  This times f2n_composite_train against the debug variant's
f2n_debug_composite_train_colors_in (the sums handed in for free) on the converged fox batch's surviving samples.  The epilogue's own
cost (a weight and a ray index per sample, a segmented row sum, ~3 float atomics per 16-sample tile and ray, a zero-fill of the
colour buffer) is NOT charged: the difference printed here is an upper bound of the saving."""
import ctypes, os, sys
os.environ["F2N_DEBUG_BUILD"] = "1"
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import f2_nerf_amd  # noqa: F401
from f2_nerf_amd import capi
src = open(os.path.join(ROOT, "tools", "gather_ab.py")).read().split("rng = np.random.default_rng(1)")[0]
ns = {"__file__": os.path.join(ROOT, "tools", "gather_ab.py"), "__name__": "gather_ab_helpers"}
exec(compile(src, "gather_ab_helpers", "exec"), ns)
sample, timeit = ns["sample"], ns["timeit"]
dev = "cuda"
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
z = np.load(os.path.join(ROOT, "tools", "data", "converged_sampler.npz"))
conv = sample(z["tree_nodes"], z["pers_trans"], z["search_order"], z["rays_o"], z["rays_d"], float(z["fineness"]))
t_all = conv["t"].cpu().numpy()
start = np.nonzero(np.r_[True, np.diff(t_all) < 0])[0]
end = np.r_[start[1:], len(t_all)]
keep = np.maximum(1, ((end - start) * 0.47).astype(np.int64))
idx = np.concatenate([np.arange(s, s + k) for s, k in zip(start, keep)])
n_rays = len(start)
se = np.zeros((n_rays, 2), np.int32); se[:, 1] = np.cumsum(keep); se[1:, 0] = se[:-1, 1]
m = len(idx)
rng = np.random.default_rng(3)
dt, t = conv["dt"][T(idx)].contiguous(), conv["t"][T(idx)].contiguous()
f0 = T((rng.standard_normal(m) * 1.5 + 4.0).astype(np.float32))
rgb = T(rng.random((m, 3), dtype=np.float32)); bg = T(rng.random((n_rays, 3), dtype=np.float32)); gt = T(rng.random((n_rays, 3), dtype=np.float32))
se_d = T(se)
outs = [dict(colors=torch.zeros((n_rays, 3), device=dev), weights=torch.zeros(m, device=dev), drgb=torch.zeros((m, 3), device=dev),
             df0=torch.zeros(m, device=dev), losses=torch.zeros(8, device=dev)) for _ in range(2)]
L = capi.lib()
P = lambda x: ctypes.c_void_p(x.data_ptr())
st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
fl, ci = ctypes.c_float, ctypes.c_int


def plain(o):
    rc = L.f2n_composite_train(st(), ci(n_rays), P(se_d), P(f0), ci(1), P(dt), P(t), P(rgb), P(bg), P(gt), fl(0.01), fl(0.0), fl(0.1), fl(1.0), ci(0), ci(16),
                               None, None, P(o["colors"]), P(o["weights"]), P(o["drgb"]), P(o["df0"]), ci(1), P(o["losses"]), ci(0))
    assert rc == 0, rc


plain(outs[0]); torch.cuda.synchronize()
wc = torch.zeros((n_rays, 3), device=dev)  # the sums an epilogue would deliver: colours minus the background term
last_trans = torch.exp(-torch.zeros(n_rays, device=dev).index_add_(0, T(np.repeat(np.arange(n_rays), keep)), torch.exp(f0 - 3.0) * dt))
wc = outs[0]["colors"] - last_trans[:, None] * bg


def handed_in(o):
    rc = L.f2n_debug_composite_train_colors_in(st(), ci(n_rays), P(se_d), P(f0), ci(1), P(dt), P(t), P(rgb), P(bg), P(gt), fl(0.01), fl(0.0), fl(0.1), fl(1.0),
                                               P(o["colors"]), P(o["weights"]), P(o["drgb"]), P(o["df0"]), P(o["losses"]), P(wc))
    assert rc == 0, rc


handed_in(outs[1]); torch.cuda.synchronize()
err = float((outs[0]["colors"] - outs[1]["colors"]).abs().max())
grad_err = max(float((outs[0]["df0"] - outs[1]["df0"]).abs().max()), float((outs[0]["drgb"] - outs[1]["drgb"]).abs().max()))
res = []
for rep in range(5):
    a = timeit(lambda: plain(outs[0]), 50) * 1e3
    b = timeit(lambda: handed_in(outs[1]), 50) * 1e3
    res.append((a, b))
a, b = np.median([r[0] for r in res]), np.median([r[1] for r in res])
print("n1_bound: %d rays, %d surviving samples (longest ray %d) | composite_train %.1f us | with the colour sums handed in %.1f us | upper bound of the saving %.1f us "
      "(median of 5 alternated pairs of 50 launches; the handed-in sums are rebuilt from the first launch's colours: colours agree to %.1e, sample gradients to %.1e)" % (n_rays, m, int(keep.max()), a, b, a - b, err, grad_err))
