#!/usr/bin/env python3
"""Times the hash-gradient scatter (f2n_hash_bwd, owner-binned path: hash_bin_kernel -> hash_bin_accumulate_kernel) alone on the
GPU, on real sample sets: the converged fox octree's training batch (the front `keep` fraction of every ray: what early stop
leaves) and the fresh scene's 8192-ray batch.  Run under `rocprofv3 --kernel-trace --stats` for the per-kernel split.
Measurement aid (round-4 verdict, weak 4: "the scatter does not scale down with the batch")."""
import argparse, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import f2_nerf_amd  # noqa: F401
from f2_nerf_amd import capi, runtime
sys.path.insert(0, os.path.join(ROOT, "tools"))
import importlib.util
ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=30)
ap.add_argument("--zero-frac", type=float, default=0.3)
ap.add_argument("--only", default="", help="substring of the one sample set to run (e.g. '47%')")
ap.add_argument("--amps", default="2e-4,0.05", help="standard deviations of the synthetic f16 gradients: a training's loss-scaled gradients sum to "
                "~1 per table slice (f2n_debug_counters()[2] of the debug variant: 0.6 converged, 1.2 fresh), which 2e-4 reproduces; "
                "0.05 drives every slice's sum of |addend| past the fixed-point route's range, i.e. times the owners' fp64 route")
args = ap.parse_args()
dev = "cuda"
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
st = dict(np.load(os.path.join(ROOT, "tests", "golden", "fox_state.npz")))
spec = importlib.util.spec_from_file_location("gather_ab_sample", os.path.join(ROOT, "tools", "gather_ab.py"))
src = open(os.path.join(ROOT, "tools", "gather_ab.py")).read().split("rng = np.random.default_rng(1)")[0]  # (its helpers only)
ns = {"__file__": os.path.join(ROOT, "tools", "gather_ab.py"), "__name__": "gather_ab_helpers"}
exec(compile(src, "gather_ab_helpers", "exec"), ns)
sample, timeit = ns["sample"], ns["timeit"]
rng = np.random.default_rng(1)
log2 = 19
nvol = int(st["n_volumes"])
prim, bias = T(st["prim_pool"]), T(st["bias_pool"])
lidx = T((np.arange(16) * (1 << log2)).astype(np.int32)); lsize = T(np.full(16, 1 << log2, np.int32))
scale = T(np.exp2(7.0 * np.arange(16) / 15.0 + 3.0).astype(np.float32))
z = np.load(os.path.join(ROOT, "tools", "data", "converged_sampler.npz"))
conv = sample(z["tree_nodes"], z["pers_trans"], z["search_order"], z["rays_o"], z["rays_d"], float(z["fineness"]))
ro, rd, _, _, _ = runtime.synthetic_ray_batch(st, 8192, rng)
fresh = sample(st["tree_nodes"], st["pers_trans"], st["search_order"], ro, rd, 16.0)


def front_part(smp, keep):
    """The first `keep` fraction of every run of equal ... rows are ray-ordered; a ray's rows are where t restarts."""
    t = smp["t"].cpu().numpy()
    start = np.nonzero(np.r_[True, np.diff(t) < 0])[0]
    end = np.r_[start[1:], len(t)]
    idx = np.concatenate([np.arange(s, s + max(1, int((e - s) * keep))) for s, e in zip(start, end)])
    return smp["pts"][T(idx)].contiguous(), smp["anchors"][T(idx)].contiguous()


sets = {"converged, 47% of every ray": front_part(conv, 0.47), "converged, 24% of every ray": front_part(conv, 0.24),
        "fresh, all samples": (fresh["pts"], fresh["anchors"])}
for amp in [float(a) for a in args.amps.split(",")]:
    for name, (pts, anchors) in sets.items():
        if args.only and args.only not in name:
            continue
        n = pts.shape[0]
        g = (torch.randn((n, 32), device=dev) * amp).to(torch.float16)
        g[torch.rand(n, device=dev) < args.zero_frac] = 0
        table = torch.zeros(16 << (log2 + 1), dtype=torch.float16, device=dev)
        f = lambda: capi.hash_bwd(n, nvol, prim, lidx, lsize, bias, scale, pts, True, anchors, 3, g, table, 1 << log2)
        ms = timeit(f, args.reps)
        capi.debug_counters(reset=True)
        table.zero_(); f()
        c = capi.debug_counters()
        print("scatter_bench gradients ~ N(0, %g^2)  %-30s n %7d  %.4f ms per scatter (both kernels)  checksum of one scatter %d  records applied by atomics %d  "
              "slices summed on the fp64 route %d of %d" % (amp, name, n, ms, int(table.view(torch.int32).to(torch.int64).sum()), c[0], c[1], 17 * (1 << log2) // 8192), flush=True)
