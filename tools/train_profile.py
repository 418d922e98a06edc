#!/usr/bin/env python3
"""Wall time of ExpRunner::Train per 1000 iterations of the 20k-iteration fox run (where the run's seconds go): measurement aid."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import f2_nerf_amd  # noqa: F401
from f2_nerf_amd import runtime, fox_data
st = fox_data.load_state()
sc, images = fox_data.scene(2)
ds = runtime.make_dataset(sc, images)
runner, cfg, _ = runtime.make_runner(st, "wanjinyou", ["train.end_iter=20000"], seed=2022)
torch.manual_seed(2022)
torch.cuda.synchronize()
t_all = time.perf_counter()
for k in range(20):
    c0 = runner.counters(); torch.cuda.synchronize(); t0 = time.perf_counter()
    s = runner.train(ds, (k + 1) * 1000, 1)
    runner.flush(); torch.cuda.synchronize(); el = time.perf_counter() - t0
    c1 = runner.counters()
    n = max(int(s["iterations"]), 1)
    print("iters %5d..%5d  %.3f s  %.3f ms/iter  rays/iter %6.0f  marched/iter %7.0f  meaningful/iter %7.0f  nodes %6d  fineness %.2f" % (
        k * 1000, (k + 1) * 1000, el, el / n * 1e3, s["total_rays"] / n, (c1["total_marched"] - c0["total_marched"]) / n,
        (c1["total_meaningful"] - c0["total_meaningful"]) / n, runner.n_nodes(), float(runner.fineness)), flush=True)
print("total %.2f s   peak HBM allocated %.2f GiB, reserved by the allocator %.2f GiB" % (
    time.perf_counter() - t_all, torch.cuda.max_memory_allocated() / 2 ** 30, torch.cuda.max_memory_reserved() / 2 ** 30))
