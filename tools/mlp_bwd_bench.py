#!/usr/bin/env python3
"""Times the two MLP backward kernels alone on a bench-sized synthetic batch (measurement aid; event timing, no profiler)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import f2_nerf_amd  # noqa: F401
from f2_nerf_amd import capi

n = int(sys.argv[1]) if len(sys.argv) > 1 else 800000
dev = "cuda"
torch.manual_seed(0)
drgb = torch.randn((n, 3), device=dev) * 1e-3
sidx = torch.randint(0, 50, (n // 100 + 1,), device=dev, dtype=torch.int32).repeat_interleave(100)[:n].contiguous()
ph2 = (torch.randn(7168, device=dev) * 0.1).to(torch.float16)
ph1 = (torch.randn(3072, device=dev) * 0.1).to(torch.float16)
sx = torch.randn((n, 32), device=dev).to(torch.float16)
dfeat = torch.zeros((n, 16), device=dev); dp2 = torch.zeros(7168, device=dev); dp1 = torch.zeros(3072, device=dev)
demb = torch.zeros((50, 16), device=dev); df0 = torch.randn(n, device=dev)
dy = torch.randn((n, 16), device=dev) * 1e-3
dx = torch.zeros((n, 32), device=dev)
sx32 = sx.float()

def timeit(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

print("n = %d" % n)
print("shade_bwd (colour MLP 32-64-64-16 + emb + df0 merge, incl. partial reductions)  %.3f ms" %
      timeit(lambda: capi.shade_bwd(n, drgb, sidx, ph2, sx, 128.0, dfeat, dp2, demb, df0)))
print("mlp_bwd NH=1 (field MLP 32-64-16, dx to fp32, incl. partial reduction)           %.3f ms" %
      timeit(lambda: capi.mlp_bwd(n, 32, 64, 1, 128.0, ph1, sx32, dy, dp1, dx)))
print("mlp_bwd NH=2                                                                    %.3f ms" %
      timeit(lambda: capi.mlp_bwd(n, 32, 64, 2, 128.0, ph2, sx32, dy, dp2, dx)))
