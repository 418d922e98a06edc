"""Test infrastructure (it builds its inputs with the oracle's hash-grid layout helpers, so it lives under tests/).  What the kernels' SOURCE TEXT moves through memory, per unit of work -- the figures DESIGN.md section 4 prices the roofline with
(592 B per marched sample for the gather, 8-byte records for the scatter, 30 B per table entry for Adam), counted
instead of argued.  The kernels are compiled for the emulated wavefront (tests/wave_emul/) with the compiler's load / store
instrumentation on top; the runtime classifies every access of a launch (private stack = registers: dropped; the dynamic LDS buffer
and the library's own statics = LDS; everything else = global memory) and counts bytes (aggregate copies the compiler expands into memcpy are not seen: noted where it matters).  No GPU needed:
`python tests/wave_emul/source_level_traffic.py > profiles/r05_source_level_traffic.txt`.  What it cannot know: what the caches make of it
(that is what the FETCH_SIZE / WRITE_SIZE passes of profiles/run_profiles.sh measure on the hardware)."""
import ctypes
import os
import sys
from collections import OrderedDict

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "wave_emul"))

F32 = np.float32


def main():
    import wemu_build
    import f2_nerf_amd  # noqa: F401
    from f2_nerf_amd import capi
    import test_gpu_parity as gp
    lib, _ = wemu_build.build(tag="traffic", traffic=True)
    L = ctypes.CDLL(lib)
    capi._lib = L
    capi._p = lambda t, kind=None, allow_none=False: ctypes.c_void_p(0 if t is None else t.data_ptr())
    capi._stream = lambda: ctypes.c_void_p(0)
    torch.cuda.synchronize = lambda *a, **k: None
    gp.DEV = "cpu"
    gp.T = lambda a: torch.from_numpy(np.array(a, copy=True, order="C"))
    gp.N = lambda t: t.detach().numpy().copy()
    T, oc = gp.T, gp.oc
    st = dict(np.load(os.path.join(ROOT, "tests", "golden", "fox_state.npz")))
    g = dict(np.load(os.path.join(ROOT, "tests", "golden", "fox_golden.npz")))

    def launches():
        out = OrderedDict()
        name = ctypes.create_string_buffer(256)
        v = (ctypes.c_ulonglong * 8)()
        for i in range(L.wemu_traffic_launches()):
            L.wemu_traffic_get(i, name, 256, v)
            k = name.value.decode().strip("()")
            acc = out.setdefault(k, [0] * 9)
            for j in range(6):
                acc[j] += v[j]
            acc[6] += v[6]; acc[7] = v[7]; acc[8] += 1
        L.wemu_traffic_reset()
        return out

    def report(title, unit, n_units, expect=""):
        print("\n== %s  (%d %ss)%s" % (title, n_units, unit, ("\n   DESIGN: " + expect) if expect else ""))
        print("   %-52s %8s %13s %13s %11s %11s" % ("kernel", "launches", "global B read", "global B wr.", "LDS B read", "LDS B wr."))
        print("   %-52s %8s %13s %13s %11s %11s" % ("", "", "/" + unit, "/" + unit, "/" + unit, "/" + unit))
        raw = [0] * 4
        for k, a in launches().items():
            cols = [a[0], a[1], a[4], a[5]]
            raw = [x + y for x, y in zip(raw, cols)]
            print("   %-52s %8d %13.1f %13.1f %11.1f %11.1f" % (k[:52], a[8], *[c / n_units for c in cols]))
        print("   %-52s %8s %13.1f %13.1f %11.1f %11.1f" % ("all launches of the call", "", *[c / n_units for c in raw]))
        return raw

    rng = np.random.default_rng(0)
    # ---- the hash gather: BASELINE config 2's table (2^19 x 16, the fox scene's 372 warps), marched fox samples ----
    grid = gp.make_grid(st, rng, 19, scale=0.3)
    gd = gp.grid_dev(grid)
    totals = {}
    for reps in (4, 8):  # (>= 16384 samples: the launcher stages the hash constants of a level pair in LDS, as at the benched size)
        pts = np.tile(g["march_pts"], (reps, 1)); anchors = np.tile(g["march_anchors"], (reps, 1))
        n = len(pts)
        planes = torch.zeros((8, n, 4), dtype=torch.float16)
        L.wemu_traffic_reset()
        capi.hash_gather_planes(n, grid.n_volumes, gd["table_h"], gd["prim"], gd["lidx"], gd["lsize"], gd["bias"], gd["scale"], T(pts), True, T(anchors), 3, planes)
        totals[n] = report("f2n_hash_gather_planes (XCD-partitioned, staged constants), fineness-16 march of the fox golden x%d" % reps, "sample", n,
                           "16 levels x 8 corners x 4 B = 512 table + 12 point + 4 warp index read, 64 B of f16 planes written = 592 B")
    (n1, a), (n2, b) = sorted(totals.items())
    stage = 2 * int(grid.n_volumes) * 6 * 4
    print("   -> global read per sample = 512 (table: 16 levels x 8 corners x 4 B) + 8 level-pair passes x (12 point + 4 warp index) = 640 B, plus the "
          "staging of a level\n      pair's hash constants into LDS, %d B per (workgroup, level pair): %.1f B per sample HERE, where a workgroup serves ONE "
          "256-sample tile per pair\n      (the grid is min(tiles, 256) workgroups per XCD); at the benched 7.9e5 samples a workgroup serves 12 tiles per "
          "pair: %.1f B per sample.\n      So the source reads 640 + ~%d B per sample where SURVEY 8(d) prices 528: the point and its warp index once per PASS, "
          "not once (+112 B), and the staged\n      constants -- coalesced reads that hit L2 (the counter passes' 0.54 x algorithmic HBM traffic contains them)."
          % (stage, 8 * stage / 256.0, 8 * stage / (12 * 256.0), round(8 * stage / (12 * 256.0))))
    assert abs(a[0] / n1 - (640 + 8 * stage / 256.0)) < 12 and abs(a[1] / n1 - 64) < 0.5, (a[0] / n1, a[1] / n1)
    # ---- the hash-gradient scatter: owner-binned path ----
    nb = 40000 + 37
    n_rays = nb // 50 + 1
    o = rng.random((n_rays, 3), dtype=F32) * F32(.6) + F32(.2)
    dvec = rng.standard_normal((n_rays, 3)).astype(F32); dvec /= np.linalg.norm(dvec, axis=1, keepdims=True)
    ray = np.repeat(np.arange(n_rays), 50)[:nb]
    q = np.clip(o[ray] + dvec[ray] * ((np.arange(nb) % 50).astype(F32) * F32(0.004))[:, None], 0.01, 0.99).astype(F32)
    vol = rng.integers(0, grid.n_volumes, n_rays).astype(np.int32)[ray]
    gin = (rng.standard_normal((nb, 32)) * 0.05).astype(np.float16)
    gtab = torch.zeros(grid.table_f32.size, dtype=torch.float16)
    L.wemu_traffic_reset()
    capi.hash_bwd(nb, grid.n_volumes, gd["prim"], gd["lidx"], gd["lsize"], gd["bias"], gd["scale"], T(q), False, T(vol), 1, T(gin), gtab, 1 << 19)
    report("f2n_hash_bwd, owner-binned (dense gradients, 50-sample rays at 0.004 spacing)", "sample", nb,
           "64 B of f16 gradient read; up to 16 levels x 8 corners x 8-byte records written by the producer and read by the owner (less what run "
           "combining removes)")
    print("   (NOT complete for the producer's stores: its records are 8-byte aggregates, which -O0 copies with memcpy -- the instrumentation sees scalar "
          "and vector\n    accesses only.  The owner's record reads are counted: that many bytes of records exist; both kernels' LDS traffic is counted)")
    # ---- Adam over the table ----
    n_tab = 17 << 15
    tp, tm, tv = torch.randn(n_tab), torch.zeros(n_tab), torch.zeros(n_tab)
    tg, th = torch.randn(n_tab).to(torch.float16), torch.zeros(n_tab, dtype=torch.float16)
    L.wemu_traffic_reset()
    capi.adam_fused([], dict(n=n_tab, param=tp, grad_h=tg, grad_scale=1.0 / 128, exp_avg=tm, exp_avg_sq=tv, param_h=th), 3, 1e-2, 0.9, 0.99, 1e-15, True)
    report("f2n_adam_fused, table group alone", "entry", n_tab, "param 4 + 4, moments 8 + 8, f16 gradient 2 read + 2 cleared, f16 copy 2 written = 30 B "
           "(x 17 * 2^19 entries = 267 MB per step)")
    return 0


if __name__ == "__main__":
    sys.exit(main())
