#!/usr/bin/env python3
"""TEST / MEASUREMENT INFRASTRUCTURE ONLY -- CPU baseline for bench.py's `cpu_baseline` object.

Times the oracle's restatement of one full training iteration of the hot path (BASELINE config 1: ngp_fox,
wanjinyou.yaml, 256 rays/batch: sampler -> no-grad field pre-pass -> early stop -> field + shader + compositing ->
losses -> backward -> dense Adam) on the host cores, with the OpenMP build of oracle/f2n_oracle.c.
Runs as its own process (F2N_ORACLE_OMP=1, no torch import: two OpenMP runtimes in one process crash).
Prints ONE JSON line.  `kind` is "port": the reference has no CPU path, this is the faithful
stand-in, a reported baseline and not a target.
"""
import json
import os
import sys
import time

os.environ["F2N_ORACLE_OMP"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

from oracle import capi as oc  # noqa: E402
from oracle import pipeline as op  # noqa: E402

F32 = np.float32


def xavier(rng, n_hidden):
    parts = []
    for rows, cols in [(64, 32)] + [(64, 64)] * (n_hidden - 1) + [(16, 64)]:
        s = np.sqrt(6.0 / (rows + cols))
        parts.append(rng.uniform(-s, s, rows * cols).astype(F32))
    return np.concatenate(parts)


def main(n_rays=256, log2_table=19, n_edge=8192, budget_s=12.0, max_iters=6, fineness=16.0):
    st = dict(np.load(os.path.join(ROOT, "tests", "golden", "fox_state.npz")))
    rng = np.random.default_rng(2022)
    nvol = int(st["n_volumes"])
    pool = (1 << log2_table) * 16
    table = ((rng.random((pool, 2), dtype=F32) * F32(.2) - F32(1.)) * F32(1e-4))
    grid = op.HashGrid(table, st["prim_pool"], st["bias_pool"], nvol, log2_table)
    p_field, p_color = xavier(rng, 1), xavier(rng, 2)
    app_emb = (rng.standard_normal((len(st["poses"]), 16)) * 0.1).astype(F32)
    n_edges = st["edge_pool"].size // 64
    adam = {k: [np.zeros_like(v), np.zeros_like(v)] for k, v in
            (("table", grid.table_f32.reshape(-1)), ("field", p_field), ("color", p_color), ("emb", app_emb.reshape(-1)))}

    def iteration(step):
        cam = st["train_set"][rng.integers(0, len(st["train_set"]), n_rays)].astype(np.int32)
        pose, K = st["poses"][cam], st["intri"][cam]
        i = rng.integers(0, 960, n_rays).astype(F32) + F32(.5)
        j = rng.integers(0, 540, n_rays).astype(F32) + F32(.5)
        d_cam = np.stack([(j - K[:, 0, 2]) / K[:, 0, 0], -(i - K[:, 1, 2]) / K[:, 1, 1], -np.ones(n_rays, F32)], -1).astype(F32)
        d = np.einsum("nij,nj->ni", pose[:, :3, :3], d_cam).astype(F32)
        d = (d / np.linalg.norm(d, axis=-1, keepdims=True)).astype(F32)
        o = np.ascontiguousarray(pose[:, :3, 3]).astype(F32)
        gt = rng.random((n_rays, 3), dtype=F32)
        noise = (((rng.random(1024 + n_rays + 10, dtype=F32) - F32(.5)) + F32(1.)) * F32(fineness)).astype(F32)
        bg = rng.random((n_rays, 3), dtype=F32)
        hits = oc.oct_intersect(st["search_order"], o, d, 0.01, 1e8, st["tree_nodes"])
        smp = oc.ray_march(o, d, noise, 1. / 256., True, *hits, st["tree_nodes"], st["pers_trans"])
        feat_all = op.field_fwd(grid, p_field, smp["pts"], smp["anchors"][:, 0])
        _, _, mask, new_se = op.early_stop(feat_all[:, 0], smp["dt"], smp["pts_idx_bounds"])
        pts, dirs, dt, t, anchors = op.compact(mask, smp["pts"], smp["dirs"], smp["dt"], smp["t"], smp["anchors"])
        m = len(pts)
        eidx = rng.integers(0, n_edges, n_edge).astype(np.int32)
        ecoord = (rng.random((n_edge, 2), dtype=F32) * F32(2.) - F32(1.)).astype(F32)
        e_pts, e_idx = oc.edge_samples(st["edge_pool"], st["pers_trans"], eidx, ecoord)
        q_pts = np.concatenate([pts, e_pts.reshape(-1, 3)], 0)
        q_vol = np.concatenate([anchors[:, 0], e_idx.reshape(-1)], 0).astype(np.int32)
        feat, fctx = op.field_fwd(grid, p_field, q_pts, q_vol, want_ctx=True)
        sidx = oc.scatter_idx(m, new_se, cam)
        rgb, sctx = op.shade_fwd(p_color, feat[:m], dirs, app_emb, sidx, want_ctx=True)
        comp = op.composite_fwd(feat[:m], dt, t, rgb, bg, new_se, want_ctx=True)
        lg = op.losses_and_grads(comp["colors"], gt, comp["disparity"], comp["weights"], new_se,
                                 feat[m:].reshape(n_edge, 2, 16), 0.0, 0.0, 0.1)
        drgb, df0 = op.composite_bwd(comp["ctx"], dt, rgb, bg, new_se, lg["dcolors"], lg["ddisparity"], None, None, 0.0)
        dp_color, dfeat_sh, demb = op.shade_bwd(p_color, sctx, drgb, len(app_emb), sidx)
        dfeat = np.zeros_like(feat)
        dfeat[:m] = dfeat_sh
        dfeat[:m, 0] += df0
        dfeat[m:] = lg["dedge"].reshape(-1, 16)
        dp_field, gtab, _ = op.field_bwd(grid, p_field, fctx, dfeat, 128.0, fp32_accumulate=True)
        for key, p, g, wd in (("table", grid.table_f32.reshape(-1), gtab.reshape(-1), 0.0), ("field", p_field, dp_field, 1e-6),
                              ("color", p_color, dp_color, 1e-6), ("emb", app_emb.reshape(-1), demb.reshape(-1), 1e-6)):
            newp, adam[key][0], adam[key][1] = op.adam_step(p, g, adam[key][0], adam[key][1], step, 1e-2, 0.9, 0.99, 1e-15, wd)
            p[...] = newp
        return len(smp["t"]), m

    iteration(1)  # warm-up
    t0 = time.time()
    n_tot = m_tot = its = 0
    while its < max_iters and (time.time() - t0) < budget_s:
        n, m = iteration(its + 2)
        n_tot += n
        m_tot += m
        its += 1
    el = time.time() - t0
    print(json.dumps({"value": m_tot / el, "unit": "ray-samples/s", "cores": oc.num_threads(), "kind": "port",
                      "sample": "%d training iterations of %d rays (ngp_fox wanjinyou, log2_table %d, fineness %g): "
                                "%d samples marched, %d meaningful, %.1f s; OpenMP port of the reference path, most of its time in the "
                                "reference-faithful DENSE per-iteration table passes (fp32->fp16 cast of 2^%d x16 entries, gradient "
                                "zero-fill / widen / divide, dense Adam), not in per-sample work" % (its, n_rays, log2_table, fineness, n_tot, m_tot, el, log2_table),
                      "rays_per_s": its * n_rays / el, "marched_samples_per_s": n_tot / el}))


if __name__ == "__main__":
    main()
