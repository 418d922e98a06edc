#!/usr/bin/env python3
"""Where do two trainings from one seed part?  Trains the fox scene twice (ExpRunner::Train, one iteration per call) and
records, after EVERY iteration, order-free integer checksums of each parameter tensor (hash table, field MLP, colour MLP,
appearance embedding), the node count, the ray count of the next batch and the running sample counters; prints the first
iteration at which the two runs differ and in which quantity.  Measurement aid (round-3 verdict: "training is not
reproducible run to run")."""
import argparse, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import f2_nerf_amd  # noqa: F401
from f2_nerf_amd import runtime, fox_data, capi

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=150)
ap.add_argument("--runs", type=int, default=2)
ap.add_argument("--factor", type=int, default=2)
ap.add_argument("--speculation", type=int, default=-1)
ap.add_argument("--stride", type=int, default=1, help="iterations per Train call (checksums every `stride` iterations)")
ap.add_argument("--overrides", nargs="*", default=[])
args = ap.parse_args()
NAMES = ["table", "field_mlp", "color_mlp", "app_emb", "nodes", "n_nodes", "batch", "marched", "meaningful"]


def csum(t):
    return int(t.detach().contiguous().view(torch.int32).to(torch.int64).sum().item())


def one_run():
    st = fox_data.load_state()
    sc, images = fox_data.scene(args.factor)
    ds = runtime.make_dataset(sc, images)
    runner, cfg, _ = runtime.make_runner(st, "wanjinyou", ["train.end_iter=20000"] + args.overrides, seed=2022)
    if args.speculation >= 0:
        runner.speculative_sampling = args.speculation
    torch.manual_seed(2022)
    rows = []
    for it in range(args.stride, args.iters + 1, args.stride):
        runner.train(ds, it, 1)
        s = runner.states()  # [0] nodes [4] table [8] field MLP [9] colour MLP [10] app_emb
        c = runner.counters()
        rows.append([csum(s[4]), csum(s[8]), csum(s[9]), csum(s[-1]), csum(s[0].view(torch.int32)) if s[0].numel() % 4 == 0 else 0,
                     runner.n_nodes(), runner.cur_batch_size(), c["total_marched"], c["total_meaningful"]])
    dbg = capi.debug_counters() if hasattr(capi, "debug_counters") else None
    return rows, dbg


runs = []
for r in range(args.runs):
    rows, dbg = one_run()
    runs.append(rows)
    print("run %d: last row %s  debug counters %s" % (r, rows[-1], dbg), flush=True)
ref = runs[0]
for r in range(1, args.runs):
    first = None
    for i, (a, b) in enumerate(zip(ref, runs[r])):
        if a != b:
            first = i
            break
    if first is None:
        print("run %d == run 0 over %d iterations (every checksum)" % (r, args.iters))
    else:
        diff = [NAMES[k] for k in range(len(NAMES)) if ref[first][k] != runs[r][first][k]]
        print("run %d parts from run 0 at iteration %d in: %s" % (r, (first + 1) * args.stride, ", ".join(diff)))
        for i in range(max(0, first - 1), min(len(ref), first + 3)):
            print("   it %4d  run0 %s\n            run%d %s" % ((i + 1) * args.stride, ref[i], r, runs[r][i]))
