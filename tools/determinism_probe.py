#!/usr/bin/env python3
"""Where do two trainings from one seed part?  Trains the fox scene twice (ExpRunner::Train, one iteration per call) and
records, after EVERY iteration, order-free integer checksums of each parameter tensor (hash table, field MLP, colour MLP,
appearance embedding), the node count, the ray count of the next batch and the running sample counters; prints the first
iteration at which the two runs differ and in which quantity.  Measurement aid (round-3 verdict: "training is not
reproducible run to run")."""
import argparse, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=150)
ap.add_argument("--runs", type=int, default=2)
ap.add_argument("--factor", type=int, default=2)
ap.add_argument("--speculation", type=int, default=-1)
ap.add_argument("--stride", type=int, default=1, help="iterations per Train call (checksums every `stride` iterations)")
ap.add_argument("--dense-from", type=int, default=0, help="checksums after EVERY iteration in [dense-from, dense-to]")
ap.add_argument("--dense-to", type=int, default=0)
ap.add_argument("--poison", type=int, default=0, help="between runs: render the test views (1) and leave 0xFF-filled blocks of "
                "every size class in the caching allocator (2: both) -- a run that reads memory it never wrote then parts from run 0")
ap.add_argument("--set", nargs="*", default=[], help="runner properties, name=int (speculation_depth=1 tail_repair=0 march_blocks=0 ...)")
ap.add_argument("--side-delay", nargs="*", default=[], help="one begin_us:complete_us:main_us:period per run AFTER run 0: "
                "run 0 trains undisturbed, the others with their streams skewed -- they must not part")
ap.add_argument("--pollute", action="store_true", help="run r > 0: garbage (different per run) left in the LDS and registers of every CU in front "
                "of every step and every speculative begin / completion (f2n_debug_pollute): a kernel that reads state it never wrote parts")
ap.add_argument("--overrides", nargs="*", default=[])
ap.add_argument("--taps", action="store_true", help="with --digest: checksums of every step's intermediate arrays too (ExpRunner.step_taps): "
                "the first ARRAY in which two runs part")
ap.add_argument("--save-taps", default="", help="file prefix: every run's tap matrix is saved (comparison across processes)")
ap.add_argument("--digest", action="store_true", help="per-STEP digest (ExpRunner.step_digest: seq, iter, rays, marched, kept, table checksum "
                "after the step) of every run, compared step by step: names the first step and quantity in which two runs part")
args = ap.parse_args()
if args.side_delay or args.pollute:  # the stream-skew / pollution hooks exist in the debug variant of the libraries only
    os.environ["F2N_DEBUG_BUILD"] = "1"
import f2_nerf_amd  # noqa: F401
from f2_nerf_amd import runtime, fox_data, capi
NAMES = ["table", "field_mlp", "color_mlp", "app_emb", "nodes", "n_nodes", "batch", "marched", "meaningful", "speculative", "fallback", "dropped", "rays_repaired"]


SCHEDULE = sorted(set(list(range(args.stride, args.iters + 1, args.stride)) +
                      (list(range(args.dense_from, min(args.dense_to, args.iters) + 1)) if args.dense_to > 0 else [])))


def csum(t):
    return int(t.detach().contiguous().view(torch.int32).to(torch.int64).sum().item())


def one_run():
    st = fox_data.load_state()
    sc, images = fox_data.scene(args.factor)
    ds = runtime.make_dataset(sc, images)
    runner, cfg, _ = runtime.make_runner(st, "wanjinyou", ["train.end_iter=20000"] + args.overrides, seed=2022)
    if args.speculation >= 0:
        runner.speculative_sampling = args.speculation
    for kv in args.set:
        k, v = kv.split("=")
        setattr(runner, k, int(v))
    torch.manual_seed(2022)
    runner.digest_table = bool(args.digest)
    runner.digest_taps = bool(args.taps)
    rows = []
    for it in SCHEDULE:
        runner.train(ds, it, 1)
        s = runner.states()  # [0] nodes [4] table [8] field MLP [9] colour MLP [10] app_emb
        c = runner.counters()
        sp = runner.speculation_counters()
        rows.append([csum(s[4]), csum(s[8]), csum(s[9]), csum(s[-1]), csum(s[0].view(torch.int32)) if s[0].numel() % 4 == 0 else 0,
                     runner.n_nodes(), runner.cur_batch_size(), c["total_marched"], c["total_meaningful"]] +
                    [int(sp[k]) for k in ("speculative", "fallback", "dropped", "rays_repaired")])
    dbg = capi.debug_counters() if hasattr(capi, "debug_counters") else None
    if args.digest:
        DIGESTS.append([tuple(int(v) for v in row) for row in runner.step_digest()])
        if args.taps:
            TAPS.append(runner.step_taps().numpy().copy())
            if args.save_taps:
                np.save("%s_run%d.npy" % (args.save_taps, len(TAPS) - 1), TAPS[-1])
    if args.poison >= 1:
        runner.test_images(ds)
    del runner
    if args.poison >= 2:
        junk = []
        for lg in range(9, 31):  # 512 B ... 1 GiB
            for _ in range(6 if lg < 24 else 2):
                junk.append(torch.full((1 << lg,), -1, dtype=torch.int8, device="cuda"))
        torch.cuda.synchronize()
        del junk
    return rows, dbg


runs = []
DIGESTS = []
TAPS = []
TAP_NAMES = ["seq", "pts", "dt", "anchors", "f0", "survivors", "bg", "edge", "colors", "table_grad", "small_grads", "table", "field_mlp", "color_mlp", "app_emb", "grad_before", "pts_all_after", "vol_all_after", "field_x", "dfeat",
             "pts_pre", "dt_pre", "anchors_pre", "pts_all_pre", "vol_all_pre"]
for r in range(args.runs):
    if args.side_delay or args.pollute:
        b, c, m, per = [int(v) for v in args.side_delay[(r - 1) % len(args.side_delay)].split(":")] if (r > 0 and args.side_delay) else (0, 0, 0, 1)
        runtime.host().ExpRunner.debug_side_delay(b, c, m, max(per, 1), (7919 * r) if args.pollute else 0)
    rows, dbg = one_run()
    runs.append(rows)
    print("run %d: last row %s  debug counters %s" % (r, rows[-1], dbg), flush=True)
ref = runs[0]
for r in range(1, args.runs):
    first = None
    for i, (a, b) in enumerate(zip(ref, runs[r])):
        if a[:-1] != b[:-1]:  # (rays_repaired: how many rays a repair touched depends on what the side stream's walk happened to see)
            first = i
            break
    if first is None:
        print("run %d == run 0 over %d iterations (every checksum)" % (r, args.iters))
    else:
        diff = [NAMES[k] for k in range(len(NAMES)) if ref[first][k] != runs[r][first][k]]
        print("run %d parts from run 0 at iteration %d in: %s" % (r, SCHEDULE[first], ", ".join(diff)))
        for i in range(max(0, first - 1), min(len(ref), first + 3)):
            print("   it %4d  run0 %s\n            run%d %s" % (SCHEDULE[i], ref[i], r, runs[r][i]))

if args.digest:
    DNAMES = ["seq", "iter", "rays", "marched", "kept", "table"]
    for r in range(1, len(DIGESTS)):
        a, b = DIGESTS[0], DIGESTS[r]
        first = next((i for i, (x, y) in enumerate(zip(a, b)) if x != y), None)
        if first is None and len(a) == len(b):
            print("digest: run %d == run 0 in every one of %d steps" % (r, len(a)))
        elif first is None:
            print("digest: run %d has %d steps, run 0 %d (equal where both exist)" % (r, len(b), len(a)))
        else:
            diff = [DNAMES[k] for k in range(len(a[first])) if a[first][k] != b[first][k]]
            print("digest: run %d parts from run 0 at step seq %d (iteration %d) in: %s" % (r, a[first][0], a[first][1], ", ".join(diff)))
            for i in range(max(0, first - 2), min(len(a), first + 3)):
                print("   run0 %s\n   run%d %s" % (a[i], r, b[i]))

if args.taps:
    for r in range(1, len(TAPS)):
        a, b = TAPS[0], TAPS[r]
        k = min(len(a), len(b))
        bad = np.nonzero((a[:k] != b[:k]).any(1))[0]
        if len(bad) == 0:
            print("taps: run %d == run 0 in every tap of %d steps" % (r, k))
        else:
            i = int(bad[0])
            cols = [TAP_NAMES[c] for c in range(a.shape[1]) if a[i, c] != b[i, c]]
            print("taps: run %d parts from run 0 at step seq %d, first in (pipeline order): %s" % (r, int(a[i, 0]), ", ".join(cols)))
            for j in range(max(0, i - 1), min(k, i + 2)):
                print("   run0 %s\n   run%d %s" % (a[j].tolist(), r, b[j].tolist()))

if args.taps:
    # inside ONE run: an array whose checksum in front of its consumer differs from the one behind it changed in between
    PAIRS = [("pts_pre", "pts"), ("dt_pre", "dt"), ("anchors_pre", "anchors"), ("pts_all_pre", "pts_all_after"), ("vol_all_pre", "vol_all_after")]
    for r, a in enumerate(TAPS):
        for pre, post in PAIRS:
            i, j = TAP_NAMES.index(pre), TAP_NAMES.index(post)
            bad = np.nonzero(a[:, i] != a[:, j])[0]
            print("pre/post: run %d  %-12s vs %-14s  steps where the array changed around its consumer: %d%s" %
                  (r, pre, post, len(bad), ("  first at seq %d" % int(a[bad[0], 0])) if len(bad) else ""))
