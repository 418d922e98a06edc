#!/usr/bin/env python3
"""Fixed cost and per-sample cost of every main-queue kernel, separately (round-5 verdict, item 1).  Brings the fox scene to its
converged state (ExpRunner::Train, --iters iterations), then runs training steps on batches of --factors x the converged batch
size with the sampler BEHIND the stat update on the main stream (no side stream runs underneath: every kernel is alone on the GPU),
one phase per factor, each behind a 0.3 s pause.  The converged batch's ray-length distribution is kept (the rays are drawn the
same way, only more or fewer of them).  Under `rocprofv3 --kernel-trace` profiles/sweep_rocpd.py turns the trace into one table:
kernel x phase -> mean duration.  Without rocprofv3 the host's HIP-event timers (F2N_TIMED_CALL) print per C-ABI call.  Measurement aid."""
import argparse, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import f2_nerf_amd  # noqa: F401
from f2_nerf_amd import runtime, fox_data

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=20000)
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--factors", default="0.46,1,2,3")
ap.add_argument("--speculation", type=int, default=0)
ap.add_argument("--fused-tail", type=int, default=1, help="0: reduce / flags / adam_fused as separate launches behind the scatter; 1: f2n_field_bwd_step_tail")
ap.add_argument("--overrides", nargs="*", default=[])
ap.add_argument("--no-events", action="store_true", help="no HIP-event timers (for a run under rocprofv3 --kernel-trace)")
args = ap.parse_args()
st = fox_data.load_state()
sc, images = fox_data.scene(2)
ds = runtime.make_dataset(sc, images)
runner, cfg, _ = runtime.make_runner(st, "wanjinyou", ["train.end_iter=%d" % max(args.iters, 1)] + args.overrides, seed=2022)
torch.manual_seed(2022)
if args.iters > 0:
    runner.train(ds, args.iters, 1)
    torch.cuda.synchronize()
runner.speculative_sampling = args.speculation
if args.speculation == 0:
    # the STREAMING step's kernels (survivor count on the device: field_shade_fwd, composite_train, the fused tail) without a next batch to
    # prefetch -- the sampler then runs inside the step on the main stream and nothing runs underneath anything
    runner.async_counts = 2
    runner.fused_tail = args.fused_tail
R0 = max(16, runner.cur_batch_size())
H = runtime.host().ExpRunner
for f in [float(v) for v in args.factors.split(",")]:
    R = max(16, int(R0 * f))
    batches = [ds.rand_rays_data(R, 1) for _ in range(4)]
    def step(i):
        b, nb = batches[i % 4], batches[(i + 1) % 4]
        if args.speculation == 0:  # no next batch: the sampler runs inside the step, on the main stream -- every kernel alone
            return runner.train_step(b[0], b[1], b[2], b[3], b[4], True)
        return runner.train_step(b[0], b[1], b[2], b[3], b[4], True, nb[0], nb[1], nb[2])
    for i in range(4):
        step(i)
    runner.flush(); torch.cuda.synchronize()
    c0 = runner.counters(); time.sleep(0.3)
    if not args.no_events:
        H.enable_kernel_timing(["*"])
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(4 + i)
    runner.flush(); torch.cuda.synchronize()
    el = time.perf_counter() - t0
    time.sleep(0.3)  # (a second marker: the next phase's batch draws and warm-up steps are not part of this one)
    tm = H.collect_kernel_timing() if not args.no_events else {}
    H.disable_kernel_timing()
    c1 = runner.counters()
    print("PHASE factor %.2f rays %d  %.3f ms/step  marched/step %d meaningful/step %d nodes %d" % (
        f, R, el / args.steps * 1e3, (c1["total_marched"] - c0["total_marched"]) // args.steps,
        (c1["total_meaningful"] - c0["total_meaningful"]) // args.steps, runner.n_nodes()), flush=True)
    print("    " + "  ".join("%s %.1f" % (k, v[1] / max(v[0], 1) * 1e3) for k, v in sorted(tm.items())) + "  (us per call, HIP events)", flush=True)
