#!/usr/bin/env python3
"""Brings the fox scene to its converged state (ExpRunner::Train for --iters iterations on the committed photographs) and
then runs --steps more training steps on pre-drawn batches after a 0.3 s pause -- the pause is the marker by which
profiles/timeline_rocpd.py finds those steps inside a rocprofv3 --kernel-trace of this script.  Measurement aid."""
import argparse, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import f2_nerf_amd  # noqa: F401
from f2_nerf_amd import runtime, fox_data

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=20000)
ap.add_argument("--steps", type=int, default=12)
ap.add_argument("--factor", type=int, default=2)
ap.add_argument("--overrides", nargs="*", default=[])
ap.add_argument("--kernel-timing", action="store_true")
ap.add_argument("--speculation", type=int, default=-1, help="0 off / 1 on / 2 auto (default: the host's default)")
ap.add_argument("--march-blocks-sweep", default="", help="v1,v2,...: repeat the timed steps with the speculative march on that many persistent blocks (0: classic)")
ap.add_argument("--block-waves-sweep", default="", help="v1,v2,...: repeat the timed steps with the persistent march in workgroups of that many waves")
ap.add_argument("--fused-tail-sweep", action="store_true", help="repeat the timed steps with the step's tail fused into the field backward's call / as separate launches")
ap.add_argument("--restore", action="store_true", help="sweeps: restart every phase from the state the training left (same batches, same work)")
ap.add_argument("--native", action="store_true", help="time ExpRunner::Train's own loop (fresh batches drawn on the device every iteration) instead of python-driven steps on resident batches")
ap.add_argument("--depth", type=int, default=-1, help="sampling pipeline depth of the timed steps (1 / 2; default: the host's)")
ap.add_argument("--draws-on-main", action="store_true", help="ExpRunner::Train's batch draws on the main queue (rounds 4-5) instead of the tail stream")
ap.add_argument("--spec-start-event", action="store_true", help="keep the spec_start event at the top of a step although the draws are off the main queue")
ap.add_argument("--env-sweep", default="", help="NAME=v1,v2,...: repeat the timed steps once per value of an environment knob (knobs exist in the debug variant only: F2N_DEBUG_BUILD=1; most are read once per process)")
args = ap.parse_args()
st = fox_data.load_state()
sc, images = fox_data.scene(args.factor)
ds = runtime.make_dataset(sc, images)
runner, cfg, _ = runtime.make_runner(st, "wanjinyou", ["train.end_iter=%d" % max(args.iters, 1)] + args.overrides, seed=2022)
torch.manual_seed(2022)
if args.iters > 0:
    torch.cuda.synchronize(); _t0 = time.perf_counter()
    runner.train(ds, args.iters, 1)
    torch.cuda.synchronize(); print("trained %d iterations in %.2f s" % (args.iters, time.perf_counter() - _t0), flush=True)
if args.draws_on_main:
    runner.draws_off_main = False
if args.spec_start_event:
    runner.spec_start_without_event = False
if args.speculation >= 0:
    runner.speculative_sampling = args.speculation
R = max(16, runner.cur_batch_size())
batches = [ds.rand_rays_data(R, 1) for _ in range(8)]
if args.depth >= 1:
    runner.speculation_depth = args.depth
def step(i):
    b, nb, nb2 = batches[i % 8], batches[(i + 1) % 8], batches[(i + 2) % 8]
    if runner.speculation_depth >= 2:
        return runner.train_step(b[0], b[1], b[2], b[3], b[4], True, nb[0], nb[1], nb[2], nb2[0], nb2[1])
    return runner.train_step(b[0], b[1], b[2], b[3], b[4], True, nb[0], nb[1], nb[2])
_snapshot = None
def timed(tag=""):
    # every phase of a sweep starts from the SAME state (parameters, octree, optimiser position): a training that carries on between the
    # phases drifts -- fewer samples per step, a compaction -- by more than the differences a sweep is after
    global _snapshot
    if args.restore:
        if _snapshot is None:
            _snapshot = ([t.clone() for t in runner.states()], runner.iter_step)
        else:
            runner.load_states([t.clone() for t in _snapshot[0]])
            runner.iter_step = _snapshot[1]
            runner.update_ada_params()
    for i in range(6):
        step(i)
    c0 = runner.counters(); torch.cuda.synchronize(); time.sleep(0.3)
    if args.kernel_timing:
        runtime.host().ExpRunner.enable_kernel_timing(["ray_march", "oct_intersect", "field_bwd", "hash_gather", "shade_bwd", "oct_repair", "march_repair",
                                                       "field_shade_fwd", "pack_samples", "early_stop", "field_prepass_fused", "field_mlp_prepass", "composite_train"])
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(6 + i)
    runner.flush(); torch.cuda.synchronize()
    el = time.perf_counter() - t0
    c1 = runner.counters()
    print("%srays %d  %.3f ms/step  marched/step %d meaningful/step %d nodes %d" % (tag, R, el / args.steps * 1e3,
          (c1["total_marched"] - c0["total_marched"]) // args.steps, (c1["total_meaningful"] - c0["total_meaningful"]) // args.steps,
          runner.n_nodes()), flush=True)
    if args.kernel_timing:
        tm = runtime.host().ExpRunner.collect_kernel_timing()
        runtime.host().ExpRunner.disable_kernel_timing()
        print("    " + "  ".join("%s %.1f us" % (k, v[1] / max(v[0], 1) * 1e3) for k, v in sorted(tm.items())), flush=True)
if args.native:
    runner.end_iter = runner.iter_step + args.steps + 64
    runner.train(ds, runner.iter_step + 16, 1); runner.flush()
    c0 = runner.counters(); torch.cuda.synchronize(); time.sleep(0.3)
    from f2_nerf_amd import capi as _capi
    _capi.debug_counters(reset=True)
    runtime.host().ExpRunner.host_profile(True)
    t0 = time.perf_counter()
    s2 = runner.train(ds, runner.iter_step + args.steps, 1); runner.flush(); torch.cuda.synchronize()
    el = time.perf_counter() - t0
    hp = runtime.host().ExpRunner.host_profile(False)
    print("host thread, us per iteration: " + "  ".join("%s %.1f (x%.1f)" % (k, v[1] / args.steps * 1e6, v[0] / args.steps) for k, v in sorted(hp.items())))
    c1 = runner.counters()
    print("native loop: %.3f ms/step  marched/step %d meaningful/step %d nodes %d  spec %s  scatter [atomic records, fp64 slices, largest slice sum, overflow-list records] %s" % (
        el / args.steps * 1e3, (c1["total_marched"] - c0["total_marched"]) // args.steps,
        (c1["total_meaningful"] - c0["total_meaningful"]) // args.steps, runner.n_nodes(), dict(runner.speculation_counters()),
        _capi.debug_counters()[:2] + [float(np.array(_capi.debug_counters()[2], np.int32).view(np.float32)), _capi.debug_counters()[3]]), flush=True)
elif args.march_blocks_sweep:
    for rep in range(2):
        for v in args.march_blocks_sweep.split(","):
            runner.march_blocks = int(v)
            timed("march_blocks=%s: " % v)
elif args.block_waves_sweep:
    for rep in range(2):
        for v in args.block_waves_sweep.split(","):
            runner.march_block_waves = int(v)
            timed("march_block_waves=%s: " % v)
elif args.fused_tail_sweep:
    for rep in range(3):
        for v in (True, False):
            runner.fused_tail = v
            timed("fused_tail=%s: " % v)
elif args.env_sweep:
    name, vals = args.env_sweep.split("=")
    for rep in range(2):
        for v in vals.split(","):
            os.environ[name] = v
            timed("%s=%s: " % (name, v))
else:
    timed()
