#!/usr/bin/env python3
"""Trains the fox scene on REAL pixels (tests/golden/fox_images_f8.npz: the reference's example photographs at 240x135)
with the loop of ExpRunner::Train (ExpRunner.cpp:82-143): adaptive ray batch (pts_batch_size / meaningful samples per
ray), Dataset::RandRaysData on the device, fused train step, octree milestones.  Reports train throughput and the test-set
PSNR (every 8th image, Dataset.cpp:105-109).  GPU box only; measurement aid + convergence evidence, not part of bench.py."""
import argparse, json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import f2_nerf_amd  # noqa: F401
from f2_nerf_amd import runtime

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=20000)
ap.add_argument("--preset", default="wanjinyou")
ap.add_argument("--log-every", type=int, default=2000)
ap.add_argument("--dump-sampler", default="", help="save the final octree + one ray batch (input of tools/sampler_bench.py)")
ap.add_argument("--breakdown", action="store_true", help="per-kernel HIP-event breakdown of 20 more steps in the final state")
args = ap.parse_args()

st = dict(np.load(os.path.join(ROOT, "tests", "golden", "fox_state.npz")))
fx = np.load(os.path.join(ROOT, "tests", "golden", "fox_images_f8.npz"))
images = torch.from_numpy(fx["images"].astype(np.float32) / np.float32(255.))
f = float(fx["factor_vs_state"])
small = dict(st)
small["image_hw"] = np.array(images.shape[1:3])
small["intri"] = st["intri"].copy(); small["intri"][:, :2, :] /= f
ds = runtime.make_dataset(small, images)
runner, cfg, _ = runtime.make_runner(st, args.preset, ["train.end_iter=%d" % args.iters], seed=2022)
torch.manual_seed(2022)
test_set = [int(v) for v in st["test_set"]]

def test_psnr_views():
    return [float(runner.test_image_psnr(ds, i)) for i in test_set]

log = []
torch.cuda.synchronize(); t0 = time.perf_counter(); n_mean = 0; n_rays = 0; t_last = t0
it = 0
while it < args.iters:  # ExpRunner::Train in windows (the C++ loop draws rays on the device and prefetches their sampling)
    target = min(args.iters, it + args.log_every)
    s = runner.train(ds, target, 1)
    it = runner.iter_step
    torch.cuda.synchronize(); now = time.perf_counter()
    n_mean += s["total_meaningful"]; n_rays += s["total_rays"]
    mse = float(s["mse"])
    rec = {"iter": it, "train_psnr_batch": round(10 * np.log10(1 / max(mse, 1e-12)), 2), "rays_per_batch": s["n_rays"],
           "meaningful_per_ray": round(runner.meaningful_per_ray, 1), "n_nodes": runner.n_nodes(),
           "samples_per_s_window": round(s["total_meaningful"] / (now - t_last)), "elapsed_s": round(now - t0, 1)}
    log.append(rec); print(json.dumps(rec), flush=True)
    t_last = now
torch.cuda.synchronize(); wall = time.perf_counter() - t0
views = test_psnr_views()
views2 = test_psnr_views()  # rendered twice: evaluation is deterministic, so the two passes must agree
print(json.dumps({"test_psnr_per_view": [round(v, 2) for v in views], "second_pass_max_abs_diff": round(max(abs(a - b) for a, b in zip(views, views2)), 4)}), flush=True)
print(json.dumps({"iters": args.iters, "train_wall_s": round(wall, 1), "ray_samples_per_s": round(n_mean / wall),
                  "rays_per_s": round(n_rays / wall), "test_psnr": round(float(np.mean(views)), 3), "test_views": test_set,
                  "image_hw": [int(v) for v in images.shape[1:3]], "data": "ngp_fox photographs at 1/8 resolution"}), flush=True)

if args.breakdown:
    host = runtime.host()
    host.ExpRunner.enable_kernel_timing(["*"])
    nm = na = 0
    torch.cuda.synchronize(); t1 = time.perf_counter()
    for it in range(20):
        b = max(16, runner.cur_batch_size())
        ro, rd, bounds, gt, cam = ds.rand_rays_data(b, 1)
        s = runner.train_step(ro, rd, bounds, gt, cam, True)
        nm += s["n_meaningful"]; na += s["n_samples"]
    torch.cuda.synchronize(); dt = (time.perf_counter() - t1) / 20
    t = host.ExpRunner.collect_kernel_timing(); host.ExpRunner.disable_kernel_timing()
    tot = sum(v[1] for v in t.values())
    print("leaf hits per ray %.1f, marched samples per ray %.1f, meaningful per ray %.1f, octree nodes %d" %
          (runner.oct_per_ray, runner.sampled_per_ray, runner.meaningful_per_ray, runner.n_nodes()))
    smp = runner.get_samples(ro, rd, bounds)
    per_ray = (smp["pts_idx_bounds"][:, 1] - smp["pts_idx_bounds"][:, 0]).cpu().numpy()
    print("marched samples per ray: mean %.1f  p50 %d  p90 %d  p99 %d  p99.9 %d  max %d  (rays with 0: %.1f%%)" %
          (per_ray.mean(), *np.percentile(per_ray, [50, 90, 99, 99.9]).astype(int), per_ray.max(), 100 * (per_ray == 0).mean()))
    print("final state: %d rays/step, %.0f marched, %.0f meaningful samples/step, %.3f ms/step (with event timing)" % (b, na / 20, nm / 20, dt * 1e3))
    for k, v in sorted(t.items(), key=lambda kv: -kv[1][1]):
        print("  %-22s launches %4.1f  %8.3f ms/step  %5.1f%%" % (k, v[0] / 20, v[1] / 20, 100 * v[1] / tot))
    print("  sum of timed calls: %.3f ms/step" % (tot / 20))

if args.dump_sampler:
    b = max(16, runner.cur_batch_size())
    ro, rd, bounds, gt, cam = ds.rand_rays_data(b, 1)
    np.savez_compressed(args.dump_sampler, tree_nodes=runner.tree_nodes().cpu().numpy(), pers_trans=st["pers_trans"],
                        search_order=st["search_order"], rays_o=ro.cpu().numpy(), rays_d=rd.cpu().numpy(),
                        fineness=np.float32(runner.fineness))
    print("dumped", args.dump_sampler)
