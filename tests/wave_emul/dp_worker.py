"""Test infrastructure.  One rank of a data-parallel training on the emulated stack (tests/test_wave_emul_cpu.py starts N of these as
processes): the emulated host module, an ExpRunner of its own -- built from a seed of its own, so that the attach has something to
replicate --, DataParallel::Attach through the shared-memory <rccl/rccl.h> of host_shim/, then train steps on its own ray batches with
the two-deep sampling pipeline, as bench.py --gpus N runs them.  Prints one JSON line: checksums of the replica, counters."""
import ctypes
import importlib.util
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "wave_emul"))


def main():
    rank, world, uid_hex, steps, rays, overlap = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])
    import numpy as np
    import torch
    import wemu_build
    ctypes.CDLL(wemu_build.LIB, mode=ctypes.RTLD_GLOBAL)
    spec = importlib.util.spec_from_file_location("_f2n_host_emul", wemu_build.build_host())
    host = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(host)
    import f2_nerf_amd  # noqa: F401
    from f2_nerf_amd import runtime
    import test_gpu_e2e as e2e
    runtime._host = host
    torch.cuda.is_available = lambda: True
    torch.cuda.set_device = lambda *a, **k: None
    torch.set_num_threads(1)
    st = dict(np.load(os.path.join(ROOT, "tests", "golden", "fox_state.npz")))
    overrides = ["field.log2_table_size=12", "pts_sampler.compact_freq=4", "pts_sampler.sub_div_milestones=[5]"]
    runner, cfg, _ = runtime.make_runner(st, "wanjinyou", overrides, seed=5 + 10 * rank, table_init=0.3)  # (replicas start DIFFERENT)
    runner.n_edge_pts = 64
    before = int(runner.states()[4].contiguous().view(torch.int32).to(torch.int64).sum())
    if world > 1 or overlap >= 0:
        runner.attach_data_parallel(rank, world, bytes.fromhex(uid_hex), bool(overlap) if overlap >= 0 else world > 1, False)
    rng = np.random.default_rng(1000 + rank)
    batches = []
    for _ in range(steps + 2):
        ro, rd, bounds, cam = e2e.fox_batch(st, rng, rays)
        batches.append([torch.from_numpy(np.ascontiguousarray(a)) for a in (ro, rd, bounds, rng.random((rays, 3), dtype=np.float32), cam)])
    if world > 1 and rank == 1 and steps >= 5:
        # one of THIS rank's batches misses the scene altogether (no sample, no backward, no step-tail call): its collectives must still pair
        # with the other ranks' -- the small buffers first, then the table's buckets, from GradSyncBegin (round 6)
        batches[3][0] = torch.full_like(batches[3][0], 100.0)
        batches[3][1] = torch.tensor([[1.0, 0.0, 0.0]]).repeat(rays, 1)
    torch.manual_seed(2022)
    losses = []
    empty_steps = 0
    for i in range(steps):
        b, nb, nb2 = batches[i], batches[i + 1], batches[i + 2]
        if runner.speculation_depth >= 2:
            s = runner.train_step(b[0], b[1], b[2], b[3], b[4], True, nb[0], nb[1], nb[2], nb2[0], nb2[1])
        else:
            s = runner.train_step(b[0], b[1], b[2], b[3], b[4], True, nb[0], nb[1], nb[2])
        losses.append(float(s["loss"]))
        empty_steps += int(s["n_samples"] == 0)
    runner.flush()
    stt = runner.states()
    csum = lambda t: int(t.contiguous().view(torch.int32).to(torch.int64).sum())  # noqa: E731
    c = runner.counters()
    print(json.dumps(dict(rank=rank, table_before_attach=before, table=csum(stt[4]), field_mlp=csum(stt[8]), color_mlp=csum(stt[9]), app_emb=csum(stt[10]),
                          nodes=int(stt[0].to(torch.int64).sum()), n_nodes=runner.n_nodes(), comm_ranks=int(runner.dp_comm_ranks()), losses=losses,
                          meaningful=c["total_meaningful"], marched=c["total_marched"], spec=dict(runner.speculation_counters()),
                          meaningful_per_ray=float(runner.meaningful_per_ray), empty_steps=empty_steps)), flush=True)


if __name__ == "__main__":
    main()
