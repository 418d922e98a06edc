"""Test infrastructure.  A race detector for the host's STREAMS (reading (b)).  The host layer orders its main stream and
its side streams with events; on the GPU a missing wait is a race that shows once in thousands of steps under load.  Here the order the
host ASKS for is kept exactly (host_shim.h reports every record / wait / host-side synchronisation to the emulated kernel library, which
keeps a vector clock per stream) and every global access of every launch is known (the "traffic" build: the compiler's load / store
instrumentation): an access that is not ordered behind a conflicting access of another stream is reported with both kernels' names.
Runs ExpRunner::TrainStep with the two-deep sampling pipeline, leaves dying under batches sampled ahead (both repairs), a compaction and
a subdivision -- the speculative-training scenario of tests/test_wave_emul_cpu.py.  Not seen: torch operations of the host (they can
hide a race, never invent one); `.item()`-style implicit synchronisations (they can invent one: triage by hand).
    gcc -shared -fPIC -o _build/libnofree.so nofree.c;  LD_PRELOAD=_build/libnofree.so python tests/wave_emul/stream_races.py"""
import ctypes
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "wave_emul"))


def report(L, what):
    n = L.wemu_hb_races()
    L.wemu_hb_race.restype = ctypes.c_long
    print("# %s; %d launches; %d distinct (kernel, kernel, kind) conflicts without an ordering" % (what, L.wemu_traffic_launches(), n))
    a, b, k = ctypes.create_string_buffer(300), ctypes.create_string_buffer(300), ctypes.create_string_buffer(64)
    addr = ctypes.c_ulonglong()
    for i in range(n):
        cnt = L.wemu_hb_race(i, a, b, k, 300, ctypes.byref(addr))
        print("%-20s %8d words   %s   <-   %s" % (k.value.decode(), cnt, b.value.decode(), a.value.decode()))
    buf = ctypes.create_string_buffer(8000)
    L.wemu_hb_shared_words(buf, 8000)
    print("#   " + buf.value.decode())


def main():
    import numpy as np
    import torch
    import wemu_build
    host_path = wemu_build.build_host()
    lib, _ = wemu_build.build(tag="traffic", traffic=True)
    L = ctypes.CDLL(lib, mode=ctypes.RTLD_GLOBAL)  # (same soname as the plain build: the host module's dependency resolves to THIS one)
    spec = importlib.util.spec_from_file_location("_f2n_host_emul", host_path)
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
    if len(sys.argv) > 1 and sys.argv[1] == "train":  # ExpRunner::Train itself: batches drawn on the device, the loop's own prefetching
        from f2_nerf_amd import fox_data
        depth = int(sys.argv[2]) if len(sys.argv) > 2 else 2
        sc, images = fox_data.scene(8)
        ds = runtime.make_dataset(sc, images)
        runner, cfg, _ = runtime.make_runner(st, "wanjinyou", ["train.end_iter=20000", "field.log2_table_size=12", "train.pts_batch_size=4096",
                                                               "pts_sampler.compact_freq=4", "pts_sampler.sub_div_milestones=[6]"], seed=2022)
        runner.speculative_sampling, runner.speculation_depth = 1, depth
        torch.manual_seed(2022)
        L.wemu_hb_enable(1)
        runner.train(ds, 5, 1)
        runner.train(ds, 10, 1)
        runner.flush()
        report(L, "ExpRunner::Train, 10 iterations in two calls on the fox photographs (128-ray batches, compactions at 0 / 4 / 8, a subdivision at 6), "
                  "speculation depth %d; %s" % (depth, dict(runner.speculation_counters())))
        return 0
    tail, depth, R, NE, ITERS = (sys.argv[1] != "full" if len(sys.argv) > 1 else True), int(sys.argv[2]) if len(sys.argv) > 2 else 3, 64, 64, 9
    overrides = ["field.log2_table_size=12", "train.learning_rate=0.0", "pts_sampler.sub_div_milestones=[7]", "pts_sampler.compact_freq=5"]
    rng0 = np.random.default_rng(31)
    batches = []
    for _ in range(ITERS + 2):
        ro, rd, bounds, cam = e2e.fox_batch(st, rng0, R)
        batches.append([torch.from_numpy(np.ascontiguousarray(a)) for a in (ro, rd, bounds, rng0.random((R, 3), dtype=np.float32), cam)])
    runner, cfg, _ = runtime.make_runner(st, "wanjinyou", overrides, seed=5, table_init=0.3)
    states = [t.clone() for t in runner.states()]
    states[8][-16 * 64:-15 * 64] *= 16.0
    runner.load_states(states)
    runner.n_edge_pts = NE
    runner.speculative_sampling, runner.tail_repair, runner.speculation_depth, runner.march_blocks = True, tail, depth, 96
    torch.manual_seed(11)
    L.wemu_hb_enable(1)
    for it in range(ITERS):
        for t in runner.occupancy_buffers()[:2]:
            t.fill_(0)
        b, nb, nb2 = batches[it], batches[it + 1], batches[it + 2]
        if depth >= 2:
            runner.train_step(b[0], b[1], b[2], b[3], b[4], True, nb[0], nb[1], nb[2], nb2[0], nb2[1])
        else:
            runner.train_step(b[0], b[1], b[2], b[3], b[4], True, nb[0], nb[1], nb[2])
    runner.flush()
    report(L, "%d steps, tail repair %s, speculation depth %d; %s" % (ITERS, tail, depth, dict(runner.speculation_counters())))
    return 0


if __name__ == "__main__":
    sys.exit(main())
