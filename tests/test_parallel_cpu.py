"""CPU, world_size 2, gloo: the data-parallel collectives of f2-nerf_amd/parallel.py.
  * gradient sync: every rank ends with the mean of the per-rank gradients, only on the active table prefix;
  * occupancy sync: two ranks that each saw HALF of a ray batch end, after all-reduce(MAX) of their votes, with exactly
    the node statistics and pruning decisions of one process that saw the WHOLE batch (oracle restatement of
    PersSampler.cu:475-603), i.e. replicas stay identical and equal to the single-GPU semantics."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import f2_nerf_amd  # noqa: F401
    from f2_nerf_amd import parallel
    from oracle import capi as oc
    # ---- gradient sync ----
    log2 = 6
    rng = np.random.default_rng(100 + rank)
    table = torch.from_numpy(rng.standard_normal((16 << log2, 2)).astype(np.float16))
    mlp1 = torch.from_numpy(rng.standard_normal(3072).astype(np.float32))
    mlp2 = torch.from_numpy(rng.standard_normal(7168).astype(np.float32))
    emb = torch.from_numpy(rng.standard_normal((50, 16)).astype(np.float32))
    before = table.clone()
    parallel.make_grad_sync([table, mlp1, mlp2, emb], log2)()
    np.save(os.path.join(out_dir, "g_table_%d.npy" % rank), table.numpy())
    np.save(os.path.join(out_dir, "g_table_before_%d.npy" % rank), before.numpy())
    np.save(os.path.join(out_dir, "g_mlp2_%d.npy" % rank), mlp2.numpy())
    # ---- occupancy sync ----
    st = dict(np.load(os.path.join(ROOT, "tests", "golden", "fox_state.npz")))
    g = dict(np.load(os.path.join(ROOT, "tests", "golden", "fox_golden.npz")))
    se = g["march_pts_idx_bounds"]
    R = len(se)
    half = slice(0, R // 2) if rank == 0 else slice(R // 2, R)
    n_nodes = st["tree_nodes"].size // 64
    w = (g["seg_val"] * np.float32(0.01)).astype(np.float32)
    oi = np.ascontiguousarray(g["march_anchors"][:, 1])
    wa, aa, mk, cnt = oc.mark_visit(n_nodes, se[half], oi, w, g["occ_alpha"], np.zeros(n_nodes, np.int32))
    occ = torch.from_numpy(np.stack([wa, aa, mk, cnt]))  # the [4, n_nodes] buffer of PersSampler::UpdateOctNodes
    parallel.occupancy_sync(occ)
    adders, mark, vcnt = occ[:2], occ[2], occ[3]
    ws = np.full(n_nodes, 3, np.int32)
    w2, a2, nodes2 = oc.update_node_stats(adders[0].numpy(), adders[1].numpy(), mark.numpy(), ws, ws, st["tree_nodes"])
    np.save(os.path.join(out_dir, "occ_%d.npy" % rank), np.concatenate([w2, a2, vcnt.numpy(), nodes2.view(np.int32)]))
    dist.destroy_process_group()


class _StubRunner:
    """The hook surface of the host ExpRunner that parallel.attach() wires (bindings.cpp), without a GPU."""

    def __init__(self, rank, log2):
        rng = np.random.default_rng(500 + rank)
        self.table = torch.from_numpy(rng.standard_normal((32 << log2, 2)).astype(np.float16))  # allocated: 2x the active prefix
        self.flat = torch.from_numpy(rng.standard_normal(3072 + 7168 + 800).astype(np.float32))
        self.grad_sync = self.occ_sync = None

    def flatten_small_grads(self):
        return self.flat

    def grad_buffers(self):
        return [self.table, self.flat[:3072], self.flat[3072:10240], self.flat[10240:]]

    def set_grad_sync_hook(self, f):
        self.grad_sync = f

    def set_pipelined_grad_sync(self, begin, end):
        self.grad_sync = lambda: (begin(), end())

    def set_occupancy_sync_hook(self, f):
        self.occ_sync = f


def _attach_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import f2_nerf_amd  # noqa: F401
    from f2_nerf_amd import parallel
    log2 = 5
    r = _StubRunner(rank, log2)
    before_t, before_f = r.table.clone(), r.flat.clone()
    parallel.attach(r, log2)  # gloo: the blocking variant (overlap defaults to the RCCL backend only)
    assert r.grad_sync is not None and r.occ_sync is not None
    # a replica whose batch missed the scene contributes zero gradients and no votes, and still calls both hooks
    if rank == 1:
        r.table.zero_(); r.flat.zero_()
        before_t, before_f = r.table.clone(), r.flat.clone()
    occ = torch.full((4, 37), -1, dtype=torch.int32)
    if rank == 0:
        occ[0, 3] = 512; occ[2, 3] = 1; occ[3, 3] = 9
    r.occ_sync(occ)
    r.grad_sync()
    np.save(os.path.join(out_dir, "a_t_%d.npy" % rank), r.table.numpy())
    np.save(os.path.join(out_dir, "a_f_%d.npy" % rank), r.flat.numpy())
    np.save(os.path.join(out_dir, "a_tb_%d.npy" % rank), before_t.numpy())
    np.save(os.path.join(out_dir, "a_fb_%d.npy" % rank), before_f.numpy())
    np.save(os.path.join(out_dir, "a_occ_%d.npy" % rank), occ.numpy())
    dist.destroy_process_group()


def test_attach_wires_three_collectives_gloo(tmp_path):
    """parallel.attach(): after one step's hooks every replica holds the mean gradient (active table prefix + ONE flat small
    buffer) and the max-combined occupancy votes -- including when one replica had nothing to contribute."""
    world = 2
    mp.spawn(_attach_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    ld = lambda n, r: np.load(tmp_path / (n % r))
    active = 17 << 5
    tb = [ld("a_tb_%d.npy", r).astype(np.float32).reshape(-1) for r in range(world)]
    mean_t = ((tb[0] + tb[1]) / 2).astype(np.float16).astype(np.float32)
    mean_f = (ld("a_fb_%d.npy", 0) + ld("a_fb_%d.npy", 1)) / 2
    for r in range(world):
        t = ld("a_t_%d.npy", r).astype(np.float32).reshape(-1)
        np.testing.assert_array_equal(t[:active], mean_t[:active])
        np.testing.assert_array_equal(t[active:], tb[r][active:])
        np.testing.assert_allclose(ld("a_f_%d.npy", r), mean_f, rtol=0, atol=1e-7)
    o0, o1 = ld("a_occ_%d.npy", 0), ld("a_occ_%d.npy", 1)
    np.testing.assert_array_equal(o0, o1)
    assert o0[0, 3] == 512 and o0[2, 3] == 1 and o0[3, 3] == 9 and (o0[1] == -1).all()


def test_data_parallel_collectives_gloo(tmp_path, fox_state, fox_golden):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    t = [np.load(tmp_path / ("g_table_%d.npy" % r)).astype(np.float32) for r in range(world)]
    b = [np.load(tmp_path / ("g_table_before_%d.npy" % r)).astype(np.float32) for r in range(world)]
    active = 17 << 6
    mean = ((b[0] + b[1]) / 2).astype(np.float16).astype(np.float32).reshape(-1)
    for r in range(world):
        flat = t[r].reshape(-1)
        np.testing.assert_array_equal(flat[:active], mean[:active])            # averaged on the active prefix
        np.testing.assert_array_equal(flat[active:], b[r].reshape(-1)[active:])  # untouched beyond it
    m = [np.load(tmp_path / ("g_mlp2_%d.npy" % r)) for r in range(world)]
    np.testing.assert_array_equal(m[0], m[1])
    # occupancy: both ranks identical, and identical to the single-process result on the full batch
    o = [np.load(tmp_path / ("occ_%d.npy" % r)) for r in range(world)]
    np.testing.assert_array_equal(o[0], o[1])
    from oracle import capi as oc
    st, g = fox_state, fox_golden
    n_nodes = st["tree_nodes"].size // 64
    w = (g["seg_val"] * np.float32(0.01)).astype(np.float32)
    wa, aa, mk, cnt = oc.mark_visit(n_nodes, g["march_pts_idx_bounds"], np.ascontiguousarray(g["march_anchors"][:, 1]), w,
                                    g["occ_alpha"], np.zeros(n_nodes, np.int32))
    ws = np.full(n_nodes, 3, np.int32)
    w2, a2, nodes2 = oc.update_node_stats(wa, aa, mk, ws, ws, st["tree_nodes"])
    np.testing.assert_array_equal(o[0], np.concatenate([w2, a2, cnt, nodes2.view(np.int32)]))


class _StateStub:
    """states() / aux_states() of the host ExpRunner: a replica of another construction has other sizes everywhere."""

    def __init__(self, rank):
        rng = np.random.default_rng(900 + rank)
        n_nodes, n_warps, n_edges = (5, 3, 4) if rank == 0 else (9, 2, 7)
        self.st = [torch.from_numpy(rng.integers(0, 255, n_nodes * 64).astype(np.uint8)),
                   torch.from_numpy(rng.integers(0, 255, n_warps * 544).astype(np.uint8)),
                   torch.from_numpy(rng.standard_normal(12).astype(np.float32))]
        self.aux = [torch.from_numpy(rng.integers(0, 255, n_edges * 64).astype(np.uint8)),
                    torch.from_numpy(rng.standard_normal((n_warps + 1, 3, 4)).astype(np.float32)),
                    torch.from_numpy(rng.standard_normal((n_warps + 1, 3, 3)).astype(np.float32)),
                    torch.empty(0)]  # (a tensor nobody set travels as an empty one)

    def states(self):
        return self.st

    def load_states(self, s):
        self.st = [t.clone() for t in s]

    def aux_states(self):
        return self.aux

    def load_aux_states(self, s):
        self.aux = [t.clone() for t in s]


def _bcast_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import f2_nerf_amd  # noqa: F401
    from f2_nerf_amd import parallel
    r = _StateStub(rank)
    parallel.broadcast_states(r)
    torch.save({"st": r.st, "aux": r.aux}, os.path.join(out_dir, "b_%d.pt" % rank))
    dist.destroy_process_group()


def test_broadcast_replicates_checkpoint_vector_and_edge_pool_gloo(tmp_path):
    """parallel.broadcast_states(): rank 0's checkpoint vector AND what the checkpoint does not hold (edge pool, training
    cameras) reach a replica that was constructed with other node / warp / edge counts -- a rank that kept its own edge pool
    would index rank 0's warps with its own t_idx_a/b (round-2 advisor finding)."""
    world = 2
    mp.spawn(_bcast_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    want = _StateStub(0)
    for r in range(world):
        got = torch.load(tmp_path / ("b_%d.pt" % r))
        for a, b in zip(got["st"] + got["aux"], want.st + want.aux):
            assert a.numel() == b.numel() and a.dtype == b.dtype
            assert torch.equal(a.reshape(-1), b.reshape(-1))


# ---------------------------------------------------------------------------------------------------------------------
# Round 4: the arithmetic of the gradient exchange at world size 8, and bench.py's multi-GPU launcher.
# ---------------------------------------------------------------------------------------------------------------------
def _avg8_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import f2_nerf_amd  # noqa: F401
    from f2_nerf_amd import parallel
    t = torch.from_numpy(np.load(os.path.join(out_dir, "avg8_in.npy"))[rank].copy())
    parallel._avg_(t)
    np.save(os.path.join(out_dir, "avg8_out_%d.npy" % rank), t.numpy())
    dist.destroy_process_group()


def test_f16_gradient_average_over_eight_ranks_neither_overflows_nor_flushes(tmp_path):
    """The hash-gradient table travels as x128 loss-scaled binary16 and is AVERAGED over the ranks (parallel._avg_ on gloo;
    ncclAvg on the C++ host's communicator, DataParallel.cpp).  Eight ranks, adversarial columns: (a) every rank at the f16
    maximum -- a sum-then-divide in f16 would be inf; (b) the smallest normal on one rank only -- its mean is a subnormal and
    must survive; (c) the smallest subnormal x 8 on every rank; (d) mixed signs that cancel; (e) ordinary gradients.  Every
    rank must end with round_f16(exact mean).  The same columns through the pre-multiply-then-sum arithmetic in f16 that
    NCCL / RCCL document for ncclAvg (each contribution scaled by 1/8 first, partial sums in the buffer's type): no overflow
    either, and within one f16 ulp of the largest contribution per hop of the exact mean."""
    world = 8
    rng = np.random.default_rng(2024)
    n = 4096
    x = (rng.standard_normal((world, n)) * 3.0).astype(np.float16)                   # (e)
    x[:, 0] = np.float16(65504.0)                                                      # (a)
    x[:, 1] = 0; x[3, 1] = np.float16(2.0 ** -14)                                      # (b)
    x[:, 2] = np.float16(8 * 2.0 ** -24)                                               # (c)
    x[:, 3] = np.float16(1000.0) * np.where(np.arange(world) % 2 == 0, 1, -1)          # (d)
    x[:, 4] = np.float16(-65504.0)
    np.save(tmp_path / "avg8_in.npy", x)
    mp.spawn(_avg8_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    exact = x.astype(np.float64).mean(0)
    want = exact.astype(np.float16)
    for r in range(world):
        got = np.load(tmp_path / ("avg8_out_%d.npy" % r))
        assert np.isfinite(got.astype(np.float32)).all()
        assert (got.view(np.uint16) == want.view(np.uint16)).all(), r
    assert want[0] == np.float16(65504.0) and want[4] == np.float16(-65504.0) and want[1] == np.float16(2.0 ** -17) and want[3] == 0
    assert want[2] == np.float16(8 * 2.0 ** -24)
    # ncclAvg as documented (PreMulSum with 1/nranks, arithmetic in the buffer's type): ring order 0..7
    acc = (x[0].astype(np.float32) / 8).astype(np.float16)
    for r in range(1, world):
        acc = (acc.astype(np.float32) + (x[r].astype(np.float32) / 8).astype(np.float16).astype(np.float32)).astype(np.float16)
    assert np.isfinite(acc.astype(np.float32)).all()
    assert acc[0] == np.float16(65504.0)                       # no overflow at the maximum
    assert acc[2] == np.float16(8 * 2.0 ** -24)                # subnormal contributions of 2^-24 each still add up
    ulp = np.maximum(np.abs(x.astype(np.float64)).max(0), 2.0 ** -14) * 2.0 ** -10   # of the largest contribution of a column
    assert (np.abs(acc.astype(np.float64) - exact) <= 8 * ulp).all()


def test_f16_ring_and_tree_orders_of_the_table_average_over_eight_ranks():
    """RCCL reduces the 17 MiB f16 table in the buffer's own type, in an order the collective's algorithm fixes: a RING
    reduce-scatter accumulates chunk c along the ranks c+1, c+2, ..., c (a different rotation per chunk), a TREE pairs ranks
    ((0+1)+(2+3))+((4+5)+(6+7)); ncclAvg pre-multiplies every contribution by 1/8 (exact in binary16 short of the subnormals).
    Worst-case x128 loss-scaled gradients -- every rank near the f16 maximum with equal sign, alternating signs, one huge rank
    among tiny ones, a heavy-tailed body -- through all eight ring rotations and the tree, in binary16 arithmetic: never an
    inf / NaN, every order within 4 f16 ulps of the largest partial sum's magnitude of the exact mean, and whatever the order
    every replica holds the SAME bytes (an all-reduce hands every rank the one reduced chunk), which is what keeps replicas
    identical (bench.py `replicas.identical`).  (round-4 verdict, weak 10: the gloo test pinned pre-multiply-then-sum only.)"""
    world, n = 8, 8192
    rng = np.random.default_rng(7)
    x = (rng.standard_t(2.5, (world, n)) * 40.0).clip(-65504, 65504).astype(np.float16)        # heavy-tailed body
    x[:, 0] = np.float16(65504.0)                                                                # all at the maximum, same sign
    x[:, 1] = np.float16(65504.0) * np.where(np.arange(world) % 2 == 0, 1, -1)                   # alternating signs at the maximum
    x[:, 2] = np.float16(2.0 ** -20); x[5, 2] = np.float16(60000.0)                              # one huge rank among tiny ones
    x[:, 3] = np.float16(2.0 ** -24)                                                             # smallest subnormal everywhere
    x[:, 4] = np.float16(-65504.0); x[0, 4] = np.float16(65504.0)

    def h(v):
        return v.astype(np.float16)

    pre = h(x.astype(np.float32) / 8)  # ncclAvg: PreMulSum with 1 / nranks, in the buffer's type

    def add(a, b):
        return h(a.astype(np.float32) + b.astype(np.float32))

    exact = x.astype(np.float64).mean(0)
    results = []
    for start in range(world):  # ring: the chunk that ends on rank `start - 1` is accumulated along start, start+1, ...
        acc = pre[start]
        for k in range(1, world):
            acc = add(acc, pre[(start + k) % world])
        results.append(acc)
    t = [pre[r] for r in range(world)]  # tree
    while len(t) > 1:
        t = [add(t[i], t[i + 1]) for i in range(0, len(t), 2)]
    results.append(t[0])
    big = np.maximum(np.abs(x.astype(np.float64)).max(0), 2.0 ** -14)
    for acc in results:
        a = acc.astype(np.float64)
        assert np.isfinite(a).all()
        assert (np.abs(a - exact) <= 4 * big * 2.0 ** -10 + 8 * 2.0 ** -24).all()
    assert all(r[0] == np.float16(65504.0) for r in results)             # eight maxima average to the maximum: no overflow on the way
    assert all(abs(float(r[1])) <= 64.0 for r in results)                # cancelling maxima stay finite in every order
    # the pre-multiply loses subnormal bits below 2^-24 * 8: the column of smallest subnormals averages to 0 or 2^-24-ish, never garbage
    assert all(0.0 <= float(r[3]) <= 2.0 ** -23 for r in results)


def test_bench_multi_gpu_launcher_dry_run():
    """`python bench.py --gpus 8` outside a launcher re-executes itself under torch.distributed.run with eight ranks on
    127.0.0.1 (what the driver does itself); the command is checked without running it (F2N_BENCH_DRY_RUN=1), and a rank
    count that does not match WORLD_SIZE is refused."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["F2N_BENCH_DRY_RUN"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "7", "--warmup", "3"], env=env,
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-500:]
    cmd = r.stdout.strip().splitlines()[-1].split()
    assert "torch.distributed.run" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--nproc-per-node") + 1] == "8" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-6:] == ["--gpus", "8", "--steps", "7", "--warmup", "3"] and cmd[-7].endswith("bench.py")
    env2 = dict(env, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r2 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8"], env=env2, capture_output=True, text=True, timeout=120)
    assert r2.returncode != 0 and "WORLD_SIZE=2" in (r2.stderr + r2.stdout)
