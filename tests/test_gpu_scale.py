"""GPU parity where the numbers are quoted (run with `-m gpu`): BASELINE config 2 at full size (8192 rays, 2^19 x 16
table), the converged 148k-node octree, a synthetic 23-level tree that fills the DFS work stack, and a multi-iteration
training trajectory across a subdivision milestone and a compaction -- all against the CPU oracle (pinned to the
reference's own kernels, tests/test_oracle_vs_ref.py) on identical state and identical explicit random draws.
Contract (north star): bit-exact sample indices; rendered RGB / PSNR within 1e-3."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import capi as oc  # noqa: E402
from oracle import pipeline as op  # noqa: E402
from oracle import octree_construct as octc  # noqa: E402
from test_gpu_e2e import fox_batch, oracle_train_iteration, rel_err  # noqa: E402

F32 = np.float32
DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def rt():
    if not torch.cuda.is_available():
        pytest.fail("no HIP device visible")
    import f2_nerf_amd  # noqa: F401
    from f2_nerf_amd import runtime
    runtime.host()
    return runtime


@pytest.fixture(scope="module")
def hip():
    if not torch.cuda.is_available():
        pytest.fail("no HIP device visible")
    import f2_nerf_amd  # noqa: F401
    from f2_nerf_amd import capi
    capi.lib()
    return capi


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def N(t):
    return t.detach().cpu().numpy()


def same_bits(a, b):
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    if a.shape != b.shape:
        return False
    if a.dtype == np.float32:
        return bool((a.view(np.uint32) == b.view(np.uint32)).all())
    return bool((a == b).all())


def single_pass_sample(hip, tree_nodes, pers_trans, search_order, rays_o, rays_d, noise, sample_l, scale_by_dis, near=0.01,
                       max_hits=1024):
    """The production sampler chain (child blocks -> single-pass DFS into strided slots -> single-pass march -> scan ->
    pack), straight through the C-ABI."""
    n = rays_o.shape[0]
    tn, tr, so = T(tree_nodes), T(pers_trans), T(search_order)
    ro, rd, nz = T(rays_o), T(rays_d), T(noise)
    n_nodes = tree_nodes.size // 64
    cb = torch.zeros(n_nodes * 8 * 32, dtype=torch.uint8, device=DEV)
    hip.oct_build_child_blocks(n_nodes, tn, cb)
    se = torch.zeros((n, 2), dtype=torch.int32, device=DEV)
    oi = torch.zeros(n * max_hits, dtype=torch.int32, device=DEV)
    nf = torch.zeros((n * max_hits, 2), device=DEV)
    otr = torch.zeros(n * max_hits, dtype=torch.int32, device=DEV)
    tot = torch.zeros(1, dtype=torch.int32, device=DEV)
    hip.oct_intersect_strided(n, max_hits, so, ro, rd, near, 1e8, tn, se, oi, nf, tot, otr, cb)
    cnt = torch.zeros(n, dtype=torch.int32, device=DEV)
    S = 1024
    s_dt = torch.zeros(n * S, device=DEV); s_t = torch.zeros(n * S, device=DEV)
    s_an = torch.zeros((n * S, 2), dtype=torch.int32, device=DEV); fod = torch.zeros(n, device=DEV)
    hip.ray_march_strided(n, sample_l, scale_by_dis, ro, rd, nz, se, oi, nf, tn, tr, cnt, None, s_dt, s_t, s_an, fod, otr)
    pse = torch.zeros((n, 2), dtype=torch.int32, device=DEV)
    tot2 = torch.zeros(1, dtype=torch.int32, device=DEV)
    hip.segment_scan(n, cnt, pse, tot2)
    m = int(tot2.item()); mm = max(m, 1)
    out = dict(pts=torch.zeros((mm, 3), device=DEV), dirs=torch.zeros((mm, 3), device=DEV), dt=torch.zeros(mm, device=DEV),
               t=torch.zeros(mm, device=DEV), anchors=torch.zeros((mm, 3), dtype=torch.int32, device=DEV))
    hip.pack_samples(n, pse, ro, rd, tr, None, s_dt, s_t, s_an, out["pts"], out["dirs"], out["dt"], out["t"], out["anchors"])
    torch.cuda.synchronize()
    res = {k: N(v)[:m] for k, v in out.items()}
    res["pts_idx_bounds"] = N(pse)
    res["first_oct_dis"] = N(fod).reshape(n, 1)
    hits = (N(se), N(oi), N(nf), int(tot.item()))
    return hits, res


def check_strided_hits(hits, ref_hits, max_hits):
    se, oi, nf, total = hits
    rse, ridx, rnf = ref_hits
    n = len(rse)
    assert total == len(ridx)
    cnt = se[:, 1] - se[:, 0]
    assert (se[:, 0] == np.arange(n) * max_hits).all() and (cnt == rse[:, 1] - rse[:, 0]).all()
    # gather the filled prefixes of all slots into the oracle's ray-ordered compact layout
    ray = np.repeat(np.arange(n), cnt)
    pos = np.arange(len(ray)) - np.repeat(rse[:, 0], cnt)
    flat = ray * max_hits + pos
    assert same_bits(oi[flat], ridx), "leaf indices"
    assert same_bits(nf[flat], rnf), "near / far"


# ---------------------------------------------------------------------------------------------------
# sampler on deep trees
# ---------------------------------------------------------------------------------------------------
def test_sampler_on_converged_octree(hip):
    """The octree of a finished fox training (148 k nodes after five subdivisions and 20 compactions; dumped by
    tools/train_fox.py --dump-sampler) with a real training batch of 13056 rays: leaf lists and every sample, bit for bit."""
    z = dict(np.load(os.path.join(ROOT, "tools", "data", "converged_sampler.npz")))
    n = z["rays_o"].shape[0]
    rd = oc.normalize_dirs(z["rays_d"])
    rng = np.random.default_rng(0)
    noise = (((rng.random(1024 + n + 10, dtype=F32) - F32(.5)) + F32(1.)) * F32(float(z["fineness"]))).astype(F32)
    hits, got = single_pass_sample(hip, z["tree_nodes"], z["pers_trans"], z["search_order"], z["rays_o"], rd, noise, 1. / 256., True)
    ref_hits = oc.oct_intersect(z["search_order"], z["rays_o"], rd, 0.01, 1e8, z["tree_nodes"], 1024)
    check_strided_hits(hits, ref_hits, 1024)
    ref = oc.ray_march(z["rays_o"], rd, noise, 1. / 256., True, *ref_hits, z["tree_nodes"], z["pers_trans"])
    per_ray = ref["pts_idx_bounds"][:, 1] - ref["pts_idx_bounds"][:, 0]
    assert per_ray.max() > 150 and per_ray.mean() < 80  # the converged regime: a long tail next to short rays
    for k in ref:
        assert same_bits(got[k], ref[k]), k


def deep_corner_tree(depth, trans_idx=0):
    """A `depth`-level tree whose (-,-,-) child is subdivided at every level, all other children valid leaves: a ray that
    leaves the corner along a near-diagonal crosses 4 children of EVERY node on the path, so the front-to-back walk has 3
    parked siblings per level on its work stack at the bottom of the path."""
    nodes = np.zeros(1 + 8 * depth, octc.NODE_DT)
    nodes["parent"] = -1
    nodes["childs"] = -1
    nodes["trans_idx"] = -1
    nodes["is_leaf_node"] = 1
    side = F32(1024.)
    nodes[0]["center"] = (side * F32(.5),) * 3
    nodes[0]["side_len"] = side
    cur, nxt = 0, 1
    for _ in range(depth):
        pc, ps = nodes[cur]["center"].copy(), nodes[cur]["side_len"]
        nodes[cur]["is_leaf_node"] = 0
        nodes[cur]["trans_idx"] = -1
        for st in range(8):
            off = np.array([((st >> 2) & 1) - .5, ((st >> 1) & 1) - .5, (st & 1) - .5], F32)
            v = nxt + st
            nodes[v]["center"] = (pc + ps * F32(.5) * off).astype(F32)
            nodes[v]["side_len"] = ps * F32(.5)
            nodes[v]["parent"] = cur
            nodes[v]["trans_idx"] = trans_idx
            nodes[cur]["childs"][st] = v
        cur, nxt = nxt, nxt + 8  # slot 0 = (-,-,-)
    return nodes


@pytest.mark.parametrize("depth", [19, 23])
def test_deep_tree_fills_the_dfs_stack(hip, fox_state, depth):
    """Paths of 19+ levels with 3 parked siblings per level need more than 56 work-stack entries (the size the first
    version of the cooperative DFS had, with a lossy overflow branch); 23 levels is the deepest path the reference's own
    48-int stack can walk (PersSampler.cu:7,70).  The leaf lists must equal the one-thread DFS of the reference."""
    st = fox_state
    nodes = deep_corner_tree(depth)
    blob = nodes.view(np.uint8).reshape(-1)
    rng = np.random.default_rng(depth)
    n = 257
    smallest = 1024. / 2 ** depth
    o = (rng.random((n, 3)) * smallest * 0.5).astype(F32)  # inside the deepest cell, next to the corner
    d = (np.array([1., 1.1, 1.2]) + rng.random((n, 3)) * 0.3)
    d[n // 2:] *= -1.0  # the other half looks away from the tree's bulk: short lists
    o[n // 2:] = (rng.random((n - n // 2, 3)) * 900 + 50).astype(F32)
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(F32)
    noise = (((rng.random(1024 + n + 10, dtype=F32) - F32(.5)) + F32(1.)) * F32(64.)).astype(F32)
    tr = np.zeros(1, octc.TRANS_DT)  # one benign warp for every leaf: affine projections (z' = 1), finite Jacobians
    tr["w2xz"][0, :, 0, :3] = rng.standard_normal((12, 3)) * 0.01
    tr["w2xz"][0, :, 1, 3] = 1.0
    tr["weight"][0] = rng.standard_normal((3, 12))
    tr["center"][0] = 512.
    tr["dis_summary"][0] = 1000.
    trans = tr.view(np.uint8).reshape(-1)
    hits, got = single_pass_sample(hip, blob, trans, st["search_order"], o, d, noise, 1. / 256., False, near=0.0)
    ref_hits = oc.oct_intersect(st["search_order"], o, d, 0.0, 1e8, blob, 1024)
    per_ray = ref_hits[0][:, 1] - ref_hits[0][:, 0]
    assert per_ray[:n // 2].min() >= 3 * depth  # every level contributes its three far siblings
    check_strided_hits(hits, ref_hits, 1024)
    ref = oc.ray_march(o, d, noise, 1. / 256., False, *ref_hits, blob, trans)
    for k in ("pts_idx_bounds", "anchors", "t", "dt", "pts"):
        assert same_bits(got[k], ref[k]), k


# ---------------------------------------------------------------------------------------------------
# BASELINE config 2, one full iteration at the benched size
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("table_init", ["reference", 0.3])
def test_config2_full_iteration_parity(rt, fox_state, table_init):
    """8192 rays, wanjinyou.yaml, 2^19 x 16 table, 8192 edge samples, fineness 16 -- exactly what bench.py times (the
    partitioned plane gather, the cached-feature field forward, the owner-binned scatter), once on the fresh table of the
    bench (nothing is stopped early) and once on a table with trained-looking magnitudes (early stop, compaction)."""
    st = fox_state
    rng = np.random.default_rng(2022)
    R, NE = 8192, 8192
    runner, cfg, arrays = rt.make_runner(st, "wanjinyou", seed=7, table_init=table_init)
    assert int(cfg["field"]["log2_table_size"]) == 19
    runner.iter_step = 1
    runner.update_ada_params()
    ro, rd, bounds, cam = fox_batch(st, rng, R)
    gt = rng.random((R, 3), dtype=F32)
    noise = (((rng.random(1024 + R + 10, dtype=F32) - F32(.5)) + F32(1.)) * F32(float(runner.fineness))).astype(F32)
    bg = rng.random((R, 3), dtype=F32)
    eidx = rng.integers(0, st["edge_pool"].size // 64, NE).astype(np.int32)
    ecoord = (rng.random((NE, 2), dtype=F32) * F32(2.) - F32(1.)).astype(F32)
    d = rt.to_dev(ro, rd, bounds, gt, cam, noise, bg, eidx, ecoord)
    runner.set_forced_randoms(d[5], d[6], d[7], d[8])
    ref = oracle_train_iteration(st, cfg, arrays, ro, rd, cam, gt, noise, bg, eidx, ecoord, iter_step=1)

    s = runner.get_samples(d[0], d[1], d[2])
    for k in ("pts_idx_bounds", "anchors", "t", "dt", "pts", "dirs"):
        assert same_bits(N(s[k]), ref["smp"][k]), k
    del s
    runner.zero_grad()
    stats = runner.train_step(d[0], d[1], d[2], d[3], d[4], False)
    assert stats["n_samples"] == len(ref["smp"]["t"])
    # early-stop threshold (T > 1e-4) against 1-ulp expf differences: a handful of samples out of ~8e5 may flip
    assert abs(stats["n_meaningful"] - ref["n_kept"]) <= 8, (stats["n_meaningful"], ref["n_kept"])
    assert abs(float(stats["loss"]) - ref["loss"]) <= 1e-3 * max(1.0, abs(ref["loss"]))
    g = {k: N(v) for k, v in runner.grads().items()}
    rg = ref["grads"]
    for k in ("color_mlp", "field_mlp", "app_emb"):
        assert rel_err(g[k], rg[k]) <= 3e-2, (k, rel_err(g[k], rg[k]))
    a, b = g["feat_pool"].reshape(-1).astype(np.float64), rg["feat_pool"].reshape(-1).astype(np.float64)
    cos = float((a * b).sum() / (np.linalg.norm(a) * np.linalg.norm(b)))
    assert cos > 0.999, cos
    assert rel_err(g["feat_pool"].reshape(-1), rg["feat_pool"].reshape(-1)) <= 5e-2
    assert abs(int((a != 0).sum()) - int((b != 0).sum())) <= 2e-3 * (b != 0).sum() + 64  # same entries touched (fp16 flush aside)

    out = runner.render_train(d[0], d[1], d[2], d[4])
    colors = N(out["colors"])
    assert np.abs(colors - ref["colors"]).max() <= 1e-3, np.abs(colors - ref["colors"]).max()
    mse_g, mse_r = float(((colors - gt) ** 2).mean()), float(((ref["colors"] - gt) ** 2).mean())
    assert abs(10 * np.log10(1 / mse_g) - 10 * np.log10(1 / mse_r)) <= 1e-3
    assert np.abs(N(out["disparity"]) - ref["disparity"]).max() <= 1e-3 * max(1.0, np.abs(ref["disparity"]).max())
    if stats["n_meaningful"] == ref["n_kept"]:
        assert (N(out["idx_start_end"]) == ref["new_se"]).all()
        assert np.abs(N(out["weights"]) - ref["weights"]).max() <= 1e-3


# ---------------------------------------------------------------------------------------------------
# a training trajectory across a subdivision milestone and a compaction
# ---------------------------------------------------------------------------------------------------
class OracleTrainer:
    """ExpRunner::Train (ExpRunner.cpp:82-143) + PersSampler::UpdateOctNodes (PersSampler.cu:536-615) on the oracle."""

    def __init__(self, st, cfg, arrays, n_images):
        self.st, self.cfg = st, cfg
        tn, tr, visit, _ms, table, prim, bias, nvol, p_field, p_color, app_emb = [np.array(a) for a in arrays]
        self.nodes = tn.view(octc.NODE_DT).copy()
        self.tr = tr
        n = len(self.nodes)
        self.w_stats = np.full(n, 1000, np.int32)
        self.a_stats = np.full(n, 1000, np.int32)
        self.visit = visit.astype(np.int32).copy()
        self.milestones = sorted(int(v) for v in cfg["pts_sampler"]["sub_div_milestones"])
        self.grid = op.HashGrid(table, prim, bias, int(nvol[0]), int(cfg["field"]["log2_table_size"]))
        self.p_field, self.p_color, self.app_emb = p_field.copy(), p_color.copy(), app_emb.copy()
        self.adam = {k: [np.zeros_like(v.reshape(-1)), np.zeros_like(v.reshape(-1))] for k, v in
                     (("table", self.grid.table_f32), ("field", self.p_field), ("color", self.p_color), ("emb", self.app_emb))}
        self.iter_step = 0
        self.optim_steps = 0
        ts = st["train_set"]
        self.w2c, self.intri, self.bounds = st["w2c"][ts], st["intri"][ts], st["bounds"][ts]

    def lr(self):
        t = self.cfg["train"]
        warm, end = int(t["learning_rate_warm_up_end_iter"]), int(t["end_iter"])
        if self.iter_step >= warm:
            p = F32(self.iter_step - warm) / F32(end - warm)
            f = (F32(1.) - F32(t["learning_rate_alpha"])) * (np.cos(p * F32(np.pi), dtype=F32) * F32(.5) + F32(.5)) + F32(t["learning_rate_alpha"])
        else:
            f = F32(self.iter_step) / F32(warm)
        return float(F32(t["learning_rate"]) * F32(f))

    def fineness(self):
        t = self.cfg["train"]
        end = int(t["ray_march_fineness_decay_end_iter"])
        if self.iter_step >= end:
            return 1.0
        p = F32(self.iter_step) / F32(end)
        return float(np.exp(np.log(F32(1.)) * p + np.log(F32(t["ray_march_init_fineness"])) * (F32(1.) - p), dtype=F32))

    def tree_blob(self):
        return self.nodes.view(np.uint8).reshape(-1)

    def proc(self, compact, subdivide, brute):
        self.nodes, self.w_stats, self.a_stats = octc.proc_octree(self.nodes, self.w_stats, self.a_stats, self.visit, compact,
                                                                  subdivide, brute)
        self.visit = np.zeros(len(self.nodes), np.int32)

    def step(self, rays_o, rays_d_raw, cam, gt, noise, bg, eidx, ecoord):
        st, cfg = self.st, self.cfg
        arrays = [self.tree_blob(), self.tr, None, None, self.grid.table_f32, self.grid.prim_pool, self.grid.bias_pool,
                  np.array([self.grid.n_volumes]), self.p_field, self.p_color, self.app_emb]
        # (the table gradient as the reference forms it: fp16 addends, fp16 running sums)
        ref = oracle_train_iteration(st, cfg, arrays, rays_o, rays_d_raw, cam, gt, noise, bg, eidx, ecoord, self.iter_step,
                                     fp32_accumulate=False)
        # occupancy update with the pre-pass weights / alphas of ALL marched samples (Renderer.cpp:140-149)
        smp = ref["smp"]
        w_add, a_add, mark, self.visit = oc.mark_visit(len(self.nodes), smp["pts_idx_bounds"], smp["anchors"][:, 1],
                                                       ref["w_pre"], ref["a_pre"], self.visit)
        self.w_stats, self.a_stats, blob = oc.update_node_stats(w_add, a_add, mark, self.w_stats, self.a_stats, self.tree_blob())
        self.nodes = blob.view(octc.NODE_DT).copy()
        while self.milestones and self.milestones[0] <= self.iter_step:  # PersSampler.cu:605-610
            self.proc(True, True, self.milestones[0] <= 0)
            self.nodes = oc.mark_invisible(self.tree_blob(), self.intri, self.w2c, self.bounds).view(octc.NODE_DT).copy()
            self.proc(True, False, False)
            self.milestones.pop(0)
        if self.iter_step % int(cfg["pts_sampler"]["compact_freq"]) == 0:
            self.proc(True, False, False)
        # Adam (the optimiser groups of Hash3DAnchored.cpp:124-150, SHShader.cpp:44-56, Renderer.cpp:238-258)
        self.optim_steps += 1
        lr = self.lr()
        g = ref["grads"]
        for key, p, grad, wd in (("table", self.grid.table_f32, g["feat_pool"], 0.0), ("field", self.p_field, g["field_mlp"], 1e-6),
                                 ("color", self.p_color, g["color_mlp"], 1e-6), ("emb", self.app_emb, g["app_emb"], 1e-6)):
            newp, self.adam[key][0], self.adam[key][1] = op.adam_step(p.reshape(-1), np.asarray(grad, F32).reshape(-1), self.adam[key][0],
                                                                      self.adam[key][1], self.optim_steps, lr, 0.9, 0.99, 1e-15, wd)
            p.reshape(-1)[...] = newp
        self.iter_step += 1
        return ref


def run_trajectory(rt, st, R, NE, ITERS, overrides, realign, render_at, seed, init_stat=2, tag="TRAJ"):
    """ITERS iterations of ExpRunner::TrainStep on the device and of OracleTrainer on the CPU from the same state with explicit
    draws.  Until the two sides FORK (a borderline occupancy vote falls differently: the MLP outputs differ by an f16 ulp,
    PersSampler.cu:497-520) everything integer is compared exactly after every iteration.  realign=True copies the device's
    statistics into the oracle at a fork, so that the exact comparison carries on; realign=False lets the fork stand: from
    then on the comparison is statistical (node / sample counts, loss, colours) and the divergence it causes is reported."""
    runner, cfg, arrays = rt.make_runner(st, "wanjinyou", overrides, seed=5, table_init=0.3)
    runner.n_edge_pts = NE
    for t in runner.occupancy_buffers()[:2]:
        t.fill_(init_stat)
    orc = OracleTrainer(st, cfg, arrays, len(st["poses"]))
    import copy
    cfg_noemb = copy.deepcopy(cfg)
    cfg_noemb["renderer"]["use_app_emb"] = False
    orc.w_stats[:] = init_stat
    orc.a_stats[:] = init_stat
    rng = np.random.default_rng(seed)
    m = dict(n_nodes_seen=set(), vote_flips=0, worst_rgb=0.0, worst_rgb_mean=0.0, fork_iter=None, forked_nodes=0, worst_loss=0.0,
             worst_sample_diff=0.0, worst_node_diff=0)
    forked = False
    for it in range(ITERS):
        assert runner.iter_step == orc.iter_step == it
        ro, rd, bounds, cam = fox_batch(st, rng, R)
        gt = rng.random((R, 3), dtype=F32)
        fin = float(runner.fineness)
        assert abs(fin - orc.fineness()) <= 1e-5 * fin
        noise = (((rng.random(1024 + R + 10, dtype=F32) - F32(.5)) + F32(1.)) * F32(fin)).astype(F32)
        bg = rng.random((R, 3), dtype=F32)
        n_edges = st["edge_pool"].size // 64
        eidx = rng.integers(0, n_edges, NE).astype(np.int32)
        ecoord = (rng.random((NE, 2), dtype=F32) * F32(2.) - F32(1.)).astype(F32)
        d = rt.to_dev(ro, rd, bounds, gt, cam, noise, bg, eidx, ecoord)
        runner.set_forced_randoms(d[5], d[6], d[7], d[8])
        ref = orc.step(ro, rd, cam, gt, noise, bg, eidx, ecoord)
        stats = runner.train_step(d[0], d[1], d[2], d[3], d[4], True)
        assert not stats["skipped_nan"]
        n_ref = len(ref["smp"]["t"])
        dl = abs(float(stats["loss"]) - ref["loss"]) / max(1.0, abs(ref["loss"]))
        m["worst_loss"] = max(m["worst_loss"], dl)
        got_nodes = N(runner.tree_nodes()).view(octc.NODE_DT)
        wst, ast, vcnt = [N(t) for t in runner.occupancy_buffers()]
        m["n_nodes_seen"].add(len(got_nodes))
        if not forked:
            assert stats["n_samples"] == n_ref, it
            assert abs(stats["n_meaningful"] - ref["n_kept"]) <= max(2, R // 1024), (it, stats["n_meaningful"], ref["n_kept"])
            assert dl <= 2e-3, (it, float(stats["loss"]), ref["loss"])
            assert len(got_nodes) == len(orc.nodes), (it, len(got_nodes), len(orc.nodes))
            for f in ("center", "side_len", "parent", "childs", "is_leaf_node"):
                assert (got_nodes[f] == orc.nodes[f]).all(), (it, f)
            assert (vcnt == orc.visit).all(), it
            # A vote compares a sample's weight / alpha with a threshold (PersSampler.cu:497-520); where the two sides' MLP outputs
            # differ by an f16 ulp a borderline vote can fall differently.  Such nodes are counted and bounded.
            bad = (wst != orc.w_stats) | (ast != orc.a_stats) | (got_nodes["trans_idx"] != orc.nodes["trans_idx"])
            if bad.any():
                m["vote_flips"] += int(bad.sum())
                assert bad.sum() <= max(8, R // 256), (it, int(bad.sum()))
                assert (np.abs(wst - orc.w_stats)[bad] <= 513).all() and (np.abs(ast - orc.a_stats)[bad] <= 33).all()  # one vote each
                if realign:  # ... and the oracle's statistics re-aligned so that one borderline vote does not fork the rest
                    orc.w_stats, orc.a_stats = wst.copy(), ast.copy()
                    orc.nodes["trans_idx"] = got_nodes["trans_idx"]
                else:
                    forked = True
                    m["fork_iter"] = it
                    m["forked_nodes"] = int(bad.sum())
        else:
            # the two sides carry different occupancy statistics for a few nodes: leaves may die (and be compacted away, or be
            # subdivided at the milestone) an iteration apart.  What must hold is that this stays a perturbation.
            sd = abs(stats["n_samples"] - n_ref) / max(n_ref, 1)
            nd = abs(len(got_nodes) - len(orc.nodes))
            m["worst_sample_diff"] = max(m["worst_sample_diff"], sd)
            m["worst_node_diff"] = max(m["worst_node_diff"], nd)
            assert sd <= 0.03, (it, stats["n_samples"], n_ref)
            assert nd <= 0.02 * len(orc.nodes) + 16, (it, len(got_nodes), len(orc.nodes))
            assert dl <= 1e-2, (it, float(stats["loss"]), ref["loss"])
        # at the listed iterations both sides render this batch with the UPDATED state (VALIDATE mode: no occupancy votes, no
        # appearance embedding; the explicit noise / background draws stay in force)
        if it in render_at:
            got = N(runner.render_rays(d[0], d[1], d[2])[0])
            arrays_now = [orc.tree_blob(), orc.tr, None, None, orc.grid.table_f32, orc.grid.prim_pool, orc.grid.bias_pool,
                          np.array([orc.grid.n_volumes]), orc.p_field, orc.p_color, orc.app_emb]
            ref2 = oracle_train_iteration(st, cfg_noemb, arrays_now, ro, rd, cam, gt, noise, bg, eidx, ecoord, orc.iter_step)
            e = np.abs(got - ref2["colors"])
            err, mean = float(e.max()), float(e.mean())
            m["worst_rgb"] = max(m["worst_rgb"], err)
            m["worst_rgb_mean"] = max(m["worst_rgb_mean"], mean)
            if not forked:
                # identical weights render within 1e-3 (test_config2_full_iteration_parity); here the weights themselves come out
                # of two optimiser trajectories, and Adam (eps 1e-15) turns one-ulp differences of f16 gradients into full-size
                # steps of the affected entries.  The device side is not bit-reproducible either (scatter order), so the late
                # bound is set from the spread of repeated runs: within 1e-3 for the first two dozen updates in every run; at
                # updates 30 / 36 between 1.5e-3 and 4.3e-3 over 14 runs of the same test
                assert err <= (1e-3 if it < 24 else 1e-2), (it, err)
            else:
                # a leaf that one side has pruned and the other has not was borderline EMPTY on both: colours move by little
                assert mean <= 2e-3 and float(np.percentile(e, 99.9)) <= 3e-2, (it, mean, err)
    states = [N(t) for t in runner.states()]
    dl_f, dl_c = np.abs(states[8] - orc.p_field), np.abs(states[9] - orc.p_color)
    tab = states[4].reshape(-1)
    ref_tab = orc.grid.table_f32.reshape(-1)
    m["table_cos"] = float((tab.astype(np.float64) * ref_tab).sum() / (np.linalg.norm(tab) * np.linalg.norm(ref_tab)))
    m["field_dl"], m["color_dl"] = dl_f, dl_c
    print("%s_METRICS rays %d iters %d realign %d flips %d fork_iter %s forked_nodes %d worst_rgb %.2e (mean %.2e) worst_loss %.2e "
          "sample_diff %.2e node_diff %d table_cos %.6f field mean %.2e p99 %.2e max %.2e colour mean %.2e p99 %.2e max %.2e" % (
              tag, R, ITERS, int(realign), m["vote_flips"], m["fork_iter"], m["forked_nodes"], m["worst_rgb"], m["worst_rgb_mean"],
              m["worst_loss"], m["worst_sample_diff"], m["worst_node_diff"], m["table_cos"], dl_f.mean(), np.percentile(dl_f, 99),
              dl_f.max(), dl_c.mean(), np.percentile(dl_c, 99), dl_c.max()))
    return m


TRAJ_OVERRIDES = ["field.log2_table_size=14", "pts_sampler.sub_div_milestones=[12]", "pts_sampler.compact_freq=8",
                  "train.learning_rate_warm_up_end_iter=20"]


@pytest.mark.parametrize("realign", [True, False])
def test_training_trajectory_across_milestone_and_compaction(rt, fox_state, realign):
    """36 iterations on both sides from the same state with explicit draws: subdivision + MarkInvisible + compaction at the
    milestone (iteration 12), compactions every 8 iterations, nodes dying in between (the occupancy statistics start at 2
    instead of 1000, so unvisited / empty leaves are pruned within the run).  Compared after EVERY iteration: the node
    array, the visit counts and the sample counts exactly; the occupancy statistics exactly up to a bounded number of
    borderline votes; the loss within 2e-3; every sixth iteration the rendered batch colours (1e-3, later 1e-2).
    realign=False (round-2 verdict): the oracle is NOT re-aligned when a borderline vote falls differently -- the two sides
    fork, and the test asserts how far (it reports the iteration of the fork and the divergence that follows)."""
    ITERS = 36
    m = run_trajectory(rt, fox_state, 256, 512, ITERS, TRAJ_OVERRIDES, realign, {it for it in range(ITERS) if it % 6 == 5 or it == ITERS - 1},
                       seed=99, tag="TRAJ")
    assert len(m["n_nodes_seen"]) >= 2 and 897 not in m["n_nodes_seen"], m["n_nodes_seen"]  # pruned at iteration 0, subdivided at the milestone
    if realign:
        assert m["vote_flips"] <= 24, m["vote_flips"]  # of ~1e5 node-iterations, clustered at the end as the weights drift apart (4..8 over 14 runs)
    else:
        assert m["fork_iter"] is None or m["fork_iter"] >= 6, m["fork_iter"]  # (the first iterations are exact in every run)
        assert m["forked_nodes"] <= 8
    # final parameters: the tables of both sides took the same trajectory
    assert m["table_cos"] > (0.9999 if realign or m["fork_iter"] is None else 0.999), m["table_cos"]
    # (Adam turns "tiny gradient or exactly zero" into a full lr-sized step per iteration: individual weights may sit a few
    # learning rates apart after 36 updates; the networks as a whole took the same path)
    for dlt in (m["field_dl"], m["color_dl"]):
        # (over 14 runs: field MLP mean 2.6..3.0e-4, p99 1.4..1.6e-3; colour MLP mean 5.2..6.5e-4, p99 4.2..6.7e-3, max 4e-2)
        assert dlt.mean() <= 2e-3 and np.percentile(dlt, 99) <= 3e-2, (float(dlt.mean()), float(np.percentile(dlt, 99)), float(dlt.max()))


def test_training_trajectory_at_the_benched_size(rt, fox_state):
    """6 iterations at BASELINE config 2's size -- 8192 rays, 2^19 x 16 table, 8192 edge samples, ~7e5 samples per iteration
    (the partitioned gather, the owner-binned scatter, the table Adam over 17 * 2^19 halves) -- across a subdivision milestone
    (iteration 3) and compactions (every 2), oracle not re-aligned.  (12 iterations ran the same way in round 3: no fork, RGB
    1.5e-4, table cosine 1.0000; the oracle costs ~16 s per iteration at this size.)"""
    ITERS = 6
    overrides = ["pts_sampler.sub_div_milestones=[3]", "pts_sampler.compact_freq=2", "train.learning_rate_warm_up_end_iter=20"]
    m = run_trajectory(rt, fox_state, 8192, 8192, ITERS, overrides, False, {2, 5}, seed=123, tag="TRAJ_BENCH_SIZE")
    assert len(m["n_nodes_seen"]) >= 2 and 897 not in m["n_nodes_seen"], m["n_nodes_seen"]
    assert m["table_cos"] > 0.999, m["table_cos"]
    for dlt in (m["field_dl"], m["color_dl"]):
        assert dlt.mean() <= 2e-3 and np.percentile(dlt, 99) <= 3e-2, (float(dlt.mean()), float(np.percentile(dlt, 99)), float(dlt.max()))


# ---------------------------------------------------------------------------------------------------
# octree maintenance on the device
# ---------------------------------------------------------------------------------------------------
def _ref_proc(nodes, w, a, visit, compact, subdivide, brute):
    """The reference's own ProcOctree where oracle/_ref is present (it travels with the snapshot), else its pinned restatement."""
    from oracle import ref
    if ref.available():
        bn, bw, ba = ref.proc_octree(nodes.view(np.uint8).reshape(-1), w, a, visit, compact, subdivide, brute)
        return bn.view(octc.NODE_DT), bw, ba
    return octc.proc_octree(nodes, w, a, visit, compact, subdivide, brute)


@pytest.mark.parametrize("scene", ["fox", "converged"])
def test_device_proc_octree_chain(rt, fox_state, scene):
    """PersOctree::ProcOctree on the device (csrc/octree.hip) against the reference's sequential algorithm through a chain
    of prune / compress / subdivide rounds with random leaf deaths and visit counts -- on the 897-node construction tree
    and on the 148 k-node tree of a finished training: node arrays field by field, statistics, visit counts."""
    st = fox_state
    rng = np.random.default_rng(4)
    runner, cfg, _ = rt.make_runner(st, "wanjinyou", ["field.log2_table_size=12"], seed=1)
    if scene == "converged":
        z = dict(np.load(os.path.join(ROOT, "tools", "data", "converged_sampler.npz")))
        nodes = z["tree_nodes"].view(octc.NODE_DT).copy()
        rounds = [(False, False, 0.3), (True, False, 0.2), (False, False, 0.7)]
    else:
        nodes = st["tree_nodes"].view(octc.NODE_DT).copy()
        rounds = [(True, False, 0.3), (False, False, 0.2), (True, True, 0.0), (False, False, 0.6), (True, False, 0.5), (False, False, 0.95)]
    for rnd, (sub, brute, kill) in enumerate(rounds):
        n = len(nodes)
        valid = np.nonzero(nodes["trans_idx"] >= 0)[0]
        dead = rng.choice(valid, int(len(valid) * kill), replace=False)
        nodes["trans_idx"][dead] = -1
        visit = rng.integers(0, 10, n).astype(np.int32)
        w = rng.integers(-5, 2000, n).astype(np.int32)
        a = rng.integers(-5, 2000, n).astype(np.int32)
        states = [t.cpu() for t in runner.states()]
        states[0] = torch.from_numpy(nodes.view(np.uint8).reshape(-1).copy())
        states[2] = torch.from_numpy(visit)
        runner.load_states(states)
        ws, as_, _ = runner.occupancy_buffers()
        ws.copy_(torch.from_numpy(w)); as_.copy_(torch.from_numpy(a))
        want_nodes, want_w, want_a = _ref_proc(nodes, w, a, visit, True, sub, brute)
        runner.proc_octree(True, sub, brute)
        got = N(runner.tree_nodes()).view(octc.NODE_DT)
        assert len(got) == len(want_nodes) == runner.n_nodes(), (rnd, len(got), len(want_nodes))
        for f in ("center", "side_len", "parent", "childs", "is_leaf_node", "trans_idx"):
            assert (got[f] == want_nodes[f]).all(), (rnd, f)
        gw, ga, gv = [N(t) for t in runner.occupancy_buffers()]
        assert (gw == want_w).all() and (ga == want_a).all() and (gv == 0).all(), rnd
        nodes = got.copy()


# ---------------------------------------------------------------------------------------------------
# BASELINE configs 3-5: other geometries, other presets
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("preset", ["llff", "nerf-360"])
def test_rig_presets_sampler_parity_and_training(rt, preset):
    """confs/llff.yaml (forward-facing rig, sample_l 1/512, no distance scaling, no appearance embedding, disparity loss) and
    confs/nerf-360.yaml (inward ring) on synthetic rigs (f2-nerf_amd/rigs.py): octree / warps / edge pool are built from the
    rig's cameras on the device; on that scene the sampler must equal the oracle bit for bit (random-pose rays, the preset's
    sample_l / near / scale_by_dis), one training iteration must match the oracle (RGB 1e-3), and training must run."""
    import f2_nerf_amd  # noqa: F401
    from f2_nerf_amd import rigs
    torch.manual_seed(3)
    runner, cfg, sc = rigs.build_runner(preset, ["field.log2_table_size=16", "train.learning_rate_warm_up_end_iter=20"], seed=11)
    ps = cfg["pts_sampler"]
    assert abs(float(ps["sample_l"]) - (1. / 512. if preset == "llff" else 1. / 256.)) < 1e-9 and not bool(ps["scale_by_dis"])
    assert not bool(cfg["renderer"]["use_app_emb"])
    n_nodes = runner.n_nodes()
    assert n_nodes > 50 and int(sc["n_volumes"]) > 8, (n_nodes, int(sc["n_volumes"]))
    rng = np.random.default_rng(5)
    R, NE = 1024, 512
    ro, rd, bounds, gt, emb = rt.synthetic_ray_batch(sc, R, rng)
    runner.n_edge_pts = NE
    runner.iter_step = 1
    runner.update_ada_params()
    fin = float(runner.fineness)
    noise = (((rng.random(1024 + R + 10, dtype=F32) - F32(.5)) + F32(1.)) * F32(fin)).astype(F32)
    bg = rng.random((R, 3), dtype=F32)
    eidx = rng.integers(0, sc["edge_pool"].size // 64, NE).astype(np.int32)
    ecoord = (rng.random((NE, 2), dtype=F32) * F32(2.) - F32(1.)).astype(F32)
    d = rt.to_dev(ro, rd, bounds, gt, emb, noise, bg, eidx, ecoord)
    runner.set_forced_randoms(d[5], d[6], d[7], d[8])
    # the oracle on the scene the device built (state-driven parity: same nodes, warps, primes, biases, weights)
    states = [N(t) for t in runner.states()]
    sc_o = dict(sc)
    sc_o["search_order"] = oc.search_order_table()
    ref = oracle_train_iteration(sc_o, cfg, states, ro, rd, emb, gt, noise, bg, eidx, ecoord, iter_step=1)
    s = runner.get_samples(d[0], d[1], d[2])
    assert len(ref["smp"]["t"]) > 20 * R // 4, len(ref["smp"]["t"])  # the rig's rays do cross the scene
    for k in ("pts_idx_bounds", "anchors", "t", "dt", "pts", "dirs"):
        assert same_bits(N(s[k]), ref["smp"][k]), (preset, k)
    stats = runner.train_step(d[0], d[1], d[2], d[3], d[4], False)
    assert stats["n_samples"] == len(ref["smp"]["t"]) and abs(stats["n_meaningful"] - ref["n_kept"]) <= 2
    assert abs(float(stats["loss"]) - ref["loss"]) <= 1e-3 * max(1.0, abs(ref["loss"]))
    out = runner.render_train(d[0], d[1], d[2], d[4])
    assert np.abs(N(out["colors"]) - ref["colors"]).max() <= 1e-3
    runner.clear_forced_randoms()
    gtc = np.tile(np.array([[0.7, 0.4, 0.1]], F32), (R, 1))
    dg = rt.to_dev(gtc)[0]
    mse = [float(runner.train_step(d[0], d[1], d[2], dg, d[4], True)["mse"]) for _ in range(50)]
    assert np.isfinite(mse).all() and min(mse[-5:]) < 0.8 * mse[0], (preset, mse[0], mse[-5:])


def test_big_table_preset_trains_at_log2_22(rt, fox_state):
    """confs/wanjinyou_big.yaml (log2_table_size 20, 50k schedule) at the 2^22 entries per level BASELINE config 5 names:
    the owner-binned scatter with 1024 table slices per level, Adam over the 17 * 2^22-half active prefix."""
    st = fox_state
    rng = np.random.default_rng(15)
    torch.manual_seed(15)
    runner, cfg, _ = rt.make_runner(st, "wanjinyou_big", ["field.log2_table_size=22", "train.learning_rate_warm_up_end_iter=40"], seed=4)
    assert int(cfg["train"]["end_iter"]) == 50000
    runner.n_edge_pts = 2048
    R = 4096
    ro, rd, bounds, cam = fox_batch(st, rng, R)
    gt = np.tile(np.array([[0.7, 0.4, 0.1]], F32), (R, 1))
    d = rt.to_dev(ro, rd, bounds, gt, cam)
    mse = []
    for it in range(40):
        s = runner.train_step(d[0], d[1], d[2], d[3], d[4], True)
        assert s["n_samples"] > 32768 and not s["skipped_nan"]
        mse.append(float(s["mse"]))
    assert np.isfinite(mse).all() and min(mse[-5:]) < 0.8 * mse[0], (mse[0], mse[-5:])
    tab = runner.states()[4]
    assert tab.shape[0] == 16 << 22 and torch.isfinite(tab).all()


# ---------------------------------------------------------------------------------------------------
# speculative sampling: sample early, repair behind the stat update
# ---------------------------------------------------------------------------------------------------
def _strided_sample(hip, tn, cb, tr, so, ro, rd, nz, n, max_hits=1024):
    """f2n_oct_intersect_strided + f2n_ray_march_strided into fresh buffers (device tensors kept)."""
    b = dict(se=torch.zeros((n, 2), dtype=torch.int32, device=DEV), oi=torch.zeros(n * max_hits, dtype=torch.int32, device=DEV),
             nf=torch.zeros((n * max_hits, 2), device=DEV), otr=torch.zeros(n * max_hits, dtype=torch.int32, device=DEV),
             tot=torch.zeros(1, dtype=torch.int32, device=DEV), cnt=torch.zeros(n, dtype=torch.int32, device=DEV),
             s_dt=torch.zeros(n * 1024, device=DEV), s_t=torch.zeros(n * 1024, device=DEV),
             s_an=torch.zeros((n * 1024, 2), dtype=torch.int32, device=DEV), fod=torch.zeros(n, device=DEV))
    hip.oct_intersect_strided(n, max_hits, so, ro, rd, 0.01, 1e8, tn, b["se"], b["oi"], b["nf"], b["tot"], b["otr"], cb)
    hip.ray_march_strided(n, 1. / 256., True, ro, rd, nz, b["se"], b["oi"], b["nf"], tn, tr, b["cnt"], None, b["s_dt"], b["s_t"],
                          b["s_an"], b["fod"], b["otr"])
    return b


def _filled_prefixes(b, n, max_hits=1024):
    """The defined part of a strided sampling result as flat arrays (slots beyond a ray's count are scratch)."""
    se, cnt = N(b["se"]), N(b["cnt"])
    k = se[:, 1] - se[:, 0]
    ray_k = np.repeat(np.arange(n), k)
    flat_k = ray_k * max_hits + (np.arange(len(ray_k)) - np.repeat(np.cumsum(k) - k, k))
    ray_s = np.repeat(np.arange(n), cnt)
    flat_s = ray_s * 1024 + (np.arange(len(ray_s)) - np.repeat(np.cumsum(cnt) - cnt, cnt))
    return dict(se=se, cnt=cnt, oi=N(b["oi"])[flat_k], nf=N(b["nf"])[flat_k], otr=N(b["otr"])[flat_k], s_dt=N(b["s_dt"])[flat_s],
                s_t=N(b["s_t"])[flat_s], s_an=N(b["s_an"])[flat_s], fod=N(b["fod"]), tot=N(b["tot"]))


@pytest.mark.parametrize("kill_frac", [0.02, 0.0005])
def test_speculative_sampling_repair(hip, kill_frac):
    """f2n_oct_intersect_repair / f2n_ray_march_repair (include/f2n_abi.h, "Speculative sampling"): a batch sampled BEFORE a stat
    update killed leaves, then repaired, equals -- slot for slot, count for count -- the same batch sampled AFTER the update.
    Converged 148 k-node octree, its real 13 056-ray batch; the update kills 2 % / 0.05 % of the valid leaves."""
    z = dict(np.load(os.path.join(ROOT, "tools", "data", "converged_sampler.npz")))
    n = z["rays_o"].shape[0]
    rd_np = oc.normalize_dirs(z["rays_d"])
    rng = np.random.default_rng(7)
    noise = (((rng.random(1024 + n + 10, dtype=F32) - F32(.5)) + F32(1.)) * F32(1.)).astype(F32)
    tn, tr, so = T(z["tree_nodes"].copy()), T(z["pers_trans"]), T(z["search_order"])
    ro, rd, nz = T(z["rays_o"]), T(rd_np), T(noise)
    n_nodes = z["tree_nodes"].size // 64
    cb = torch.zeros(n_nodes * 8 * 32, dtype=torch.uint8, device=DEV)
    hip.oct_build_child_blocks(n_nodes, tn, cb)
    spec = _strided_sample(hip, tn, cb, tr, so, ro, rd, nz, n)  # the speculative result: the tree before the update
    before = _filled_prefixes(spec, n)
    # the stat update: the victims were visited, got no positive vote and stand at 0 -> -1 -> dead (PersSampler.cu:579-603)
    nodes = z["tree_nodes"].view(octc.NODE_DT)
    valid = np.nonzero((nodes["trans_idx"] >= 0) & (nodes["childs"] < 0).all(1))[0]
    hit_leaves = np.unique(before["oi"])
    victims = rng.choice(hit_leaves, max(1, int(len(hit_leaves) * kill_frac)), replace=False)  # leaves this batch crosses
    assert np.isin(victims, valid).all()
    w_stats = np.full(n_nodes, 1000, np.int32); a_stats = np.full(n_nodes, 1000, np.int32)
    w_stats[victims] = 0
    mark = np.zeros(n_nodes, np.int32); mark[victims] = 1
    adders = np.full((2, n_nodes), -1, np.int32)
    died_at = torch.zeros(n_nodes, dtype=torch.int32, device=DEV)
    death_epoch = torch.zeros(1, dtype=torch.int32, device=DEV)
    d_add, d_mark, d_w, d_a = T(adders), T(mark), T(w_stats), T(a_stats)
    EPOCH = 5
    hip.oct_update_stats_ex(n_nodes, d_add[0], d_add[1], d_mark, d_w, d_a, tn, cb, True, died_at, EPOCH, death_epoch)
    assert int(death_epoch.item()) == EPOCH
    got_died = np.nonzero(N(died_at) == EPOCH)[0]
    assert (np.sort(got_died) == np.sort(victims)).all()
    assert (N(tn).view(octc.NODE_DT)["trans_idx"][victims] == -1).all()
    # a second update of a later epoch in which nothing dies leaves the stamps alone
    hip.oct_update_stats_ex(n_nodes, d_add[0], d_add[1], d_mark, d_w, d_a, tn, cb, True, died_at, EPOCH + 1, death_epoch)
    assert int(death_epoch.item()) == EPOCH and (N(died_at) == EPOCH).sum() == len(victims)
    # repair asked for a LATER epoch than any death: returns on the device without touching anything
    flags = torch.full((n,), 7, dtype=torch.int32, device=DEV)
    n_rep = torch.zeros(1, dtype=torch.int32, device=DEV)
    hip.oct_intersect_repair(n, 1024, so, ro, rd, 0.01, 1e8, tn, spec["se"], spec["oi"], spec["nf"], spec["tot"], spec["otr"], cb, died_at,
                             EPOCH + 1, death_epoch, flags, n_rep)
    hip.ray_march_repair(n, 1. / 256., True, ro, rd, nz, spec["se"], spec["oi"], spec["nf"], tn, tr, spec["cnt"], None, spec["s_dt"],
                         spec["s_t"], spec["s_an"], spec["fod"], spec["otr"], flags, death_epoch, EPOCH + 1)
    assert (N(flags) == 7).all() and int(n_rep.item()) == 0
    untouched = _filled_prefixes(spec, n)
    for k in before:
        assert same_bits(untouched[k], before[k]), k
    # the real repair
    hip.oct_intersect_repair(n, 1024, so, ro, rd, 0.01, 1e8, tn, spec["se"], spec["oi"], spec["nf"], spec["tot"], spec["otr"], cb, died_at,
                             EPOCH, death_epoch, flags, n_rep)
    hip.ray_march_repair(n, 1. / 256., True, ro, rd, nz, spec["se"], spec["oi"], spec["nf"], tn, tr, spec["cnt"], None, spec["s_dt"],
                         spec["s_t"], spec["s_an"], spec["fod"], spec["otr"], flags, death_epoch, EPOCH)
    repaired = _filled_prefixes(spec, n)
    fresh = _filled_prefixes(_strided_sample(hip, tn, cb, tr, so, ro, rd, nz, n), n)  # the batch sampled after the update
    for k in fresh:
        assert same_bits(repaired[k], fresh[k]), k
    fl = N(flags)
    assert set(np.unique(fl)) <= {0, 1} and int(n_rep.item()) == int(fl.sum())
    # exactly the rays whose speculative list held a victim were walked again
    k_per_ray = before["se"][:, 1] - before["se"][:, 0]
    ray_of = np.repeat(np.arange(n), k_per_ray)
    want = np.zeros(n, np.int32)
    want[np.unique(ray_of[np.isin(before["oi"], victims)])] = 1
    assert (fl == want).all() and 0 < fl.sum() < n
    assert not same_bits(before["cnt"], fresh["cnt"])  # (the deaths did change the batch)
    # against the CPU oracle on the updated tree
    ref_hits = oc.oct_intersect(z["search_order"], z["rays_o"], rd_np, 0.01, 1e8, N(tn), 1024)
    assert same_bits(repaired["oi"], ref_hits[1]) and same_bits(repaired["nf"], ref_hits[2])
    # f2n_pack_samples_repair: the pack of a batch that was packed optimistically before the update runs again only if a leaf
    # died in an epoch >= spec_epoch, and then equals the pack of the repaired slots
    pse = torch.zeros((n, 2), dtype=torch.int32, device=DEV); tot = torch.zeros(1, dtype=torch.int32, device=DEV)
    hip.segment_scan(n, spec["cnt"], pse, tot)
    m = max(int(tot.item()), 1)

    def packed(fill):
        return dict(pts=torch.full((m, 3), fill, device=DEV), dirs=torch.full((m, 3), fill, device=DEV), dt=torch.full((m,), fill, device=DEV),
                    t=torch.full((m,), fill, device=DEV), anchors=torch.full((m, 3), -7, dtype=torch.int32, device=DEV))
    want_p, got_p, kept_p = packed(0.0), packed(3.0), packed(3.0)
    hip.pack_samples(n, pse, ro, rd, tr, None, spec["s_dt"], spec["s_t"], spec["s_an"], *[want_p[k] for k in ("pts", "dirs", "dt", "t", "anchors")])
    hip.pack_samples_repair(n, pse, ro, rd, tr, None, spec["s_dt"], spec["s_t"], spec["s_an"],
                            *[got_p[k] for k in ("pts", "dirs", "dt", "t", "anchors")], death_epoch, EPOCH)
    hip.pack_samples_repair(n, pse, ro, rd, tr, None, spec["s_dt"], spec["s_t"], spec["s_an"],
                            *[kept_p[k] for k in ("pts", "dirs", "dt", "t", "anchors")], death_epoch, EPOCH + 1)
    for k in want_p:
        assert same_bits(N(got_p[k]), N(want_p[k])), k
        assert (N(kept_p[k]) == (-7 if k == "anchors" else 3.0)).all(), k  # (no leaf died since EPOCH + 1: not touched)


@pytest.mark.parametrize("kill_frac,max_hits", [(0.02, 1024), (0.0005, 1024), (0.02, 24)])
def test_speculative_tail_repair(hip, kill_frac, max_hits):
    """Tail repair of a speculatively sampled batch (include/f2n_abi.h, ABI v10): the march records a resumable state per
    leaf-list entry (f2n_ray_march_strided_rec); after a stat update killed leaves, f2n_oct_list_repair removes the dead
    entries from the lists in place, f2n_oct_intersect_repair_flagged walks the lists that were cut at max_hits again, and
    f2n_ray_march_repair_tail resumes every invalidated ray behind its first dead leaf.  The result equals -- slot for slot --
    the batch sampled after the update, and the recording march equals the plain one.  max_hits = 24 cuts most lists of the
    converged scene, so that the F2N_REPAIR_FULL path runs too."""
    z = dict(np.load(os.path.join(ROOT, "tools", "data", "converged_sampler.npz")))
    n = z["rays_o"].shape[0]
    rd_np = oc.normalize_dirs(z["rays_d"])
    rng = np.random.default_rng(11)
    noise = (((rng.random(1024 + n + 10, dtype=F32) - F32(.5)) + F32(1.)) * F32(1.)).astype(F32)
    tn, tr, so = T(z["tree_nodes"].copy()), T(z["pers_trans"]), T(z["search_order"])
    ro, rd, nz = T(z["rays_o"]), T(rd_np), T(noise)
    n_nodes = z["tree_nodes"].size // 64
    cb = torch.zeros(n_nodes * 8 * 32, dtype=torch.uint8, device=DEV)
    hip.oct_build_child_blocks(n_nodes, tn, cb)
    plain = _strided_sample(hip, tn, cb, tr, so, ro, rd, nz, n, max_hits)

    def rec_sample():
        b = dict(se=torch.zeros((n, 2), dtype=torch.int32, device=DEV), oi=torch.zeros(n * max_hits, dtype=torch.int32, device=DEV),
                 nf=torch.zeros((n * max_hits, 2), device=DEV), otr=torch.zeros(n * max_hits, dtype=torch.int32, device=DEV),
                 tot=torch.zeros(1, dtype=torch.int32, device=DEV), cnt=torch.zeros(n, dtype=torch.int32, device=DEV),
                 s_dt=torch.zeros(n * 1024, device=DEV), s_t=torch.zeros(n * 1024, device=DEV),
                 s_an=torch.zeros((n * 1024, 2), dtype=torch.int32, device=DEV), fod=torch.zeros(n, device=DEV),
                 ls=torch.full((n * max_hits, 2), -1, dtype=torch.int32, device=DEV), reached=torch.full((n,), -9, dtype=torch.int32, device=DEV))
        hip.oct_intersect_strided(n, max_hits, so, ro, rd, 0.01, 1e8, tn, b["se"], b["oi"], b["nf"], b["tot"], b["otr"], cb)
        hip.ray_march_strided_rec(n, max_hits, 1. / 256., True, ro, rd, nz, b["se"], b["oi"], b["nf"], tn, tr, b["cnt"], None, b["s_dt"],
                                  b["s_t"], b["s_an"], b["fod"], b["otr"], b["ls"], b["reached"])
        return b
    spec = rec_sample()
    before = _filled_prefixes(spec, n, max_hits)
    want0 = _filled_prefixes(plain, n, max_hits)
    for k in want0:
        assert same_bits(before[k], want0[k]), k  # recording changes nothing
    k_per_ray = before["se"][:, 1] - before["se"][:, 0]
    reached = N(spec["reached"])
    assert ((reached >= 0) & (reached <= k_per_ray)).all()
    # the stat update
    nodes = z["tree_nodes"].view(octc.NODE_DT)
    hit_leaves = np.unique(before["oi"])
    victims = rng.choice(hit_leaves, max(1, int(len(hit_leaves) * kill_frac)), replace=False)
    w_stats = np.full(n_nodes, 1000, np.int32); a_stats = np.full(n_nodes, 1000, np.int32)
    w_stats[victims] = 0
    mark = np.zeros(n_nodes, np.int32); mark[victims] = 1
    adders = np.full((2, n_nodes), -1, np.int32)
    died_at = torch.zeros(n_nodes, dtype=torch.int32, device=DEV)
    death_epoch = torch.zeros(1, dtype=torch.int32, device=DEV)
    d_add, d_mark, d_w, d_a = T(adders), T(mark), T(w_stats), T(a_stats)
    EPOCH = 3
    hip.oct_update_stats_ex(n_nodes, d_add[0], d_add[1], d_mark, d_w, d_a, tn, cb, True, died_at, EPOCH, death_epoch)
    assert int(death_epoch.item()) == EPOCH
    frm = torch.full((n,), 7, dtype=torch.int32, device=DEV)
    n_rep = torch.zeros(1, dtype=torch.int32, device=DEV); n_full = torch.zeros(1, dtype=torch.int32, device=DEV)

    def repair(epoch):
        hip.oct_list_repair(n, max_hits, spec["se"], spec["oi"], spec["nf"], spec["otr"], spec["tot"], died_at, epoch, death_epoch,
                            spec["reached"], frm, n_rep, n_full)
        hip.oct_intersect_repair_flagged(n, max_hits, so, ro, rd, 0.01, 1e8, tn, spec["se"], spec["oi"], spec["nf"], spec["tot"], spec["otr"],
                                         cb, death_epoch, epoch, frm, n_full)
        hip.ray_march_repair_tail(n, max_hits, 1. / 256., True, ro, rd, nz, spec["se"], spec["oi"], spec["nf"], tn, tr, spec["cnt"], None,
                                  spec["s_dt"], spec["s_t"], spec["s_an"], spec["fod"], spec["otr"], spec["ls"], spec["reached"], frm,
                                  death_epoch, epoch)
    repair(EPOCH + 1)  # asked for a later epoch than any death: returns on the device without touching anything
    assert (N(frm) == 7).all() and int(n_rep.item()) == 0 and int(n_full.item()) == 0
    untouched = _filled_prefixes(spec, n, max_hits)
    for k in before:
        assert same_bits(untouched[k], before[k]), k
    repair(EPOCH)
    repaired = _filled_prefixes(spec, n, max_hits)
    fresh = _filled_prefixes(_strided_sample(hip, tn, cb, tr, so, ro, rd, nz, n, max_hits), n, max_hits)
    for k in fresh:
        assert same_bits(repaired[k], fresh[k]), k
    assert not same_bits(before["cnt"], fresh["cnt"])  # (the deaths did change the batch)
    fr = N(frm)
    ray_of = np.repeat(np.arange(n), k_per_ray)
    holds_victim = np.zeros(n, bool)
    holds_victim[np.unique(ray_of[np.isin(before["oi"], victims)])] = True
    assert (fr[~holds_victim] == -1).all()                      # rays without a dead leaf: nothing to do
    cut = k_per_ray >= max_hits
    if max_hits == 1024:
        assert int(n_full.item()) == 0 and (fr >= -1).all()
        # a ray resumes at its first dead entry -- unless its march never got that far
        pos_in_ray = np.arange(len(ray_of)) - np.repeat(np.cumsum(k_per_ray) - k_per_ray, k_per_ray)
        first_dead = np.full(n, 1 << 30)
        dead_e = np.isin(before["oi"], victims)
        np.minimum.at(first_dead, ray_of[dead_e], pos_in_ray[dead_e])
        want = np.where(holds_victim & (first_dead <= reached), first_dead, -1)
        assert (fr == want).all()
        assert 0 < (fr >= 0).sum() and (fr > 0).sum() > 0       # tails were resumed, not only whole rays
    else:
        assert int(n_full.item()) == int((cut & holds_victim).sum()) > 0   # cut lists that lost an entry were walked again ...
        assert (fr[cut & holds_victim] == 0).all()                          # ... and are marched from their origin
    assert int(n_rep.item()) == int((fr >= 0).sum())
    ref_hits = oc.oct_intersect(z["search_order"], z["rays_o"], rd_np, 0.01, 1e8, N(tn), max_hits)
    assert same_bits(repaired["oi"], ref_hits[1]) and same_bits(repaired["nf"], ref_hits[2])


@pytest.mark.parametrize("view", ["after_first_update", "nodes_new_child_blocks_old", "nodes_old_child_blocks_new"])
@pytest.mark.parametrize("max_hits", [1024, 24])
def test_speculative_walk_that_saw_a_later_tree(hip, view, max_hits):
    """A batch begun ahead is walked on a side stream that is only ordered behind the updates issued BEFORE it was begun: when it
    actually runs, later updates may be finished, or in flight.  Whatever mixture of old and new node records / child blocks the
    walk and the march saw, the batch completed behind those updates (repair from spec_epoch) must equal the batch sampled on
    the final tree.  Two updates (epochs 3 and 4); the walk runs after the first one -- seeing all of it, or only its node
    records, or only its child blocks -- with spec_epoch = 3."""
    z = dict(np.load(os.path.join(ROOT, "tools", "data", "converged_sampler.npz")))
    n = z["rays_o"].shape[0]
    rd_np = oc.normalize_dirs(z["rays_d"])
    rng = np.random.default_rng(23)
    noise = (((rng.random(1024 + n + 10, dtype=F32) - F32(.5)) + F32(1.)) * F32(1.)).astype(F32)
    tn, tr, so = T(z["tree_nodes"].copy()), T(z["pers_trans"]), T(z["search_order"])
    ro, rd, nz = T(z["rays_o"]), T(rd_np), T(noise)
    n_nodes = z["tree_nodes"].size // 64
    cb = torch.zeros(n_nodes * 8 * 32, dtype=torch.uint8, device=DEV)
    hip.oct_build_child_blocks(n_nodes, tn, cb)
    first = _filled_prefixes(_strided_sample(hip, tn, cb, tr, so, ro, rd, nz, n, max_hits), n, max_hits)
    hit_leaves = np.unique(first["oi"])
    victims = rng.choice(hit_leaves, max(2, int(len(hit_leaves) * 0.04)), replace=False)
    v1, v2 = victims[: len(victims) // 2], victims[len(victims) // 2:]
    died_at = torch.zeros(n_nodes, dtype=torch.int32, device=DEV)
    death_epoch = torch.zeros(1, dtype=torch.int32, device=DEV)
    d_add = T(np.full((2, n_nodes), -1, np.int32))

    def update(vs, epoch):
        w_stats = np.full(n_nodes, 1000, np.int32); w_stats[vs] = 0
        mark = np.zeros(n_nodes, np.int32); mark[vs] = 1
        hip.oct_update_stats_ex(n_nodes, d_add[0], d_add[1], T(mark), T(w_stats), T(np.full(n_nodes, 1000, np.int32)), tn, cb, True, died_at,
                                epoch, death_epoch)
    tn_old, cb_old = tn.clone(), cb.clone()
    update(v1, 3)
    tn_seen = tn_old if view == "nodes_old_child_blocks_new" else tn
    cb_seen = cb_old if view == "nodes_new_child_blocks_old" else cb
    spec = dict(se=torch.zeros((n, 2), dtype=torch.int32, device=DEV), oi=torch.zeros(n * max_hits, dtype=torch.int32, device=DEV),
                nf=torch.zeros((n * max_hits, 2), device=DEV), otr=torch.zeros(n * max_hits, dtype=torch.int32, device=DEV),
                tot=torch.zeros(1, dtype=torch.int32, device=DEV), cnt=torch.zeros(n, dtype=torch.int32, device=DEV),
                s_dt=torch.zeros(n * 1024, device=DEV), s_t=torch.zeros(n * 1024, device=DEV),
                s_an=torch.zeros((n * 1024, 2), dtype=torch.int32, device=DEV), fod=torch.zeros(n, device=DEV),
                ls=torch.full((n * max_hits, 2), -1, dtype=torch.int32, device=DEV), reached=torch.full((n,), -9, dtype=torch.int32, device=DEV))
    hip.oct_intersect_strided(n, max_hits, so, ro, rd, 0.01, 1e8, tn_seen, spec["se"], spec["oi"], spec["nf"], spec["tot"], spec["otr"], cb_seen)
    hip.ray_march_strided_rec(n, max_hits, 1. / 256., True, ro, rd, nz, spec["se"], spec["oi"], spec["nf"], tn_seen, tr, spec["cnt"], None,
                              spec["s_dt"], spec["s_t"], spec["s_an"], spec["fod"], spec["otr"], spec["ls"], spec["reached"])
    update(v2, 4)
    frm = torch.full((n,), 7, dtype=torch.int32, device=DEV)
    n_rep = torch.zeros(1, dtype=torch.int32, device=DEV); n_full = torch.zeros(1, dtype=torch.int32, device=DEV)
    hip.oct_list_repair(n, max_hits, spec["se"], spec["oi"], spec["nf"], spec["otr"], spec["tot"], died_at, 3, death_epoch, spec["reached"],
                        frm, n_rep, n_full)
    hip.oct_intersect_repair_flagged(n, max_hits, so, ro, rd, 0.01, 1e8, tn, spec["se"], spec["oi"], spec["nf"], spec["tot"], spec["otr"], cb,
                                     death_epoch, 3, frm, n_full)
    hip.ray_march_repair_tail(n, max_hits, 1. / 256., True, ro, rd, nz, spec["se"], spec["oi"], spec["nf"], tn, tr, spec["cnt"], None,
                              spec["s_dt"], spec["s_t"], spec["s_an"], spec["fod"], spec["otr"], spec["ls"], spec["reached"], frm, death_epoch, 3)
    repaired = _filled_prefixes(spec, n, max_hits)
    fresh = _filled_prefixes(_strided_sample(hip, tn, cb, tr, so, ro, rd, nz, n, max_hits), n, max_hits)
    for k in fresh:
        assert same_bits(repaired[k], fresh[k]), k
    assert not same_bits(first["cnt"], fresh["cnt"])


@pytest.mark.parametrize("n_blocks,record,block_waves", [(64, True, 1), (512, False, 1), (512, True, 8), (100, False, 16)])
def test_persistent_march(hip, n_blocks, record, block_waves):
    """f2n_ray_march_persistent (a few persistent waves -- one per workgroup or, round 6, workgroups of several --, rays sorted by
    leaf count, groups of four off a counter) fills the same slots, counts and -- when recording -- the same resumable states as
    one block per four rays."""
    z = dict(np.load(os.path.join(ROOT, "tools", "data", "converged_sampler.npz")))
    n = z["rays_o"].shape[0]
    rd_np = oc.normalize_dirs(z["rays_d"])
    rng = np.random.default_rng(3)
    noise = (((rng.random(1024 + n + 10, dtype=F32) - F32(.5)) + F32(1.)) * F32(1.)).astype(F32)
    tn, tr, so = T(z["tree_nodes"].copy()), T(z["pers_trans"]), T(z["search_order"])
    ro, rd, nz = T(z["rays_o"]), T(rd_np), T(noise)
    n_nodes = z["tree_nodes"].size // 64
    cb = torch.zeros(n_nodes * 8 * 32, dtype=torch.uint8, device=DEV)
    hip.oct_build_child_blocks(n_nodes, tn, cb)
    want = _strided_sample(hip, tn, cb, tr, so, ro, rd, nz, n)
    ls_w = torch.full((n * 1024, 2), -1, dtype=torch.int32, device=DEV); re_w = torch.full((n,), -9, dtype=torch.int32, device=DEV)
    if record:
        hip.ray_march_strided_rec(n, 1024, 1. / 256., True, ro, rd, nz, want["se"], want["oi"], want["nf"], tn, tr, want["cnt"], None,
                                  want["s_dt"], want["s_t"], want["s_an"], want["fod"], want["otr"], ls_w, re_w)
    got = dict(want)
    for k in ("cnt", "s_dt", "s_t", "s_an", "fod"):
        got[k] = torch.zeros_like(want[k])
    ls_g = torch.full((n * 1024, 2), -1, dtype=torch.int32, device=DEV); re_g = torch.full((n,), -9, dtype=torch.int32, device=DEV)
    order = torch.zeros(n, dtype=torch.int32, device=DEV); counter = torch.full((1,), 12345, dtype=torch.int32, device=DEV)
    hip.ray_march_persistent(n, 1024, n_blocks, 1. / 256., True, ro, rd, nz, got["se"], got["oi"], got["nf"], tn, tr, got["cnt"], None,
                             got["s_dt"], got["s_t"], got["s_an"], got["fod"], got["otr"], ls_g if record else None,
                             re_g if record else None, order, counter, block_waves=block_waves)
    a, b = _filled_prefixes(want, n), _filled_prefixes(got, n)
    for k in a:
        assert same_bits(a[k], b[k]), k
    o = N(order)
    assert (np.sort(o) == np.arange(n)).all()                    # a permutation of the rays ...
    k_per_ray = a["se"][:, 1] - a["se"][:, 0]
    assert (np.diff(k_per_ray[o]) <= 0).all()                    # ... longest leaf list first
    assert int(counter.item()) >= (n + 3) // 4                   # every group of four was handed out
    if record:
        assert (N(re_g) == N(re_w)).all()
        assert (N(ls_g) == N(ls_w)).all()                        # (untouched entries keep the fill value on both sides)


@pytest.mark.parametrize("tail,depth", [(True, 3), (True, 2), (True, 1), (False, 1)])
def test_speculative_training_equals_sampling_after_the_update(rt, fox_state, tail, depth):
    """ExpRunner::TrainStep with the next batch's sampling issued speculatively (Renderer::PreSampleSpecBegin / Complete)
    against the same steps with the sampling behind the stat update: per-step sample counts, node array and occupancy
    statistics identical.  Learning rate 0 keeps the weights -- and so both runs -- deterministic, while the statistics are
    re-armed at 0 before every step so that every visited leaf without a positive vote in THAT batch dies at once (and rays
    of the next, already marched batch are repaired) in most steps; a compaction and a subdivision fall inside the run.
    The batches are drawn on the device right before each step, as ExpRunner::Train does (the speculative sampling must be
    ordered behind the kernels that write its rays)."""
    st = fox_state
    overrides = ["field.log2_table_size=14", "train.learning_rate=0.0", "pts_sampler.sub_div_milestones=[9]", "pts_sampler.compact_freq=6"]
    R, NE, ITERS = 1024, 512, 16
    rng0 = np.random.default_rng(31)
    host_batches = []
    for _ in range(ITERS + 2):
        ro, rd, bounds, cam = fox_batch(st, rng0, R)
        host_batches.append([torch.from_numpy(np.ascontiguousarray(a)).pin_memory() for a in (ro, rd, bounds, rng0.random((R, 3), dtype=F32), cam)])
    busy = torch.randn(4096, 4096, device=DEV)
    logs = {}
    for spec in (True, False):
        runner, cfg, _ = rt.make_runner(st, "wanjinyou", overrides, seed=5, table_init=0.3)
        # (a random table under a Xavier network is a near-uniform fog in which every visited leaf earns a positive vote: the
        # density row of the field MLP's output layer is scaled up so that the scene has opaque and empty stretches)
        states = [t.cpu().clone() for t in runner.states()]
        states[8][-16 * 64:-15 * 64] *= 16.0
        runner.load_states(states)
        runner.n_edge_pts = NE
        runner.speculative_sampling = spec
        runner.tail_repair = tail  # (repair by list compaction + tail march, or by a second walk + march from the origin)
        # depth 3: the batch after next is ALWAYS walked two steps ahead of its use, on a small persistent grid; 2 (the default):
        # only for big octrees -- on this small one it is begun when a step's backward has been queued (Renderer::SpecBeginAtStepEnd);
        # 1: the next batch only, from the top of the step
        runner.speculation_depth = depth
        runner.march_blocks = 96
        torch.manual_seed(11)  # the same noise / background / edge draws in both runs
        log = []
        nb = [t.to(DEV, non_blocking=True) for t in host_batches[0]]
        nb2 = [t.to(DEV, non_blocking=True) for t in host_batches[1]]
        for it in range(ITERS):
            for t in runner.occupancy_buffers()[:2]:
                t.fill_(0)
            b, nb = nb, nb2
            for _ in range(8):  # the device is kept behind the host: the next batches' rays are still to be written when ...
                busy = (busy @ busy).clamp_(-1.0, 1.0)
            nb2 = [t.to(DEV, non_blocking=True) for t in host_batches[it + 2]]  # ... their (speculative) sampling is queued
            if depth >= 2:
                s = runner.train_step(b[0], b[1], b[2], b[3], b[4], True, nb[0], nb[1], nb[2], nb2[0], nb2[1])
            else:
                s = runner.train_step(b[0], b[1], b[2], b[3], b[4], True, nb[0], nb[1], nb[2])
            runner.flush()
            w, a, v = [N(t).copy() for t in runner.occupancy_buffers()]
            log.append(dict(n_samples=s["n_samples"], kept=runner.counters()["total_meaningful"], nodes=N(runner.tree_nodes()).copy(),
                            w=w, a=a, v=v, loss=float(s["loss"])))
        logs[spec] = (log, runner.speculation_counters())
    (la, ca), (lb, cb_) = logs[True], logs[False]
    print("SPEC_COUNTERS on %s | off %s" % (ca, cb_))
    assert ca["speculative"] >= ITERS - 6 and ca["rays_repaired"] > 0, ca   # most steps speculated, and deaths did invalidate rays
    assert cb_["speculative"] == 0 and cb_["rays_repaired"] == 0, cb_
    n_nodes = set()
    for it in range(ITERS):
        x, y = la[it], lb[it]
        assert x["n_samples"] == y["n_samples"] and x["kept"] == y["kept"], (it, x["n_samples"], y["n_samples"], x["kept"], y["kept"])
        assert x["nodes"].shape == y["nodes"].shape and (x["nodes"] == y["nodes"]).all(), it
        assert (x["w"] == y["w"]).all() and (x["a"] == y["a"]).all() and (x["v"] == y["v"]).all(), it
        assert abs(x["loss"] - y["loss"]) <= 1e-6 * max(1.0, abs(y["loss"])), (it, x["loss"], y["loss"])
        n_nodes.add(x["nodes"].size // 64)
    assert len(n_nodes) >= 3, n_nodes  # compaction and subdivision happened inside the run
