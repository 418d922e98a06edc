"""CPU: randomized bit-exact comparison of the oracle restatement against the reference's own kernels
(oracle/_ref, built from /root/reference by oracle/build_ref.py).  Skipped where _ref is not built; the
committed goldens (test_golden.py) are the travelling pin."""
import numpy as np
import pytest

from oracle import capi, ref

pytestmark = pytest.mark.skipif(not ref.available(), reason="oracle/_ref not built (needs /root/reference)")


def bits(a):
    a = np.ascontiguousarray(a)
    return a.view(np.uint32) if a.dtype == np.float32 else a


def same(a, b):
    assert a.shape == b.shape and a.dtype == b.dtype
    np.testing.assert_array_equal(bits(a), bits(b))


def _rays(st, rng, n, all_cams=True):
    cams = np.arange(len(st["poses"])) if all_cams else st["train_set"]
    cam = cams[rng.integers(0, len(cams), n)].astype(np.int32)
    ij = np.stack([rng.integers(0, 960, n), rng.integers(0, 540, n)], -1).astype(np.float32) + np.float32(.5)
    o, d = ref.img2world(st["poses"], st["intri"], st["dist_params"], cam, ij)
    return o, (d / np.linalg.norm(d, axis=-1, keepdims=True)).astype(np.float32), cam


def test_img2world_rays_bit_exact(fox_state):
    """Dataset.cu:13-123 (pixel -> world ray with Newton undistortion): restatement == reference kernel, bit for bit,
    on the fox cameras and on synthetic cameras with strong distortion (many Newton iterations)."""
    st = fox_state
    rng = np.random.default_rng(3)
    n = 5000
    cam = rng.integers(0, len(st["poses"]), n).astype(np.int32)
    ij = np.stack([rng.integers(0, 960, n), rng.integers(0, 540, n)], -1).astype(np.int32)
    for dist in (st["dist_params"], (rng.standard_normal(st["dist_params"].shape) * [0.3, 0.1, 0.02, 0.02]).astype(np.float32),
                 np.zeros_like(st["dist_params"])):
        want_o, want_d = ref.img2world(st["poses"], st["intri"], dist, cam, ij.astype(np.float32) + np.float32(.5))
        got_o, got_d = capi.img2world(st["poses"], st["intri"], dist, cam, ij)
        assert (got_o.view(np.uint32) == want_o.view(np.uint32)).all()
        nan = np.isnan(want_d)  # a diverged Newton iteration (possible under the strong synthetic distortion): NaN on both
        assert (np.isnan(got_d) == nan).all() and nan.mean() < 0.01
        assert (got_d.view(np.uint32)[~nan] == want_d.view(np.uint32)[~nan]).all()


def test_layout_facts():
    lay = ref.struct_layout().tolist()
    assert lay == [64, 0, 12, 16, 20, 52, 56, 544, 0, 384, 528, 540, 64, 0, 4, 8, 20, 32]
    same(ref.search_order_table(), capi.search_order_table())


@pytest.mark.parametrize("seed,fineness,scale_by_dis,max_hits", [(1, 4.0, True, 1024), (2, 1.0, False, 1024),
                                                                  (3, 16.0, True, 5)])
def test_sampler_random(fox_state, seed, fineness, scale_by_dis, max_hits):
    st = fox_state
    rng = np.random.default_rng(seed)
    n = 600
    o, d, _ = _rays(st, rng, n)
    # a few degenerate rays: axis-parallel directions (the +-1e-6 guard) and a ray that misses everything
    d[0] = [1, 0, 0]; d[1] = [0, -1, 0]; d[2] = [0, 0, 1]; o[3] = [1e4, 1e4, 1e4]; d[3] = [1, 0, 0]
    a = ref.oct_intersect(st["search_order"], o, d, 0.01, 1e8, st["tree_nodes"], max_hits)
    b = capi.oct_intersect(st["search_order"], o, d, 0.01, 1e8, st["tree_nodes"], max_hits)
    for x, y in zip(a, b):
        same(x, y)
    noise = ((rng.random(1024 + n + 10, dtype=np.float32) - np.float32(.5)) + np.float32(1.)) * np.float32(fineness)
    ma = ref.ray_march(o, d, noise, 1. / 256., scale_by_dis, *a, st["tree_nodes"], st["pers_trans"])
    mb = capi.ray_march(o, d, noise, 1. / 256., scale_by_dis, *b, st["tree_nodes"], st["pers_trans"])
    for k in mb:
        same(ma[k], mb[k])
    counts = mb["pts_idx_bounds"][:, 1] - mb["pts_idx_bounds"][:, 0]
    assert counts.max() <= 1024 and counts[3] == 0


def test_hash_random(fox_state):
    st = fox_state
    rng = np.random.default_rng(5)
    nv = int(st["n_volumes"])
    for log2_t in (10, 14):
        local = 1 << log2_t
        table = (rng.standard_normal(16 * local * 2).astype(np.float32)).astype(np.float16)
        li = (np.arange(16) * local).astype(np.int32)
        ls = np.full(16, local, np.int32)
        n = 3000
        q = (rng.random((n, 3), dtype=np.float32) * np.float32(1.6) - np.float32(0.3))  # includes q01 < 0 (saturation)
        vol = rng.integers(0, nv, n).astype(np.int32)
        fa = ref.hash_fwd(table.view(np.uint16), st["prim_pool"], li, ls, st["bias_pool"], q, vol, nv)
        fb = capi.hash_fwd(table.view(np.uint16), st["prim_pool"], li, ls, st["bias_pool"], q, vol, nv)
        same(fa, fb)
        gin = (rng.standard_normal((n, 32)) * 0.1).astype(np.float16)
        gin[::5] = 0
        ga = ref.hash_bwd(table.size, st["prim_pool"], li, ls, st["bias_pool"], q, vol, nv, gin.view(np.uint16))
        gb = capi.hash_bwd(table.size, st["prim_pool"], li, ls, st["bias_pool"], q, vol, nv, gin.view(np.uint16))
        same(ga, gb)
    # zero bias (rand_bias: false): all queries with negative coordinates saturate to cell 0
    zb = np.zeros_like(st["bias_pool"])
    same(ref.hash_fwd(table.view(np.uint16), st["prim_pool"], li, ls, zb, q, vol, nv),
         capi.hash_fwd(table.view(np.uint16), st["prim_pool"], li, ls, zb, q, vol, nv))


def test_small_ops_random():
    rng = np.random.default_rng(9)
    R = 300
    cnt = rng.integers(0, 40, R)
    cnt[::17] = 0
    end = np.cumsum(cnt)
    se = np.stack([end - cnt, end], -1).astype(np.int32)
    n = int(end[-1])
    val, vec = rng.random(n, dtype=np.float32), rng.standard_normal((n, 3)).astype(np.float32)
    d1, d3 = rng.random(R, dtype=np.float32), rng.random((R, 3), dtype=np.float32)
    same(ref.flex_sum(val, se), capi.flex_sum(val, se))
    same(ref.flex_sum(vec, se), capi.flex_sum(vec, se))
    same(ref.flex_sum_bwd(d1, se, n), capi.flex_sum_bwd(d1, se, n))
    same(ref.flex_sum_bwd(d3, se, n), capi.flex_sum_bwd(d3, se, n))
    for inc in (False, True):
        same(ref.flex_acc(val, se, inc), capi.flex_acc(val, se, inc))
        same(ref.flex_acc_bwd(val, se, inc), capi.flex_acc_bwd(val, se, inc))
    same(ref.weight_var(val, se), capi.weight_var(val, se))
    same(ref.weight_var_bwd(val, se, d1), capi.weight_var_bwd(val, se, d1))
    for p in (0.0, 0.3, 1.0):
        same(ref.grad_scaling_bwd(vec, se, p), capi.grad_scaling_bwd(vec, se, p))
        same(ref.grad_scaling_bwd(val, se, p), capi.grad_scaling_bwd(val, se, p))
    mask = (val > 0.5).astype(np.int32)
    same(ref.count_valid(se, mask), capi.count_valid(se, mask))
    dirs = rng.standard_normal((500, 3)).astype(np.float32)
    dirs /= np.linalg.norm(dirs, axis=-1, keepdims=True)
    for deg in (1, 2, 3, 4, 5, 6, 7, 8):  # SHShader.cu:32-102, every degree the reference kernel offers
        same(ref.sh_encode(dirs, deg), capi.sh_encode(dirs, deg))
    emb = rng.standard_normal((43, 16)).astype(np.float32)
    eidx = rng.integers(0, 43, R).astype(np.int32)
    ai = capi.scatter_idx(n, se, eidx)
    same(ref.scatter_idx(n, se, eidx), ai)
    ta = rng.standard_normal((n, 16)).astype(np.float32)
    same(ref.scatter_add(emb, ai, ta), capi.scatter_add(emb, ai, ta))
    same(ref.scatter_add_bwd(43, ai, ta), capi.scatter_add_bwd(43, ai, ta))


def test_occupancy_random(fox_state):
    st = fox_state
    rng = np.random.default_rng(11)
    o, d, _ = _rays(st, rng, 400, all_cams=False)
    hits = capi.oct_intersect(st["search_order"], o, d, 0.01, 1e8, st["tree_nodes"])
    noise = np.full(1024 + 400 + 10, 8.0, np.float32)
    m = capi.ray_march(o, d, noise, 1. / 256., True, *hits, st["tree_nodes"], st["pers_trans"])
    n = len(m["t"])
    n_nodes = st["tree_nodes"].size // 64
    w = (rng.random(n, dtype=np.float32) ** 8).astype(np.float32) * np.float32(0.2)
    a = (rng.random(n, dtype=np.float32) ** 8).astype(np.float32) * np.float32(0.3)
    oi = np.ascontiguousarray(m["anchors"][:, 1])
    cnt0 = rng.integers(0, 3, n_nodes).astype(np.int32)
    ra = ref.mark_visit(n_nodes, m["pts_idx_bounds"], oi, w, a, cnt0)
    rb = capi.mark_visit(n_nodes, m["pts_idx_bounds"], oi, w, a, cnt0)
    for x, y in zip(ra, rb):
        same(x, y)
    # stats update (torch integer ops restated) followed by MarkInvalidNodes (reference kernel)
    ws = rng.integers(-2, 5, n_nodes).astype(np.int32)
    as_ = rng.integers(-2, 5, n_nodes).astype(np.int32)
    w2, a2, nodes2 = capi.update_node_stats(rb[0], rb[1], rb[2], ws, as_, st["tree_nodes"])
    occ = (rb[0] > 0).astype(np.int32)
    exp_w = np.clip(np.maximum(ws, occ * rb[0]) + rb[2] * (1 - occ) * rb[0], -100, 1 << 20)
    same(w2, exp_w.astype(np.int32))
    same(nodes2, ref.mark_invalid(w2, a2, st["tree_nodes"]))


def test_octree_maintenance_and_edge_pool_pinned(fox_state):
    """The comparators of the device-side octree maintenance are pinned here: oracle/octree_construct.py's proc_octree and
    construct_edge_pool against PersOctree::ProcOctree / ConstructEdgePool of the reference itself (PersSampler.cpp:120-330,
    :614-659, compiled in place by oracle/build_ref.py) -- node arrays byte for byte, statistics exactly -- on the fox
    octree with random kills / visit counts, through a chain of subdivisions, brute-force subdivision and compactions."""
    from oracle import octree_construct as octc
    st = fox_state
    rng = np.random.default_rng(17)
    nodes = st["tree_nodes"].view(octc.NODE_DT).copy()
    w = np.full(len(nodes), 1000, np.int32)
    a = np.full(len(nodes), 1000, np.int32)
    # the edge pool of the fixture tree
    want_e = ref.construct_edge_pool(st["tree_nodes"]).view(octc.EDGE_DT)  # (the struct's 20 padding bytes are indeterminate)
    for got_e in (octc.construct_edge_pool(nodes), st["edge_pool"].view(octc.EDGE_DT)):
        assert len(got_e) == len(want_e) > 1000
        for f in ("t_idx_a", "t_idx_b", "center", "dir_0", "dir_1"):
            assert (got_e[f] == want_e[f]).all(), f
    for rnd, (sub, brute, kill) in enumerate([(True, False, 0.3), (False, False, 0.2), (True, True, 0.0), (False, False, 0.6),
                                              (True, False, 0.5), (False, False, 0.9)]):
        n = len(nodes)
        valid = np.nonzero(nodes["trans_idx"] >= 0)[0]
        dead = rng.choice(valid, int(len(valid) * kill), replace=False)
        nodes["trans_idx"][dead] = -1
        visit = rng.integers(0, 10, n).astype(np.int32)
        w = rng.integers(-5, 2000, n).astype(np.int32)
        a = rng.integers(-5, 2000, n).astype(np.int32)
        blob = nodes.view(np.uint8).reshape(-1)
        want_nodes, want_w, want_a = ref.proc_octree(blob, w, a, visit, True, sub, brute)
        got_nodes, got_w, got_a = octc.proc_octree(nodes, w, a, visit, True, sub, brute)
        want = want_nodes.view(octc.NODE_DT)
        assert len(want) == len(got_nodes), (rnd, len(want), len(got_nodes))
        for f in ("center", "side_len", "parent", "childs", "is_leaf_node", "trans_idx"):
            assert (want[f] == got_nodes[f]).all(), (rnd, f)
        assert (want_w == got_w).all() and (want_a == got_a).all(), rnd
        nodes = got_nodes.copy()
        assert len(nodes) > 1


def test_warp_construction_pinned_against_the_reference_host_code(fox_state):
    """the comparators the device-side octree / warp builder is tested against
    (oracle/octree_construct.py: distance_summary, get_visi_cams, construct_trans) against the reference's OWN
    DistanceSummary / GetVisiCams / PCA / PersOctree::ConstructTrans (PersSampler.cpp:16-66, 423-612), compiled in place against
    this image's libtorch on the CPU (oracle/build_ref_torch.py).  Visible-camera lists exactly; the 544-byte TransInfo -- twelve
    2x4 projection frames, the 3x12 PCA weight matrix, centre, distance summary -- to rounding (both sides run the same
    libtorch ops; the angle-axis rotation is Eigen on one side, numpy on the other).  The first camera of the farthest-point
    selection is a random draw in the reference: both sides are given the same one through the process-wide generator."""
    import torch
    from oracle import ref_torch, octree_construct as octc
    if not ref_torch.available():
        pytest.skip("oracle/_ref/libf2n_ref_torch.so not built (needs /root/reference)")
    st = fox_state
    ts = st["train_set"]
    c2w = torch.from_numpy(st["poses"][ts].astype(np.float32))
    K = torch.from_numpy(st["intri"][ts].astype(np.float32))
    bnd = torch.from_numpy(st["bounds"][ts].astype(np.float32))
    rng = np.random.default_rng(5)
    for n in (1, 2, 7, 43, 500):  # DistanceSummary: lower-quartile log mean
        d = (rng.random(n) * 3 + 0.05).astype(np.float32)
        a, e = ref_torch.distance_summary(d), octc.distance_summary(torch.from_numpy(d))
        assert abs(a - e) <= 2e-7 * abs(e), (n, a, e)
    ctx = octc._VisiCtx(c2w, K, bnd)
    # boxes: real leaves of the fox octree (the ones the reference built a warp for) plus a few that see few or no cameras
    nodes = st["tree_nodes"].view(octc.NODE_DT)
    leaves = np.nonzero(nodes["trans_idx"] >= 0)[0]
    picks = leaves[rng.choice(len(leaves), 8, replace=False)]
    boxes = [(float(nodes["side_len"][u]), [float(v) for v in nodes["center"][u]]) for u in picks]
    boxes += [(0.25, [1.0, 0.5, -0.5]), (8.0, [0., 0., 0.]), (0.05, [30., 0., 0.])]
    n_warps = 0
    for trial, (side, center) in enumerate(boxes):
        cen = torch.tensor(center, dtype=torch.float32)
        visi = octc.get_visi_cams(ctx, side, cen)
        assert visi == ref_torch.get_visi_cams(side, np.array(center, np.float32), c2w.numpy(), K.numpy(), bnd.numpy()), trial
        if trial >= len(picks) or len(visi) < 6:
            continue
        vc = c2w[visi].contiguous()
        pts = ((torch.from_numpy(rng.random((4096, 3)).astype(np.float32)) - .5) * side + cen).contiguous()
        torch.manual_seed(100 + trial)
        raw = ref_torch.construct_trans(pts.numpy(), vc.numpy(), K[0].numpy(), cen.numpy())
        torch.manual_seed(100 + trial)
        first = int(torch.randint(len(visi), (1,), dtype=torch.int32))
        got = octc.construct_trans(pts, vc, K[0], cen, None, first_cam=first)
        want = raw.view(octc.TRANS_DT)[0]
        # (the PCA eigenvectors amplify the rounding-level differences of the frames by the inverse eigenvalue gaps: 1e-5 .. 1e-4)
        for f, tol in (("w2xz", 5e-6), ("weight", 5e-4), ("center", 0.0), ("dis_summary", 2e-7)):
            x, y = np.asarray(want[f], np.float64), np.asarray(got[f], np.float64)
            assert np.abs(x - y).max() <= tol * max(np.abs(x).max(), 1e-30), (trial, f, np.abs(x - y).max())
        n_warps += 1
    assert n_warps >= 6 and ref_torch.check_failures() == 0


def test_whole_octree_construction_pinned_against_the_reference_host_code(fox_state):
    """PersOctree::PersOctree's node / warp construction for the fox cameras (PersSampler.cpp:70-82: ConstructTreeNode's
    recursion :359-421 around GetVisiCams, DistanceSummary and ConstructTrans), run by the reference's own code (compiled in
    place against libtorch, CPU) and by the restatement the device builder is compared with, from the same generator state and
    with the reference's draw order: every field of every node equal; every warp's projection frames to 1e-5, its PCA weights
    to rounding in the median (a leaf whose covariance has two nearly equal eigenvalues may rotate them: bounded at 5e-2)."""
    import torch
    from oracle import ref_torch, octree_construct as octc
    if not ref_torch.available():
        pytest.skip("oracle/_ref/libf2n_ref_torch.so not built (needs /root/reference)")
    st = fox_state
    ts = st["train_set"]
    c2w = torch.from_numpy(st["poses"][ts].astype(np.float32))
    K = torch.from_numpy(st["intri"][ts].astype(np.float32))
    bnd = torch.from_numpy(st["bounds"][ts].astype(np.float32))
    w2c = torch.from_numpy(st["w2c"][ts].astype(np.float32))
    max_depth, bbox, thres = 16, 512.0, 1.5  # wanjinyou.yaml: max_level 16, bbox_levels 10 (side 2^9), split_dist_thres 1.5
    torch.manual_seed(7)
    rn, rtr = ref_torch.build_octree(max_depth, bbox, thres, c2w.numpy(), K.numpy(), bnd.numpy())
    got_nodes, got_trans, _ = octc.PersOctreeBuilder(max_depth, bbox, thres, c2w, w2c, K, bnd, seed=7, draws_like_reference=True).arrays()
    want_nodes, want_trans = rn.view(octc.NODE_DT), rtr.view(octc.TRANS_DT)
    assert len(want_nodes) == len(got_nodes) > 500 and len(want_trans) == len(got_trans) > 200
    assert len(want_nodes) == st["tree_nodes"].size // 64  # (the committed fixture is this tree)
    for f in ("center", "side_len", "parent", "childs", "is_leaf_node", "trans_idx"):
        assert (want_nodes[f] == got_nodes[f]).all(), f

    def rel(f):
        x, y = np.asarray(want_trans[f], np.float64), np.asarray(got_trans[f], np.float64)
        return np.abs(x - y).reshape(len(x), -1).max(1) / np.abs(x).reshape(len(x), -1).max(1)
    assert rel("w2xz").max() <= 1e-5
    assert np.median(rel("weight")) <= 1e-5 and rel("weight").max() <= 5e-2
    assert (np.asarray(want_trans["center"]) == np.asarray(got_trans["center"])).all() and rel("dis_summary").max() <= 1e-6
    assert ref_torch.check_failures() == 0


def test_training_schedules_of_the_host_equal_the_reference_statements():
    """SURVEY 8(a) a18, the schedule part: march fineness, learning rate (warm-up + cosine), gradient-scaling progress and the
    variance-loss ramp as the C++ host computes them (ExpRunner::ScheduleAt, the pure function behind UpdateAdaParams /
    CurVarLossWeight; bound as host.schedule_at, no device needed) against the reference's own statements
    (ExpRunner.cpp:108-114, 221-254, compiled in place by oracle/build_ref.py) for every iteration of the shipped
    configurations and some adversarial ones: equal floats, bit for bit."""
    import f2_nerf_amd  # noqa: F401
    from f2_nerf_amd import config, runtime
    host = runtime.host()
    cases = []
    for name in ("wanjinyou", "wanjinyou_big", "llff", "nerf-360", "free"):
        c = config.preset(name)
        t = c["train"]
        cases.append((int(t["end_iter"]), float(t["ray_march_init_fineness"]), int(t["ray_march_fineness_decay_end_iter"]),
                      int(t["var_loss_start"]), int(t["var_loss_end"]), int(t["gradient_scaling_start"]), int(t["gradient_scaling_end"]),
                      float(t["learning_rate"]), float(t["learning_rate_alpha"]), float(t["learning_rate_warm_up_end_iter"]),
                      float(t["var_loss_weight"])))
    cases += [(300, 16.0, 100, 10, 20, 5, 50, 1e-2, 0.1, 17.0, 1e-2), (1000, 3.7, 999, 0, 1, 0, 1, 5e-3, 0.0, 1.0, 0.3),
              (64, 1.0, 1, 70, 80, 100, 200, 1e-1, 1.0, 30.0, 0.0)]
    checked = 0
    for (end, fin0, fin_end, vs, ve, gs, ge, lr, alpha, warm, vw) in cases:
        iters = sorted(set(list(range(0, min(end, 2200))) + list(range(max(0, end - 1200), end + 3)) + [fin_end - 1, fin_end, vs, vs + 1, ve, ve + 1]))
        for it in iters:
            if it < 0:
                continue
            want = ref.schedules(it, end, fin0, fin_end, vs, ve, gs, ge, lr, alpha, warm, vw)
            got = np.array(host.schedule_at(fin0, fin_end, lr, alpha, warm, end, float(gs), float(ge), vw, vs, ve, it), np.float32)
            assert (want.view(np.uint32) == got.view(np.uint32)).all(), (it, end, want, got)
            checked += 1
    assert checked > 10000


def test_scene_preparation_of_the_launcher_equals_the_reference_function():
    """The launcher's scene preparation (f2-nerf_amd/rigs.py::prepare_scene: scene centre / radius, normalised poses, w2c,
    relaxed and clamped bounds) against the reference's own Dataset::NormalizeScene + bounds relaxation
    (Dataset.cpp:73-76, 127-146, compiled in place against libtorch on the CPU) on a forward-facing and an inward-ring rig:
    every output equal, bit for bit."""
    from oracle import ref_torch
    if not ref_torch.available():
        pytest.skip("oracle/_ref/libf2n_ref_torch.so not built (needs /root/reference)")
    import f2_nerf_amd  # noqa: F401
    from f2_nerf_amd import rigs
    rng = np.random.default_rng(3)
    for meta, hw in (rigs.forward_facing(rng), rigs.inward_ring(rng)):
        cam = np.asarray(meta, np.float64).astype(np.float32).reshape(-1, 27)
        for bf in ((0.5, 4.0), (0.1, 128.0)):
            want = ref_torch.normalize_scene(cam[:, :12].reshape(-1, 3, 4), cam[:, 25:27], bf)
            got = rigs.prepare_scene(meta, hw, 1.0, bf)
            for k in ("poses", "bounds", "w2c", "center", "radius"):
                same(np.asarray(want[k], np.float32), np.asarray(got[k], np.float32))


def test_pose_interpolation_of_the_host_against_the_reference_function():
    """PoseInterpolate (Utils/CameraUtils.cpp:11-44: Eigen rotation-matrix -> quaternion, slerp, back; translation lerp), used
    by RenderPath / RandRaysWholeSpace: the host's Eigen-free spelling (csrc/host/Dataset.cpp, bound without a device) against
    the reference's function compiled in place, on random pose pairs incl. equal rotations and the end points: within a few
    float32 ulps (5e-7 absolute on unit-scale entries) -- not bit-exact: Eigen's slerp / normalisation round differently."""
    import torch
    from oracle import ref_torch
    if not ref_torch.available():
        pytest.skip("oracle/_ref/libf2n_ref_torch.so not built (needs /root/reference)")
    import f2_nerf_amd  # noqa: F401
    from f2_nerf_amd import runtime
    host = runtime.host()
    rng = np.random.default_rng(0)

    def rand_rot():
        q = rng.standard_normal(4)
        w, x, y, z = q / np.linalg.norm(q)
        return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                         [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                         [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
    worst = 0.0
    for t in range(300):
        A = np.concatenate([rand_rot(), rng.standard_normal((3, 1))], 1).astype(np.float32)
        B = np.concatenate([A[:, :3] if t % 5 == 0 else rand_rot(), rng.standard_normal((3, 1))], 1).astype(np.float32)
        for al in (0.0, 0.25, 0.5, 0.9, 1.0):
            want = ref_torch.pose_interpolate(A, B, al)
            got = host.Dataset.pose_interpolate(torch.from_numpy(A), torch.from_numpy(B), al).numpy()
            worst = max(worst, float(np.abs(want[:, :3] - got[:, :3]).max()))
            assert np.abs(want[:, 3] - got[:, 3]).max() <= 1e-6 * max(1.0, float(np.abs(want[:, 3]).max()))
    assert worst <= 1e-6, worst
