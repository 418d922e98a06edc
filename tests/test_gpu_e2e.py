"""GPU end-to-end parity (run with `-m gpu`): the C++/LibTorch host layer (ExpRunner -> Renderer -> PersSampler /
Hash3DAnchored / SHShader over the C-ABI) against the CPU oracle pipeline on BASELINE config 1 (ngp_fox,
wanjinyou.yaml, 256 rays), identical state and identical explicit random draws on both sides.
Contract (north star): bit-exact sample indices; rendered RGB within 1e-3."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import capi as oc  # noqa: E402
from oracle import pipeline as op  # noqa: E402
from oracle import octree_construct as octc  # noqa: E402

F32 = np.float32
N_EDGE = 2048


@pytest.fixture(scope="module")
def rt():
    if not torch.cuda.is_available():
        pytest.fail("no HIP device visible")
    import f2_nerf_amd  # noqa: F401
    from f2_nerf_amd import runtime
    runtime.host()
    return runtime


def fox_batch(st, rng, n):
    cam = st["train_set"][rng.integers(0, len(st["train_set"]), n)].astype(np.int32)
    pose, K = st["poses"][cam], st["intri"][cam]
    i = rng.integers(0, 960, n).astype(F32) + F32(.5)
    j = rng.integers(0, 540, n).astype(F32) + F32(.5)
    d_cam = np.stack([(j - K[:, 0, 2]) / K[:, 0, 0], -(i - K[:, 1, 2]) / K[:, 1, 1], -np.ones(n, F32)], -1).astype(F32)
    d = np.einsum("nij,nj->ni", pose[:, :3, :3], d_cam).astype(F32)
    return np.ascontiguousarray(pose[:, :3, 3]).astype(F32), d, st["bounds"][cam].astype(F32), cam


def oracle_train_iteration(st, cfg, arrays, rays_o, rays_d_raw, emb_idx, gt, noise, bg, edge_idx, edge_coords, iter_step,
                           fp32_accumulate=True):
    """ExpRunner::Train body on the oracle: returns outputs and true gradients."""
    tn, tr, _, _, table, prim, bias, nvol, p_field, p_color, app_emb = arrays
    nvol = int(nvol[0])
    log2 = int(cfg["field"]["log2_table_size"])
    grid = op.HashGrid(table, prim, bias, nvol, log2)
    rays_d = oc.normalize_dirs(rays_d_raw)  # PersSampler.cu:319 with the fixed summation order (f2n_normalize_dirs)
    ps = cfg["pts_sampler"]
    hits = oc.oct_intersect(st["search_order"], rays_o, rays_d, float(ps["near"]), 1e8, tn, int(ps["max_oct_intersect_per_ray"]))
    smp = oc.ray_march(rays_o, rays_d, noise, float(ps["sample_l"]), bool(ps["scale_by_dis"]), *hits, tn, tr)
    fsh = dict(d_hidden=int(cfg["field"]["mlp_hidden_dim"]), n_hidden=int(cfg["field"]["n_hidden_layers"]))
    ssh = dict(d_hidden=int(cfg["shader"]["d_hidden"]), n_hidden=int(cfg["shader"]["n_hiddens"]), degree=int(cfg["shader"]["degree"]))
    feat_all = op.field_fwd(grid, p_field, smp["pts"], smp["anchors"][:, 0], **fsh)
    w_pre, a_pre, mask, new_se = op.early_stop(feat_all[:, 0], smp["dt"], smp["pts_idx_bounds"])
    pts, dirs, dt, t, anchors = op.compact(mask, smp["pts"], smp["dirs"], smp["dt"], smp["t"], smp["anchors"])
    m = len(pts)
    e_pts, e_idx = oc.edge_samples(st["edge_pool"], tr, edge_idx, edge_coords)
    q_pts = np.concatenate([pts, e_pts.reshape(-1, 3)], 0)
    q_vol = np.concatenate([anchors[:, 0], e_idx.reshape(-1)], 0).astype(np.int32)
    feat, fctx = op.field_fwd(grid, p_field, q_pts, q_vol, want_ctx=True, **fsh)
    scene_feat, edge_feat = feat[:m], feat[m:].reshape(len(edge_idx), 2, 16)
    use_emb = bool(cfg["renderer"]["use_app_emb"])
    sidx = oc.scatter_idx(m, new_se, emb_idx) if use_emb else None
    rgb, sctx = op.shade_fwd(p_color, scene_feat, dirs, app_emb if use_emb else None, sidx, want_ctx=True, **ssh)
    comp = op.composite_fwd(scene_feat, dt, t, rgb, bg, new_se, want_ctx=True)
    tcfg = cfg["train"]
    var_w = 0.0
    if iter_step > tcfg["var_loss_end"]:
        var_w = tcfg["var_loss_weight"]
    elif iter_step > tcfg["var_loss_start"]:
        var_w = (iter_step - tcfg["var_loss_start"]) / (tcfg["var_loss_end"] - tcfg["var_loss_start"]) * tcfg["var_loss_weight"]
    lg = op.losses_and_grads(comp["colors"], gt, comp["disparity"], comp["weights"], new_se, edge_feat, var_w,
                             float(tcfg["disp_loss_weight"]), float(tcfg["tv_loss_weight"]))
    gs0, gs1 = float(tcfg["gradient_scaling_start"]), float(tcfg["gradient_scaling_end"])
    gs = 1.0 if iter_step >= gs1 else max(0.0, (iter_step - gs0) / (gs1 - gs0 + 1e-9))
    drgb, df0 = op.composite_bwd(comp["ctx"], dt, rgb, bg, new_se, lg["dcolors"], lg["ddisparity"], None,
                                 lg["dweights"] if var_w != 0 else None, gs)
    dp_color, dfeat_sh, demb = op.shade_bwd(p_color, sctx, drgb, len(app_emb) if use_emb else 0, sidx)
    dfeat = np.zeros_like(feat)
    dfeat[:m] = dfeat_sh
    dfeat[:m, 0] += df0
    dfeat[m:] = lg["dedge"].reshape(-1, 16)
    dp_field, gtab, _ = op.field_bwd(grid, p_field, fctx, dfeat, 128.0, fp32_accumulate=fp32_accumulate)
    return dict(smp=smp, hits=hits, mask=mask, new_se=new_se, n_kept=m, colors=comp["colors"], disparity=comp["disparity"],
                depth=comp["depth"], weights=comp["weights"], edge_feat=edge_feat, loss=lg["loss"], w_pre=w_pre, a_pre=a_pre,
                grads=dict(feat_pool=gtab.reshape(-1, 2), field_mlp=dp_field, color_mlp=dp_color, app_emb=demb))


def rel_err(a, b):
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


def test_config1_end_to_end_parity(rt, fox_state):
    st = fox_state
    rng = np.random.default_rng(42)
    R = 256
    overrides = ["field.log2_table_size=14"]
    runner, cfg, arrays = rt.make_runner(st, "wanjinyou", overrides, seed=7, table_init=0.3)
    runner.n_edge_pts = N_EDGE
    runner.iter_step = 1  # keep the iteration-0 compaction (iter % compact_freq == 0) out of the comparison
    runner.update_ada_params()
    ro, rd, bounds, cam = fox_batch(st, rng, R)
    gt = rng.random((R, 3), dtype=F32)
    fineness = float(runner.fineness)
    noise = (((rng.random(1024 + R + 10, dtype=F32) - F32(.5)) + F32(1.)) * F32(fineness)).astype(F32)
    bg = rng.random((R, 3), dtype=F32)
    n_edges = st["edge_pool"].size // 64
    eidx = rng.integers(0, n_edges, N_EDGE).astype(np.int32)
    ecoord = (rng.random((N_EDGE, 2), dtype=F32) * F32(2.) - F32(1.)).astype(F32)
    d = rt.to_dev(ro, rd, bounds, gt, cam, noise, bg, eidx, ecoord)
    runner.set_forced_randoms(d[5], d[6], d[7], d[8])

    # --- sampler: bit-exact sample indices and positions ---
    s = runner.get_samples(d[0], d[1], d[2])
    ref = oracle_train_iteration(st, cfg, arrays, ro, rd, cam, gt, noise, bg, eidx, ecoord, iter_step=1)
    for k in ("pts_idx_bounds", "anchors", "t", "dt", "pts", "dirs"):
        got = s[k].cpu().numpy()
        exp = ref["smp"][k]
        assert got.shape == exp.shape, k
        assert (got.view(np.uint32) == exp.view(np.uint32)).all() if got.dtype == F32 else (got == exp).all(), k

    # --- one training iteration without the optimiser step ---
    stats = runner.train_step(d[0], d[1], d[2], d[3], d[4], False)
    assert stats["n_samples"] == len(ref["smp"]["t"])
    assert abs(stats["n_meaningful"] - ref["n_kept"]) <= 2  # early-stop threshold vs 1-ulp expf differences
    out = runner.render_train(d[0], d[1], d[2], d[4])
    colors = out["colors"].detach().cpu().numpy()
    assert np.abs(colors - ref["colors"]).max() <= 1e-3, np.abs(colors - ref["colors"]).max()  # north-star RGB tolerance
    # ... and the same against the oracle's OTHER reading of the unpinned MLP contract (binary16 accumulator fragments,
    # oracle/f2n_oracle.c: oracle_set_mlp_accumulator): the tolerance covers the accumulator nobody can pin
    with oc.mlp_accumulator(1):
        ref_h = oracle_train_iteration(st, cfg, arrays, ro, rd, cam, gt, noise, bg, eidx, ecoord, iter_step=1)
    assert np.abs(colors - ref_h["colors"]).max() <= 1e-3, np.abs(colors - ref_h["colors"]).max()
    mse_g, mse_r = float(((colors - gt) ** 2).mean()), float(((ref["colors"] - gt) ** 2).mean())
    assert abs(10 * np.log10(1 / mse_g) - 10 * np.log10(1 / mse_r)) <= 1e-3  # PSNR within 1e-3 dB
    assert np.abs(out["disparity"].detach().cpu().numpy() - ref["disparity"]).max() <= 1e-3 * max(1.0, np.abs(ref["disparity"]).max())
    if stats["n_meaningful"] == ref["n_kept"]:
        assert (out["idx_start_end"].cpu().numpy() == ref["new_se"]).all()
        assert np.abs(out["weights"].detach().cpu().numpy() - ref["weights"]).max() <= 1e-3
        assert np.abs(out["edge_feats"].detach().cpu().numpy() - ref["edge_feat"]).max() <= 4 * 2.0 ** -11 * max(1.0, np.abs(ref["edge_feat"]).max())
    assert abs(float(stats["loss"]) - ref["loss"]) <= 1e-3 * max(1.0, abs(ref["loss"]))

    # --- gradients (true, unscaled) ---
    runner.zero_grad()
    stats = runner.train_step(d[0], d[1], d[2], d[3], d[4], False)
    g = {k: v.cpu().numpy() for k, v in runner.grads().items()}
    rg = ref["grads"]
    assert rel_err(g["color_mlp"], rg["color_mlp"]) <= 3e-2, rel_err(g["color_mlp"], rg["color_mlp"])
    assert rel_err(g["field_mlp"], rg["field_mlp"]) <= 3e-2, rel_err(g["field_mlp"], rg["field_mlp"])
    assert rel_err(g["app_emb"], rg["app_emb"]) <= 3e-2, rel_err(g["app_emb"], rg["app_emb"])
    gt_tab, rt_tab = g["feat_pool"].reshape(-1), rg["feat_pool"].reshape(-1)
    cos = float((gt_tab * rt_tab).sum() / (np.linalg.norm(gt_tab) * np.linalg.norm(rt_tab)))
    assert cos > 0.999, cos
    assert rel_err(gt_tab, rt_tab) <= 5e-2, rel_err(gt_tab, rt_tab)

    # --- the untaped train step (what bench.py times) and the autograd-taped one are the same computation ---
    stats_a = runner.train_step_autograd(d[0], d[1], d[2], d[3], d[4], False)
    ga = {k: v.cpu().numpy() for k, v in runner.grads().items()}
    assert stats_a["n_samples"] == stats["n_samples"] and stats_a["n_meaningful"] == stats["n_meaningful"]
    assert abs(float(stats_a["loss"]) - float(stats["loss"])) <= 1e-6 * max(1.0, abs(float(stats["loss"])))
    assert abs(float(stats_a["mse"]) - float(stats["mse"])) <= 1e-6
    for k in ("color_mlp", "field_mlp", "app_emb"):
        assert rel_err(g[k], ga[k]) <= 2e-3, (k, rel_err(g[k], ga[k]))
    ta = ga["feat_pool"].reshape(-1)
    assert float((gt_tab * ta).sum() / (np.linalg.norm(gt_tab) * np.linalg.norm(ta))) > 0.99999
    assert rel_err(gt_tab, ta) <= 1e-2, rel_err(gt_tab, ta)


def test_validate_render_and_training_sanity(rt, fox_state):
    st = fox_state
    rng = np.random.default_rng(5)
    torch.manual_seed(5)
    runner, cfg, _ = rt.make_runner(st, "wanjinyou", ["field.log2_table_size=16", "train.learning_rate_warm_up_end_iter=10"], seed=3)
    runner.n_edge_pts = 1024
    R = 1024
    ro, rd, bounds, cam = fox_batch(st, rng, R)
    gt = np.tile(np.array([[0.2, 0.5, 0.8]], F32), (R, 1))
    d = rt.to_dev(ro, rd, bounds, gt, cam)
    cols0 = runner.render_rays(d[0], d[1], d[2])[0].cpu().numpy()
    assert cols0.shape == (R, 3) and np.isfinite(cols0).all()
    losses = []
    for it in range(60):
        s = runner.train_step(d[0], d[1], d[2], d[3], d[4], True)
        losses.append(float(s["mse"]))
        assert not s["skipped_nan"]
    assert runner.iter_step == 60
    assert losses[-1] < 0.5 * losses[0], (losses[0], losses[-1])  # it learns the constant target
    cols1 = runner.render_rays(d[0], d[1], d[2])[0].cpu().numpy()
    assert ((cols1 - gt) ** 2).mean() < ((cols0 - gt) ** 2).mean()
    # checkpoint round trip in the reference's state order
    states = runner.states()
    assert len(states) == 11
    runner2, _, _ = rt.make_runner(st, "wanjinyou", ["field.log2_table_size=16"], seed=99)
    runner2.load_states([t.cpu() for t in states])
    runner2.iter_step = runner.iter_step  # scalars.pt of the reference checkpoint (ExpRunner.cpp:188-194)
    runner2.update_ada_params()
    a = runner.render_rays(d[0], d[1], d[2])[0].cpu().numpy()
    b = runner2.render_rays(d[0], d[1], d[2])[0].cpu().numpy()
    assert np.abs(a - b).max() <= 1e-6


def test_proc_octree_matches_restatement(rt, fox_state):
    """Host-side octree maintenance (compaction / path compression / subdivision) against the oracle's restatement
    of PersSampler.cpp:120-330 on the same node array and statistics."""
    st = fox_state
    runner, cfg, _ = rt.make_runner(st, "wanjinyou", ["field.log2_table_size=12"], seed=1)
    nodes = st["tree_nodes"].view(octc.NODE_DT).copy()
    n = len(nodes)
    rng = np.random.default_rng(8)
    # kill a random third of the valid leaves, give the rest random visit counts
    valid = np.nonzero(nodes["trans_idx"] >= 0)[0]
    dead = rng.choice(valid, len(valid) // 3, replace=False)
    nodes["trans_idx"][dead] = -1
    visit = rng.integers(0, 10, n).astype(np.int32)
    states = runner.states()
    states = [t.cpu() for t in states]
    states[0] = torch.from_numpy(nodes.view(np.uint8).reshape(-1).copy())
    states[2] = torch.from_numpy(visit)
    runner.load_states(states)
    w = np.full(n, 1000, np.int32)
    exp_nodes, exp_w, exp_a = octc.proc_octree(nodes, w, w, visit, True, True, False)
    runner.proc_octree(True, True, False)
    got = runner.tree_nodes().cpu().numpy().view(octc.NODE_DT)
    assert len(got) == len(exp_nodes) == runner.n_nodes()
    for f in ("center", "side_len", "parent", "childs", "is_leaf_node", "trans_idx"):
        assert (got[f] == exp_nodes[f]).all(), f
    wst, ast, vcnt = [t.cpu().numpy() for t in runner.occupancy_buffers()]
    assert (wst == exp_w).all() and (ast == exp_a).all() and (vcnt == 0).all()
    # compaction only
    exp2, _, _ = octc.proc_octree(exp_nodes, exp_w, exp_a, np.zeros(len(exp_nodes), np.int32), True, False, False)
    runner.proc_octree(True, False, False)
    got2 = runner.tree_nodes().cpu().numpy().view(octc.NODE_DT)
    assert len(got2) == len(exp2)
    for f in ("parent", "childs", "is_leaf_node", "trans_idx"):
        assert (got2[f] == exp2[f]).all(), f


def test_dataset_rays_and_whole_image_render(rt, fox_state):
    """device-resident ray generation (host `Dataset`) and whole-image rendering."""
    st = fox_state
    H, W = [int(v) for v in st["image_hw"]]
    rng = np.random.default_rng(8)
    n_img = len(st["poses"])
    images = torch.from_numpy(rng.random((n_img, H // 8, W // 8, 3), dtype=F32))  # small synthetic "photos"
    small = dict(st); small["image_hw"] = np.array([H // 8, W // 8]); small["intri"] = st["intri"].copy()
    small["intri"][:, :2, :] /= 8.0
    ds = rt.make_dataset(small, images)
    torch.manual_seed(5)
    ro, rd, bounds, gt, cam = ds.rand_rays_data(4096, 1)
    cam_np, ij_np = cam.cpu().numpy(), ds.last_ij.cpu().numpy()
    assert set(cam_np.tolist()) <= set(int(v) for v in st["train_set"]) and len(set(cam_np.tolist())) > 20
    assert ij_np[:, 0].max() < H // 8 and ij_np[:, 1].max() < W // 8 and ij_np.min() >= 0
    from oracle import capi as oc
    eo, ed = oc.img2world(small["poses"], small["intri"], small["dist_params"], cam_np, ij_np)
    assert (ro.cpu().numpy().view(np.uint32) == eo.view(np.uint32)).all() and (rd.cpu().numpy().view(np.uint32) == ed.view(np.uint32)).all()
    assert (gt.cpu().numpy() == images.numpy()[cam_np, ij_np[:, 0], ij_np[:, 1]]).all()
    assert (bounds.cpu().numpy() == st["bounds"][cam_np]).all()
    # random-pose rays: unit-norm rotation, origin inside the hull of the blended cameras' window
    wo, wd, wb = ds.rand_rays_whole_space(1000)
    assert wo.shape == (1000, 3) and torch.isfinite(wd).all() and float((wo - wo[0]).abs().max()) == 0.0
    pa = ds.pose_interpolate(torch.from_numpy(st["poses"][3]), torch.from_numpy(st["poses"][7]), 0.25).numpy()
    R = pa[:, :3]
    assert np.abs(R @ R.T - np.eye(3)).max() < 1e-5 and np.allclose(pa[:, 3], 0.75 * st["poses"][3][:, 3] + 0.25 * st["poses"][7][:, 3], atol=1e-6)
    assert np.allclose(ds.pose_interpolate(torch.from_numpy(st["poses"][3]), torch.from_numpy(st["poses"][7]), 0.0).numpy(), st["poses"][3], atol=1e-5)
    # whole-image render == the same rays rendered chunk by chunk; PSNR helper agrees with its definition
    runner, cfg, _ = rt.make_runner(st, "wanjinyou", ["field.log2_table_size=14"], seed=3, table_init=0.3)
    co, cd, cb = ds.rays_of_camera(int(st["test_set"][0]))
    assert co.shape[0] == (H // 8) * (W // 8)
    img, first_oct, disp = runner.render_whole_image(co, cd, cb)
    chunks = [runner.render_rays(co[i:i + 8192], cd[i:i + 8192], cb[i:i + 8192])[0] for i in range(0, co.shape[0], 8192)]
    assert (torch.cat(chunks) == img).all() and float(disp.max()) == 1.0 and torch.isfinite(first_oct).all()
    # the forward-only inference path (one fused field + colour launch on device-side counts, 65536-ray chunks) is the default;
    # the taped Render() in 8192-ray chunks -- the reference's structure -- must give the same image, bit for bit
    assert runner.forward_render
    runner.forward_render = False
    img_t, first_t, disp_t = runner.render_whole_image(co, cd, cb)
    four = runner.render_rays(co[:3000], cd[:3000], cb[:3000])
    runner.forward_render = True
    runner.render_chunk_rays = 3000  # (several chunks, a ragged last one)
    img_c, _, disp_c = runner.render_whole_image(co, cd, cb)
    four_f = runner.render_rays(co[:3000], cd[:3000], cb[:3000])
    runner.render_chunk_rays = 65536
    assert (img_t == img).all() and (first_t == first_oct).all() and (disp_t == disp).all()
    assert (img_c == img).all() and (disp_c == disp).all()
    for a, b in zip(four, four_f):  # colours, disparity, first_oct_dis, depth
        assert (a == b).all()
    psnr = runner.test_image_psnr(ds, int(st["test_set"][0]))
    # the reference's definition (ExpRunner.cpp:360-369): the prediction is quantised to 8 bit before the error is measured
    quant = (img.cpu().clip(0, 1) * 255.).to(torch.uint8).to(torch.float32) / 255.
    mse = float(((quant - images[int(st["test_set"][0])].reshape(-1, 3)) ** 2).mean())
    assert abs(psnr - 20 * np.log10(1.0 / np.sqrt(mse))) < 1e-3
    per_view = runner.test_images(ds)
    assert len(per_view) == len(st["test_set"]) + 1 and abs(per_view[0] - psnr) < 1e-4
    assert abs(per_view[-1] - np.mean(per_view[:-1])) < 1e-4
    # RenderPath frame: colours | first-hit disparity | disparity side by side, at 1/2 resolution
    frame = runner.render_path_frame(ds, torch.from_numpy(st["render_poses"][3]), 2)
    assert tuple(frame.shape) == ((H // 8) // 2, 3 * ((W // 8) // 2), 3) and torch.isfinite(frame).all()
    frames = []
    runner.render_path(ds, torch.from_numpy(st["render_poses"][3:5]), lambda i, im: frames.append((i, im.clone())), 2)
    assert [i for i, _ in frames] == [0, 1] and torch.equal(frames[0][1], frame)
    # checkpoint files in the reference's container format and order: renderer.pt (tensor list) + scalars.pt
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        runner.iter_step = 1234
        runner.update_ada_params()
        runner.save_checkpoint(tmp)
        import os as _os
        assert _os.path.exists(tmp + "/renderer.pt") and _os.path.exists(tmp + "/scalars.pt")
        runner2, _, _ = rt.make_runner(st, "wanjinyou", ["field.log2_table_size=14"], seed=77)
        runner2.load_checkpoint(tmp)
        assert runner2.iter_step == 1234 and abs(runner2.cur_lr - runner.cur_lr) < 1e-12
        for a, b in zip(runner.states(), runner2.states()):
            assert torch.equal(a.cpu(), b.cpu())
        img1, _, _ = runner.render_whole_image(co, cd, cb)  # (at iteration 1234: the march fineness follows the schedule)
        img2, _, _ = runner2.render_whole_image(co, cd, cb)
        assert torch.equal(img2, img1)


def test_data_parallel_hooks_on_one_gpu(rt, fox_state):
    """The RCCL path of bench.py --gpus N, exercised with a world of one rank: attach() (flat small-gradient buffer,
    three collectives per step) must leave training bit-identical to a runner without hooks."""
    import socket
    import torch.distributed as dist
    from f2_nerf_amd import parallel
    st = fox_state
    rng = np.random.default_rng(77)
    R = 512
    ro, rd, bounds, cam = fox_batch(st, rng, R)
    gt = rng.random((R, 3), dtype=F32)
    d = rt.to_dev(ro, rd, bounds, gt, cam)
    noise = torch.from_numpy((((rng.random(1024 + R + 10, dtype=F32) - F32(.5)) + F32(1.)) * F32(8.)).astype(F32)).cuda()
    bg = torch.from_numpy(rng.random((R, 3), dtype=F32)).cuda()
    eidx = torch.from_numpy(rng.integers(0, st["edge_pool"].size // 64, 256).astype(np.int32)).cuda()
    ecoord = torch.from_numpy((rng.random((256, 2), dtype=F32) * F32(2.) - F32(1.)).astype(F32)).cuda()
    outs = []
    sock = socket.socket(); sock.bind(("127.0.0.1", 0)); port = sock.getsockname()[1]; sock.close()
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        for mode in ("none", "blocking", "overlapped", "prefetch", "overlapped+prefetch", "native", "native-blocking+prefetch"):
            runner, cfg, _ = rt.make_runner(st, "wanjinyou", ["field.log2_table_size=14"], seed=11, table_init=0.3)
            runner.n_edge_pts = 256
            runner.set_forced_randoms(noise, bg, eidx, ecoord)
            if mode in ("blocking", "overlapped", "overlapped+prefetch"):
                parallel.attach(runner, 14, overlap=mode.startswith("overlapped"), native=False)  # torch.distributed hooks
            if mode.startswith("native"):  # the C++ host's own RCCL communicator (DataParallel.cpp), one-rank world
                # (hooks_for_one_rank: a one-rank world would otherwise skip its exchanges)
                parallel.attach(runner, 14, overlap=(mode == "native"), native=True, hooks_for_one_rank=True)
            # "prefetch": the next iteration's rays (here: the same batch) are sampled on a side stream during this one
            nxt = (d[0], d[1], d[2]) if "prefetch" in mode else (None, None, None)
            losses = [float(runner.train_step(d[0], d[1], d[2], d[3], d[4], True, *nxt)["loss"]) for _ in range(5)]
            assert runner.iter_step == 5
            outs.append((losses, [t.clone() for t in runner.states()]))  # states() completes a pending overlapped step
            if mode in ("blocking", "overlapped", "native"):
                # a replica whose batch misses the scene still takes part in all three exchanges (no hang, no throw)
                far = d[0] + 1e4
                s_empty = runner.train_step(far, d[1], d[2], d[3], d[4], True)
                assert s_empty["n_samples"] == 0 and runner.iter_step == 6
                runner.flush()
                for t in runner.states():
                    if t.dtype.is_floating_point:
                        assert torch.isfinite(t).all()
    finally:
        dist.destroy_process_group()
    for other in outs[1:]:
        # (the REPORTED loss of a streaming step comes from f2n_composite_train's pre-scaled partial sums, that of a synchronous
        # step from f2n_train_loss: equal to rounding; parameters and gradients -- below -- are compared exactly)
        assert np.allclose(outs[0][0], other[0], rtol=2e-6, atol=0), (outs[0][0], other[0])
        for a, b in zip(outs[0][1], other[1]):
            assert a.shape == b.shape
            if a.dtype.is_floating_point:
                # The hash table's fp64 LDS accumulation order is not deterministic, and a streaming step lays its rows out
                # differently (edge samples first), which shifts the 16-sample rows the scatter combines runs in: individual
                # f16 records round differently, and for entries whose gradient sits at the f16 underflow boundary Adam (eps
                # 1e-15) turns "zero or not" into a full lr-sized step.  Compare closely instead of bitwise.
                d = (a.float() - b.float()).abs()
                assert float(d.max()) <= 2e-3 * max(float(a.float().abs().max()), 1e-6)
                assert float(d.mean()) <= 2e-6 * max(float(a.float().abs().max()), 1e-6)
            else:
                assert (a == b).all()


def test_streaming_step_equals_synchronous_step(rt, fox_state):
    """A streaming training step keeps the survivor count on the device (f2n_*_dyn entry points: buffers sized for the
    marched count, edge samples first, no host read-back in the middle of the iteration).  Same state, same explicit
    draws: loss, colours and gradients must equal those of the synchronous step (the reference's order of operations)."""
    st = fox_state
    rng = np.random.default_rng(61)
    R, NE = 1024, 512
    ro, rd, bounds, cam = fox_batch(st, rng, R)
    gt = rng.random((R, 3), dtype=F32)
    noise = (((rng.random(1024 + R + 10, dtype=F32) - F32(.5)) + F32(1.)) * F32(8.)).astype(F32)
    bg = rng.random((R, 3), dtype=F32)
    eidx = rng.integers(0, st["edge_pool"].size // 64, NE).astype(np.int32)
    ecoord = (rng.random((NE, 2), dtype=F32) * F32(2.) - F32(1.)).astype(F32)
    d = rt.to_dev(ro, rd, bounds, gt, cam, noise, bg, eidx, ecoord)
    res = []
    for mode in (0, 2):  # 0: read the count back (as Render does); 2: never
        runner, cfg, _ = rt.make_runner(st, "wanjinyou", ["field.log2_table_size=16"], seed=21, table_init=0.5)
        states = [t.cpu().clone() for t in runner.states()]
        states[8][-16 * 64:-15 * 64] *= 16.0  # the density row of the field MLP's output layer: opaque stretches -> early stop
        runner.load_states(states)
        runner.n_edge_pts = NE
        runner.async_counts = mode
        runner.iter_step = 1
        runner.update_ada_params()
        runner.set_forced_randoms(d[5], d[6], d[7], d[8])
        runner.zero_grad()
        stats = runner.train_step(d[0], d[1], d[2], d[3], d[4], False)
        g = {k: v.cpu().numpy().copy() for k, v in runner.grads().items()}
        c = runner.counters()
        res.append((float(stats["loss"]), stats["n_samples"], c["total_meaningful"], g))
    (l0, n0, m0, g0), (l1, n1, m1, g1) = res
    assert n0 == n1 and m0 == m1 and 0.99 * n0 > m0 > 32768, (n0, m0)  # early stop did something; the binned scatter is in play
    assert l0 == l1
    for k in ("color_mlp", "field_mlp", "app_emb"):
        assert rel_err(g1[k], g0[k]) <= 2e-3, (k, rel_err(g1[k], g0[k]))  # (per-block partial sums see the rows in another order)
    a, b = g0["feat_pool"].reshape(-1).astype(np.float64), g1["feat_pool"].reshape(-1).astype(np.float64)
    assert float((a * b).sum() / (np.linalg.norm(a) * np.linalg.norm(b))) > 0.999999
    assert np.abs(a - b).max() <= 2.0 ** -9 * np.abs(a).max()


def test_prefetched_sampling_is_not_used_by_a_render(rt, fox_state):
    """The sampling a training step prefetches for the next batch is made with training noise / fineness: a render of the
    very same ray tensors (or of rays that recycle their address) must sample afresh.  Regression test: a test image
    rendered right after training once in a while came out at 15 dB."""
    st = fox_state
    rng = np.random.default_rng(31)
    torch.manual_seed(31)
    runner, cfg, _ = rt.make_runner(st, "wanjinyou", seed=4)
    runner.n_edge_pts = 512
    R = 512
    ro, rd, bounds, cam = fox_batch(st, rng, R)
    gt = np.tile(np.array([[0.7, 0.4, 0.1]], F32), (R, 1))
    d = rt.to_dev(ro, rd, bounds, gt, cam)
    ro2, rd2, bounds2, _ = fox_batch(st, rng, R)
    n = rt.to_dev(ro2, rd2, bounds2)
    for _ in range(3):
        runner.train_step(d[0], d[1], d[2], d[3], d[4], True, n[0], n[1], n[2])  # prefetches the sampling of `n`
    first = runner.render_rays(n[0], n[1], n[2])[0].clone()
    second = runner.render_rays(n[0], n[1], n[2])[0].clone()
    assert torch.equal(first, second)
    # and the training path still picks its prefetch up: same result as a runner that never prefetched
    twin, _, _ = rt.make_runner(st, "wanjinyou", seed=4)
    twin.n_edge_pts = 512
    assert twin.iter_step == 0 and runner.iter_step == 3


def test_deferred_finiteness_flags_when_prefetching(rt, fox_state):
    """A train_step that is handed the next batch does not wait for its own finiteness flags: the update is predicated on
    the device, and the host reaction (iteration counter, loss scale) arrives with the next step or flush()."""
    st = fox_state
    rng = np.random.default_rng(21)
    torch.manual_seed(21)
    runner, cfg, _ = rt.make_runner(st, "wanjinyou", seed=4)
    runner.n_edge_pts = 512
    R = 512
    ro, rd, bounds, cam = fox_batch(st, rng, R)
    gt = np.tile(np.array([[0.7, 0.4, 0.1]], F32), (R, 1))
    d = rt.to_dev(ro, rd, bounds, gt, cam)
    bad_gt = d[3].clone()
    bad_gt[5, 1] = float("nan")  # -> non-finite loss -> non-finite gradients everywhere
    nxt = (d[0], d[1], d[2])
    s = runner.train_step(d[0], d[1], d[2], d[3], d[4], True, *nxt)
    assert not s["skipped_nan"] and runner.iter_step == 1
    before = [t.clone() for t in runner.states()]
    s = runner.train_step(d[0], d[1], d[2], bad_gt, d[4], True, *nxt)
    assert not s["skipped_nan"] and runner.iter_step == 2  # not known yet
    s = runner.train_step(d[0], d[1], d[2], d[3], d[4], True, *nxt)
    assert s["skipped_nan"] and runner.iter_step == 2  # the bad step was taken back, this one counted
    s = runner.train_step(d[0], d[1], d[2], bad_gt, d[4], True, *nxt)
    assert runner.iter_step == 3
    runner.flush()  # resolves the last step's flags
    assert runner.iter_step == 2
    for t in runner.states():
        if t.dtype.is_floating_point:
            assert torch.isfinite(t).all()
    # the dropped step changed no parameter: one good step from `before` on a twin runner gives the same state
    twin, _, _ = rt.make_runner(st, "wanjinyou", seed=4)
    twin.n_edge_pts = 512
    torch.manual_seed(21)
    twin.train_step(d[0], d[1], d[2], d[3], d[4], True, *nxt)
    twin.flush()
    for a, b in zip(before, twin.states()):
        assert a.shape == b.shape


@pytest.mark.parametrize("preset,overrides", [("llff", []), ("nerf-360", []), ("wanjinyou_big", ["field.log2_table_size=20"]), ("free", [])])
def test_other_presets_train(rt, fox_state, preset, overrides):
    """BASELINE configs 3-5 as plumbing cases: the presets that differ in sampler step (sample_l 1/512), scale_by_dis,
    appearance embedding and table size must train (finite, decreasing loss) through the same fused step."""
    st = fox_state
    rng = np.random.default_rng(15)
    torch.manual_seed(15)  # noise / background / edge draws come from torch's generator
    runner, cfg, _ = rt.make_runner(st, preset, overrides + ["train.learning_rate_warm_up_end_iter=40"], seed=4)
    runner.n_edge_pts = 512
    R = 512
    ro, rd, bounds, cam = fox_batch(st, rng, R)
    gt = np.tile(np.array([[0.7, 0.4, 0.1]], F32), (R, 1))
    d = rt.to_dev(ro, rd, bounds, gt, cam)
    mse, skipped = [], 0
    for it in range(60):
        s = runner.train_step(d[0], d[1], d[2], d[3], d[4], True)
        assert s["n_samples"] > 0
        skipped += bool(s["skipped_nan"])  # a non-finite fp16 gradient drops the iteration (ExpRunner.cpp:131-134); rare
        mse.append(float(s["mse"]))
    assert skipped <= 2 and runner.iter_step == 60 - skipped
    assert np.isfinite(mse).all() and min(mse[-5:]) < 0.8 * mse[0], (preset, mse[0], mse[-5:])
    cols = runner.render_rays(d[0], d[1], d[2])[0]
    assert torch.isfinite(cols).all()


def test_octree_construction_from_cameras(rt, fox_state):
    """the product builds the octree / warps / edge pool from the training cameras on the device.
    The fixture tests/golden/fox_state.npz was built by the oracle restatement of the reference's constructor from the
    same cameras: topology, node numbering, leaf validity and the edge pool must be identical (they do not depend on
    random draws); the warps depend on random points, so they are compared with the oracle's ConstructTrans on the
    same explicit draws."""
    from oracle import octree_construct as ocn
    st = fox_state
    host = rt.host()
    ts = st["train_set"]
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32))
    torch.manual_seed(1)
    built = host.build_octree(T(st["poses"][ts]), T(st["intri"][ts]), T(st["bounds"][ts]), 16, float(1 << 9), 1.5, 4096)
    nodes = built["tree_nodes"].numpy().view(ocn.NODE_DT)
    ref_nodes = st["tree_nodes"].view(ocn.NODE_DT)
    assert len(nodes) == len(ref_nodes), (len(nodes), len(ref_nodes))
    for f in ("center", "side_len", "parent", "childs", "is_leaf_node", "trans_idx"):
        assert (nodes[f] == ref_nodes[f]).all(), f
    assert built["n_volumes"] == int(st["n_volumes"])
    ref_edges = st["edge_pool"].view(ocn.EDGE_DT)
    edges = built["edge_pool"].numpy().view(ocn.EDGE_DT)
    assert len(edges) == len(ref_edges)
    for f in ("t_idx_a", "t_idx_b", "center", "dir_0", "dir_1"):
        assert (edges[f] == ref_edges[f]).all(), f
    assert (host.construct_edge_pool(torch.from_numpy(st["tree_nodes"])).numpy() == st["edge_pool"]).all()
    # warps: same explicit random points and first camera through the product and through the oracle
    rng = np.random.default_rng(3)
    c2w, intri = torch.from_numpy(st["poses"][ts]), torch.from_numpy(st["intri"][ts])
    ctx = ocn._VisiCtx(c2w, intri, torch.from_numpy(st["bounds"][ts]))
    leaves = np.nonzero(ref_nodes["trans_idx"] >= 0)[0]
    for u in leaves[rng.integers(0, len(leaves), 4)]:
        center = torch.from_numpy(ref_nodes["center"][u].copy()); side = float(ref_nodes["side_len"][u])
        visi = ocn.get_visi_cams(ctx, side, center)
        pts = ((torch.from_numpy(rng.random((8192, 3), dtype=F32)) - .5) * side + center[None]).contiguous()
        first = int(rng.integers(0, len(visi)))
        want = ocn.construct_trans(pts, c2w[visi], intri[0], center, None, first_cam=first)
        got = host.construct_trans(pts.cuda(), c2w[visi], intri[0], center, first).numpy().view(ocn.TRANS_DT)[0]
        assert np.abs(got["w2xz"] - want["w2xz"]).max() <= 1e-4 * np.abs(want["w2xz"]).max()
        assert abs(float(got["dis_summary"]) - float(want["dis_summary"])) <= 1e-5 * float(want["dis_summary"])
        for r in range(3):  # principal axes are defined up to sign
            a, b = got["weight"][r], want["weight"][r]
            err = min(np.abs(a - b).max(), np.abs(a + b).max())
            assert err <= 2e-3 * np.abs(b).max(), (r, err, np.abs(b).max())
    # a runner on the freshly built scene trains
    torch.manual_seed(2)
    runner, cfg, _ = rt.make_runner_from_cameras(st["poses"], st["intri"], st["bounds"], ts, "wanjinyou",
                                                 ["field.log2_table_size=16", "train.learning_rate_warm_up_end_iter=10"])
    assert runner.n_nodes() == len(ref_nodes) and runner.n_volumes == int(st["n_volumes"])
    runner.n_edge_pts = 512
    rng = np.random.default_rng(5)
    ro, rd, bounds, cam = fox_batch(st, rng, 512)
    gt = np.tile(np.array([[0.3, 0.6, 0.2]], F32), (512, 1))
    d = rt.to_dev(ro, rd, bounds, gt, cam)
    mse = [float(runner.train_step(d[0], d[1], d[2], d[3], d[4], True)["mse"]) for _ in range(40)]
    assert np.isfinite(mse).all() and mse[-1] < 0.7 * mse[0], (mse[0], mse[-1])


def test_launcher_train_test_render_path(rt, tmp_path):
    """f2_nerf_amd.run == scripts/run.py + main.cpp + ExpRunner::Execute of the reference, on a data directory in the
    reference's layout (cams_meta.npy, images_4/*.png, poses_render.npy) written from a small synthetic forward-facing rig:
    mode=train leaves checkpoints (renderer.pt + scalars.pt, `latest` links), train_info.txt and test_images/info.yaml;
    mode=test with is_continue reproduces the PSNR from the checkpoint; mode=render_path writes the novel views; mode=render_all
    the side-by-side images of every view."""
    import yaml
    from PIL import Image
    from f2_nerf_amd import rigs, run
    rng = np.random.default_rng(2)
    meta, hw = rigs.forward_facing(rng, n_side=(5, 4), hw=(48, 64), focal=56.0)
    meta[:, 12:14] *= 4.0; meta[:, 14] *= 4.0; meta[:, 16:18] *= 4.0  # intrinsics on disk refer to the factor-1 images (Dataset.cpp:49)
    data = tmp_path / "data" / "synth" / "rig"
    (data / "images_4").mkdir(parents=True)
    np.save(data / "cams_meta.npy", meta)
    for i in range(len(meta)):
        Image.fromarray(rng.integers(0, 255, (48, 64, 3), dtype=np.uint8)).save(data / "images_4" / ("%03d.png" % i))
    np.save(data / "poses_render.npy", meta[:3, :12].reshape(3, 3, 4))
    common = ["--config-name=llff", "dataset_name=synth", "case_name=rig", "exp_name=t", "+work_dir=%s" % tmp_path,
              "field.log2_table_size=14", "train.end_iter=60", "train.save_freq=30", "train.learning_rate_warm_up_end_iter=10",
              "pts_sampler.sub_div_milestones=[20]", "pts_sampler.compact_freq=25", "train.pts_batch_size=32768"]
    assert run.main(common + ["mode=train"]) == 0
    exp = tmp_path / "exp" / "rig" / "t"
    assert (exp / "checkpoints" / "00000030" / "renderer.pt").exists() and (exp / "checkpoints" / "00000060" / "scalars.pt").exists()
    assert os.path.realpath(exp / "checkpoints" / "latest" / "renderer.pt") == str(exp / "checkpoints" / "00000060" / "renderer.pt")
    assert (exp / "train_info.txt").exists() and (exp / "record" / "runtime_config.yaml").exists()
    info = yaml.safe_load(open(exp / "test_images" / "info.yaml"))
    assert set(info) == {"0", "8", "16", "mean_psnr"} and np.isfinite(info["mean_psnr"])
    assert run.main(common + ["mode=test", "is_continue=true"]) == 0
    info2 = yaml.safe_load(open(exp / "test_images" / "info.yaml"))
    assert abs(info2["mean_psnr"] - info["mean_psnr"]) < 1e-3  # same weights, same octree, deterministic evaluation
    assert run.main(common + ["mode=render_path", "is_continue=true"]) == 0
    frames = sorted(os.listdir(exp / "novel_images"))
    assert frames == ["60_000.png", "60_001.png", "60_002.png"]
    assert Image.open(exp / "novel_images" / frames[0]).size == (3 * 64, 48)
    # mode=render_all (ExpRunner::RenderAllImages): every image of the data set as VisualizeImage writes it -- gt | colours |
    # first-hit disparity | disparity side by side
    assert run.main(common + ["mode=render_all", "is_continue=true"]) == 0
    imgs = sorted(os.listdir(exp / "images"))
    assert len(imgs) == len(meta) and all(f.startswith("60_") for f in imgs)
    assert Image.open(exp / "images" / imgs[0]).size == (4 * 64, 48)


def test_bucketed_table_exchange_on_one_gpu(rt, fox_state):
    """The bucketed table exchange of a data-parallel step (host/DataParallel.cpp: the scatter's owner launch cut into four launches,
    each range's all-reduce started from the library's callback while the next owners run) through RCCL with a one-rank world, at a
    batch that takes the owner-binned scatter: parameters after three steps equal a hook-less run bit for bit (an average over one
    rank is the identity), pipelined and blocking, and the callbacks did fire (3 steps x 4 buckets)."""
    import socket
    import torch.distributed as dist
    from f2_nerf_amd import parallel, runtime
    st = fox_state
    rng = np.random.default_rng(5)
    batches = [runtime.to_dev(*runtime.synthetic_ray_batch(st, 2048, rng)) for _ in range(4)]
    sock = socket.socket(); sock.bind(("127.0.0.1", 0)); port = sock.getsockname()[1]; sock.close()
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        outs = {}
        for mode in ("none", "native", "native-blocking"):
            runner, cfg, _ = rt.make_runner(st, "wanjinyou", ["field.log2_table_size=15"], seed=3, table_init=0.3)
            torch.manual_seed(9)
            if mode != "none":
                parallel.attach(runner, 15, overlap=(mode == "native"), native=True, hooks_for_one_rank=True)
            for i in range(3):
                b, nb = batches[i], batches[i + 1]
                s = runner.train_step(b[0], b[1], b[2], b[3], b[4], True, nb[0], nb[1], nb[2])
                assert s["n_samples"] >= 32768, s["n_samples"]  # (the owner-binned scatter: F2N_BIN_MIN_N)
            outs[mode] = ([t.clone() for t in runner.states()], runner.dp_bucket_callbacks())
            if mode != "none":  # (round 6: every step's small gradient buffers left from the step's tail chain, beside the scatter)
                assert runner.dp_small_exchanges_early() == 3, (mode, runner.dp_small_exchanges_early())
            del runner
        for mode in ("native", "native-blocking"):
            assert outs[mode][1] == 12, (mode, outs[mode][1])
            for a, b in zip(outs["none"][0], outs[mode][0]):
                assert torch.equal(a, b), mode
        assert outs["none"][1] == 0
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("log2", [15, 21])
def test_fused_step_tail_equals_separate_launches(rt, fox_state, log2):
    """The step's tail inside the field backward's call (round 6; ExpRunner.fused_tail, f2n_field_bwd_step_tail: deferred reductions,
    finiteness flags and the small groups' Adam on the tail stream beside the scatter's producers, the table's Adam in the scatter's
    owner blocks) against the separate launches behind the scatter (reduce -> flags -> f2n_adam_fused): streaming steps at a batch
    that takes the owner-binned scatter and at one that does not, one of them with a non-finite loss (dropped on the device, taken
    back by the host one step later) -- every state tensor, the iteration counter and the loss scales end identical."""
    from f2_nerf_amd import runtime
    st = fox_state
    rng = np.random.default_rng(15)
    big = [runtime.to_dev(*runtime.synthetic_ray_batch(st, 2048, rng)) for _ in range(5)]
    small = [runtime.to_dev(*runtime.synthetic_ray_batch(st, 128, rng)) for _ in range(3)]
    bad_gt = big[2][3].clone()
    bad_gt[5, 1] = float("nan")
    outs = {}
    for fused in (True, False):
        runner, cfg, _ = rt.make_runner(st, "wanjinyou", ["field.log2_table_size=%d" % log2], seed=3, table_init=0.3)  # (21: 8704 owner blocks)
        torch.manual_seed(9)
        runner.fused_tail = fused
        runner.exact_flag_order = True  # (the fused tail reads the previous step's flags in front of the backward: ask the other for the same)
        losses = []
        for i in range(4):
            b, nb = big[i], big[i + 1]
            s = runner.train_step(b[0], b[1], b[2], bad_gt if i == 2 else b[3], b[4], True, nb[0], nb[1], nb[2])
            assert s["n_samples"] >= 32768, s["n_samples"]  # (the owner-binned scatter: F2N_BIN_MIN_N)
            losses.append(float(s["loss"]))
        runner.flush()
        mid = ([t.clone() for t in runner.states()], runner.iter_step)
        for i in range(2):  # small batches: the scatter's atomics, the ordinary table pass behind them
            b, nb = small[i], small[i + 1]
            s = runner.train_step(b[0], b[1], b[2], b[3], b[4], True, nb[0], nb[1], nb[2])
            assert 0 < s["n_samples"] < 32768
            losses.append(float(s["loss"]))
        runner.flush()
        outs[fused] = (mid, [t.clone() for t in runner.states()], runner.iter_step, losses, {k: v.clone() for k, v in runner.grads().items()})
        del runner
    a, b = outs[True], outs[False]
    assert a[0][1] == b[0][1] == 3 and a[2] == b[2] == 5  # four steps, one taken back; two more
    assert [x for x in a[3][:4] if x == x] == [x for x in b[3][:4] if x == x] and sum(x != x for x in a[3]) == 1
    for x, y in zip(a[0][0], b[0][0]):  # behind the owner-binned steps: identical
        assert torch.equal(x, y)
    # behind the small batches: their scatter adds f16 atomics in arrival order, two runs of ONE program differ in low-order bits
    for x, y in zip(a[1], b[1]):
        if x.dtype.is_floating_point:
            assert float((x.double() - y.double()).abs().max()) <= 1e-4
        else:
            assert torch.equal(x, y)
    assert max(abs(u - v) for u, v in zip(a[3][4:], b[3][4:])) <= 1e-5
    for k in a[4]:
        assert float(a[4][k].abs().sum()) == 0.0 and float(b[4][k].abs().sum()) == 0.0, k  # every gradient buffer consumed and cleared


def test_two_rank_bench_when_two_gpus_are_visible():
    """bench.py --gpus 2 through its own torch.distributed.run launch: two ranks over RCCL (the native communicator of
    csrc/host/DataParallel.cpp), replicas bit-identical after 20 pipelined steps.  Needs two devices: on a one-GPU lease
    it skips and says so (no multi-rank measurement exists for this code: DESIGN.md section 5)."""
    import json
    import subprocess
    import sys
    n = torch.cuda.device_count()
    if n < 2:
        pytest.skip("only %d HIP device(s) visible: the two-rank RCCL run cannot execute on this lease" % n)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "3", "--no-converged",
                        "--no-cpu-baseline"], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["config"]["parallelism"] == "ray-dp2"
    rep = line["replicas"]
    assert rep["rccl_comm_ranks"] == 2 and rep["torch_distributed_world"] == 2, rep
    assert rep["identical"], rep
    assert line["value"] > 0 and line["ms_per_step"] > 0


def test_aux_states_and_deferred_reset(rt, fox_state):
    """Round-2 advisor items on the device: (1) what a data-parallel attach replicates besides the checkpoint vector -- the
    edge pool and the training cameras -- round-trips through aux_states / load_aux_states (a replica that kept its own edge
    pool would index rank 0's warps); (2) partial sums a failed backward left registered are dropped by the next ZeroGrad
    (f2n_deferred_reset) instead of being folded into the next step's gradients."""
    import f2_nerf_amd  # noqa: F401
    from f2_nerf_amd import capi
    st = fox_state
    runner, cfg, _ = rt.make_runner(st, "wanjinyou", ["field.log2_table_size=14"], seed=3, table_init=0.3)
    aux = [t.clone() for t in runner.aux_states()]
    assert aux[0].numel() == st["edge_pool"].size and (aux[0].cpu().numpy() == st["edge_pool"]).all()
    assert aux[1].shape[0] == len(st["train_set"]) and aux[2].shape == (len(st["train_set"]), 3, 3)
    other, _, _ = rt.make_runner(st, "wanjinyou", ["field.log2_table_size=14"], seed=4, table_init=0.3)
    other.set_edge_pool(torch.from_numpy(st["edge_pool"][:64 * 10].copy()))  # a replica with another (shorter) edge pool
    assert other.aux_states()[0].numel() == 640
    other.load_states(runner.states())
    other.load_aux_states(aux)
    for a, b in zip(other.aux_states(), aux):
        assert a.shape == b.shape and torch.equal(a, b)
    # (2) a stale deferred reduction must not survive a ZeroGrad
    rng = np.random.default_rng(2)
    R = 256
    ro, rd, bounds, cam = fox_batch(st, rng, R)
    d = rt.to_dev(ro, rd, bounds, rng.random((R, 3), dtype=F32), cam)
    runner.n_edge_pts = 256
    fr = rt.to_dev((((rng.random(1024 + R + 10, dtype=F32) - F32(.5)) + F32(1.)) * F32(float(runner.fineness))).astype(F32),
                   rng.random((R, 3), dtype=F32), rng.integers(0, st["edge_pool"].size // 64, 256).astype(np.int32),
                   (rng.random((256, 2), dtype=F32) * F32(2.) - F32(1.)).astype(F32))
    runner.set_forced_randoms(*fr)  # (the same draws in both steps below)
    runner.zero_grad()
    runner.train_step(d[0], d[1], d[2], d[3], d[4], False)
    g0 = {k: v.clone() for k, v in runner.grads().items()}
    poison = torch.full((7168,), 1e6, device="cuda")
    lib = capi.lib()
    import ctypes
    # a deferring backward whose step "threw" before f2n_reduce_deferred: emulate by calling the deferring entry point directly
    n = 64
    x = torch.zeros((n, 32), dtype=torch.float16, device="cuda")
    drgb = torch.ones((n, 3), device="cuda")
    dfeat = torch.zeros((n, 16), device="cuda")
    ph = torch.zeros(7168, dtype=torch.float16, device="cuda")
    rc = lib.f2n_shade_bwd_dyn(ctypes.c_void_p(torch.cuda.current_stream().cuda_stream), n, ctypes.c_void_p(0), ctypes.c_void_p(drgb.data_ptr()),
                               ctypes.c_void_p(0), ctypes.c_void_p(ph.data_ptr()), ctypes.c_void_p(x.data_ptr()), ctypes.c_float(128.0),
                               ctypes.c_void_p(dfeat.data_ptr()), ctypes.c_void_p(poison.data_ptr()), ctypes.c_void_p(0), 0, ctypes.c_void_p(0), 1)
    assert rc == 0
    runner.zero_grad()   # -> f2n_deferred_reset
    runner.async_counts = 2  # a streaming step: its backward defers its reductions and folds "everything registered" in one launch
    runner.train_step(d[0], d[1], d[2], d[3], d[4], False)
    runner.async_counts = 1
    runner.flush()
    torch.cuda.synchronize()
    assert float(poison.max()) == 1e6 and float(poison.min()) == 1e6  # the stale registration was never folded
    g1 = runner.grads()
    for k in ("color_mlp", "field_mlp"):
        assert float((g0[k] - g1[k]).abs().max()) <= 2e-3 * float(g0[k].abs().max()) + 1e-9, k


def test_prefetched_samples_do_not_survive_a_state_load(rt, fox_state):
    """A streaming step prefetches the NEXT batch's samples against the octree of the moment.  If the octree is replaced before
    that batch is trained on (load_states / install_octree between two steps), the prefetched samples belong to node indices
    that no longer exist: the renderer must notice (PersOctree::generation_) and sample again."""
    st = fox_state
    rng = np.random.default_rng(12)
    R = 512
    runner, cfg, _ = rt.make_runner(st, "wanjinyou", ["field.log2_table_size=14"], seed=2, table_init=0.3)
    runner.n_edge_pts = 256
    fr = rt.to_dev((((rng.random(1024 + R + 10, dtype=F32) - F32(.5)) + F32(1.)) * F32(float(runner.fineness))).astype(F32),
                   rng.random((R, 3), dtype=F32), rng.integers(0, st["edge_pool"].size // 64, 256).astype(np.int32),
                   (rng.random((256, 2), dtype=F32) * F32(2.) - F32(1.)).astype(F32))
    runner.set_forced_randoms(*fr)
    b = []
    for _ in range(2):
        ro, rd, bounds, cam = fox_batch(st, rng, R)
        b.append(rt.to_dev(ro, rd, bounds, rng.random((R, 3), dtype=F32), cam))
    for mode in (1, 0):  # speculative prefetch / prefetch behind the update
        runner.speculative_sampling = mode
        runner.iter_step = 1  # (iteration 0 compacts the octree: no speculation in that one)
        runner.update_ada_params()
        runner.train_step(b[0][0], b[0][1], b[0][2], b[0][3], b[0][4], True, b[1][0], b[1][1], b[1][2])  # prefetches batch 1
        n_old = int(runner.get_samples(b[1][0], b[1][1], b[1][2])["t"].shape[0])
        z = dict(np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "data", "converged_sampler.npz")))
        states = [t.cpu().clone() for t in runner.states()]
        saved = [t.clone() for t in states]
        states[0] = torch.from_numpy(z["tree_nodes"].copy())                       # the 148 k-node tree of a finished training
        states[2] = torch.zeros(z["tree_nodes"].size // 64, dtype=torch.int32)
        runner.load_states(states)
        n_new = int(runner.get_samples(b[1][0], b[1][1], b[1][2])["t"].shape[0])
        assert n_new != n_old
        s = runner.train_step(b[1][0], b[1][1], b[1][2], b[1][3], b[1][4], False)
        assert s["n_samples"] == n_new, (mode, s["n_samples"], n_new, n_old)
        runner.load_states(saved)
