"""GPU parity tests (run with `-m gpu` on an MI355X): every hot-path kernel is called THROUGH THE C-ABI
(f2-nerf_amd/capi.py -> libf2n_hip.so) and compared with the CPU oracle on the same seeded inputs and with the
committed golden vectors.  Integer / index outputs are bit-exact; fp32 sampler outputs are bit-exact (both
sides are built with -ffp-contract=off and IEEE div/sqrt); fp16 hash features are bit-exact; MLP-dependent
values use the tolerances stated next to each assert (north-star contract: rendered RGB within 1e-3)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import capi as oc  # noqa: E402
from oracle import pipeline as op  # noqa: E402

DEV = "cuda:0"
F32 = np.float32


@pytest.fixture(scope="module")
def hip():
    if not torch.cuda.is_available():
        pytest.fail("no HIP device visible: GPU tests must run on the MI355X box")
    import f2_nerf_amd  # noqa: F401
    from f2_nerf_amd import capi
    capi.lib()  # fails loudly if the native library is missing
    return capi


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def N(t):
    return t.detach().cpu().numpy()


def bits(a):
    a = np.ascontiguousarray(a)
    return a.view(np.uint32) if a.dtype == np.float32 else a


def assert_same(a, b, what=""):
    assert a.shape == b.shape, (what, a.shape, b.shape)
    bad = bits(a) != bits(b)
    assert not bad.any(), "%s: %d / %d elements differ" % (what, int(bad.sum()), bad.size)


# ---------------------------------------------------------------------------------------------------
# sampler
# ---------------------------------------------------------------------------------------------------
def gpu_sample(hip, st, rays_o, rays_d, noise, sample_l, scale_by_dis, near=0.01, far=1e8, max_hits=1024):
    R = rays_o.shape[0]
    so, tn, tr = T(st["search_order"]), T(st["tree_nodes"]), T(st["pers_trans"])
    ro, rd, nz = T(rays_o), T(rays_d), T(noise)
    cnt = torch.empty(R, dtype=torch.int32, device=DEV)
    se = torch.empty((R, 2), dtype=torch.int32, device=DEV)
    tot = torch.zeros(1, dtype=torch.int32, device=DEV)
    hip.oct_intersect_count(R, max_hits, so, ro, rd, near, far, tn, cnt)
    hip.segment_scan(R, cnt, se, tot)
    K = int(tot.item())
    oidx = torch.empty(max(K, 1), dtype=torch.int32, device=DEV)
    onf = torch.empty((max(K, 1), 2), dtype=torch.float32, device=DEV)
    hip.oct_intersect_fill(R, so, ro, rd, near, far, tn, se, oidx, onf)
    pcnt = torch.empty(R, dtype=torch.int32, device=DEV)
    pse = torch.empty((R, 2), dtype=torch.int32, device=DEV)
    hip.ray_march_count(R, sample_l, scale_by_dis, ro, rd, nz, se, oidx, onf, tn, tr, pcnt)
    hip.segment_scan(R, pcnt, pse, tot)
    n = int(tot.item())
    m = max(n, 1)
    out = dict(pts=torch.empty((m, 3), device=DEV), dirs=torch.empty((m, 3), device=DEV), dt=torch.empty(m, device=DEV),
               t=torch.empty(m, device=DEV), anchors=torch.zeros((m, 3), dtype=torch.int32, device=DEV),
               first_oct_dis=torch.empty(R, device=DEV))
    hip.ray_march_fill(R, sample_l, scale_by_dis, ro, rd, nz, se, oidx, onf, tn, tr, pse, out["pts"], out["dirs"],
                       out["dt"], out["t"], out["anchors"], out["first_oct_dis"])
    torch.cuda.synchronize()
    res = {k: N(v)[:n] if k != "first_oct_dis" else N(v).reshape(R, 1) for k, v in out.items()}
    res["pts_idx_bounds"] = N(pse)
    return (N(se), N(oidx)[:K], N(onf)[:K]), res


def fox_rays(st, rng, n):
    cam = st["train_set"][rng.integers(0, len(st["train_set"]), n)].astype(np.int32)
    pose = st["poses"][cam]
    K = st["intri"][cam]
    i = rng.integers(0, 960, n).astype(F32) + F32(.5)
    j = rng.integers(0, 540, n).astype(F32) + F32(.5)
    d_cam = np.stack([(j - K[:, 0, 2]) / K[:, 0, 0], -(i - K[:, 1, 2]) / K[:, 1, 1], -np.ones(n, F32)], -1).astype(F32)
    d = np.einsum("nij,nj->ni", pose[:, :3, :3], d_cam).astype(F32)
    d = (d / np.linalg.norm(d, axis=-1, keepdims=True)).astype(F32)
    return np.ascontiguousarray(pose[:, :3, 3]).astype(F32), d, cam


def test_sampler_golden(hip, fox_state, fox_golden):
    g = fox_golden
    hits, m = gpu_sample(hip, fox_state, g["rays_o"], g["rays_d"], g["noise"], float(g["sample_l"]), True)
    assert_same(hits[0], g["oct_start_end"], "oct_start_end")
    assert_same(hits[1], g["oct_idx"], "oct_idx")
    assert_same(hits[2], g["oct_near_far"], "oct_near_far")
    for k in ("pts", "dirs", "dt", "t", "anchors", "pts_idx_bounds", "first_oct_dis"):
        assert_same(m[k], g["march_" + k], k)


@pytest.mark.parametrize("seed,fineness,scale_by_dis,max_hits,n", [(1, 8.0, True, 1024, 1500), (2, 2.0, False, 1024, 700),
                                                                    (3, 16.0, True, 5, 513)])
def test_sampler_vs_oracle(hip, fox_state, seed, fineness, scale_by_dis, max_hits, n):
    st = fox_state
    rng = np.random.default_rng(seed)
    o, d, _ = fox_rays(st, rng, n)
    d[0] = [1, 0, 0]; d[1] = [0, -1, 0]; d[2] = [0, 0, 1]; o[3] = [1e4, 1e4, 1e4]; d[3] = [1, 0, 0]
    noise = ((rng.random(1024 + n + 10, dtype=F32) - F32(.5)) + F32(1.)) * F32(fineness)
    hits, m = gpu_sample(hip, st, o, d, noise, 1. / 256., scale_by_dis, max_hits=max_hits)
    ref_hits = oc.oct_intersect(st["search_order"], o, d, 0.01, 1e8, st["tree_nodes"], max_hits)
    for a, b, w in zip(hits, ref_hits, ("se", "idx", "nf")):
        assert_same(a, b, w)
    ref = oc.ray_march(o, d, noise, 1. / 256., scale_by_dis, *ref_hits, st["tree_nodes"], st["pers_trans"])
    for k in ref:
        assert_same(m[k], ref[k], k)
    # single-pass strided variant: same per-ray leaf lists in fixed slots, and the march accepts that layout
    so, tn, tr = T(st["search_order"]), T(st["tree_nodes"]), T(st["pers_trans"])
    se2 = torch.zeros((n, 2), dtype=torch.int32, device=DEV)
    oi2 = torch.zeros(n * max_hits, dtype=torch.int32, device=DEV)
    nf2 = torch.zeros((n * max_hits, 2), device=DEV)
    tot = torch.zeros(1, dtype=torch.int32, device=DEV)
    tr2 = torch.full((n * max_hits,), -9, dtype=torch.int32, device=DEV)
    n_nodes = st["tree_nodes"].size // 64
    cb = torch.zeros(n_nodes * 8 * 32, dtype=torch.uint8, device=DEV)  # the one-read-per-node view of the tree
    hip.oct_build_child_blocks(n_nodes, tn, cb)
    hip.oct_intersect_strided(n, max_hits, so, T(o), T(d), 0.01, 1e8, tn, se2, oi2, nf2, tot, tr2, cb if seed != 3 else None)
    se2n, oi2n, nf2n = N(se2), N(oi2), N(nf2)
    rse = ref_hits[0]
    assert int(tot.item()) == len(ref_hits[1])
    assert (se2n[:, 0] == np.arange(n) * max_hits).all() and ((se2n[:, 1] - se2n[:, 0]) == (rse[:, 1] - rse[:, 0])).all()
    for r in range(0, n, 37):
        c = rse[r, 1] - rse[r, 0]
        assert_same(oi2n[r * max_hits:r * max_hits + c], ref_hits[1][rse[r, 0]:rse[r, 1]], "strided idx")
        assert_same(nf2n[r * max_hits:r * max_hits + c], ref_hits[2][rse[r, 0]:rse[r, 1]], "strided near/far")
        node_trans = st["tree_nodes"].view(np.int32).reshape(-1, 16)[:, 14]  # TreeNode.trans_idx @56
        assert_same(N(tr2)[r * max_hits:r * max_hits + c], node_trans[oi2n[r * max_hits:r * max_hits + c]], "strided trans idx")
    # the same walk out of LDS-resident child records (interior nodes only): every output word identical
    words = st["tree_nodes"].view(np.int32).reshape(-1, 16)
    is_interior = (words[:, 5:13] >= 0).any(1)
    assert 1 <= int(is_interior.sum()) <= hip.oct_lds_max_interior()
    interior_nodes = np.nonzero(is_interior)[0].astype(np.int32)
    rank_of = (np.cumsum(is_interior) - 1).astype(np.int32)
    se3 = torch.zeros_like(se2); oi3 = torch.zeros_like(oi2); nf3 = torch.zeros_like(nf2); tr3 = torch.full_like(tr2, -9)
    tot3 = torch.zeros(1, dtype=torch.int32, device=DEV)
    hip.oct_intersect_strided_lds(n, max_hits, so, T(o), T(d), 0.01, 1e8, tn, se3, oi3, nf3, tot3, tr3, cb, T(interior_nodes), T(rank_of))
    assert int(tot3.item()) == int(tot.item())
    assert_same(N(se3), se2n, "LDS walk: segments")
    used = np.zeros(n * max_hits, dtype=bool)
    for r in range(n):
        used[r * max_hits:r * max_hits + (se2n[r, 1] - se2n[r, 0])] = True
    assert_same(N(oi3)[used], oi2n[used], "LDS walk: leaf lists")
    assert_same(N(nf3)[used], nf2n[used], "LDS walk: near / far")
    assert_same(N(tr3)[used], N(tr2)[used], "LDS walk: trans idx")
    pcnt = torch.zeros(n, dtype=torch.int32, device=DEV)
    hip.ray_march_count(n, 1. / 256., scale_by_dis, T(o), T(d), T(noise), se2, oi2, nf2, tn, tr, pcnt)
    assert_same(N(pcnt), (ref["pts_idx_bounds"][:, 1] - ref["pts_idx_bounds"][:, 0]).astype(np.int32), "march count on strided hits")
    # single-pass march (strided slots + pack) == the count / scan / fill triple, bit for bit
    S = 1024
    cnt2 = torch.zeros(n, dtype=torch.int32, device=DEV)
    s_pts = torch.zeros((n * S, 3), device=DEV); s_dt = torch.zeros(n * S, device=DEV); s_t = torch.zeros(n * S, device=DEV)
    s_an = torch.zeros((n * S, 2), dtype=torch.int32, device=DEV); fod = torch.zeros(n, device=DEV)
    hip.ray_march_strided(n, 1. / 256., scale_by_dis, T(o), T(d), T(noise), se2, oi2, nf2, tn, tr, cnt2, s_pts, s_dt, s_t, s_an, fod,
                          tr2 if seed != 2 else None)
    assert_same(N(cnt2), N(pcnt), "single-pass counts")
    pse2 = torch.zeros((n, 2), dtype=torch.int32, device=DEV)
    hip.segment_scan(n, cnt2, pse2, tot)
    tot_n = int(tot.item()); mm = max(tot_n, 1)
    packed = dict(pts=torch.zeros((mm, 3), device=DEV), dirs=torch.zeros((mm, 3), device=DEV), dt=torch.zeros(mm, device=DEV),
                  t=torch.zeros(mm, device=DEV), anchors=torch.full((mm, 3), -7, dtype=torch.int32, device=DEV))
    hip.pack_samples(n, pse2, T(o), T(d), tr, s_pts, s_dt, s_t, s_an, packed["pts"], packed["dirs"], packed["dt"], packed["t"],
                     packed["anchors"])
    assert_same(N(pse2), ref["pts_idx_bounds"], "single-pass bounds")
    for k in ("pts", "dirs", "dt", "t", "anchors"):
        assert_same(N(packed[k])[:tot_n], ref[k], "single-pass " + k)
    # without a slot buffer for the warped points: the pack computes them (same bits)
    hip.ray_march_strided(n, 1. / 256., scale_by_dis, T(o), T(d), T(noise), se2, oi2, nf2, tn, tr, cnt2, None, s_dt, s_t, s_an, fod, tr2)
    packed["pts"].zero_()
    hip.pack_samples(n, pse2, T(o), T(d), tr, None, s_dt, s_t, s_an, packed["pts"], packed["dirs"], packed["dt"], packed["t"],
                     packed["anchors"])
    for k in ("pts", "dirs", "dt", "t", "anchors"):
        assert_same(N(packed[k])[:tot_n], ref[k], "pack-computed " + k)
    assert_same(N(fod).reshape(n, 1), ref["first_oct_dis"], "single-pass first_oct_dis")


def test_sampler_sample_cap(hip, fox_state):
    """Maximum size: with a very fine step every ray runs into the 1024-samples-per-ray cap (PersSampler.cu:9); counts,
    bounds and the samples themselves must still match the oracle bit for bit."""
    st = fox_state
    rng = np.random.default_rng(9)
    n = 64
    o, d, _ = fox_rays(st, rng, n)
    noise = ((rng.random(1024 + n + 10, dtype=F32) - F32(.5)) + F32(1.)) * F32(0.02)
    hits, m = gpu_sample(hip, st, o, d, noise, 1. / 256., True)
    ref_hits = oc.oct_intersect(st["search_order"], o, d, 0.01, 1e8, st["tree_nodes"], 1024)
    ref = oc.ray_march(o, d, noise, 1. / 256., True, *ref_hits, st["tree_nodes"], st["pers_trans"])
    per_ray = ref["pts_idx_bounds"][:, 1] - ref["pts_idx_bounds"][:, 0]
    assert per_ray.max() == 1024 and (per_ray == 1024).mean() > 0.5
    for k in ref:
        assert_same(m[k], ref[k], k)


def test_sampler_full_size_properties(hip, fox_state):
    """BASELINE config 2 size (8192 rays): size-independent properties + run-to-run determinism."""
    st = fox_state
    rng = np.random.default_rng(7)
    R = 8192
    o, d, _ = fox_rays(st, rng, R)
    noise = ((rng.random(1024 + R + 10, dtype=F32) - F32(.5)) + F32(1.)) * F32(16.0)
    hits, m = gpu_sample(hip, st, o, d, noise, 1. / 256., True)
    hits2, m2 = gpu_sample(hip, st, o, d, noise, 1. / 256., True)
    for k in m:
        assert_same(m[k], m2[k], "determinism " + k)
    se = m["pts_idx_bounds"]
    assert se[0, 0] == 0 and (se[1:, 0] == se[:-1, 1]).all() and (se[:, 1] >= se[:, 0]).all()
    assert se[-1, 1] == len(m["t"]) and (se[:, 1] - se[:, 0]).max() <= 1024
    # t strictly increases along every ray; samples lie inside the leaf they are anchored to
    ray_of = np.repeat(np.arange(R), se[:, 1] - se[:, 0])
    same_ray = ray_of[1:] == ray_of[:-1]
    assert (np.diff(m["t"])[same_ray] > 0).all()
    from oracle.octree_construct import NODE_DT
    nodes = st["tree_nodes"].view(NODE_DT)
    nd = nodes[m["anchors"][:, 1]]
    assert (nd["trans_idx"] == m["anchors"][:, 0]).all()
    world = o[ray_of] + d[ray_of] * m["t"][:, None]
    assert (np.abs(world - nd["center"]).max(-1) <= nd["side_len"] * 0.5 * (1 + 1e-4) + 1e-5).all()
    # a subset checked bit-exactly against the oracle
    sub = slice(0, 512)
    ref_hits = oc.oct_intersect(st["search_order"], o[sub], d[sub], 0.01, 1e8, st["tree_nodes"])
    ref = oc.ray_march(o[sub], d[sub], noise[:1024 + 512 + 10], 1. / 256., True, *ref_hits, st["tree_nodes"], st["pers_trans"])
    # the noise window of ray r is noise[r : r+1024] on both sides, independent of the batch size
    n_sub = int(se[511, 1])
    assert_same(m["t"][:n_sub], ref["t"], "t subset")
    assert_same(m["pts"][:n_sub], ref["pts"], "pts subset")


def test_normalize_dirs(hip):
    rng = np.random.default_rng(17)
    d = (rng.standard_normal((5000, 3)) * np.exp(rng.standard_normal((5000, 1)) * 3)).astype(F32)
    out = torch.empty((5000, 3), device=DEV)
    hip.normalize_dirs(5000, T(d), out)
    assert_same(N(out), oc.normalize_dirs(d), "normalized dirs")


def test_segment_scan(hip):
    rng = np.random.default_rng(0)
    for n in (1, 63, 64, 4096, 4097, 8192, 100003):
        c = rng.integers(0, 1000, n).astype(np.int32)
        se = torch.empty((n, 2), dtype=torch.int32, device=DEV)
        tot = torch.zeros(1, dtype=torch.int32, device=DEV)
        hip.segment_scan(n, T(c), se, tot)
        end = np.cumsum(c).astype(np.int32)
        assert_same(N(se)[:, 1], end)
        assert_same(N(se)[:, 0], (end - c).astype(np.int32))
        assert int(tot.item()) == int(end[-1])
    tot = torch.full((1,), 5, dtype=torch.int32, device=DEV)
    hip.segment_scan(0, torch.empty(1, dtype=torch.int32, device=DEV), torch.empty((1, 2), dtype=torch.int32, device=DEV), tot)
    assert int(tot.item()) == 0  # empty input


def test_segment_scan_and_flags_write_their_host_mirror(hip):
    """f2n_segment_scan_ex / f2n_nonfinite_flags_ex: the launch that produces a count (or the finiteness flags) also writes
    the host's copy into mapped host memory -- what the host reads behind an event instead of queueing a copy."""
    rng = np.random.default_rng(3)
    for n in (1, 4097, 100003):
        c = rng.integers(0, 1000, n).astype(np.int32)
        se = torch.empty((n, 2), dtype=torch.int32, device=DEV)
        tot = torch.zeros(1, dtype=torch.int32, device=DEV)
        also = torch.tensor([41, 42], dtype=torch.int32, device=DEV)
        mirror = torch.full((3,), -1, dtype=torch.int32).pin_memory()
        hip.segment_scan_ex(n, T(c), se, tot, mirror, also)
        torch.cuda.synchronize()
        assert mirror.tolist() == [41, 42, int(c.sum())] and int(tot.item()) == int(c.sum())
        assert_same(N(se)[:, 1], np.cumsum(c).astype(np.int32))
        mirror1 = torch.full((1,), -1, dtype=torch.int32).pin_memory()
        hip.segment_scan_ex(n, T(c), se, tot, mirror1)
        torch.cuda.synchronize()
        assert mirror1.tolist() == [int(c.sum())]
    a = rng.standard_normal(3072).astype(F32); b = rng.standard_normal(7168).astype(F32)
    b[17] = np.inf
    flags = torch.full((3,), 7, dtype=torch.int32, device=DEV)
    mirror = torch.full((4,), 9, dtype=torch.int32).pin_memory()
    hip.nonfinite_flags(a.size, T(a), b.size, T(b), flags, mirror)
    torch.cuda.synchronize()
    assert N(flags).tolist() == [0, 1, 1] and mirror.tolist() == [0, 1, 1, 9]


def test_edge_samples_and_occupancy(hip, fox_state, fox_golden):
    st, g = fox_state, fox_golden
    n = len(g["edge_idx"])
    pts = torch.empty((n, 2, 3), device=DEV)
    idx = torch.empty((n, 2), dtype=torch.int32, device=DEV)
    hip.edge_samples(n, T(st["edge_pool"]), T(st["pers_trans"]), T(g["edge_idx"]), T(g["edge_coord"]), pts, idx)
    assert_same(N(pts), g["edge_pts"], "edge pts")
    assert_same(N(idx), g["edge_out_idx"], "edge idx")
    # occupancy votes on the golden march
    n_nodes = st["tree_nodes"].size // 64
    w = (g["seg_val"] * F32(0.01)).astype(F32)
    wa = torch.full((n_nodes,), -1, dtype=torch.int32, device=DEV)
    aa = torch.full((n_nodes,), -1, dtype=torch.int32, device=DEV)
    mk = torch.zeros(n_nodes, dtype=torch.int32, device=DEV)
    cnt = torch.zeros(n_nodes, dtype=torch.int32, device=DEV)
    R = len(g["cam"])
    hip.oct_mark_visit(R, T(g["march_pts_idx_bounds"]), T(g["march_anchors"]), 3, T(w), T(g["occ_alpha"]), wa, aa, mk, cnt)
    for a, b in ((wa, "occ_w_adder"), (aa, "occ_a_adder"), (mk, "occ_mark"), (cnt, "occ_cnt")):
        assert_same(N(a), g[b], b)
    rng = np.random.default_rng(3)
    ws = rng.integers(-2, 5, n_nodes).astype(np.int32)
    as_ = rng.integers(-2, 5, n_nodes).astype(np.int32)
    ew, ea, enodes = oc.update_node_stats(N(wa), N(aa), N(mk), ws, as_, st["tree_nodes"])
    tw, ta, tn = T(ws), T(as_), T(st["tree_nodes"])
    hip.oct_update_stats(n_nodes, wa, aa, mk, tw, ta, tn)
    assert_same(N(tw), ew); assert_same(N(ta), ea); assert_same(N(tn), enodes)
    tw, ta, tn = T(ws), T(as_), T(st["tree_nodes"])  # same update, vote buffers re-armed for the next iteration
    hip.oct_update_stats(n_nodes, wa, aa, mk, tw, ta, tn, reset_votes=True)
    assert_same(N(tw), ew); assert_same(N(ta), ea); assert_same(N(tn), enodes)
    assert (N(wa) == -1).all() and (N(aa) == -1).all() and (N(mk) == 0).all()
    ts = st["train_set"]
    tn2 = T(st["tree_nodes"])
    hip.oct_mark_invisible(n_nodes, len(ts), tn2, T(st["intri"][ts]), T(st["w2c"][ts]), T(st["bounds"][ts]))
    from oracle.octree_construct import NODE_DT
    assert_same(N(tn2).view(NODE_DT)["trans_idx"].copy(), g["invisible_trans_idx"], "invisible")


@pytest.mark.parametrize("n_nodes", [897, 200000])  # block-local vote images in LDS / global atomics (more nodes than fit)
def test_early_stop_and_votes_in_one_launch(hip, fox_golden, n_nodes):
    """f2n_early_stop_votes = f2n_early_stop + f2n_oct_mark_visit: weights, alphas, mask, kept counts and all four vote /
    mark / visit-count arrays equal those of the two separate launches, bit for bit, in both vote modes; ragged and empty
    rays included."""
    g = fox_golden
    rng = np.random.default_rng(8)
    se, anchors, dt = g["march_pts_idx_bounds"].copy(), g["march_anchors"].copy(), g["march_dt"]
    n, R = len(dt), len(se)
    se[3] = (se[3, 0], se[3, 0])  # an empty ray (its samples are simply not referenced)
    if n_nodes > 897:  # spread the leaves over a big node array so that runs of equal leaves stay runs
        anchors[:, 1] = (anchors[:, 1].astype(np.int64) * 211) % n_nodes
    f0 = (rng.standard_normal(n) * 2.0 + 1.0).astype(F32)  # dense enough for the early stop to bite
    d_se, d_f0, d_dt, d_an = T(se), T(f0), T(dt), T(anchors)

    def buffers():
        return dict(w=torch.full((n,), -5., device=DEV), a=torch.full((n,), -5., device=DEV),
                    m=torch.full((n,), -5, dtype=torch.int32, device=DEV), k=torch.full((R,), -5, dtype=torch.int32, device=DEV),
                    wa=torch.full((n_nodes,), -1, dtype=torch.int32, device=DEV), aa=torch.full((n_nodes,), -1, dtype=torch.int32, device=DEV),
                    mk=torch.zeros(n_nodes, dtype=torch.int32, device=DEV),
                    vc=torch.from_numpy(rng.integers(0, 3, n_nodes).astype(np.int32)).to(DEV))
    rng = np.random.default_rng(9); a = buffers()
    rng = np.random.default_rng(9); b = buffers()
    hip.early_stop(R, d_se, d_f0, 1, d_dt, a["w"], a["a"], a["m"], a["k"])
    hip.oct_mark_visit(R, d_se, d_an, 3, a["w"], a["a"], a["wa"], a["aa"], a["mk"], a["vc"])
    hip.early_stop_votes(R, d_se, d_f0, 1, d_dt, b["w"], b["a"], b["m"], b["k"], d_an, 3, b["wa"], b["aa"], b["mk"], b["vc"])
    for key in a:
        assert_same(N(a[key]), N(b[key]), key)
    assert int(N(a["k"])[3]) == 0 and 0 < int(N(a["k"]).sum()) < n and int(N(a["mk"]).sum()) > 10


def test_edge_samples_from_uniforms_and_sampler_prologue(hip, fox_state, fox_golden):
    """f2n_edge_samples_ex: three uniforms per point instead of (randint, uniform(-1,1)) draws -- idx = floor(u0 * n_edges),
    coords = 2u - 1 --, the index output at the stride of an anchors array and a second destination: all equal to the
    reference kernel (pinned golden / oracle) on the draws they imply.  f2n_sampler_prologue = normalize_dirs + zero fill +
    noise map in one launch, bit for bit."""
    st, g = fox_state, fox_golden
    rng = np.random.default_rng(12)
    n_edges = st["edge_pool"].size // 64
    n = 1000
    u = rng.random((n, 3), dtype=F32)
    u[0, 0] = F32(1.) - F32(2.) ** -24  # the largest float below one: the index clamp
    eidx = np.minimum((u[:, 0] * F32(n_edges)).astype(np.int32), n_edges - 1)
    ecoord = (u[:, 1:] * F32(2.) - F32(1.)).astype(F32)
    ref_pts = torch.empty((n, 2, 3), device=DEV); ref_idx = torch.empty((n, 2), dtype=torch.int32, device=DEV)
    hip.edge_samples(n, T(st["edge_pool"]), T(st["pers_trans"]), T(eidx), T(ecoord), ref_pts, ref_idx)
    epts, eoidx = oc.edge_samples(st["edge_pool"], st["pers_trans"], eidx, ecoord)
    assert_same(N(ref_pts), epts.reshape(n, 2, 3), "edge pts vs oracle"); assert_same(N(ref_idx), eoidx.reshape(n, 2), "edge idx vs oracle")
    pts1 = torch.zeros((n, 2, 3), device=DEV); anc = torch.full((2 * n, 3), -9, dtype=torch.int32, device=DEV)
    pts2 = torch.zeros((n, 2, 3), device=DEV); idx2 = torch.zeros((n, 2), dtype=torch.int32, device=DEV)
    hip.edge_samples_ex(n, T(st["edge_pool"]), n_edges, T(st["pers_trans"]), None, None, T(u), pts1, anc, 3, pts2, idx2, 1)
    assert_same(N(pts1), N(ref_pts)); assert_same(N(pts2), N(ref_pts)); assert_same(N(idx2), N(ref_idx))
    a = N(anc)
    assert_same(a[:, 0].reshape(n, 2), N(ref_idx)); assert (a[:, 1:] == -9).all()
    # prologue
    R = 777
    dirs = rng.normal(size=(R, 3)).astype(F32) * F32(3.)
    uu = rng.random(1024 + R + 10, dtype=F32)
    out = torch.zeros((R, 3), device=DEV); zero = torch.full((2,), 5, dtype=torch.int32, device=DEV); nz = T(uu.copy())
    hip.sampler_prologue(R, T(dirs), out, zero, nz, 3.5, nz)  # in place, as the host does
    assert_same(N(out), oc.normalize_dirs(dirs)); assert (N(zero) == 0).all()
    assert_same(N(nz), ((uu - F32(.5)) + F32(1.)) * F32(3.5))


# ---------------------------------------------------------------------------------------------------
# hash grid
# ---------------------------------------------------------------------------------------------------
def make_grid(st, rng, log2_t, scale=1.0, zeros=False):
    local = 1 << log2_t
    table = np.zeros((16 * local, 2), F32) if zeros else (rng.standard_normal((16 * local, 2)).astype(F32) * F32(scale))
    return op.HashGrid(table, st["prim_pool"], st["bias_pool"], int(st["n_volumes"]), log2_t)


def grid_dev(grid):
    return dict(table_h=T(grid.table_h.view(np.float16)), prim=T(grid.prim_pool), lidx=T(grid.local_idx),
                lsize=T(grid.local_size), bias=T(grid.bias_pool), scale=T(grid.scales))


@pytest.mark.parametrize("log2_t,n", [(10, 1000), (19, 5000)])
def test_hash_forward_bit_exact(hip, fox_state, log2_t, n):
    rng = np.random.default_rng(log2_t)
    grid = make_grid(fox_state, rng, log2_t)
    q = (rng.random((n, 3), dtype=F32) * F32(1.6) - F32(0.3))  # includes q < 0: saturating float->u32
    vol = rng.integers(0, grid.n_volumes, n).astype(np.int32)
    gd = grid_dev(grid)
    out = torch.zeros((n, 32), dtype=torch.float16, device=DEV)
    hip.hash_fwd(n, grid.n_volumes, gd["table_h"], gd["prim"], gd["lidx"], gd["lsize"], gd["bias"], gd["scale"], T(q), False,
                 T(vol), 1, out)
    ref = oc.hash_fwd(grid.table_h, grid.prim_pool, grid.local_idx, grid.local_size, grid.bias_pool, q, vol, grid.n_volumes)
    assert_same(N(out).view(np.uint16), ref, "hash features")
    # warped input + strided volume index (anchors[:,0] read in place)
    anchors = np.zeros((n, 3), np.int32)
    anchors[:, 0] = vol
    pw = (q * F32(2.) - F32(1.)).astype(F32)
    out2 = torch.zeros((n, 32), dtype=torch.float16, device=DEV)
    hip.hash_fwd(n, grid.n_volumes, gd["table_h"], gd["prim"], gd["lidx"], gd["lsize"], gd["bias"], gd["scale"], T(pw), True,
                 T(anchors), 3, out2)
    q2 = ((pw + F32(1.)) * F32(.5)).astype(F32)
    ref2 = oc.hash_fwd(grid.table_h, grid.prim_pool, grid.local_idx, grid.local_size, grid.bias_pool, q2, vol, grid.n_volumes)
    assert_same(N(out2).view(np.uint16), ref2, "hash features (warped)")


def test_hash_forward_golden_and_linearity(hip, fox_state, fox_golden):
    import sys, os
    sys.path.insert(0, os.path.dirname(__file__))
    from test_golden import _hash_setup
    g = fox_golden
    table, li, ls, q01, vol = _hash_setup(fox_state, g)
    n = len(vol)
    scale = T(oc.level_scales())
    out = torch.zeros((n, 32), dtype=torch.float16, device=DEV)
    args = (n, int(fox_state["n_volumes"]))
    rest = (T(fox_state["prim_pool"]), T(li), T(ls), T(fox_state["bias_pool"]), scale, T(q01), False, T(vol), 1)
    hip.hash_fwd(*args, T(table), *rest, out)
    assert_same(N(out).view(np.uint16), g["hash_feat"], "golden hash features")
    # linearity in the table: scaling every entry by 2 (exact in fp16) doubles every feature exactly
    out2 = torch.zeros_like(out)
    hip.hash_fwd(*args, T((table.astype(F32) * 2).astype(np.float16)), *rest, out2)
    a, b = N(out).astype(F32), N(out2).astype(F32)
    normal = np.abs(a) >= 2.0 ** -13  # doubling commutes with fp16 rounding only outside the subnormal range
    assert normal.mean() > 0.99
    assert_same((a[normal] * 2).astype(np.float16).view(np.uint16), b[normal].astype(np.float16).view(np.uint16), "linearity")


def test_hash_backward(hip, fox_state):
    rng = np.random.default_rng(11)
    grid = make_grid(fox_state, rng, 12)
    n = 4000
    q = rng.random((n, 3), dtype=F32)
    vol = rng.integers(0, grid.n_volumes, n).astype(np.int32)
    gin = (rng.standard_normal((n, 32)) * 0.05).astype(np.float16)
    gin[::7] = 0
    gd = grid_dev(grid)
    gtab = torch.zeros(grid.table_f32.size, dtype=torch.float16, device=DEV)
    hip.hash_bwd(n, grid.n_volumes, gd["prim"], gd["lidx"], gd["lsize"], gd["bias"], gd["scale"], T(q), False, T(vol), 1, T(gin),
                 gtab)
    ref32 = oc.hash_bwd(grid.table_f32.size, grid.prim_pool, grid.local_idx, grid.local_size, grid.bias_pool, q, vol,
                        grid.n_volumes, gin.view(np.uint16), fp32_accumulate=True)
    got = N(gtab).astype(F32)
    # fp16 atomics accumulate in arbitrary order: compare with the fp32-accumulated oracle at fp16 resolution of the
    # running sums (each of <= ~40 additions may round by half an ulp of the partial sum)
    tol = 2e-3 * np.abs(ref32).max() + 8 * 2.0 ** -11 * np.abs(ref32)
    assert (np.abs(got - ref32) <= tol).all(), float(np.abs(got - ref32).max())
    assert (got != 0).sum() == (ref32 != 0).sum() or abs(int((got != 0).sum()) - int((ref32 != 0).sum())) < 50
    # collision-free case is bit-exact against the sequential fp16 oracle: one point only
    gtab1 = torch.zeros_like(gtab)
    hip.hash_bwd(1, grid.n_volumes, gd["prim"], gd["lidx"], gd["lsize"], gd["bias"], gd["scale"], T(q[1:2]), False, T(vol[1:2]),
                 1, T(gin[1:2]), gtab1)
    ref1 = oc.hash_bwd(grid.table_f32.size, grid.prim_pool, grid.local_idx, grid.local_size, grid.bias_pool, q[1:2], vol[1:2],
                       grid.n_volumes, gin[1:2].view(np.uint16))
    same_cell_twice = False
    got1 = N(gtab1).view(np.uint16)
    if not same_cell_twice:
        mism = (got1 != ref1).sum()
        assert mism <= 4, mism  # two corners of one level may hash to the same entry (order-dependent rounding)


@pytest.mark.parametrize("log2", [14, 20, 22])  # 22: the table size BASELINE config 5 names (F2N_BIN_MAX_BINS is sized for it)
def test_hash_backward_owner_binned(hip, fox_state, log2):
    """Large batches take the owner-binned scatter (queues -> LDS accumulation -> plain stores).  It must agree with
    the direct packed-f16 atomics of the small-batch path and with the fp32-accumulated oracle; rays of consecutive,
    closely spaced samples exercise the in-row run combining, a short ragged tail the chunk boundaries."""
    rng = np.random.default_rng(12)
    grid = make_grid(fox_state, rng, log2, zeros=log2 >= 22)  # (the backward never reads the table values)
    n = 40000 + 37
    n_rays = n // 50 + 1
    o = rng.random((n_rays, 3), dtype=F32) * F32(.6) + F32(.2)
    dvec = rng.standard_normal((n_rays, 3)).astype(F32); dvec /= np.linalg.norm(dvec, axis=1, keepdims=True)
    ray = np.repeat(np.arange(n_rays), 50)[:n]
    step = (np.arange(n) % 50).astype(F32) * F32(0.004)
    q = np.clip(o[ray] + dvec[ray] * step[:, None], 0.01, 0.99).astype(F32)
    vol = rng.integers(0, grid.n_volumes, n_rays).astype(np.int32)[ray]
    gin = (rng.standard_normal((n, 32)) * 0.05).astype(np.float16)
    gin[rng.random(n) < 0.5] = 0
    gd = grid_dev(grid)
    args = (n, grid.n_volumes, gd["prim"], gd["lidx"], gd["lsize"], gd["bias"], gd["scale"], T(q), False, T(vol), 1, T(gin))
    g_atomic = torch.zeros(grid.table_f32.size, dtype=torch.float16, device=DEV)
    hip.hash_bwd(*args, g_atomic)  # level_entries = 0: direct atomics
    g_binned = torch.zeros_like(g_atomic)
    hip.hash_bwd(*args, g_binned, 1 << log2)
    ref32 = oc.hash_bwd(grid.table_f32.size, grid.prim_pool, grid.local_idx, grid.local_size, grid.bias_pool, q, vol,
                        grid.n_volumes, gin.view(np.uint16), fp32_accumulate=True)
    a, b = N(g_atomic).astype(F32), N(g_binned).astype(F32)
    tol = 2e-3 * np.abs(ref32).max() + 8 * 2.0 ** -11 * np.abs(ref32)
    assert (np.abs(b - ref32) <= tol).all(), float(np.abs(b - ref32).max())
    # the binned path sums in fp32 and rounds once per (slice, level): it is the closer of the two to the oracle
    assert np.abs(b - ref32).sum() <= np.abs(a - ref32).sum()
    assert np.abs(b - ref32).max() <= 2.0 ** -9 * np.abs(ref32).max()
    # accumulates into an existing table (even / odd level overlap, second call)
    hip.hash_bwd(*args, g_binned, 1 << log2)
    assert np.abs(N(g_binned).astype(F32) - 2 * ref32).max() <= 2.0 ** -8 * np.abs(ref32).max()


@pytest.mark.parametrize("log2", [20, 22])
def test_hash_backward_overflow_lists_keep_the_sums_order_free(hip, fox_state, log2):
    """Records that find their queue segment full (round 6; tables of more than 2^19 entries per level): every sample inside one tiny
    cube, consecutive samples in four alternating transforms so that no run combines -- a level's 8 x chunk records go to 32 table
    entries, i.e. to a few slices, more than a segment's share.  They travel through their producer block's overflow list and are summed by the owners like every other record:
    no packed-f16 atomic (f2n_debug_counters()[0] stays 0), the table within f16 resolution of the fp32-accumulated oracle, and two
    scatters of the same input leave the same bits."""
    rng = np.random.default_rng(77)
    grid = make_grid(fox_state, rng, log2, zeros=True)
    n = 32768 + 1024 + 5
    q = (F32(0.371) + rng.random((n, 3), dtype=F32) * F32(2e-4)).astype(F32)
    vol = (np.arange(n) % 4).astype(np.int32)
    gin = (rng.standard_normal((n, 32)) * 0.01).astype(np.float16)
    gd = grid_dev(grid)
    args = (n, grid.n_volumes, gd["prim"], gd["lidx"], gd["lsize"], gd["bias"], gd["scale"], T(q), False, T(vol), 1, T(gin))
    tabs = []
    hip.debug_counters(reset=True)
    for _ in range(2):
        g_binned = torch.zeros(grid.table_f32.size, dtype=torch.float16, device=DEV)
        hip.hash_bwd(*args, g_binned, 1 << log2)
        tabs.append(N(g_binned).view(np.uint16).copy())
    c = hip.debug_counters()
    assert c[0] == 0 and c[3] > 0, c  # nothing through the atomics, something through the lists
    assert (tabs[0] == tabs[1]).all()
    ref32 = oc.hash_bwd(grid.table_f32.size, grid.prim_pool, grid.local_idx, grid.local_size, grid.bias_pool, q, vol,
                        grid.n_volumes, gin.view(np.uint16), fp32_accumulate=True)
    got = tabs[0].view(np.float16).astype(F32)
    assert ((got != 0) == (ref32 != 0)).all()
    assert np.abs(got - ref32).max() <= 2.0 ** -9 * np.abs(ref32).max(), float(np.abs(got - ref32).max())


# ---------------------------------------------------------------------------------------------------
# MLP
# ---------------------------------------------------------------------------------------------------
def rand_params(rng, n_hidden):
    n = oc.mlp_n_params(32, 64, n_hidden)
    p = np.zeros(n, F32)
    off = 0
    for rows, cols in [(64, 32)] + [(64, 64)] * (n_hidden - 1) + [(16, 64)]:
        s = np.sqrt(6.0 / (rows + cols))
        p[off:off + rows * cols] = rng.uniform(-s, s, rows * cols).astype(F32)
        off += rows * cols
    return p


def torch_mlp_ref(params, x, n_hidden):
    """Plain fp32 PyTorch reference of the same network (no fp16 rounding) for a sanity bound."""
    h = torch.from_numpy(x)
    off = 0
    dims = [(64, 32)] + [(64, 64)] * (n_hidden - 1) + [(16, 64)]
    for li, (rows, cols) in enumerate(dims):
        w = torch.from_numpy(params[off:off + rows * cols].reshape(rows, cols))
        off += rows * cols
        h = h @ w.t()
        if li < len(dims) - 1:
            h = torch.relu(h)
    return h.numpy()


@pytest.mark.parametrize("n_hidden,n", [(1, 1000), (2, 1000), (1, 17), (2, 16)])
def test_mlp_forward(hip, n_hidden, n):
    rng = np.random.default_rng(n_hidden * 100 + n)
    params = rand_params(rng, n_hidden)
    x = rng.standard_normal((n, 32)).astype(F32)
    ph = T(oc.f2h(params).view(np.float16))
    out = torch.zeros((n, 16), dtype=torch.float16, device=DEV)
    hip.mlp_fwd(n, 32, 64, n_hidden, ph, T(x), out)
    got = N(out).astype(F32)
    ref = oc.h2f(oc.mlp_fwd(params, x, 64, n_hidden))
    # same rounding points (fp16 weights/activations, fp32 accumulate); only the fp32 summation order differs, which
    # can flip an fp16 rounding of a hidden activation: tolerance = a few fp16 ulps of the output scale
    tol = 4 * 2.0 ** -11 * max(1.0, np.abs(ref).max())
    assert np.abs(got - ref).max() <= tol, (np.abs(got - ref).max(), tol)
    assert (np.abs(got - ref) > 0).mean() < 0.2
    full = torch_mlp_ref(params, x, n_hidden)
    assert np.abs(got - full).max() <= 2e-2 * max(1.0, np.abs(full).max())


@pytest.mark.parametrize("n_hidden,n", [(1, 2048), (2, 2048), (1, 33), (2, 31)])
def test_mlp_backward(hip, n_hidden, n):
    rng = np.random.default_rng(n_hidden * 7 + n)
    params = rand_params(rng, n_hidden)
    x = rng.standard_normal((n, 32)).astype(F32)
    dy = (rng.standard_normal((n, 16)) * 1e-2).astype(F32)
    ph = T(oc.f2h(params).view(np.float16))
    dparams = torch.zeros(params.size, device=DEV)
    dx = torch.zeros((n, 32), device=DEV)
    hip.mlp_bwd(n, 32, 64, n_hidden, 128.0, ph, T(x), T(dy), dparams, dx)
    _, acts = oc.mlp_fwd(params, x, 64, n_hidden, want_acts=True)
    rdp, rdx, _ = oc.mlp_bwd(params, x, acts, dy, 64, n_hidden, 128.0)
    gdp = N(dparams) / F32(128.)
    gdx = N(dx)
    assert np.abs(gdx - rdx).max() <= 3e-3 * np.abs(rdx).max() + 1e-7, np.abs(gdx - rdx).max()
    # the oracle rounds dparams to fp16 twice (reference behaviour); compare at that resolution
    assert np.abs(gdp - rdp).max() <= 4e-3 * np.abs(rdp).max() + 1e-7, (np.abs(gdp - rdp).max(), np.abs(rdp).max())
    # independent fp32 autograd check (no fp16 anywhere): loose bound
    xt = torch.from_numpy(x).requires_grad_(True)
    pt = torch.from_numpy(params).requires_grad_(True)
    off, h = 0, xt
    dims = [(64, 32)] + [(64, 64)] * (n_hidden - 1) + [(16, 64)]
    for li, (rows, cols) in enumerate(dims):
        h = h @ pt[off:off + rows * cols].reshape(rows, cols).t()
        off += rows * cols
        if li < len(dims) - 1:
            h = torch.relu(h)
    (h * torch.from_numpy(dy)).sum().backward()
    assert np.abs(gdp - pt.grad.numpy()).max() <= 8e-2 * np.abs(pt.grad.numpy()).max()
    edx = np.abs(gdx - xt.grad.numpy())  # fp16 activations flip a few ReLU masks relative to the fp32 network
    assert edx.max() <= 0.3 * np.abs(xt.grad.numpy()).max() and edx.mean() <= 3e-3 * np.abs(xt.grad.numpy()).max()


# ---------------------------------------------------------------------------------------------------
# fused field / shader
# ---------------------------------------------------------------------------------------------------
def test_partitioned_gather_equals_fused_forward(hip, fox_state, fox_golden):
    """The large-batch path (XCD-partitioned gather into planes + MLP on planes) and the single fused launch compute
    the same bits; both are reachable through f2n_field_fwd depending on n."""
    st, g = fox_state, fox_golden
    rng = np.random.default_rng(22)
    grid = make_grid(st, rng, 14, scale=0.5)
    params = rand_params(rng, 1)
    pts, anchors = g["march_pts"], g["march_anchors"]
    n = len(pts)
    gd = grid_dev(grid)
    ph = T(oc.f2h(params).view(np.float16))
    feat_a = torch.zeros((n, 16), device=DEV); f0_a = torch.zeros(n, device=DEV); sx_a = torch.zeros((n, 32), dtype=torch.float16, device=DEV)
    hip.field_fwd(n, grid.n_volumes, gd["table_h"], gd["prim"], gd["lidx"], gd["lsize"], gd["bias"], gd["scale"], T(pts),
                  T(anchors), 3, ph, feat_a, f0_a, sx_a)
    planes = torch.zeros((8, n, 4), dtype=torch.float16, device=DEV)
    hip.hash_gather_planes(n, grid.n_volumes, gd["table_h"], gd["prim"], gd["lidx"], gd["lsize"], gd["bias"], gd["scale"], T(pts),
                           True, T(anchors), 3, planes)
    feat_b = torch.zeros_like(feat_a); f0_b = torch.zeros_like(f0_a); sx_b = torch.zeros_like(sx_a)
    hip.field_mlp_planes(n, planes, ph, feat_b, f0_b, sx_b)
    assert_same(N(sx_a).view(np.uint16), N(sx_b).view(np.uint16), "features")
    assert_same(N(feat_a), N(feat_b), "field output")
    assert_same(N(f0_a), N(f0_b), "f0")
    # plane p holds features 4p..4p+3 of every sample
    assert_same(N(planes).view(np.uint16).transpose(1, 0, 2).reshape(n, 32), N(sx_a).view(np.uint16), "plane layout")


@pytest.mark.parametrize("log2_t,n,p0,clump", [(21, 100003, 3, False), (21, 70001, 0, True), (22, 131072, 5, False), (21, 1, 0, False),
                                                   (21, 1537, 1, True), (21, 0, 0, False),
                                                   (20, 120001, 1, False), (20, 66000, 1, True)])  # 2^20: wanjinyou_big.yaml's own size (round 5)
def test_binned_gather_equals_partitioned_gather(hip, fox_state, log2_t, n, p0, clump):
    """The slice-binned gather of the big tables (requests binned by 4096-entry table slice, slices served from LDS) writes the
    planes of the partitioned gather bit for bit -- for clumped points (a few slices take most of a chunk's requests), for
    batches of one sample, of one sample more than a chunk, of none, and whichever level pair the binned part starts at."""
    st = fox_state
    rng = np.random.default_rng(log2_t + p0)
    grid = make_grid(st, rng, log2_t, scale=0.5)
    gd = grid_dev(grid)
    pts = (rng.random((n, 3), dtype=F32) * F32(2) - F32(1))
    if clump:
        pts[n // 3:] = pts[7]  # two thirds of the batch in one cell: a handful of slices take most of every chunk's requests
    anchors = np.zeros((n, 3), np.int32)
    anchors[:, 0] = rng.integers(0, grid.n_volumes, n)
    a = torch.zeros((8, n, 4), dtype=torch.float16, device=DEV)
    b = torch.full((8, n, 4), float("nan"), dtype=torch.float16, device=DEV)
    hip.hash_gather_planes(n, grid.n_volumes, gd["table_h"], gd["prim"], gd["lidx"], gd["lsize"], gd["bias"], gd["scale"], T(pts),
                           True, T(anchors), 3, a)
    for _ in range(2):  # the second call reuses the workspace
        hip.hash_gather_planes_binned(n, grid.n_volumes, gd["table_h"], gd["prim"], gd["lidx"], gd["lsize"], gd["bias"], gd["scale"],
                                      T(pts), True, T(anchors), 3, b, 1 << log2_t, p0)
        assert_same(N(b).view(np.uint16), N(a).view(np.uint16), "planes")


@pytest.mark.parametrize("n_use", [None, 1500])
def test_field_fused_forward_backward(hip, fox_state, fox_golden, n_use):
    st, g = fox_state, fox_golden
    rng = np.random.default_rng(21)
    grid = make_grid(st, rng, 14, scale=0.5)
    params = rand_params(rng, 1)
    pts, anchors = g["march_pts"][:n_use], g["march_anchors"][:n_use]
    n = len(pts)
    gd = grid_dev(grid)
    ph = T(oc.f2h(params).view(np.float16))
    feat = torch.zeros((n, 16), device=DEV)
    f0 = torch.zeros(n, device=DEV)
    sx = torch.zeros((n, 32), dtype=torch.float16, device=DEV)
    hip.field_fwd(n, grid.n_volumes, gd["table_h"], gd["prim"], gd["lidx"], gd["lsize"], gd["bias"], gd["scale"], T(pts),
                  T(anchors), 3, ph, feat, f0, sx)
    rfeat, ctx = op.field_fwd(grid, params, pts, anchors[:, 0], want_ctx=True)
    assert_same(N(sx).view(np.uint16), ctx["x_h"], "fused hash features (saved)")
    tol = 4 * 2.0 ** -11 * max(1.0, np.abs(rfeat).max())
    assert np.abs(N(feat) - rfeat).max() <= tol
    assert_same(N(f0), N(feat)[:, 0].copy(), "f0 == feat[:,0]")
    # backward
    dfeat = (rng.standard_normal((n, 16)) * 1e-3).astype(F32)
    dparams = torch.zeros(params.size, device=DEV)
    gtab = torch.zeros(grid.table_f32.size, dtype=torch.float16, device=DEV)
    hip.field_bwd(n, grid.n_volumes, gd["prim"], gd["lidx"], gd["lsize"], gd["bias"], gd["scale"], T(pts), T(anchors), 3, ph, sx,
                  T(dfeat), 128.0, dparams, gtab, 1 << 14)
    rdp, rgt, _ = op.field_bwd(grid, params, ctx, dfeat, 128.0, fp32_accumulate=True)
    gdp = N(dparams) / F32(128.)
    assert np.abs(gdp - rdp).max() <= 4e-3 * np.abs(rdp).max() + 1e-7
    ggt = N(gtab).astype(F32) / F32(128.)
    assert np.abs(ggt - rgt).max() <= 1e-2 * np.abs(rgt).max(), (np.abs(ggt - rgt).max(), np.abs(rgt).max())
    cos = float((ggt * rgt).sum() / (np.linalg.norm(ggt) * np.linalg.norm(rgt)))
    assert cos > 0.9999, cos


def test_workspace_growth_with_queued_kernels(hip, fox_state, fox_golden):
    """The internal workspaces grow on demand.  Kernels that were handed the previous buffer may still be queued when a
    later call of the same iteration needs a bigger one (edge-sample forward, then the field backward): the forward's
    results must survive that (regression test of an intermittent memory-access fault in long trainings)."""
    st, g = fox_state, fox_golden
    rng = np.random.default_rng(77)
    grid = make_grid(st, rng, 14, scale=0.5)
    gd = grid_dev(grid)
    ph = T(oc.f2h(rand_params(rng, 1)).view(np.float16))
    pts0, anc0 = g["march_pts"], g["march_anchors"]
    def batch(n):
        idx = rng.integers(0, len(pts0), n)
        return T(pts0[idx]), T(np.ascontiguousarray(anc0[idx]))
    n_small = 16384  # partitioned forward: uses the plane workspace
    ps, as_ = batch(n_small)
    ref = torch.zeros((n_small, 16), device=DEV)
    hip.field_fwd(n_small, grid.n_volumes, gd["table_h"], gd["prim"], gd["lidx"], gd["lsize"], gd["bias"], gd["scale"], ps, as_, 3,
                  ph, ref, None, None)
    torch.cuda.synchronize()
    n_big = 40000
    for rep in range(12):  # every round asks for a bigger backward workspace while the forward is still queued
        n_big = int(n_big * 1.4)
        pb, ab = batch(n_big)
        sx = torch.zeros((n_big, 32), dtype=torch.float16, device=DEV)
        dfeat = torch.zeros((n_big, 16), device=DEV)
        dparams = torch.zeros(3072, device=DEV)
        gtab = torch.zeros(grid.table_f32.size, dtype=torch.float16, device=DEV)
        out = torch.zeros((n_small, 16), device=DEV)
        for _ in range(4):  # a few queued users of the current buffer
            hip.field_fwd(n_small, grid.n_volumes, gd["table_h"], gd["prim"], gd["lidx"], gd["lsize"], gd["bias"], gd["scale"], ps,
                          as_, 3, ph, out, None, None)
        hip.field_bwd(n_big, grid.n_volumes, gd["prim"], gd["lidx"], gd["lsize"], gd["bias"], gd["scale"], pb, ab, 3, ph, sx, dfeat,
                      128.0, dparams, gtab, 1 << 14)
        torch.cuda.synchronize()
        assert_same(N(out), N(ref), "forward queued in front of a workspace growth")


def test_field_forward_from_prepass_cache(hip, fox_state, fox_golden):
    """f2n_field_fwd_cached (hash features memoised by the pre-pass) must equal a fresh f2n_field_fwd bit for bit,
    for an arbitrary (compaction-like, order-preserving) subset and for the identity mapping."""
    st, g = fox_state, fox_golden
    rng = np.random.default_rng(23)
    grid = make_grid(st, rng, 14, scale=0.5)
    params = rand_params(rng, 1)
    pts, anchors = g["march_pts"], g["march_anchors"]
    n = len(pts)
    gd = grid_dev(grid)
    ph = T(oc.f2h(params).view(np.float16))
    f0_pre = torch.zeros(n, device=DEV)
    cache = torch.zeros((n, 32), dtype=torch.float16, device=DEV)
    hip.field_fwd(n, grid.n_volumes, gd["table_h"], gd["prim"], gd["lidx"], gd["lsize"], gd["bias"], gd["scale"], T(pts),
                  T(anchors), 3, ph, None, f0_pre, cache)
    rows = np.sort(rng.choice(n, size=n // 3, replace=False)).astype(np.int32)
    for src in (rows, None):
        idx = rows if src is not None else np.arange(n, dtype=np.int32)
        m = len(idx)
        want_feat = torch.zeros((m, 16), device=DEV); want_sx = torch.zeros((m, 32), dtype=torch.float16, device=DEV)
        hip.field_fwd(m, grid.n_volumes, gd["table_h"], gd["prim"], gd["lidx"], gd["lsize"], gd["bias"], gd["scale"],
                      T(pts[idx]), T(anchors[idx]), 3, ph, want_feat, None, want_sx)
        got_feat = torch.zeros((m, 16), device=DEV); got_f0 = torch.zeros(m, device=DEV)
        got_sx = torch.zeros((m, 32), dtype=torch.float16, device=DEV)
        hip.field_fwd_cached(m, n, None if src is None else T(src), cache, ph, got_feat, got_f0, got_sx)
        assert_same(N(got_feat), N(want_feat), "cached forward == fresh forward")
        assert_same(N(got_sx).view(np.uint16), N(want_sx).view(np.uint16), "cached saved features")
        assert_same(N(got_f0), N(f0_pre)[idx], "f0 of the pre-pass")
    with pytest.raises(hip.F2nError):  # identity mapping over more samples than the cache holds
        hip.field_fwd_cached(n + 1, n, None, cache, ph, torch.zeros((n + 1, 16), device=DEV), None, None)


@pytest.mark.parametrize("use_emb,use_ndev", [(True, False), (False, True)])
def test_field_and_shade_forward_in_one_launch(hip, fox_state, fox_golden, use_emb, use_ndev):
    """f2n_field_shade_fwd_dyn (field MLP on cached hash features -> colour path, `feat` never in memory) must equal
    f2n_field_fwd_cached followed by f2n_shade_fwd bit for bit: rgb, f0, both saved network inputs.  With a device-side
    count only the first *n_dev rows are produced."""
    st, g = fox_state, fox_golden
    rng = np.random.default_rng(29)
    grid = make_grid(st, rng, 14, scale=0.5)
    pf, pc = rand_params(rng, 1), rand_params(rng, 2)
    pts, anchors, dirs, se = g["march_pts"], g["march_anchors"], g["march_dirs"], g["march_pts_idx_bounds"]
    n = len(pts)
    gd = grid_dev(grid)
    phf, phc = T(oc.f2h(pf).view(np.float16)), T(oc.f2h(pc).view(np.float16))
    cache = torch.zeros((n, 32), dtype=torch.float16, device=DEV)
    hip.field_fwd(n, grid.n_volumes, gd["table_h"], gd["prim"], gd["lidx"], gd["lsize"], gd["bias"], gd["scale"], T(pts),
                  T(anchors), 3, phf, None, torch.zeros(n, device=DEV), cache)
    rows = np.sort(rng.choice(n, size=n // 2 + 7, replace=False)).astype(np.int32)
    m = len(rows)
    emb = T((rng.standard_normal((50, 16)) * 0.1).astype(F32)) if use_emb else None
    sidx = T(oc.scatter_idx(n, se, g["cam"])[rows]) if use_emb else None
    d_rows, d_dirs = T(rows), T(dirs[rows])
    # separate kernels
    feat = torch.zeros((m, 16), device=DEV); f0 = torch.zeros(m, device=DEV)
    fx = torch.zeros((m, 32), dtype=torch.float16, device=DEV); sx = torch.zeros((m, 32), dtype=torch.float16, device=DEV)
    rgb = torch.zeros((m, 3), device=DEV)
    hip.field_fwd_cached(m, n, d_rows, cache, phf, feat, f0, fx)
    hip.shade_fwd(m, feat, d_dirs, emb, sidx, phc, rgb, sx)
    # one launch
    m_dyn = m - 1000 if use_ndev else m
    n_dev = torch.tensor([m_dyn], dtype=torch.int32, device=DEV) if use_ndev else None
    f0b = torch.full((m,), -7.0, device=DEV); rgbb = torch.full((m, 3), -7.0, device=DEV)
    fxb = torch.zeros((m, 32), dtype=torch.float16, device=DEV); sxb = torch.zeros((m, 32), dtype=torch.float16, device=DEV)
    hip.field_shade_fwd(m, d_rows, cache, phf, d_dirs, emb, sidx, phc, f0b, fxb, sxb, rgbb, n_dev=n_dev)
    assert_same(N(rgbb)[:m_dyn], N(rgb)[:m_dyn], "rgb")
    assert_same(N(f0b)[:m_dyn], N(f0)[:m_dyn], "f0")
    assert_same(N(fxb).view(np.uint16)[:m_dyn], N(fx).view(np.uint16)[:m_dyn], "field input rows")
    assert_same(N(sxb).view(np.uint16)[:m_dyn], N(sx).view(np.uint16)[:m_dyn], "colour input rows")
    if use_ndev:  # rows past the device-side count are untouched
        assert (N(rgbb)[m_dyn:] == -7.0).all() and (N(f0b)[m_dyn:] == -7.0).all()
    # ... and with "extra" rows riding along (the edge samples of the TV loss: field MLP only, fp32 feature rows out):
    # the survivors' outputs do not change, the extra rows equal f2n_field_fwd_cached on them
    n_ex = 1000 + 5
    ex_cache = cache[n - n_ex:].contiguous()
    feat_ex = torch.zeros((n_ex, 16), device=DEV); fx_ex = torch.zeros((n_ex, 32), dtype=torch.float16, device=DEV)
    hip.field_fwd_cached(n_ex, n_ex, None, ex_cache, phf, feat_ex, torch.zeros(n_ex, device=DEV), fx_ex)
    f0c_ = torch.full((m,), -7.0, device=DEV); rgbc = torch.full((m, 3), -7.0, device=DEV)
    fxc = torch.zeros((m, 32), dtype=torch.float16, device=DEV); sxc = torch.zeros((m, 32), dtype=torch.float16, device=DEV)
    feat_ex2 = torch.full((n_ex, 16), -7.0, device=DEV); fx_ex2 = torch.zeros((n_ex, 32), dtype=torch.float16, device=DEV)
    hip.field_shade_fwd(m, d_rows, cache, phf, d_dirs, emb, sidx, phc, f0c_, fxc, sxc, rgbc, n_dev=n_dev, x_extra_h=ex_cache,
                        feat_extra=feat_ex2, save_x_extra_h=fx_ex2)
    assert_same(N(rgbc), N(rgbb), "rgb with extra rows"); assert_same(N(f0c_), N(f0b), "f0 with extra rows")
    assert_same(N(feat_ex2), N(feat_ex), "extra feature rows")
    assert_same(N(fx_ex2).view(np.uint16), N(fx_ex).view(np.uint16), "extra input rows")


@pytest.mark.parametrize("n_emb", [0, 50, 240, 1500])  # 240: the largest per-block LDS image (64-bit fixed point); above: global atomics
def test_shade_fused_forward_backward(hip, fox_golden, n_emb):
    use_emb = n_emb > 0
    g = fox_golden
    rng = np.random.default_rng(31)
    params = rand_params(rng, 2)
    dirs, se = g["march_dirs"], g["march_pts_idx_bounds"]
    n = len(dirs)
    feat = rng.standard_normal((n, 16)).astype(F32)
    emb = (rng.standard_normal((n_emb, 16)) * 0.1).astype(F32) if use_emb else None
    sidx = oc.scatter_idx(n, se, g["cam"]) if use_emb else None
    ph = T(oc.f2h(params).view(np.float16))
    rgb = torch.zeros((n, 3), device=DEV)
    sx = torch.zeros((n, 32), dtype=torch.float16, device=DEV)
    d_sidx = None
    if use_emb:
        d_sidx = torch.zeros(n, dtype=torch.int32, device=DEV)
        hip.scatter_idx(len(se), T(se), T(g["cam"]), d_sidx)
        assert_same(N(d_sidx), sidx, "scatter idx")
    hip.shade_fwd(n, T(feat), T(dirs), T(emb) if use_emb else None, d_sidx, ph, rgb, sx)
    rrgb, ctx = op.shade_fwd(params, feat, dirs, emb, sidx, want_ctx=True)
    assert_same(N(sx).view(np.uint16), oc.f2h(ctx["x"]), "colour MLP input (h16)")
    assert np.abs(N(rgb) - rrgb).max() <= 1e-3  # north-star tolerance on rendered colour
    drgb = (rng.standard_normal((n, 3)) * 1e-3).astype(F32)
    dfeat = torch.full((n, 16), 7.0, device=DEV)
    dparams = torch.zeros(params.size, device=DEV)
    demb = torch.zeros((n_emb, 16), device=DEV) if use_emb else None
    hip.shade_bwd(n, T(drgb), d_sidx, ph, sx, 128.0, dfeat, dparams, demb)
    rdp, rdfeat, rdemb = op.shade_bwd(params, ctx, drgb, n_emb, sidx, 128.0)
    gdf = N(dfeat)
    assert (gdf[:, 0] == 7.0).all()  # column 0 untouched
    assert np.abs(gdf[:, 1:] - rdfeat[:, 1:]).max() <= 4e-3 * np.abs(rdfeat).max() + 1e-8
    assert np.abs(N(dparams) / F32(128.) - rdp).max() <= 4e-3 * np.abs(rdp).max() + 1e-8
    if use_emb:
        assert np.abs(N(demb) - rdemb).max() <= 4e-3 * np.abs(rdemb).max() + 1e-8
    # column 0 merged from a compact array instead of left untouched: same rows otherwise
    df0 = torch.from_numpy(rng.standard_normal(n).astype(F32)).to(DEV)
    dfeat2 = torch.full((n, 16), 7.0, device=DEV)
    demb2 = torch.zeros((n_emb, 16), device=DEV) if use_emb else None
    hip.shade_bwd(n, T(drgb), d_sidx, ph, sx, 128.0, dfeat2, torch.zeros(params.size, device=DEV), demb2, df0=df0)
    assert_same(N(dfeat2)[:, 0].copy(), N(df0)); assert_same(N(dfeat2)[:, 1:].copy(), gdf[:, 1:].copy())
    if use_emb:  # the embedding gradient's column 0 is the shading input's gradient, not the merged density gradient
        assert np.abs(N(demb2) - rdemb).max() <= 4e-3 * np.abs(rdemb).max() + 1e-8


def test_sh_encode(hip, fox_golden):
    d = fox_golden["march_dirs"][::7]
    for deg, key in ((4, "sh4"),):
        out = torch.zeros((len(d), deg * deg), device=DEV)
        hip.sh_encode(len(d), deg, T(d), out)
        assert_same(N(out), fox_golden[key], key)
    d3 = fox_golden["march_dirs"][::31]
    out = torch.zeros((len(d3), 9), device=DEV)
    hip.sh_encode(len(d3), 3, T(d3), out)
    assert_same(N(out), fox_golden["sh3"], "sh3")
    for deg in (5, 6, 7, 8):  # SHShader.cu:51-102: the oracle is pinned to the reference kernel for every degree
        out = torch.zeros((len(d3), deg * deg), device=DEV)
        hip.sh_encode(len(d3), deg, T(d3), out)
        assert_same(N(out), oc.sh_encode(d3, deg), "sh%d" % deg)
    with pytest.raises(Exception):
        hip.sh_encode(4, 9, T(d[:4]), torch.zeros((4, 81), device=DEV))  # unsupported degree fails loudly


# ---------------------------------------------------------------------------------------------------
# renderer glue
# ---------------------------------------------------------------------------------------------------
def ragged(rng, R, hi=60):
    cnt = rng.integers(0, hi, R)
    cnt[::13] = 0
    end = np.cumsum(cnt)
    return np.stack([end - cnt, end], -1).astype(np.int32), int(end[-1])


def test_segmented_ops_bit_exact(hip, fox_golden):
    g = fox_golden
    se, val, vec, dsum = g["march_pts_idx_bounds"], g["seg_val"], g["seg_vec"], g["seg_dsum"]
    R, n = len(se), len(val)
    o1 = torch.zeros(R, device=DEV)
    hip.flex_sum_fwd(R, 1, T(val), T(se), o1)
    assert_same(N(o1), g["flex_sum"])
    o3 = torch.zeros((R, 3), device=DEV)
    hip.flex_sum_fwd(R, 3, T(vec), T(se), o3)
    assert_same(N(o3), g["flex_sum_vec"])
    for inc, kf, kb in ((False, "flex_acc_excl", "flex_acc_bwd_excl"), (True, "flex_acc_incl", "flex_acc_bwd_incl")):
        o = torch.zeros(n, device=DEV)
        hip.flex_acc_fwd(R, inc, T(val), T(se), o)
        assert_same(N(o), g[kf])
        hip.flex_acc_bwd(R, inc, T(val), T(se), o)
        assert_same(N(o), g[kb])
    o = torch.zeros(n, device=DEV)
    hip.flex_sum_bwd(R, 1, T(dsum), T(se), o)
    assert_same(N(o), g["flex_sum_bwd"])
    w = (val * F32(0.01)).astype(F32)
    ov = torch.zeros(R, device=DEV)
    hip.weight_var_fwd(R, T(w), T(se), ov)
    assert_same(N(ov), g["weight_var"])
    od = torch.zeros(n, device=DEV)
    hip.weight_var_bwd(R, T(w), T(se), T(dsum), od)
    assert_same(N(od), g["weight_var_bwd"])


def test_early_stop_and_compaction(hip):
    rng = np.random.default_rng(5)
    R = 700
    se, n = ragged(rng, R, 90)
    f0 = (rng.standard_normal(n) * 2 + 3).astype(F32)
    dt = (rng.random(n, dtype=F32) * F32(0.05)).astype(F32)
    w = torch.zeros(n, device=DEV); a = torch.zeros(n, device=DEV)
    mask = torch.zeros(n, dtype=torch.int32, device=DEV); kept = torch.zeros(R, dtype=torch.int32, device=DEV)
    feat = np.zeros((n, 16), F32); feat[:, 0] = f0
    hip.early_stop(R, T(se), T(feat), 16, T(dt), w, a, mask, kept)
    rw, ra, rmask, rse = op.early_stop(f0, dt, se)
    assert np.abs(N(w) - rw).max() <= 2e-6 and np.abs(N(a) - ra).max() <= 2e-6
    # expf differs by an ulp between libms: the mask may flip only where T is within 1e-9 of the threshold
    flips = (N(mask) != rmask).sum()
    assert flips <= 2, flips
    new_se = torch.zeros((R, 2), dtype=torch.int32, device=DEV); tot = torch.zeros(1, dtype=torch.int32, device=DEV)
    hip.segment_scan(R, kept, new_se, tot)
    if flips == 0:
        assert_same(N(new_se), rse, "FilterIdxBounds")
    m = int(tot.item())
    pts = rng.standard_normal((n, 3)).astype(F32); dirs = rng.standard_normal((n, 3)).astype(F32)
    t = rng.random(n, dtype=F32); anchors = rng.integers(0, 100, (n, 3)).astype(np.int32)
    o = [torch.zeros((m, 3), device=DEV), torch.zeros((m, 3), device=DEV), torch.zeros(m, device=DEV), torch.zeros(m, device=DEV),
         torch.zeros((m, 3), dtype=torch.int32, device=DEV)]
    hip.compact_samples(R, T(se), new_se, mask, T(pts), T(dirs), T(dt), T(t), T(anchors), *o)
    exp = op.compact(N(mask), pts, dirs, dt, t, anchors)
    for got, e in zip(o, exp):
        assert_same(N(got), e, "compaction")
    o2 = [torch.zeros_like(x) for x in o]
    src = torch.zeros(max(m, 1), dtype=torch.int32, device=DEV)
    hip.compact_samples_src(R, T(se), new_se, mask, T(pts), T(dirs), T(dt), T(t), T(anchors), *o2, src)
    for got, e in zip(o2, exp):
        assert_same(N(got), e, "compaction (+src)")
    assert_same(N(src)[:m], np.nonzero(N(mask))[0].astype(np.int32), "source rows")
    # ScatterIdx of the survivors riding along: equals f2n_scatter_idx on the new bounds
    ray_val = torch.from_numpy(rng.integers(0, 50, R).astype(np.int32)).to(DEV)
    o3 = [torch.zeros_like(x) for x in o]
    got_rv = torch.full((max(m, 1),), -7, dtype=torch.int32, device=DEV)
    hip.compact_samples_src(R, T(se), new_se, mask, T(pts), T(dirs), T(dt), T(t), T(anchors), *o3, src, None, ray_val, got_rv)
    exp_rv = torch.full((max(m, 1),), -7, dtype=torch.int32, device=DEV)
    hip.scatter_idx(R, new_se, ray_val, exp_rv)
    assert_same(N(got_rv), N(exp_rv), "fused scatter_idx")


@pytest.mark.parametrize("gs", [1.0, 0.3])
def test_composite_forward_backward(hip, gs):
    rng = np.random.default_rng(9)
    R = 600
    se, n = ragged(rng, R, 70)
    feat = rng.standard_normal((n, 16)).astype(F32); feat[:, 0] = feat[:, 0] * 2 + 2
    dt = (rng.random(n, dtype=F32) * F32(0.03)).astype(F32)
    t = np.sort(rng.random(n, dtype=F32) * 5)
    rgb = rng.random((n, 3), dtype=F32); bg = rng.random((R, 3), dtype=F32)
    col = torch.zeros((R, 3), device=DEV); disp = torch.zeros(R, device=DEV); dep = torch.zeros(R, device=DEV)
    wts = torch.zeros(n, device=DEV)
    hip.composite_fwd(R, T(se), T(feat), T(dt), T(t), T(rgb), T(bg), col, disp, dep, wts)
    ref = op.composite_fwd(feat, dt, t, rgb, bg, se, want_ctx=True)
    for got, k in ((col, "colors"), (disp, "disparity"), (wts, "weights")):
        assert np.abs(N(got) - ref[k]).max() <= 2e-5 * max(1.0, np.abs(ref[k]).max()), k
    # depth divides by (1 - T_last + 1e-4): for nearly transparent rays an ulp of expf() is amplified ~1e4 times
    assert (np.abs(N(dep) - ref["depth"]) <= 5e-3 * np.abs(ref["depth"]) + 1e-5).all(), "depth"
    # conservation: sum of weights + last transmittance == 1 per non-empty ray
    wsum = oc.flex_sum(N(wts), se)
    assert np.abs(wsum + ref["ctx"]["last_trans"] - 1)[se[:, 1] > se[:, 0]].max() < 1e-4
    dcol = rng.standard_normal((R, 3)).astype(F32); ddisp = rng.standard_normal(R).astype(F32)
    ddep = rng.standard_normal(R).astype(F32) * F32(0.1); dw = rng.standard_normal(n).astype(F32) * F32(0.1)
    ddep[ref["ctx"]["last_trans"] > 0.9] = 0  # keep the ill-conditioned depth denominator out of the gradient check
    drgb = torch.zeros((n, 3), device=DEV); dfeat = torch.full((n, 16), 5.0, device=DEV)
    hip.composite_bwd(R, T(se), T(feat), T(dt), T(t), T(rgb), T(bg), T(dcol), T(ddisp), T(ddep), T(dw), gs, drgb, dfeat)
    rdrgb, rdf0 = op.composite_bwd(ref["ctx"], dt, rgb, bg, se, dcol, ddisp, ddep, dw, gs)
    assert np.abs(N(drgb) - rdrgb).max() <= 1e-5 * max(1.0, np.abs(rdrgb).max())
    assert np.abs(N(dfeat)[:, 0] - rdf0).max() <= 2e-4 * max(1.0, np.abs(rdf0).max()), np.abs(N(dfeat)[:, 0] - rdf0).max()
    assert (N(dfeat)[:, 1:] == 5.0).all()
    # compact density arrays (stride 1) in and out: bit-identical to the strided column-0 accesses
    f0c = T(np.ascontiguousarray(feat[:, 0]))
    col2 = torch.zeros((R, 3), device=DEV); disp2 = torch.zeros(R, device=DEV); dep2 = torch.zeros(R, device=DEV)
    wts2 = torch.zeros(n, device=DEV)
    hip.composite_fwd(R, T(se), f0c, T(dt), T(t), T(rgb), T(bg), col2, disp2, dep2, wts2, f0_stride=1)
    assert_same(N(col2), N(col)); assert_same(N(disp2), N(disp)); assert_same(N(dep2), N(dep)); assert_same(N(wts2), N(wts))
    drgb2 = torch.zeros((n, 3), device=DEV); df0c = torch.zeros(n, device=DEV)
    hip.composite_bwd(R, T(se), f0c, T(dt), T(t), T(rgb), T(bg), T(dcol), T(ddisp), T(ddep), T(dw), gs, drgb2, df0c, f0_stride=1,
                      df0_stride=1)
    assert_same(N(drgb2), N(drgb)); assert_same(N(df0c), N(dfeat)[:, 0].copy())
    # WeightVarLoss folded into the two launches: bit-identical to the separate kernels
    var_sep = torch.zeros(R, device=DEV); hip.weight_var_fwd(R, wts, T(se), var_sep)
    col3 = torch.zeros((R, 3), device=DEV); disp3 = torch.zeros(R, device=DEV); dep3 = torch.zeros(R, device=DEV)
    wts3 = torch.zeros(n, device=DEV); var_fused = torch.full((R,), -1.0, device=DEV)
    hip.composite_fwd(R, T(se), f0c, T(dt), T(t), T(rgb), T(bg), col3, disp3, dep3, wts3, f0_stride=1, out_vars=var_fused)
    assert_same(N(var_fused), N(var_sep), "fused weight variance"); assert_same(N(wts3), N(wts))
    dvar = torch.from_numpy(rng.standard_normal(R).astype(F32) * F32(0.1)).to(DEV)
    dw_sep = torch.zeros(n, device=DEV); hip.weight_var_bwd(R, wts, T(se), dvar, dw_sep)
    drgb_a = torch.zeros((n, 3), device=DEV); df0_a = torch.zeros(n, device=DEV)
    hip.composite_bwd(R, T(se), f0c, T(dt), T(t), T(rgb), T(bg), T(dcol), T(ddisp), None, dw_sep, gs, drgb_a, df0_a, f0_stride=1,
                      df0_stride=1)
    drgb_b = torch.zeros((n, 3), device=DEV); df0_b = torch.zeros(n, device=DEV)
    hip.composite_bwd(R, T(se), f0c, T(dt), T(t), T(rgb), T(bg), T(dcol), T(ddisp), None, None, gs, drgb_b, df0_b, f0_stride=1,
                      df0_stride=1, var_weights=wts, dvars=dvar)
    assert_same(N(drgb_b), N(drgb_a), "fused weight-variance backward (drgb)"); assert_same(N(df0_b), N(df0_a), "... (d f0)")
    # empty batch and NULL gradient inputs are accepted
    hip.composite_bwd(R, T(se), T(feat), T(dt), T(t), T(rgb), T(bg), T(dcol), None, None, None, 1.0, drgb, dfeat)
    rdrgb2, _ = op.composite_bwd(ref["ctx"], dt, rgb, bg, se, dcol)
    assert np.abs(N(drgb) - rdrgb2).max() <= 1e-5 * max(1.0, np.abs(rdrgb2).max())


@pytest.mark.parametrize("gs,var_w", [(1.0, 0.01), (0.3, 0.0)])
def test_composite_train_equals_three_launches(hip, gs, var_w):
    """f2n_composite_train (compositing forward + the loss of ExpRunner::Train + compositing backward in one launch) against
    f2n_composite_fwd -> f2n_train_loss -> f2n_composite_bwd: colours, weights, d rgb, d f0 and the TV gradient bit for bit, the
    reported loss values to rounding (they sum pre-scaled terms).  Ragged rays including empty ones and rays longer than a row."""
    rng = np.random.default_rng(41)
    R, E, D = 1000, 300, 16
    se, n = ragged(rng, R, 90)
    f0 = (rng.standard_normal(n) * 2 + 2).astype(F32)
    dt = (rng.random(n, dtype=F32) * F32(0.03)).astype(F32)
    t = np.sort(rng.random(n, dtype=F32) * 5)
    rgb = rng.random((n, 3), dtype=F32); bg = rng.random((R, 3), dtype=F32); gt = rng.random((R, 3), dtype=F32)
    edge = rng.standard_normal((E, 2, D)).astype(F32)
    disp_w, tv_w = 0.0005, 0.1
    d = dict(se=T(se), f0=T(f0), dt=T(dt), t=T(t), rgb=T(rgb), bg=T(bg), gt=T(gt), edge=T(edge))
    # the three launches
    col = torch.zeros((R, 3), device=DEV); disp = torch.zeros(R, device=DEV); dep = torch.zeros(R, device=DEV)
    wts = torch.zeros(n, device=DEV); var = torch.zeros(R, device=DEV)
    hip.composite_fwd(R, d["se"], d["f0"], d["dt"], d["t"], d["rgb"], d["bg"], col, disp, dep, wts, f0_stride=1, out_vars=var)
    losses = torch.zeros(8, device=DEV)
    dc = torch.zeros((R, 3), device=DEV); dd = torch.zeros(R, device=DEV); dv = torch.zeros(R, device=DEV)
    de = torch.zeros((E, 2, D), device=DEV)
    hip.train_loss(R, col, d["gt"], disp, var, E, D, d["edge"], var_w, disp_w, tv_w, losses, dc, dd, dv, de)
    drgb = torch.zeros((n, 3), device=DEV); df0 = torch.zeros(n, device=DEV)
    hip.composite_bwd(R, d["se"], d["f0"], d["dt"], d["t"], d["rgb"], d["bg"], dc, dd, None, None, gs, drgb, df0, f0_stride=1, df0_stride=1,
                      var_weights=wts, dvars=dv)
    # one launch
    col2 = torch.full((R, 3), -7.0, device=DEV); wts2 = torch.full((n,), -7.0, device=DEV)
    drgb2 = torch.full((n, 3), -7.0, device=DEV); df02 = torch.full((n,), -7.0, device=DEV)
    de2 = torch.full((E, 2, D), -7.0, device=DEV); losses2 = torch.full((8,), 123.0, device=DEV)
    hip.composite_train(R, d["se"], d["f0"], 1, d["dt"], d["t"], d["rgb"], d["bg"], d["gt"], var_w, disp_w, tv_w, gs, E, D, d["edge"], de2,
                        col2, wts2, drgb2, df02, 1, losses2)
    assert_same(N(col2), N(col), "colours"); assert_same(N(wts2), N(wts), "weights")
    assert_same(N(de2), N(de), "TV gradient")
    assert_same(N(drgb2), N(drgb), "d rgb"); assert_same(N(df02), N(df0), "d f0")
    a, b = N(losses2), N(losses)
    assert np.abs(a[:6] - b[:6]).max() <= 3e-6 * np.abs(b[:6]).max(), (a, b)
    assert a[6] == 0.0 and a[7] == 0.0
    # no edge samples: the TV terms vanish, the rest stands
    losses3 = torch.zeros(8, device=DEV)
    hip.composite_train(R, d["se"], d["f0"], 1, d["dt"], d["t"], d["rgb"], d["bg"], d["gt"], var_w, disp_w, tv_w, gs, 0, D, None, None,
                        col2, wts2, drgb2, df02, 1, losses3)
    assert_same(N(drgb2), N(drgb)); assert float(losses3[4]) == 0.0 and abs(float(losses3[1]) - float(losses[1])) <= 3e-6


def test_adam(hip):
    rng = np.random.default_rng(13)
    n = 4096 * 3
    p = rng.standard_normal(n).astype(F32); g = (rng.standard_normal(n) * 1e-3).astype(F32); g[::3] = 0
    m = (rng.standard_normal(n) * 1e-3).astype(F32); v = (rng.random(n) * 1e-6).astype(F32)
    for step, wd in ((1, 0.0), (10, 1e-6)):
        tp, tm, tv = T(p), T(m), T(v)
        ph = torch.zeros(n, dtype=torch.float16, device=DEV)
        hip.adam_step(n, tp, T(g * 128), 1.0 / 128, False, tm, tv, step, 1e-2, 0.9, 0.99, 1e-15, wd, ph)
        rp, rm, rv = op.adam_step(p, g, m, v, step, 1e-2, 0.9, 0.99, 1e-15, wd)
        assert np.abs(N(tp) - rp).max() <= 1e-6 * np.abs(rp).max() + 1e-7
        assert np.abs(N(tm) - rm).max() <= 1e-9 and np.abs(N(tv) - rv).max() <= 1e-12
        assert_same(N(ph).view(np.uint16), oc.f2h(N(tp)), "h16 refresh")
    # h16-gradient variant (hash table): grad is consumed and re-zeroed
    gh = (g * 128).astype(np.float16)
    tp, tm, tv, tg = T(p), T(m), T(v), T(gh)
    ph = torch.zeros(n, dtype=torch.float16, device=DEV)
    hip.adam_step_h16grad(n, tp, tg, 1.0 / 128, tm, tv, 3, 1e-2, 0.9, 0.99, 1e-15, 0.0, ph, True)
    rp, rm, rv = op.adam_step(p, gh.astype(F32) / F32(128), m, v, 3, 1e-2)
    assert np.abs(N(tp) - rp).max() <= 1e-6 * np.abs(rp).max() + 1e-7
    assert (N(tg).view(np.uint16) == 0).all()
    assert_same(N(ph).view(np.uint16), oc.f2h(N(tp)), "table refresh")
    # entries with zero gradient and zero moments do not move (dense Adam == sparse Adam there)
    z = np.zeros(8, F32)
    tz = T(z + 1); hip.adam_step(8, tz, T(z), 1.0, False, T(z), T(z), 5, 1e-2, 0.9, 0.99, 1e-15, 0.0, None)
    assert (N(tz) == 1).all()


def test_train_loss_and_gradients(hip):
    """f2n_train_loss against the reference's formulas (ExpRunner.cpp:95-120) evaluated in float64."""
    rng = np.random.default_rng(17)
    R, E, D = 777, 300, 16
    pred = rng.random((R, 3), dtype=F32); gt = rng.random((R, 3), dtype=F32)
    disp = rng.random(R, dtype=F32) * F32(3.); var = rng.random(R, dtype=F32) * F32(.2)
    edge = rng.standard_normal((E, 2, D)).astype(F32)
    var_w, disp_w, tv_w = 0.01, 0.0005, 0.1
    losses = torch.zeros(8, device=DEV)
    dc = torch.zeros((R, 3), device=DEV); dd = torch.zeros(R, device=DEV); dv = torch.zeros(R, device=DEV)
    de = torch.zeros((E, 2, D), device=DEV)
    hip.train_loss(R, T(pred), T(gt), T(disp), T(var), E, D, T(edge), var_w, disp_w, tv_w, losses, dc, dd, dv, de)
    p64, g64 = pred.astype(np.float64), gt.astype(np.float64)
    r = np.sqrt((p64 - g64) ** 2 + 1e-4)
    color, mse = r.mean(), ((p64 - g64) ** 2).mean()
    dl, vl = (disp.astype(np.float64) ** 2).mean(), np.sqrt(var.astype(np.float64) + 1e-2).mean()
    diff = edge[:, 0].astype(np.float64) - edge[:, 1].astype(np.float64)
    tv = (diff ** 2).mean()
    want = np.array([color + var_w * vl + disp_w * dl + tv_w * tv, color, vl, dl, tv, mse])
    assert np.abs(N(losses)[:6] - want).max() <= 2e-6 * np.abs(want).max(), (N(losses), want)
    close = lambda a, b: np.abs(a - b).max() <= 2e-6 * np.abs(b).max()
    assert close(N(dc), (p64 - g64) / r / (3 * R))
    assert close(N(dd), disp_w * 2 * disp.astype(np.float64) / R)
    assert close(N(dv), var_w * .5 / np.sqrt(var.astype(np.float64) + 1e-2) / R)
    assert close(N(de)[:, 0], tv_w * 2 * diff / (E * D)) and close(N(de)[:, 1], -tv_w * 2 * diff / (E * D))
    # the no-sample case of Renderer.cpp:83-97: colour term only
    hip.train_loss(R, T(pred), T(gt), None, None, 0, 0, None, var_w, disp_w, tv_w, losses, None, None, None, None)
    assert abs(float(losses[0]) - color) <= 2e-6 * color and float(losses[2]) == 0.0 and float(losses[4]) == 0.0


def test_nonfinite_flags_and_skipped_adam(hip):
    rng = np.random.default_rng(19)
    a = rng.standard_normal(3072).astype(F32); b = rng.standard_normal(7168).astype(F32)
    flags = torch.full((3,), 7, dtype=torch.int32, device=DEV)
    hip.nonfinite_flags(a.size, T(a), b.size, T(b), flags)
    assert N(flags).tolist() == [0, 0, 0]
    b2 = b.copy(); b2[5000] = np.inf
    hip.nonfinite_flags(a.size, T(a), b.size, T(b2), flags)
    assert N(flags).tolist() == [0, 1, 1]
    a2 = a.copy(); a2[-1] = np.nan
    hip.nonfinite_flags(a.size, T(a2), b.size, T(b), flags)
    assert N(flags).tolist() == [1, 0, 1]
    # Adam predicated on flags[2]: nothing moves, but a consumed h16 gradient table is still cleared
    n = 4096
    p = rng.standard_normal(n).astype(F32); g = rng.standard_normal(n).astype(F32)
    m = rng.standard_normal(n).astype(F32); v = rng.random(n, dtype=F32)
    tp, tm, tv = T(p), T(m), T(v)
    ph = torch.ones(n, dtype=torch.float16, device=DEV)
    hip.adam_step(n, tp, T(g), 1.0, False, tm, tv, 2, 1e-2, 0.9, 0.99, 1e-15, 1e-6, ph, flags[2:])
    assert_same(N(tp), p); assert_same(N(tm), m); assert_same(N(tv), v); assert (N(ph) == 1).all()
    tg = T(g.astype(np.float16))
    hip.adam_step_h16grad(n, tp, tg, 1.0 / 128, tm, tv, 2, 1e-2, 0.9, 0.99, 1e-15, 0.0, ph, True, flags[2:])
    assert_same(N(tp), p); assert_same(N(tm), m); assert (N(tg).view(np.uint16) == 0).all() and (N(ph) == 1).all()
    flags.zero_()
    hip.adam_step(n, tp, T(g), 1.0, False, tm, tv, 2, 1e-2, 0.9, 0.99, 1e-15, 1e-6, ph, flags[2:])
    assert (N(tp) != p).any()
    # fused zero_grad: the fp32 gradient is cleared after use -- on the applied and on the skipped path
    for bad in (False, True):
        tg = T(g)
        hip.nonfinite_flags(a.size, T(a2 if bad else a), b.size, T(b), flags)
        before = N(tp).copy()
        hip.adam_step(n, tp, tg, 1.0, False, tm, tv, 3, 1e-2, 0.9, 0.99, 1e-15, 1e-6, ph, flags[2:], zero_grad=True)
        assert (N(tg) == 0).all() and bool((N(tp) == before).all()) == bad


def test_adam_small_groups_equals_separate_launches(hip):
    """f2n_adam_small_groups = f2n_nonfinite_flags + one f2n_adam_step per group, bit for bit, on the applied and on the
    dropped path."""
    rng = np.random.default_rng(23)
    sizes = (3072, 7168, 800)
    for bad in (False, True):
        base = []
        for k, n in enumerate(sizes):
            g = rng.standard_normal(n).astype(F32)
            if bad and k == 1:
                g[n // 2] = np.inf
            base.append(dict(p=rng.standard_normal(n).astype(F32), g=g, m=rng.standard_normal(n).astype(F32) * F32(0.1),
                             v=rng.random(n, dtype=F32) * F32(0.01)))
        scales, wds, rounds = (1.0 / 128, 1.0 / 64, 1.0), (1e-6, 1e-6, 1e-6), (True, True, False)
        # separate launches
        sep = [{k: T(v) for k, v in b.items()} for b in base]
        sep_h = [torch.zeros(n, dtype=torch.float16, device=DEV) for n in sizes]
        flags_a = torch.full((3,), 9, dtype=torch.int32, device=DEV)
        hip.nonfinite_flags(sizes[0], sep[0]["g"], sizes[1], sep[1]["g"], flags_a)
        for k in range(3):
            hip.adam_step(sizes[k], sep[k]["p"], sep[k]["g"], scales[k], rounds[k], sep[k]["m"], sep[k]["v"], 7, 3e-3, 0.9, 0.99, 1e-15,
                          wds[k], sep_h[k] if k < 2 else None, flags_a[2:], zero_grad=True)
        # one launch
        fus = [{k: T(v) for k, v in b.items()} for b in base]
        fus_h = [torch.zeros(n, dtype=torch.float16, device=DEV) for n in sizes]
        flags_b = torch.full((3,), 9, dtype=torch.int32, device=DEV)
        hip.adam_small_groups([dict(param=fus[k]["p"], grad=fus[k]["g"], exp_avg=fus[k]["m"], exp_avg_sq=fus[k]["v"],
                                    param_h=fus_h[k] if k < 2 else None, grad_scale=scales[k], weight_decay=wds[k],
                                    grad_round_h16=rounds[k], check_finite=k < 2) for k in range(3)],
                              7, 3e-3, 0.9, 0.99, 1e-15, True, flags_b)
        assert N(flags_a).tolist() == N(flags_b).tolist() == ([0, 1, 1] if bad else [0, 0, 0])
        for k in range(3):
            for key in ("p", "m", "v", "g"):
                assert_same(N(fus[k][key]), N(sep[k][key]), "group %d %s" % (k, key))
            assert (N(fus[k]["g"]) == 0).all()
            if bad:
                assert_same(N(fus[k]["p"]), base[k]["p"])
            if k < 2:
                assert_same(N(fus_h[k]).view(np.uint16), N(sep_h[k]).view(np.uint16))


def test_adam_fused_equals_separate_launches(hip):
    """f2n_adam_fused (every small group + the h16-gradient table in ONE launch, predicated on a flag computed before) =
    one f2n_adam_step per group + f2n_adam_step_h16grad, bit for bit, on the applied and on the dropped path."""
    rng = np.random.default_rng(29)
    sizes = (3072, 7168, 800)
    n_tab = 17 << 12
    for bad in (False, True):
        base = [dict(p=rng.standard_normal(n).astype(F32), g=rng.standard_normal(n).astype(F32), m=rng.standard_normal(n).astype(F32) * F32(0.1),
                     v=rng.random(n, dtype=F32) * F32(0.01)) for n in sizes]
        tab = dict(p=rng.standard_normal(n_tab).astype(F32), g=(rng.standard_normal(n_tab) * 3).astype(np.float16),
                   m=rng.standard_normal(n_tab).astype(F32) * F32(0.1), v=rng.random(n_tab, dtype=F32) * F32(0.01))
        scales, wds, rounds = (1.0 / 128, 1.0 / 64, 1.0), (1e-6, 1e-6, 1e-6), (True, True, False)
        flag = torch.tensor([1 if bad else 0], dtype=torch.int32, device=DEV)
        out = []
        for fused in (False, True):
            grp = [{k: T(v) for k, v in b.items()} for b in base]
            hs = [torch.zeros(n, dtype=torch.float16, device=DEV) for n in sizes]
            tb = {k: T(v) for k, v in tab.items()}
            th = torch.zeros(n_tab, dtype=torch.float16, device=DEV)
            if fused:
                hip.adam_fused([dict(param=grp[k]["p"], grad=grp[k]["g"], exp_avg=grp[k]["m"], exp_avg_sq=grp[k]["v"],
                                     param_h=hs[k] if k < 2 else None, grad_scale=scales[k], weight_decay=wds[k], grad_round_h16=rounds[k])
                                for k in range(3)],
                               dict(param=tb["p"], grad_h=tb["g"], exp_avg=tb["m"], exp_avg_sq=tb["v"], param_h=th, grad_scale=1.0 / 128, n=n_tab),
                               7, 3e-3, 0.9, 0.99, 1e-15, True, flag)
            else:
                for k in range(3):
                    hip.adam_step(sizes[k], grp[k]["p"], grp[k]["g"], scales[k], rounds[k], grp[k]["m"], grp[k]["v"], 7, 3e-3, 0.9, 0.99, 1e-15,
                                  wds[k], hs[k] if k < 2 else None, flag, zero_grad=True)
                hip.adam_step_h16grad(n_tab, tb["p"], tb["g"], 1.0 / 128, tb["m"], tb["v"], 7, 3e-3, 0.9, 0.99, 1e-15, 0.0, th, True, flag)
            out.append((grp, hs, tb, th))
        (ga, ha, ta, tha), (gb, hb, tb2, thb) = out
        for k in range(3):
            for key in ("p", "m", "v", "g"):
                assert_same(N(ga[k][key]), N(gb[k][key]), "group %d %s" % (k, key))
            if k < 2:
                assert_same(N(ha[k]).view(np.uint16), N(hb[k]).view(np.uint16))
        for key in ("p", "m", "v"):
            assert_same(N(ta[key]), N(tb2[key]), "table " + key)
        assert (N(tb2["g"]).view(np.uint16) == 0).all() and (N(ta["g"]).view(np.uint16) == 0).all()
        assert_same(N(tha).view(np.uint16), N(thb).view(np.uint16))
        if bad:
            assert_same(N(tb2["p"]), tab["p"])


@pytest.mark.parametrize("n_use,bad,side,dirty,leave,amp", [(40000, False, False, False, False, 1e-3), (40000, False, True, True, False, 1e-3), (40000, True, True, False, False, 1e-3),
                                                             (33000, False, False, True, False, 1e-3), (900, False, True, False, False, 1e-3), (900, True, False, False, False, 1e-3),
                                                             (40000, False, True, True, True, 1e-3), (40000, True, False, False, True, 1e-3),
                                                             (33000, False, False, True, False, 1e-6), (40000, False, True, False, False, 1e-6)])
def test_step_tail_equals_separate_launches(hip, fox_state, fox_golden, n_use, bad, side, dirty, leave, amp):
    """f2n_field_bwd_step_tail (round 6) -- the field backward with the REST of the training step re-ordered around its scatter:
    deferred reductions, finiteness flags and the small groups' Adam behind the field-MLP backward (on a second stream when one is
    given), the hash table's Adam applied by the scatter's owner blocks to the slices they have just summed -- leaves every
    parameter, moment, f16 working copy, flag and (zeroed) gradient buffer bit-identical to f2n_field_bwd_dyn -> f2n_reduce_deferred
    -> f2n_nonfinite_flags -> f2n_adam_fused.  Large batch (binned scatter: the owners step the table) and small batch (atomics: the
    ordinary table pass runs behind them), applied and dropped (non-finite MLP gradient) iterations, a gradient table that was
    not clean on entry (what the full-queue fallback's atomics leave: summed in, then cleared).  leave: a data-parallel host's form of the
    call -- everything but the table's Adam, which the caller launches itself once its exchange is through (F2nStepTail::
    leave_table_to_caller), and a callback between the reductions and the flags (after_reduce: where the small buffers' exchange goes)."""
    st, g = fox_state, fox_golden
    rng = np.random.default_rng(64)
    LOG2 = 14
    grid = make_grid(st, rng, LOG2, scale=0.5)
    gd = grid_dev(grid)
    fparams = rand_params(rng, 1)
    idx = rng.integers(0, len(g["march_pts"]), n_use)
    pts, anchors = np.ascontiguousarray(g["march_pts"][idx]), np.ascontiguousarray(g["march_anchors"][idx])
    n = n_use
    ph = T(oc.f2h(fparams).view(np.float16))
    sx = T(oc.f2h((rng.standard_normal((n, 32)) * 0.3).astype(F32)).view(np.float16))
    # amp (round 6): 1e-3 puts ~500 of |addend| into each of this small table's 68 slices -- the owners leave their fixed-point image for
    # the fp64 route, with and without the table's Adam --, 1e-6 keeps them on it (f2n_debug_counters()[1])
    dfeat = (rng.standard_normal((n, 16)) * amp).astype(F32)
    if bad:
        dfeat[7, 3] = np.inf
    n_tab = 17 << LOG2
    n_full = grid.table_f32.size
    sizes = (3072, 7168, 800)
    base = [dict(p=rng.standard_normal(k).astype(F32), m=rng.standard_normal(k).astype(F32) * F32(0.1), v=rng.random(k, dtype=F32) * F32(0.01)) for k in sizes]
    g_pre = [None, (rng.standard_normal(sizes[1]) * 0.1).astype(F32), (rng.standard_normal(sizes[2]) * 0.1).astype(F32)]  # (what shade_bwd left)
    tab = dict(p=(rng.standard_normal(n_full) * 1e-2).astype(F32), m=(rng.standard_normal(n_full) * 1e-3).astype(F32), v=rng.random(n_full, dtype=F32) * F32(1e-6))
    g0 = np.zeros(n_full, np.float16)
    if dirty:
        where = rng.integers(0, n_tab, 500)
        g0[where] = (rng.standard_normal(500) * 0.05).astype(np.float16)
    scales, wds, rounds = (1.0 / 128, 1.0 / 128, 1.0), (1e-6, 1e-6, 1e-6), (True, True, False)
    tail_stream = torch.cuda.Stream() if side else None
    out = []
    for fused in (False, True):
        grp = [{k: T(v) for k, v in b.items()} for b in base]
        grads = [torch.zeros(sizes[0], device=DEV), T(g_pre[1]), T(g_pre[2])]
        hs = [torch.zeros(k, dtype=torch.float16, device=DEV) for k in sizes]
        tb = {k: T(v) for k, v in tab.items()}
        th = T(oc.f2h(tab["p"]).view(np.float16))
        gtab = T(g0.copy())
        flags = torch.full((4,), -7, dtype=torch.int32, device=DEV)
        groups = [dict(param=grp[k]["p"], grad=grads[k], exp_avg=grp[k]["m"], exp_avg_sq=grp[k]["v"], param_h=hs[k] if k < 2 else None,
                       grad_scale=scales[k], weight_decay=wds[k], grad_round_h16=rounds[k]) for k in range(3)]
        args = (n, None, 0, grid.n_volumes, gd["prim"], gd["lidx"], gd["lsize"], gd["bias"], gd["scale"], T(pts), T(anchors), 3, ph, sx, T(dfeat), 128.0,
                grads[0], gtab, 1 << LOG2)
        hip.deferred_reset()
        hip.debug_counters(reset=True)
        by_owners = None
        called = []
        if fused:
            by_owners = hip.field_bwd_step_tail(*args, grads[0], grads[1], flags, groups,
                                                dict(param=tb["p"], exp_avg=tb["m"], exp_avg_sq=tb["v"], param_h=th, grad_scale=1.0 / 128, n=n_tab),
                                                7, 3e-3, 0.9, 0.99, 1e-15, tail_stream=tail_stream, leave_table_to_caller=leave,
                                                after_reduce=(lambda chain: called.append(chain)) if leave else None)
            if leave:
                assert by_owners == 0 and len(called) == 1
                hip.adam_fused([], dict(param=tb["p"], grad_h=gtab, exp_avg=tb["m"], exp_avg_sq=tb["v"], param_h=th, grad_scale=1.0 / 128, n=n_tab),
                               7, 3e-3, 0.9, 0.99, 1e-15, True, flags[2:3])
        else:
            hip.field_bwd_dyn(*args, 1)
            hip.reduce_deferred()
            hip.nonfinite_flags(sizes[0], grads[0], sizes[1], grads[1], flags)
            hip.adam_fused(groups, dict(param=tb["p"], grad_h=gtab, exp_avg=tb["m"], exp_avg_sq=tb["v"], param_h=th, grad_scale=1.0 / 128, n=n_tab),
                           7, 3e-3, 0.9, 0.99, 1e-15, True, flags[2:3])
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        if n_use >= 32768:  # (the binned scatter ran: which route its owners took)
            f64_slices = hip.debug_counters()[1]
            assert (f64_slices == 0) if (amp < 1e-5 and not bad) else (f64_slices > 0), (fused, amp, f64_slices)
        out.append((grp, grads, hs, tb, th, gtab, flags, by_owners))
    (ga, gra, ha, ta, tha, gta, fa, _), (gb, grb, hb, tb2, thb, gtb, fb, by_owners) = out
    assert by_owners == (1 if (n_use >= 32768 and not leave) else 0)
    assert (N(fa)[:3] == N(fb)[:3]).all() and int(N(fa)[2]) == (1 if bad else 0)
    for k in range(3):
        for key in ("p", "m", "v"):
            assert_same(N(ga[k][key]), N(gb[k][key]), "group %d %s" % (k, key))
        assert (N(gra[k]) == 0).all() and (N(grb[k]) == 0).all()  # zero_grad, also on the dropped path
        if k < 2:
            assert_same(N(ha[k]).view(np.uint16), N(hb[k]).view(np.uint16))
    exact = n_use >= 32768 or bad or not torch.cuda.is_available()
    for key in ("p", "m", "v"):
        if exact:
            assert_same(N(ta[key]), N(tb2[key]), "table " + key)
        else:  # the small-batch scatter adds f16 atomics in arrival order: two runs of it differ in a few low-order bits of the gradient
            assert np.abs(N(ta[key]) - N(tb2[key])).max() <= 1e-4 and (N(ta[key]) != N(tb2[key])).mean() < 0.02, key
    if exact:
        assert_same(N(tha).view(np.uint16), N(thb).view(np.uint16), "f16 working table")
    assert (N(gta).view(np.uint16)[:n_tab] == 0).all() and (N(gtb).view(np.uint16)[:n_tab] == 0).all()
    if bad:
        assert_same(N(tb2["p"]), tab["p"])
    else:
        assert (N(tb2["p"])[:n_tab] != tab["p"][:n_tab]).mean() > 0.99  # every slice was stepped, also the ones nothing landed in
        assert_same(N(tb2["p"])[n_tab:], tab["p"][n_tab:], "beyond the active prefix")


def test_img2world_rays_and_pixel_gather(hip, fox_state, fox_golden):
    """Dataset.cu:93-123 on the device: bit-exact against the oracle (itself pinned on the reference kernel), for the fox
    cameras, for strongly distorted synthetic cameras, and against the committed golden rays."""
    st, g = fox_state, fox_golden
    rng = np.random.default_rng(41)
    n = 20000
    H, W = [int(v) for v in st["image_hw"]]
    cam = rng.integers(0, len(st["poses"]), n).astype(np.int32)
    ij = np.stack([rng.integers(0, H, n), rng.integers(0, W, n)], -1).astype(np.int32)
    strong = (rng.standard_normal(st["dist_params"].shape) * [0.2, 0.05, 0.01, 0.01]).astype(F32)
    for dist in (st["dist_params"], strong):
        o = torch.zeros((n, 3), device=DEV); d = torch.zeros((n, 3), device=DEV)
        hip.img2world_rays(n, T(st["poses"]), T(st["intri"]), T(dist), T(cam), T(ij), o, d)
        ro, rd = oc.img2world(st["poses"], st["intri"], dist, cam, ij)
        assert_same(N(o), ro, "ray origins")
        ok = ~np.isnan(rd)
        assert ok.mean() > 0.99 and (np.isnan(N(d)) == ~ok).all()
        assert (N(d).view(np.uint32)[ok] == rd.view(np.uint32)[ok]).all(), "ray directions"
    gij = (g["ij"] - F32(.5)).astype(np.int32)
    m = len(gij)
    o = torch.zeros((m, 3), device=DEV); d = torch.zeros((m, 3), device=DEV)
    hip.img2world_rays(m, T(st["poses"]), T(st["intri"]), T(st["dist_params"]), T(g["cam"]), T(gij), o, d)
    assert_same(N(o), g["rays_o"], "golden origins"); assert_same(N(d), g["rays_d_raw"], "golden directions")
    # pixel / bounds gather from resident images
    C, h, w = 5, 13, 17
    images = rng.random((C, h, w, 3), dtype=F32)
    cb = rng.random((C, 2), dtype=F32)
    cam2 = rng.integers(0, C, 3000).astype(np.int32)
    ij2 = np.stack([rng.integers(0, h, 3000), rng.integers(0, w, 3000)], -1).astype(np.int32)
    col = torch.zeros((3000, 3), device=DEV); bnd = torch.zeros((3000, 2), device=DEV)
    hip.gather_pixels(3000, h, w, T(images), T(cb), T(cam2), T(ij2), col, bnd)
    assert_same(N(col), images[cam2, ij2[:, 0], ij2[:, 1]], "pixels"); assert_same(N(bnd), cb[cam2], "bounds")
    # Dataset::RandRaysData in one launch (f2n_draw_ray_batch): the draws it reports, pushed through the two kernels above, give
    # the rays / colours / bounds it wrote; images come from the set only, pixels cover the whole image
    nC = len(st["poses"])
    imgs = rng.random((nC, 9, 11, 3), dtype=F32); cbn = rng.random((nC, 2), dtype=F32)
    subset = np.array([i for i in range(nC) if i % 8 != 0], np.int32)
    nb = 50000
    u = rng.random((nb, 3), dtype=F32); u[0] = [0.0, 0.0, 0.0]; u[1] = np.nextafter(F32(1), F32(0))
    cam3 = torch.zeros(nb, dtype=torch.int32, device=DEV); ij3 = torch.zeros((nb, 2), dtype=torch.int32, device=DEV)
    o3 = torch.zeros((nb, 3), device=DEV); d3 = torch.zeros((nb, 3), device=DEV)
    c3 = torch.zeros((nb, 3), device=DEV); b3 = torch.zeros((nb, 2), device=DEV)
    hip.draw_ray_batch(nb, T(u), T(subset), 9, 11, T(st["poses"]), T(st["intri"]), T(st["dist_params"]), T(imgs), T(cbn), cam3, ij3, o3, d3,
                       c3, b3)
    cam_h, ij_h = N(cam3), N(ij3)
    assert np.isin(cam_h, subset).all() and set(np.unique(cam_h)) == set(subset.tolist())
    assert ij_h.min() == 0 and ij_h[:, 0].max() == 8 and ij_h[:, 1].max() == 10
    assert cam_h[0] == subset[0] and (ij_h[0] == 0).all() and cam_h[1] == subset[-1] and (ij_h[1] == [8, 10]).all()
    o4 = torch.zeros((nb, 3), device=DEV); d4 = torch.zeros((nb, 3), device=DEV)
    hip.img2world_rays(nb, T(st["poses"]), T(st["intri"]), T(st["dist_params"]), cam3, ij3, o4, d4)
    assert_same(N(o3), N(o4), "drawn origins"); assert_same(N(d3).view(np.uint32), N(d4).view(np.uint32), "drawn directions")
    assert_same(N(c3), imgs[cam_h, ij_h[:, 0], ij_h[:, 1]], "drawn colours"); assert_same(N(b3), cbn[cam_h], "drawn bounds")


def test_empty_and_ragged_inputs(hip, fox_state):
    """Every entry point accepts an empty batch (status OK, nothing written), and the per-ray kernels accept rays
    without samples between rays with samples."""
    st = fox_state
    S = 123.0
    f = lambda *shape: torch.full(shape, S, device=DEV)
    i = lambda *shape: torch.full(shape, 77, dtype=torch.int32, device=DEV)
    h = lambda *shape: torch.full(shape, S, dtype=torch.float16, device=DEV)
    ph1, ph2 = h(3072), h(7168)
    log2 = 10
    local = 1 << log2
    lidx = T((np.arange(16) * local).astype(np.int32)); lsize = T(np.full(16, local, np.int32)); scale = T(oc.level_scales())
    prim, bias, nv = T(st["prim_pool"]), T(st["bias_pool"]), int(st["n_volumes"])
    table = h(16 * local, 2)
    outs = []
    def keep(t):
        outs.append(t)
        return t
    hip.normalize_dirs(0, f(4, 3), keep(f(4, 3)))
    hip.hash_fwd(0, nv, table, prim, lidx, lsize, bias, scale, f(4, 3), True, i(4), 1, keep(h(4, 32)))
    hip.hash_bwd(0, nv, prim, lidx, lsize, bias, scale, f(4, 3), True, i(4), 1, h(4, 32), keep(h(16 * local, 2)), local)
    hip.mlp_fwd(0, 32, 64, 1, ph1, f(4, 32), keep(h(4, 16)))
    hip.mlp_bwd(0, 32, 64, 2, 128.0, ph2, f(4, 32), f(4, 16), keep(f(7168)), keep(f(4, 32)))
    hip.field_fwd(0, nv, table, prim, lidx, lsize, bias, scale, f(4, 3), i(4), 1, ph1, keep(f(4, 16)), keep(f(4)), keep(h(4, 32)))
    hip.field_bwd(0, nv, prim, lidx, lsize, bias, scale, f(4, 3), i(4), 1, ph1, h(4, 32), f(4, 16), 128.0, keep(f(3072)),
                  keep(h(16 * local, 2)), local)
    hip.sh_encode(0, 4, f(4, 3), keep(f(4, 16)))
    hip.shade_fwd(0, f(4, 16), f(4, 3), None, None, ph2, keep(f(4, 3)), keep(h(4, 32)))
    hip.shade_bwd(0, f(4, 3), None, ph2, h(4, 32), 128.0, keep(f(4, 16)), keep(f(7168)), None)
    hip.scatter_idx(0, i(4, 2), i(4), keep(i(4)))
    hip.early_stop(0, i(4, 2), f(4), 1, f(4), keep(f(4)), keep(f(4)), keep(i(4)), keep(i(4)))
    hip.composite_fwd(0, i(4, 2), f(4, 16), f(4), f(4), f(4, 3), f(4, 3), keep(f(4, 3)), keep(f(4)), keep(f(4)), keep(f(4)))
    hip.weight_var_fwd(0, f(4), i(4, 2), keep(f(4)))
    hip.adam_step(0, keep(f(8)), f(8), 1.0, False, keep(f(8)), keep(f(8)), 1, 1e-2, 0.9, 0.99, 1e-15, 0.0, None)
    hip.adam_step_h16grad(0, keep(f(8)), h(8), 1.0, keep(f(8)), keep(f(8)), 1, 1e-2, 0.9, 0.99, 1e-15, 0.0, h(8), True)
    hip.img2world_rays(0, f(1, 3, 4), f(1, 3, 3), f(1, 4), i(4), i(4, 2), keep(f(4, 3)), keep(f(4, 3)))
    torch.cuda.synchronize()
    for t in outs:
        v = t.float()
        assert bool(((v == S) | (v == 77)).all()), "an empty call wrote to its outputs"
    # segment_scan of nothing: total 0
    tot = i(1)
    hip.segment_scan(0, i(4), i(4, 2), tot)
    assert int(tot[0]) == 0
    # ragged rays: empty segments in front, in the middle and at the end
    se = np.array([[0, 0], [0, 5], [5, 5], [5, 5], [5, 21], [21, 21]], np.int32)
    n, R = 21, len(se)
    rng = np.random.default_rng(5)
    feat = rng.standard_normal((n, 16)).astype(F32); dt = rng.random(n, dtype=F32) * F32(0.01) + F32(1e-3)
    tt = np.cumsum(dt).astype(F32); rgb = rng.random((n, 3), dtype=F32); bg = rng.random((R, 3), dtype=F32)
    colors, disp, depth, w = f(R, 3), f(R), f(R), f(n)
    hip.composite_fwd(R, T(se), T(feat), T(dt), T(tt), T(rgb), T(bg), colors, disp, depth, w)
    ref = op.composite_fwd(feat, dt, tt, rgb, bg, se)
    for got, k in ((colors, "colors"), (disp, "disparity"), (w, "weights")):
        assert np.abs(N(got) - ref[k]).max() <= 2e-5 * max(1.0, np.abs(ref[k]).max()), k
    for r in (0, 2, 3, 5):
        assert_same(N(colors)[r], bg[r])  # nothing on the ray: the background


def test_errors_are_loud(hip):
    x = torch.zeros((4, 32), device=DEV)
    with pytest.raises(Exception):  # a width no tcnn FullyFusedMLP has (16 / 32 / 64 / 128 run: tests/test_gpu_mlp_shapes.py)
        hip.mlp_fwd(4, 32, 48, 1, torch.zeros(1 << 14, dtype=torch.float16, device=DEV), x, torch.zeros((4, 16), dtype=torch.float16, device=DEV))
    with pytest.raises(Exception):
        hip.mlp_fwd(4, 32, 64, 1, torch.zeros(3072, dtype=torch.float16), x, torch.zeros((4, 16), dtype=torch.float16, device=DEV))  # CPU tensor
