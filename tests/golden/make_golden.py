#!/usr/bin/env python3
"""Generates the committed fixtures under tests/golden/ (run once, in the build container, where
/root/reference exists):

  fox_state.npz   -- serialised sampler/field state for ngp_fox in the reference's checkpoint byte
                     layout (TreeNode[64 B], TransInfo[544 B], EdgePool[64 B], primes, biases) built by
                     oracle/octree_construct.py from /root/reference/data/example/ngp_fox/cams_meta.npy
                     with the wanjinyou.yaml settings, plus the normalised cameras.
  fox_golden.npz  -- a fixed ray batch and the outputs of the REFERENCE'S OWN KERNELS on it (compiled
                     for CPU by oracle/build_ref.py, called through oracle/ref.py): leaf hit lists, march
                     samples, hash-grid features and gradients, SH basis, FlexOps, CustomOps, Scatter,
                     occupancy votes.  tests/test_golden.py holds the oracle restatement to these bits.

Usage:  python tests/golden/make_golden.py            (needs oracle/_ref/libf2n_ref.so)
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import octree_construct as oc  # noqa: E402
from oracle import ref  # noqa: E402

FOX = "/root/reference/data/example/ngp_fox"
N_LEVELS = 16


def gen_primes(rng, count, lo=1 << 28, hi=1 << 30):
    """Hash3DAnchored.cpp:24-44: rejection-sample primes in [2^28, 2^30)."""
    sieve = np.ones(32769, bool)
    sieve[:2] = False
    for i in range(2, 182):
        if sieve[i]:
            sieve[i * i::i] = False
    small = np.nonzero(sieve)[0].astype(np.int64)
    out = []
    while len(out) < count:
        cand = rng.integers(lo, hi, size=4 * (count - len(out)) + 64, dtype=np.int64)
        ok = np.ones(len(cand), bool)
        for p in small:
            ok &= (cand % p) != 0
        out.extend(cand[ok].tolist())
    return np.array(out[:count], np.int32)


def make_state(path):
    torch.set_num_threads(8)
    sc = oc.load_scene(os.path.join(FOX, "cams_meta.npy"), factor=2.0, bounds_factor=(0.5, 4.0))
    tr = torch.tensor(sc["train_set"])
    # confs/pts_sampler/perspective.yaml + wanjinyou.yaml: bbox_levels 10, max_level 16, split_dist_thres 1.5
    b = oc.PersOctreeBuilder(16, float(1 << 9), 1.5, sc["c2w"][tr].contiguous(), sc["w2c"][tr].contiguous(),
                             sc["intri"][tr].contiguous(), sc["bounds"][tr].contiguous(), seed=2022)
    nodes, trans, edges = b.arrays()
    n_vol = len(trans)
    rng = np.random.default_rng(19970826)
    prim = gen_primes(rng, 3 * N_LEVELS * n_vol).reshape(N_LEVELS, n_vol, 3)
    bias = (rng.random((N_LEVELS * n_vol, 3), dtype=np.float32) * np.float32(1000.) + np.float32(100.))
    render_poses = np.load(os.path.join(FOX, "poses_render.npy")).astype(np.float32).reshape(-1, 3, 4)
    render_poses[:, :3, 3] = (render_poses[:, :3, 3] - sc["center"].numpy()[None]) / np.float32(sc["radius"])
    np.savez_compressed(
        path,
        tree_nodes=nodes.view(np.uint8).reshape(-1), pers_trans=trans.view(np.uint8).reshape(-1),
        edge_pool=edges.view(np.uint8).reshape(-1), search_order=oc.build_search_order(),
        prim_pool=prim, bias_pool=bias.astype(np.float32), n_volumes=np.int32(n_vol),
        poses=sc["poses"].numpy(), intri=sc["intri"].numpy(), dist_params=sc["dist_params"].numpy(),
        bounds=sc["bounds"].numpy(), w2c=sc["w2c"].numpy(), center=sc["center"].numpy(),
        radius=np.float32(sc["radius"]), train_set=np.array(sc["train_set"], np.int32),
        test_set=np.array(sc["test_set"], np.int32), render_poses=render_poses,
        image_hw=np.array([960, 540], np.int32))
    print("state: nodes", len(nodes), "warps", n_vol, "edges", len(edges))


def make_golden(state_path, path):
    st = np.load(state_path)
    rng = np.random.default_rng(2022)
    R = 24
    H, W = st["image_hw"]
    cam = st["train_set"][rng.integers(0, len(st["train_set"]), R)].astype(np.int32)
    ij = np.stack([rng.integers(0, H, R), rng.integers(0, W, R)], -1).astype(np.float32) + np.float32(.5)
    rays_o, rays_d_raw = ref.img2world(st["poses"], st["intri"], st["dist_params"], cam, ij)
    rays_d = (rays_d_raw / np.linalg.norm(rays_d_raw, axis=-1, keepdims=True)).astype(np.float32)
    noise = ((rng.random(1024 + R + 10, dtype=np.float32) - np.float32(.5)) + np.float32(1.)) * np.float32(8.)
    g = dict(cam=cam, ij=ij, rays_o=rays_o, rays_d_raw=rays_d_raw, rays_d=rays_d, noise=noise,
             near=np.float32(0.01), far=np.float32(1e8), sample_l=np.float32(1. / 256.))
    se, oidx, onf = ref.oct_intersect(st["search_order"], rays_o, rays_d, 0.01, 1e8, st["tree_nodes"])
    g.update(oct_start_end=se, oct_idx=oidx, oct_near_far=onf)
    m = ref.ray_march(rays_o, rays_d, noise, 1. / 256., True, se, oidx, onf, st["tree_nodes"], st["pers_trans"])
    for k in ("pts", "dirs", "dt", "t", "anchors", "pts_idx_bounds", "first_oct_dis"):
        g["march_" + k] = m[k]
    m2 = ref.ray_march(rays_o, rays_d, noise, 1. / 256., False, se, oidx, onf, st["tree_nodes"], st["pers_trans"])
    g["march_noscale_bounds"] = m2["pts_idx_bounds"]
    g["march_noscale_t_sum"] = np.float64(m2["t"].astype(np.float64).sum())
    n = len(m["t"])
    print("golden: rays", R, "hits", len(oidx), "samples", n)
    # edge samples
    n_edges = st["edge_pool"].size // 64
    eidx = rng.integers(0, n_edges, 256).astype(np.int32)
    ecoord = (rng.random((256, 2), dtype=np.float32) * np.float32(2.) - np.float32(1.))
    epts, eout = ref.edge_samples(st["edge_pool"], st["pers_trans"], eidx, ecoord)
    g.update(edge_idx=eidx, edge_coord=ecoord, edge_pts=epts, edge_out_idx=eout)
    # hash grid: small table (log2 12) so that the fixture stays small; seeded, regenerated by the test
    log2_t = 12
    local = 1 << log2_t
    pool_halves = N_LEVELS * local * 2
    table = (rng.random(pool_halves, dtype=np.float32) * np.float32(2.) - np.float32(1.)).astype(np.float16)
    local_idx = (np.arange(N_LEVELS) * local).astype(np.int32)
    local_size = np.full(N_LEVELS, local, np.int32)
    q01 = ((m["pts"] + np.float32(1.)) * np.float32(.5)).astype(np.float32)
    vol = np.ascontiguousarray(m["anchors"][:, 0])
    nv = int(st["n_volumes"])
    feat = ref.hash_fwd(table.view(np.uint16), st["prim_pool"], local_idx, local_size, st["bias_pool"], q01, vol, nv)
    gin = (rng.standard_normal((n, 32)).astype(np.float32) * np.float32(0.05)).astype(np.float16)
    gin[rng.random(n) < 0.1] = 0
    gout = ref.hash_bwd(pool_halves, st["prim_pool"], local_idx, local_size, st["bias_pool"], q01, vol, nv,
                        gin.view(np.uint16))
    g.update(hash_log2=np.int32(log2_t), hash_table_seed_check=table[:16].view(np.uint16), hash_feat=feat,
             hash_grad_in=gin.view(np.uint16), hash_grad_out=gout)
    # SH
    g["sh4"] = ref.sh_encode(m["dirs"][::7], 4)
    g["sh3"] = ref.sh_encode(m["dirs"][::31], 3)
    # segmented ops on the march segments
    pse = m["pts_idx_bounds"]
    val = rng.random(n, dtype=np.float32)
    vec = rng.random((n, 3), dtype=np.float32)
    g.update(seg_val=val, seg_vec=vec, flex_sum=ref.flex_sum(val, pse), flex_sum_vec=ref.flex_sum(vec, pse),
             flex_acc_excl=ref.flex_acc(val, pse, False), flex_acc_incl=ref.flex_acc(val, pse, True),
             flex_acc_bwd_excl=ref.flex_acc_bwd(val, pse, False), flex_acc_bwd_incl=ref.flex_acc_bwd(val, pse, True))
    dsum = rng.random(R, dtype=np.float32)
    g.update(seg_dsum=dsum, flex_sum_bwd=ref.flex_sum_bwd(dsum, pse, n))
    w = (val * np.float32(0.01)).astype(np.float32)
    g.update(weight_var=ref.weight_var(w, pse), weight_var_bwd=ref.weight_var_bwd(w, pse, dsum),
             grad_scaling=ref.grad_scaling_bwd(vec, pse, 0.25))
    mask = (val > np.float32(0.3)).astype(np.int32)
    g.update(count_valid=ref.count_valid(pse, mask))
    emb = rng.standard_normal((50, 16)).astype(np.float32)
    sub = pse[:6]  # first 6 rays only, to keep the fixture small
    n_sub = int(sub[-1, 1])
    all_idx = ref.scatter_idx(n_sub, sub, cam[:6])
    to_add = rng.standard_normal((n_sub, 16)).astype(np.float32)
    g.update(emb=emb, scatter_idx=all_idx, scatter_to_add=to_add, scatter_add=ref.scatter_add(emb, all_idx, to_add),
             scatter_add_bwd=ref.scatter_add_bwd(50, all_idx, to_add))
    # occupancy votes
    alphas = rng.random(n, dtype=np.float32) * np.float32(0.05)
    n_nodes = st["tree_nodes"].size // 64
    wa, aa, mk, cnt = ref.mark_visit(n_nodes, pse, np.ascontiguousarray(m["anchors"][:, 1]), w, alphas,
                                     np.zeros(n_nodes, np.int32))
    g.update(occ_alpha=alphas, occ_w_adder=wa, occ_a_adder=aa, occ_mark=mk, occ_cnt=cnt)
    inv = ref.mark_invisible(st["tree_nodes"], st["intri"][st["train_set"]], st["w2c"][st["train_set"]],
                             st["bounds"][st["train_set"]])
    g["invisible_trans_idx"] = inv.view(oc.NODE_DT)["trans_idx"].copy()
    np.savez_compressed(path, **g)


if __name__ == "__main__":
    if not ref.available():
        sys.exit("build oracle/_ref first: python oracle/build_ref.py")
    sp = os.path.join(HERE, "fox_state.npz")
    if "--keep-state" not in sys.argv or not os.path.exists(sp):
        make_state(sp)
    make_golden(sp, os.path.join(HERE, "fox_golden.npz"))
    for f in ("fox_state.npz", "fox_golden.npz"):
        print(f, os.path.getsize(os.path.join(HERE, f)) // 1024, "KiB")
