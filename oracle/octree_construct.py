"""TEST INFRASTRUCTURE ONLY -- restatement of the reference's scene normalisation and PersOctree
construction, used to generate the committed fixture tests/golden/fox_state.npz.

Follows (paths under /root/reference/src):
  Dataset/Dataset.cpp:16-146          cams_meta.npy parsing, NormalizeScene, bounds relaxation, split
  PtsSampler/PersSampler.cpp:16-66    DistanceSummary, GetVisiCams
  PtsSampler/PersSampler.cpp:70-118   PersOctree ctor (node / stat / search-order tensors)
  PtsSampler/PersSampler.cpp:359-421  ConstructTreeNode
  PtsSampler/PersSampler.cpp:423-612  PCA, ConstructTrans
  PtsSampler/PersSampler.cpp:614-659  ConstructEdgePool
  PtsSampler/PersSampler.cpp:120-330  ProcOctree (compact / path compression / subdivide)

The reference draws from torch's CUDA/CPU generators of LibTorch 1.13 (random points per node, the
first "good" camera); those streams cannot be reproduced, so parity is STATE-driven: whatever tree this
produces is serialised in the reference's checkpoint byte layout and fed identically to the oracle and
to the HIP build.
"""
import math

import numpy as np
import torch

N_PROS = 12
INIT_NODE_STAT = 1000

NODE_DT = np.dtype([("center", "<f4", 3), ("side_len", "<f4"), ("parent", "<i4"), ("childs", "<i4", 8),
                    ("is_leaf_node", "u1"), ("pad0", "u1", 3), ("trans_idx", "<i4"), ("pad1", "u1", 4)])
TRANS_DT = np.dtype([("w2xz", "<f4", (N_PROS, 2, 4)), ("weight", "<f4", (3, N_PROS)), ("center", "<f4", 3),
                     ("dis_summary", "<f4")])
EDGE_DT = np.dtype([("t_idx_a", "<i4"), ("t_idx_b", "<i4"), ("center", "<f4", 3), ("dir_0", "<f4", 3),
                    ("dir_1", "<f4", 3), ("pad", "u1", 20)])
assert NODE_DT.itemsize == 64 and TRANS_DT.itemsize == 544 and EDGE_DT.itemsize == 64


# ------------------------------------------------------------------------------------------------
# Dataset.cpp:16-146
# ------------------------------------------------------------------------------------------------
def load_scene(cams_meta_path, factor=2.0, bounds_factor=(0.5, 4.0)):
    cam = torch.from_numpy(np.load(cams_meta_path)).to(torch.float32)
    n = cam.shape[0]
    cam = cam.reshape(n, 27)
    poses = cam[:, 0:12].reshape(-1, 3, 4).contiguous()
    intri = cam[:, 12:21].reshape(-1, 3, 3).contiguous()
    intri[:, 0:2, 0:3] = intri[:, 0:2, 0:3] / factor
    dist = cam[:, 21:25].contiguous()
    bounds = cam[:, 25:27].contiguous()
    # NormalizeScene, :127-146
    cam_pos = poses[:, :3, 3].clone()
    center = cam_pos.mean(0)
    radius = torch.linalg.norm(cam_pos - center[None], 2, -1).max().item()
    poses[:, :3, 3] = (cam_pos - center[None]) / radius
    c2w = poses.clone()
    w2c = torch.eye(4)[None].repeat(n, 1, 1)
    w2c[:, :3, :] = c2w
    w2c = torch.linalg.inv(w2c)[:, :3, :].contiguous()
    bounds = bounds / radius
    # relax bounds, :73-76
    bounds = torch.stack([bounds[:, 0] * bounds_factor[0], bounds[:, 1] * bounds_factor[1]], -1)
    bounds = bounds.clamp(1e-2, 1e9).contiguous()
    test = [i for i in range(n) if i % 8 == 0]  # no split.npy for fox -> every 8th, :105-109
    train = [i for i in range(n) if i % 8 != 0]
    return dict(n_images=n, poses=poses, c2w=c2w, w2c=w2c, intri=intri, dist_params=dist, bounds=bounds,
                center=center, radius=radius, train_set=train, test_set=test)


# ------------------------------------------------------------------------------------------------
# PersSampler.cpp:16-66
# ------------------------------------------------------------------------------------------------
def distance_summary(dis):
    dis = dis.reshape(-1)
    if dis.numel() <= 0:
        return 1e8
    log_dis = torch.log(dis)
    thres = torch.quantile(log_dis, 0.25).item()
    mask = (log_dis < thres).to(torch.float32)
    if mask.sum().item() < 1e-3:
        return float(np.exp(np.float32(log_dis.mean().item())))
    return float(np.exp(np.float32(((log_dis * mask).sum() / mask.sum()).item())))


class _VisiCtx:
    """Per-octree constants of GetVisiCams (ray bundle of every train camera at 128 x res_h pixels)."""

    def __init__(self, c2w, intri, bound):
        half_w = intri[0, 0, 2].item()
        half_h = intri[0, 1, 2].item()
        cx, cy, fx, fy = half_w, half_h, intri[0, 0, 0].item(), intri[0, 1, 1].item()
        res_w = 128
        res_h = int(round(res_w / half_w * half_h))
        i = torch.linspace(.5, half_h * 2. - .5, res_h)
        j = torch.linspace(.5, half_w * 2. - .5, res_w)
        ii, jj = torch.meshgrid(i, j, indexing="ij")
        ii, jj = ii.reshape(-1), jj.reshape(-1)
        cam = torch.stack([(jj - cx) / fx, -(ii - cy) / fy, -torch.ones_like(jj)], -1)  # [n_pix, 3]
        self.rays_d = torch.matmul(c2w[:, None, :3, :3], cam[None, :, :, None])[..., 0]  # [n_cams, n_pix, 3]
        self.rays_o = c2w[:, None, :3, 3]  # broadcast [n_cams, 1, 3]
        self.bound = bound


def get_visi_cams(ctx, side_len, center):
    a = ((center - side_len * .5)[None, None] - ctx.rays_o) / ctx.rays_d
    b = ((center + side_len * .5)[None, None] - ctx.rays_o) / ctx.rays_d
    a = torch.nan_to_num(a, 0., 1e6, -1e6)
    b = torch.nan_to_num(b, 0., 1e6, -1e6)
    far = torch.maximum(a, b).min(-1)[0]
    near = torch.minimum(a, b).max(-1)[0]
    far = torch.minimum(far, ctx.bound[:, None, 1])
    near = torch.maximum(near, ctx.bound[:, None, 0])
    mask = (far > near).to(torch.float32).sum(-1)
    return torch.where(mask > 0)[0].tolist()


# ------------------------------------------------------------------------------------------------
# PersSampler.cpp:423-612
# ------------------------------------------------------------------------------------------------
def pca(pts):
    mean = pts.mean(0, keepdim=True)
    moved = pts - mean
    cov = torch.matmul(moved[:, :, None], moved[:, None, :]).mean(0)
    L, V = torch.linalg.eigh(cov)
    L, V = L.to(torch.float32), V.to(torch.float32)
    _, indices = torch.sort(L, 0, descending=True)
    V = V.permute(1, 0).contiguous()[indices].permute(1, 0).contiguous()
    return L[indices].contiguous(), V


def _angle_axis(angle, axis):
    x, y, z = [float(v) for v in axis]
    c, s = math.cos(angle), math.sin(angle)
    C = 1. - c
    return np.array([[c + C * x * x, C * x * y - s * z, C * x * z + s * y],
                     [C * x * y + s * z, c + C * y * y, C * y * z - s * x],
                     [C * x * z - s * y, C * y * z + s * x, c + C * z * z]], np.float32)


def construct_trans(rand_pts, c2w, intri, center, gen, first_cam=None):
    n_virt = N_PROS // 2
    n_cur = c2w.shape[0]
    cam_pos = c2w[:, :3, 3].contiguous()
    cam_axes = torch.linalg.inv(c2w[:, :3, :3]).contiguous()
    dis = torch.linalg.norm(cam_pos - center[None], 2, -1)
    dis_summary = distance_summary(dis)
    normed = (cam_pos - center[None]) / dis[:, None]
    dis_pairs = torch.linalg.norm(normed[None] - normed[:, None], 2, -1).numpy()
    # greedy farthest-point selection of 6 spread cameras, :461-488
    good = [int(torch.randint(n_cur, (1,), generator=gen).item()) if first_cam is None else int(first_cam)]
    marks = np.zeros(n_cur, bool)
    marks[good[0]] = True
    for _ in range(1, min(n_virt, n_cur)):
        cur = np.where(marks[None, :], dis_pairs, np.float32(1e8)).min(1)
        cur[marks] = -2.
        candi = int(np.argmax(cur))  # first maximum, like the strict '>' scan
        marks[candi] = True
        good.append(candi)
    i = 0
    while len(good) < n_virt:
        good.append(good[i])
        i += 1
    cam_scale = (dis / dis_summary).clip(1., 1e9)
    rel = (cam_pos - center[None]) / dis[:, None] * dis[:, None].clip(dis_summary, 1e9)
    g = torch.tensor(good, dtype=torch.long)
    good_cam_pos = rel[g] + center[None]
    good_rel = rel[g]
    good_axis = cam_axes[g].clone()
    good_scale = cam_scale[g]
    expect_z = good_rel / torch.linalg.norm(good_rel, 2, -1, keepdim=True)
    rots = torch.zeros(n_virt, 3, 3)
    for k in range(n_virt):
        fz = good_axis[k, 2, :].numpy().astype(np.float32)
        tz = expect_z[k].numpy().astype(np.float32)
        crossed = np.cross(fz, tz).astype(np.float32)
        cos_val = float(np.dot(fz, tz))
        sin_val = float(np.linalg.norm(crossed))
        angle = math.asin(min(sin_val, 1.0))
        if cos_val < 0.:
            angle = math.pi - angle
        if sin_val > 0:
            crossed = crossed / np.float32(sin_val)
        rots[k] = torch.from_numpy(_angle_axis(angle, crossed))
    good_axis = torch.matmul(good_axis, rots.transpose(1, 2))
    x_axis = good_axis[:, 0, :].contiguous()
    y_axis = good_axis[:, 1, :].contiguous()
    z_axis = good_axis[:, 2, :].contiguous()
    assert (z_axis - expect_z).abs().max().item() < 1e-3
    focal = (intri[0, 0] / intri[0, 2]).item()
    x_axis = x_axis * focal * good_scale[:, None]
    y_axis = y_axis * focal * good_scale[:, None]
    x_axis = torch.cat([x_axis, y_axis], 0)
    z_axis = torch.cat([z_axis, z_axis], 0)
    wp_pos = torch.cat([good_cam_pos, good_cam_pos], 0)
    frame = torch.zeros(N_PROS, 2, 4)
    frame[:, 0, :3] = x_axis
    frame[:, 1, :3] = z_axis
    frame[:, 0, 3] = -(x_axis * wp_pos).sum(-1)
    frame[:, 1, 3] = -(z_axis * wp_pos).sum(-1)
    # PCA weights, :568-597
    tp = torch.matmul(frame[None, :, :, :3], rand_pts[:, None, :, None])[..., 0] + frame[None, :, :, 3]
    dv_da = 1. / tp[:, :, 1]
    dv_db = tp[:, :, 0] / -tp[:, :, 1].square()
    dv_dab = torch.stack([dv_da, dv_db], -1)
    dv_dxyz = torch.matmul(dv_dab[:, :, None, :], frame[None, :, :, :3])[:, :, 0, :]  # [n_pts, 12, 3]
    assert tp[:, :, 1].max().item() < 0.
    tv = tp[:, :, 0] / tp[:, :, 1]
    assert torch.isfinite(tv.mean())
    _, V = pca(tv)
    V = V.permute(1, 0)[:3].contiguous()  # [3, 12]
    jac = torch.matmul(V[None], dv_dxyz)
    jac_w2i = torch.matmul(dv_dxyz, torch.linalg.inv(jac))
    jac_max = jac_w2i.abs().max(1)[0]
    mean_step = (1. / jac_max).mean(0)
    V = V / mean_step[:, None]
    assert torch.isfinite(V).all() and torch.isfinite(frame).all()
    t = np.zeros((), TRANS_DT)
    t["w2xz"] = frame.numpy()
    t["weight"] = V.numpy()
    t["center"] = center.numpy()
    t["dis_summary"] = np.float32(dis_summary)
    return t


# ------------------------------------------------------------------------------------------------
# PersSampler.cpp:70-118, 359-421, 614-659
# ------------------------------------------------------------------------------------------------
class PersOctreeBuilder:
    def __init__(self, max_depth, bbox_side_len, split_dist_thres, c2w, w2c, intri, bound, seed=2022,
                 n_rand_pts=32 * 32 * 32, verbose=False, draws_like_reference=False):
        # draws_like_reference: the reference draws the 32^3 random points of a node BEFORE it knows whether the node becomes a
        # leaf with a warp (PersSampler.cpp:377), i.e. for every visited node; by default only the leaves that need them draw
        # (same trees, fewer draws).  With the flag the generator is consumed exactly as the reference consumes it, which is what
        # the pin against the reference's own code needs (tests/test_oracle_vs_ref.py).
        self.draws_like_reference = draws_like_reference
        self.max_depth, self.split_dist_thres = max_depth, split_dist_thres
        self.c2w, self.w2c, self.intri, self.bound = c2w, w2c, intri, bound
        self.gen = torch.Generator().manual_seed(seed)
        self.n_rand_pts = n_rand_pts
        self.verbose = verbose
        self.ctx = _VisiCtx(c2w, intri, bound)
        self.nodes = [self._new_node()]
        self.nodes[0]["parent"] = -1
        self.trans = []
        self._construct(0, 0, np.zeros(3, np.float32), np.float32(bbox_side_len))
        self.edge_pool = self.construct_edge_pool()

    @staticmethod
    def _new_node():
        nd = np.zeros((), NODE_DT)
        nd["childs"] = -1
        return nd

    def _construct(self, u, depth, center, side_len):
        nd = self.nodes[u]
        nd["center"], nd["side_len"], nd["is_leaf_node"], nd["trans_idx"] = center, side_len, 0, -1
        nd["childs"] = -1
        if depth > self.max_depth:
            nd["is_leaf_node"] = 1
            return
        center_ts = torch.from_numpy(np.asarray(center, np.float32))
        rand_pts = None
        if self.draws_like_reference:
            rand_pts = (torch.rand(self.n_rand_pts, 3, generator=self.gen) - .5) * float(side_len) + center_ts[None]
        visi = get_visi_cams(self.ctx, float(side_len), center_ts)
        cam_dis = torch.linalg.norm(self.c2w[:, :3, 3] - center_ts[None], 2, -1)
        dsum = distance_summary(cam_dis[visi]) if len(visi) else 1e8
        unaddressed = (len(visi) >= N_PROS // 2) and (dsum < float(side_len) * self.split_dist_thres)
        if unaddressed:
            for st in range(8):
                v = len(self.nodes)
                self.nodes.append(self._new_node())
                off = np.array([((st >> 2) & 1) - .5, ((st >> 1) & 1) - .5, (st & 1) - .5], np.float32)
                sub_center = (center + side_len * np.float32(.5) * off).astype(np.float32)
                self.nodes[u]["childs"][st] = v
                self.nodes[v]["parent"] = u
                self._construct(v, depth + 1, sub_center, np.float32(side_len * np.float32(.5)))
        elif len(visi) < N_PROS // 2:
            nd["is_leaf_node"] = 1  # leaf but invalid: not enough visible cameras
        else:
            nd["is_leaf_node"] = 1
            nd["trans_idx"] = len(self.trans)
            if rand_pts is None:
                rand_pts = (torch.rand(self.n_rand_pts, 3, generator=self.gen) - .5) * float(side_len) + center_ts[None]
            self.trans.append(construct_trans(rand_pts, self.c2w[visi], self.intri[0], center_ts, self.gen))
            if self.verbose and len(self.trans) % 50 == 0:
                print("  leaves with warps:", len(self.trans), "nodes:", len(self.nodes), flush=True)

    def construct_edge_pool(self):
        return construct_edge_pool(np.array(self.nodes, NODE_DT))

    def arrays(self):
        return np.array(self.nodes, NODE_DT), np.array(self.trans, TRANS_DT), self.edge_pool


def construct_edge_pool(nodes):
    valid = np.where(nodes["trans_idx"] >= 0)[0]
    out = []
    cen, side, tidx = nodes["center"], nodes["side_len"], nodes["trans_idx"]
    faces = [(0, +1), (0, -1), (1, +1), (1, -1), (2, +1), (2, -1)]
    for ai, a in enumerate(valid):
        bs = valid[ai + 1:]
        if len(bs) == 0:
            break
        a_small = ~(side[a] > side[bs])  # u = a unless a is strictly larger (then swap)
        u = np.where(a_small, a, bs)
        v = np.where(a_small, bs, a)
        len_u = (side[u] * np.float32(.5)).astype(np.float32)
        hit = np.zeros((len(bs), 6), bool)
        pts = np.zeros((len(bs), 6, 3), np.float32)
        for f, (ax, sg) in enumerate(faces):
            p = cen[u].copy()
            p[:, ax] = p[:, ax] + np.float32(sg) * len_u
            bias = (p - cen[v]) / side[v][:, None] * np.float32(2.)
            hit[:, f] = np.abs(bias).max(1) < np.float32(1. + 1e-4)
            pts[:, f] = p
        for bi, f in zip(*np.nonzero(hit)):
            ax = faces[f][0]
            e = np.zeros((), EDGE_DT)
            e["t_idx_a"], e["t_idx_b"], e["center"] = tidx[a], tidx[bs[bi]], pts[bi, f]
            d0, d1 = [k for k in range(3) if k != ax]
            e["dir_0"][d0] = len_u[bi]
            e["dir_1"][d1] = len_u[bi]
            out.append(e)
    return np.array(out, EDGE_DT) if out else np.zeros(0, EDGE_DT)


def build_search_order():
    """PersSampler.cpp:106-117 (std::sort with the bit-trick comparator)."""
    import functools
    out = []
    for st in range(8):
        def less(a, b, st=st):
            bt = (a ^ b) & -(a ^ b)
            return ((a & bt) ^ (st & bt)) != 0
        out.extend(sorted(range(8), key=functools.cmp_to_key(lambda a, b: -1 if less(a, b) else (1 if less(b, a) else 0))))
    return np.array(out, np.uint8)


# ------------------------------------------------------------------------------------------------
# PersSampler.cpp:120-330  ProcOctree
# ------------------------------------------------------------------------------------------------
def proc_octree(nodes, w_stats, a_stats, visit_cnt, compact, subdivide, brute_force):
    """Returns (new_nodes, new_w_stats, new_a_stats).  nodes: NODE_DT array (not modified)."""
    n = len(nodes)
    center, side = nodes["center"].copy(), nodes["side_len"].copy()
    parent, childs = nodes["parent"].copy(), nodes["childs"].copy()
    leaf, trans = nodes["is_leaf_node"].astype(bool).copy(), nodes["trans_idx"].copy()
    while compact:
        for u in range(n):
            if not leaf[u]:
                continue
            if trans[u] < 0 and parent[u] >= 0:
                row = childs[parent[u]]
                row[row == u] = -1
        update = False
        for u in range(1, n):  # the root can not become a leaf
            if not (childs[u] >= 0).any():
                if not leaf[u]:
                    update = True
                leaf[u] = True
        if not update:
            break
    if compact:  # path compression: splice out chains of single-child interior nodes
        def single_child(u):
            idx = np.nonzero(childs[u] >= 0)[0]
            return int(idx[-1]) if len(idx) == 1 else -1
        for u in range(n):
            if leaf[u] and trans[u] < 0:
                continue
            v = parent[u]
            while v >= 0 and parent[v] >= 0 and single_child(v) >= 0:
                vv = parent[v]
                row = childs[vv]
                row[row == v] = u
                parent[u] = vv
                trans[v] = -1
                leaf[v] = True  # the flag to remove it
                v = vv
    keep = (~leaf) | (trans >= 0)
    new_idx = np.full(n, -1, np.int64)
    new_idx[keep] = np.arange(int(keep.sum()))
    inv_idx = np.nonzero(keep)[0]
    assert new_idx[0] == 0 and inv_idx[0] == 0
    k_center, k_side, k_leaf, k_trans = center[keep], side[keep], leaf[keep], trans[keep]
    k_parent, k_childs = parent[keep].copy(), childs[keep].copy()
    pm = k_parent >= 0
    k_parent[pm] = new_idx[k_parent[pm]]
    assert (k_parent[pm] >= 0).all()
    cm = k_childs >= 0
    k_childs[cm] = new_idx[k_childs[cm]]
    assert (k_childs[cm] >= 0).all()
    k_w, k_a = np.asarray(w_stats)[keep].astype(np.int32), np.asarray(a_stats)[keep].astype(np.int32)
    if not subdivide:
        out = np.zeros(len(k_side), NODE_DT)
        out["center"], out["side_len"], out["parent"], out["childs"] = k_center, k_side, k_parent, k_childs
        out["is_leaf_node"], out["trans_idx"] = k_leaf, k_trans
        return out, k_w, k_a
    # sub-divide: depth-first rebuild; visited leaves split into 8 children that inherit the warp
    o_center, o_side, o_parent, o_childs, o_leaf, o_trans, o_w, o_a = [], [], [], [], [], [], [], []

    def push(c, s, p, ch, lf, tr, w, a):
        o_center.append(np.asarray(c, np.float32)); o_side.append(np.float32(s)); o_parent.append(int(p))
        o_childs.append(np.array(ch, np.int32)); o_leaf.append(bool(lf)); o_trans.append(int(tr))
        o_w.append(int(w)); o_a.append(int(a))
        return len(o_side) - 1

    def rec(u, pa):
        new_u = push(k_center[u], k_side[u], pa, k_childs[u], k_leaf[u], k_trans[u], k_w[u], k_a[u])
        if k_leaf[u]:
            assert k_trans[u] >= 0
            if not brute_force and visit_cnt[inv_idx[u]] <= 4:
                return new_u
            for st in range(8):
                off = np.array([((st >> 2) & 1) - .5, ((st >> 1) & 1) - .5, (st & 1) - .5], np.float32)
                sub = (o_center[new_u] + o_side[new_u] * np.float32(.5) * off).astype(np.float32)
                v = push(sub, o_side[new_u] * np.float32(.5), new_u, [-1] * 8, True, o_trans[new_u], o_w[new_u],
                         o_a[new_u])
                o_childs[new_u][st] = v
            o_leaf[new_u] = False
            o_trans[new_u] = -1
            o_w[new_u] = INIT_NODE_STAT
            o_a[new_u] = INIT_NODE_STAT
        else:
            assert k_trans[u] < 0
            for st in range(8):
                c = int(o_childs[new_u][st])
                if c >= 0:
                    o_childs[new_u][st] = rec(c, new_u)
        return new_u

    import sys
    sys.setrecursionlimit(20000)
    rec(0, -1)
    out = np.zeros(len(o_side), NODE_DT)
    out["center"], out["side_len"], out["parent"] = np.array(o_center), np.array(o_side), np.array(o_parent)
    out["childs"], out["is_leaf_node"], out["trans_idx"] = np.array(o_childs), np.array(o_leaf), np.array(o_trans)
    return out, np.array(o_w, np.int32), np.array(o_a, np.int32)
