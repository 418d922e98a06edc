"""TEST INFRASTRUCTURE ONLY -- ctypes binding of oracle/_ref/libf2n_ref.so: the reference's own
``__global__`` kernels compiled for CPU by oracle/build_ref.py (see that file and ref_driver.inc).

``available()`` is False on a clean clone without /root/reference (then the committed golden vectors
under tests/golden/, which were generated through this module, are the pin).
"""
import ctypes
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "_ref", "libf2n_ref.so")

_lib = None


def available():
    return os.path.exists(SO)


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(SO)
    return _lib


def _p(a):
    if a is None:
        return ctypes.c_void_p(0)
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(ctypes.c_void_p)  # keeps a reference to `a` alive for the duration of the call


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _u8(a):
    return np.ascontiguousarray(a, dtype=np.uint8)


def _u16(a):
    return np.ascontiguousarray(a, dtype=np.uint16)


def struct_layout():
    out = np.zeros(18, np.int32)
    lib().ref_struct_layout(_p(out))
    return out


def search_order_table():
    out = np.zeros(64, np.int32)
    lib().ref_search_order(_p(out))
    return out.astype(np.uint8)


def oct_intersect(search_order, rays_o, rays_d, near, far, tree_nodes, max_hits=1024):
    rays_o, rays_d = _f32(rays_o), _f32(rays_d)
    n = rays_o.shape[0]
    bounds = np.stack([np.full(n, near, np.float32), np.full(n, far, np.float32)], -1).copy()
    se = np.zeros((n, 2), np.int32)
    cap = max(1, n * 64)
    while True:
        idx = np.empty(cap, np.int32)
        nf = np.empty((cap, 2), np.float32)
        k = lib().ref_oct_intersect(ctypes.c_int(n), ctypes.c_int(max_hits), _p(_u8(search_order)), _p(rays_o),
                                    _p(rays_d), _p(bounds), _p(_u8(tree_nodes)), _p(se), ctypes.c_int(cap), _p(idx),
                                    _p(nf))
        if k >= 0:
            return se, idx[:k].copy(), nf[:k].copy()
        cap = -k


def ray_march(rays_o, rays_d, noise, sample_l, scale_by_dis, oct_se, oct_idx, near_far, tree_nodes, transes):
    rays_o, rays_d = _f32(rays_o), _f32(rays_d)
    n = rays_o.shape[0]
    se = np.zeros((n, 2), np.int32)
    first = np.zeros(n, np.float32)
    cap = max(1, n * 128)
    oct_idx = _i32(oct_idx) if len(oct_idx) else np.zeros(1, np.int32)
    near_far = _f32(near_far) if len(near_far) else np.zeros((1, 2), np.float32)
    while True:
        world = np.empty((cap, 3), np.float32)
        pts = np.empty((cap, 3), np.float32)
        dirs = np.empty((cap, 3), np.float32)
        dt = np.empty(cap, np.float32)
        t = np.empty(cap, np.float32)
        anchors = np.zeros((cap, 3), np.int32)
        soct = np.empty(cap, np.int32)
        k = lib().ref_ray_march(ctypes.c_int(n), ctypes.c_float(sample_l), ctypes.c_int(int(scale_by_dis)), _p(rays_o),
                                _p(rays_d), _p(_f32(noise)), _p(_i32(oct_se)), _p(oct_idx), _p(near_far),
                                _p(_u8(tree_nodes)), _p(_u8(transes)), _p(se), ctypes.c_int(cap), _p(world), _p(pts),
                                _p(dirs), _p(anchors), _p(dt), _p(t), _p(soct), _p(first))
        if k >= 0:
            a = anchors[:k].copy()
            a[:, 2] = 0  # uninitialised in the reference
            return dict(pts=pts[:k].copy(), dirs=dirs[:k].copy(), dt=dt[:k].copy(), t=t[:k].copy(), anchors=a,
                        pts_idx_bounds=se, first_oct_dis=first.reshape(n, 1), world_pts=world[:k].copy())
        cap = -k


def edge_samples(edge_pool, transes, edge_idx, edge_coords):
    n = len(edge_idx)
    out_pts = np.empty((n, 2, 3), np.float32)
    out_idx = np.empty((n, 2), np.int32)
    lib().ref_edge_samples(ctypes.c_int(n), _p(_u8(edge_pool)), _p(_u8(transes)), _p(_i32(edge_idx)),
                           _p(_f32(edge_coords)), _p(out_pts), _p(out_idx))
    return out_pts, out_idx


def mark_visit(n_nodes, pts_se, oct_indices, weights, alphas, visit_cnt):
    w_add = np.full(n_nodes, -1, np.int32)
    a_add = np.full(n_nodes, -1, np.int32)
    mark = np.zeros(n_nodes, np.int32)
    cnt = _i32(visit_cnt).copy()
    pts_se = _i32(pts_se)
    lib().ref_mark_visit(ctypes.c_int(pts_se.shape[0]), _p(pts_se), _p(_i32(oct_indices)), _p(_f32(weights)),
                         _p(_f32(alphas)), _p(w_add), _p(a_add), _p(mark), _p(cnt))
    return w_add, a_add, mark, cnt


def mark_invalid(w_stats, a_stats, tree_nodes):
    nodes = _u8(tree_nodes).copy()
    lib().ref_mark_invalid(ctypes.c_int(len(w_stats)), _p(_i32(w_stats)), _p(_i32(a_stats)), _p(nodes))
    return nodes


def mark_invisible(tree_nodes, intris, w2cs, bounds):
    nodes = _u8(tree_nodes).copy()
    intris, w2cs, bounds = _f32(intris), _f32(w2cs), _f32(bounds)
    lib().ref_mark_invisible(ctypes.c_int(nodes.size // 64), ctypes.c_int(intris.shape[0]), _p(nodes), _p(intris),
                             _p(w2cs), _p(bounds))
    return nodes


def hash_fwd(feat_pool_h, prim_pool, local_idx, local_size, bias_pool, points01, volume_idx, n_volumes):
    points01 = _f32(points01)
    n = points01.shape[0]
    out = np.zeros((n, 32), np.uint16)
    lib().ref_hash_fwd(ctypes.c_int(n), ctypes.c_int(n_volumes), _p(_u16(feat_pool_h)), _p(_i32(prim_pool)),
                       _p(_i32(local_idx)), _p(_i32(local_size)), _p(_f32(bias_pool)), _p(points01),
                       _p(_i32(volume_idx)), _p(out))
    return out


def hash_bwd(pool_halves, prim_pool, local_idx, local_size, bias_pool, points01, volume_idx, n_volumes, grad_in_h):
    points01 = _f32(points01)
    n = points01.shape[0]
    out = np.zeros(pool_halves, np.uint16)
    lib().ref_hash_bwd(ctypes.c_int(n), ctypes.c_int(n_volumes), _p(_i32(prim_pool)), _p(_i32(local_idx)),
                       _p(_i32(local_size)), _p(_f32(bias_pool)), _p(points01), _p(_i32(volume_idx)),
                       _p(_u16(grad_in_h)), _p(out))
    return out


def sh_encode(dirs, degree=4):
    dirs = _f32(dirs)
    out = np.empty((dirs.shape[0], degree * degree), np.float32)
    lib().ref_sh(ctypes.c_int(dirs.shape[0]), ctypes.c_int(degree), _p(dirs), _p(out))
    return out


def count_valid(se, mask):
    se = _i32(se)
    out = np.empty(se.shape[0], np.int32)
    lib().ref_count_valid(ctypes.c_int(se.shape[0]), _p(se), _p(_i32(mask)), _p(out))
    return out


def flex_sum(val, se):
    val, se = _f32(val), _i32(se)
    if val.ndim == 1:
        out = np.empty(se.shape[0], np.float32)
        lib().ref_flex_sum_fwd(ctypes.c_int(se.shape[0]), _p(val), _p(se), _p(out))
    else:
        out = np.empty((se.shape[0], val.shape[1]), np.float32)
        lib().ref_flex_sum_vec_fwd(ctypes.c_int(se.shape[0]), ctypes.c_int(val.shape[1]), _p(val), _p(se), _p(out))
    return out


def flex_sum_bwd(dsum, se, n_all):
    dsum, se = _f32(dsum), _i32(se)
    if dsum.ndim == 1:
        out = np.zeros(n_all, np.float32)
        lib().ref_flex_sum_bwd(ctypes.c_int(se.shape[0]), _p(dsum), _p(se), _p(out))
    else:
        out = np.zeros((n_all, dsum.shape[1]), np.float32)
        lib().ref_flex_sum_vec_bwd(ctypes.c_int(se.shape[0]), ctypes.c_int(dsum.shape[1]), _p(dsum), _p(se), _p(out))
    return out


def flex_acc(val, se, include_this):
    val, se = _f32(val), _i32(se)
    out = np.zeros_like(val)
    lib().ref_flex_acc_fwd(ctypes.c_int(se.shape[0]), ctypes.c_int(int(include_this)), _p(val), _p(se), _p(out))
    return out


def flex_acc_bwd(dsum, se, include_this):
    dsum, se = _f32(dsum), _i32(se)
    out = np.zeros_like(dsum)
    lib().ref_flex_acc_bwd(ctypes.c_int(se.shape[0]), ctypes.c_int(int(include_this)), _p(dsum), _p(se), _p(out))
    return out


def weight_var(weights, se):
    weights, se = _f32(weights), _i32(se)
    out = np.empty(se.shape[0], np.float32)
    lib().ref_weight_var_fwd(ctypes.c_int(se.shape[0]), _p(weights), _p(se), _p(out))
    return out


def weight_var_bwd(weights, se, dvar):
    weights, se = _f32(weights), _i32(se)
    out = np.zeros_like(weights)
    lib().ref_weight_var_bwd(ctypes.c_int(se.shape[0]), _p(weights), _p(se), _p(_f32(dvar)), _p(out))
    return out


def grad_scaling_bwd(vals, se, progress):
    vals, se = _f32(vals).copy(), _i32(se)
    c = 1 if vals.ndim == 1 else vals.shape[1]
    lib().ref_grad_scaling_bwd(ctypes.c_int(se.shape[0]), ctypes.c_int(c), ctypes.c_float(progress), _p(se), _p(vals))
    return vals


def scatter_idx(n_all, se, emb_idx):
    se = _i32(se)
    out = np.zeros(n_all, np.int32)
    lib().ref_scatter_idx(ctypes.c_int(se.shape[0]), _p(se), _p(_i32(emb_idx)), _p(out))
    return out


def scatter_add(emb, idx, to_add):
    emb, out = _f32(emb), _f32(to_add).copy()
    lib().ref_scatter_add_fwd(ctypes.c_int(out.shape[0]), ctypes.c_int(emb.shape[1]), _p(emb), _p(_i32(idx)), _p(out))
    return out


def scatter_add_bwd(n_emb, idx, dsum):
    dsum = _f32(dsum)
    out = np.zeros((n_emb, dsum.shape[1]), np.float32)
    lib().ref_scatter_add_bwd(ctypes.c_int(n_emb), ctypes.c_int(dsum.shape[0]), ctypes.c_int(dsum.shape[1]),
                              _p(_i32(idx)), _p(dsum), _p(out))
    return out


def img2world(poses, intri, dist, cam_idx, ij_shifted):
    n = len(cam_idx)
    o = np.zeros((n, 3), np.float32)
    d = np.zeros((n, 3), np.float32)
    lib().ref_img2world(ctypes.c_int(n), _p(_f32(poses)), _p(_f32(intri)), _p(_f32(dist)), _p(_i32(cam_idx)),
                        _p(_f32(ij_shifted)), _p(o), _p(d))
    return o, d


def proc_octree(tree_nodes, w_stats, a_stats, visit_cnt, compact, subdivide, brute_force):
    """PersOctree::ProcOctree (PersSampler.cpp:120-330), the reference's own vector algorithm.  tree_nodes: uint8 blob."""
    nodes = _u8(tree_nodes)
    n = nodes.size // 64
    cap = max(16, 9 * n + 16)
    out = np.zeros(cap * 64, np.uint8)
    ow, oa = np.zeros(cap, np.int32), np.zeros(cap, np.int32)
    m = lib().ref_proc_octree(ctypes.c_int(n), _p(nodes), _p(_i32(w_stats)), _p(_i32(a_stats)), _p(_i32(visit_cnt)),
                              ctypes.c_int(int(compact)), ctypes.c_int(int(subdivide)), ctypes.c_int(int(brute_force)),
                              ctypes.c_int(cap), _p(out), _p(ow), _p(oa))
    assert m >= 0, m
    return out[:m * 64].copy(), ow[:m].copy(), oa[:m].copy()


def construct_edge_pool(tree_nodes):
    """PersOctree::ConstructEdgePool (PersSampler.cpp:614-659)."""
    nodes = _u8(tree_nodes)
    n = nodes.size // 64
    cap = 1024
    while True:
        out = np.zeros(cap * 64, np.uint8)
        m = lib().ref_construct_edge_pool(ctypes.c_int(n), _p(nodes), ctypes.c_int(cap), _p(out))
        if m >= 0:
            return out[:m * 64].copy()
        cap = -m


def schedules(iter_step, end_iter, init_fineness, fineness_decay_end, var_start, var_end, gs_start, gs_end, lr, lr_alpha,
              lr_warm_up_end, var_w):
    """ExpRunner::UpdateAdaParams (ExpRunner.cpp:221-254) + the variance-loss ramp (:108-114): (fineness, lr,
    gradient_scaling_progress, var_loss_weight) at iter_step, by the reference's own statements."""
    out = np.zeros(4, np.float32)
    lib().ref_schedules(ctypes.c_uint(iter_step), ctypes.c_uint(end_iter), ctypes.c_float(init_fineness),
                        ctypes.c_int(fineness_decay_end), ctypes.c_int(var_start), ctypes.c_int(var_end), ctypes.c_int(gs_start),
                        ctypes.c_int(gs_end), ctypes.c_float(lr), ctypes.c_float(lr_alpha), ctypes.c_float(lr_warm_up_end),
                        ctypes.c_float(var_w), _p(out))
    return out
