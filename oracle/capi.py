"""TEST INFRASTRUCTURE ONLY -- ctypes binding of the C oracle (oracle/f2n_oracle.c).

Every wrapper takes/returns numpy arrays in exactly the layouts of SURVEY.md section 8(a).
"""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
# Two builds of the same source: the serial one is what the tests load (safe next to torch's bundled OpenMP
# runtime); the OpenMP one is only loaded when F2N_ORACLE_OMP=1, by bench.py's cpu_baseline leg in a
# subprocess that does not import torch (two OpenMP runtimes in one process crash).
USE_OMP = os.environ.get("F2N_ORACLE_OMP", "0") == "1"
SO = os.path.join(HERE, "libf2n_oracle_omp.so" if USE_OMP else "libf2n_oracle.so")
SRC = os.path.join(HERE, "f2n_oracle.c")

TREE_NODE_BYTES = 64
TRANS_INFO_BYTES = 544
EDGE_POOL_BYTES = 64
MAX_SAMPLE_PER_RAY = 1024
N_LEVELS = 16

_lib = None


def build(force=False):
    """gcc-compile the oracle (seconds).  -ffp-contract=off is part of the contract."""
    if force or not os.path.exists(SO) or os.path.getmtime(SO) < os.path.getmtime(SRC):
        cmd = ["gcc", "-O2", "-std=c11", "-ffp-contract=off", "-fno-fast-math", "-shared", "-fPIC", SRC, "-o", SO, "-lm"]
        if USE_OMP:
            cmd.insert(1, "-fopenmp")
        else:
            cmd.insert(1, "-Wno-unknown-pragmas")
        subprocess.check_call(cmd)
    return SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(SO)
    return _lib


def _p(a):
    if a is None:
        return ctypes.c_void_p(0)
    assert a.flags["C_CONTIGUOUS"], "array must be contiguous"
    return a.ctypes.data_as(ctypes.c_void_p)  # keeps a reference to `a` alive for the duration of the call


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _u8(a):
    return np.ascontiguousarray(a, dtype=np.uint8)


def _u16(a):
    return np.ascontiguousarray(a, dtype=np.uint16)


def f2h(x):
    x = _f32(x)
    out = np.empty(x.shape, np.uint16)
    lib().oracle_f2h(ctypes.c_int(x.size), _p(x), _p(out))
    return out


def h2f(h):
    h = _u16(h)
    out = np.empty(h.shape, np.float32)
    lib().oracle_h2f(ctypes.c_int(h.size), _p(h), _p(out))
    return out


def level_scales():
    out = np.empty(16, np.float32)
    lib().oracle_level_scales(_p(out))
    return out


def search_order_table():
    """PersSampler.cpp:106-117 restated: child visiting order per ray octant, uint8[64]."""
    import functools
    out = []
    for st in range(8):
        def less(a, b, st=st):
            bt = (a ^ b) & -(a ^ b)
            return ((a & bt) ^ (st & bt)) != 0

        def cmp(a, b):
            if less(a, b):
                return -1
            if less(b, a):
                return 1
            return 0
        out.extend(sorted(range(8), key=functools.cmp_to_key(cmp)))
    return np.array(out, np.uint8)


def normalize_dirs(dirs):
    dirs = _f32(dirs)
    out = np.empty_like(dirs)
    lib().oracle_normalize_dirs(ctypes.c_int(dirs.shape[0]), _p(dirs), _p(out))
    return out


def img2world(poses, intri, dist, cam_idx, ij):
    """ij: integer (row, column) pixel indices; the half-pixel shift is applied inside (Dataset.cu:126)."""
    poses, intri, dist = _f32(poses), _f32(intri), _f32(dist)
    cam_idx = np.ascontiguousarray(cam_idx, np.int32)
    ij = np.ascontiguousarray(ij, np.int32)
    n = cam_idx.shape[0]
    o = np.empty((n, 3), np.float32)
    d = np.empty((n, 3), np.float32)
    lib().oracle_img2world(ctypes.c_int(n), _p(poses), _p(intri), _p(dist), _p(cam_idx), _p(ij), _p(o), _p(d))
    return o, d


def oct_intersect(search_order, rays_o, rays_d, near, far, tree_nodes, max_hits=1024):
    rays_o, rays_d = _f32(rays_o), _f32(rays_d)
    n = rays_o.shape[0]
    se = np.zeros((n, 2), np.int32)
    cap = max(1, n * 64)
    while True:
        idx = np.empty(cap, np.int32)
        nf = np.empty((cap, 2), np.float32)
        k = lib().oracle_oct_intersect(ctypes.c_int(n), ctypes.c_int(max_hits), _p(_u8(search_order)), _p(rays_o),
                                       _p(rays_d), ctypes.c_float(near), ctypes.c_float(far), _p(_u8(tree_nodes)),
                                       _p(se), ctypes.c_int(cap), _p(idx), _p(nf))
        if k >= 0:
            return se, idx[:k].copy(), nf[:k].copy()
        cap = -k


def ray_march(rays_o, rays_d, noise, sample_l, scale_by_dis, oct_se, oct_idx, near_far, tree_nodes, transes):
    rays_o, rays_d = _f32(rays_o), _f32(rays_d)
    n = rays_o.shape[0]
    noise = _f32(noise)
    assert noise.size >= MAX_SAMPLE_PER_RAY + n + 10
    se = np.zeros((n, 2), np.int32)
    first = np.zeros(n, np.float32)
    cap = max(1, n * 128)
    oct_idx = _i32(oct_idx) if len(oct_idx) else np.zeros(1, np.int32)
    near_far = _f32(near_far) if len(near_far) else np.zeros((1, 2), np.float32)
    while True:
        pts = np.empty((cap, 3), np.float32)
        dirs = np.empty((cap, 3), np.float32)
        dt = np.empty(cap, np.float32)
        t = np.empty(cap, np.float32)
        anchors = np.zeros((cap, 3), np.int32)
        k = lib().oracle_ray_march(ctypes.c_int(n), ctypes.c_float(sample_l), ctypes.c_int(int(scale_by_dis)),
                                   _p(rays_o), _p(rays_d), _p(noise), _p(_i32(oct_se)), _p(oct_idx), _p(near_far),
                                   _p(_u8(tree_nodes)), _p(_u8(transes)), _p(se), ctypes.c_int(cap), _p(pts),
                                   _p(dirs), _p(dt), _p(t), _p(anchors), _p(first))
        if k >= 0:
            return dict(pts=pts[:k].copy(), dirs=dirs[:k].copy(), dt=dt[:k].copy(), t=t[:k].copy(),
                        anchors=anchors[:k].copy(), pts_idx_bounds=se, first_oct_dis=first.reshape(n, 1))
        cap = -k


def edge_samples(edge_pool, transes, edge_idx, edge_coords):
    n = len(edge_idx)
    out_pts = np.empty((n, 2, 3), np.float32)
    out_idx = np.empty((n, 2), np.int32)
    lib().oracle_edge_samples(ctypes.c_int(n), _p(_u8(edge_pool)), _p(_u8(transes)), _p(_i32(edge_idx)),
                              _p(_f32(edge_coords)), _p(out_pts), _p(out_idx))
    return out_pts, out_idx


def mark_visit(n_nodes, pts_se, oct_indices, weights, alphas, visit_cnt):
    w_add = np.full(n_nodes, -1, np.int32)
    a_add = np.full(n_nodes, -1, np.int32)
    mark = np.zeros(n_nodes, np.int32)
    cnt = _i32(visit_cnt).copy()
    pts_se = _i32(pts_se)
    lib().oracle_mark_visit(ctypes.c_int(pts_se.shape[0]), _p(pts_se), _p(_i32(oct_indices)), _p(_f32(weights)),
                            _p(_f32(alphas)), _p(w_add), _p(a_add), _p(mark), _p(cnt))
    return w_add, a_add, mark, cnt


def update_node_stats(w_add, a_add, mark, w_stats, a_stats, tree_nodes):
    w_stats, a_stats, nodes = _i32(w_stats).copy(), _i32(a_stats).copy(), _u8(tree_nodes).copy()
    lib().oracle_update_node_stats(ctypes.c_int(len(w_stats)), _p(_i32(w_add)), _p(_i32(a_add)), _p(_i32(mark)),
                                   _p(w_stats), _p(a_stats), _p(nodes))
    return w_stats, a_stats, nodes


def mark_invisible(tree_nodes, intris, w2cs, bounds):
    nodes = _u8(tree_nodes).copy()
    intris, w2cs, bounds = _f32(intris), _f32(w2cs), _f32(bounds)
    lib().oracle_mark_invisible(ctypes.c_int(nodes.size // TREE_NODE_BYTES), ctypes.c_int(intris.shape[0]),
                                _p(nodes), _p(intris), _p(w2cs), _p(bounds))
    return nodes


def hash_fwd(feat_pool_h, prim_pool, local_idx, local_size, bias_pool, points01, volume_idx, n_volumes,
             scales=None, vol_stride=1):
    points01 = _f32(points01)
    n = points01.shape[0]
    out = np.empty((n, 32), np.uint16)
    scales = level_scales() if scales is None else _f32(scales)
    lib().oracle_hash_fwd(ctypes.c_int(n), ctypes.c_int(n_volumes), _p(_u16(feat_pool_h)), _p(_i32(prim_pool)),
                          _p(_i32(local_idx)), _p(_i32(local_size)), _p(_f32(bias_pool)), _p(scales), _p(points01),
                          _p(_i32(volume_idx)), ctypes.c_int(vol_stride), _p(out))
    return out


def hash_bwd(pool_halves, prim_pool, local_idx, local_size, bias_pool, points01, volume_idx, n_volumes, grad_in_h,
             scales=None, vol_stride=1, fp32_accumulate=False):
    points01 = _f32(points01)
    n = points01.shape[0]
    scales = level_scales() if scales is None else _f32(scales)
    if fp32_accumulate:
        out = np.zeros(pool_halves, np.float32)
        fn = lib().oracle_hash_bwd_f32
    else:
        out = np.zeros(pool_halves, np.uint16)
        fn = lib().oracle_hash_bwd
    fn(ctypes.c_int(n), ctypes.c_int(n_volumes), _p(_i32(prim_pool)), _p(_i32(local_idx)), _p(_i32(local_size)),
       _p(_f32(bias_pool)), _p(scales), _p(points01), _p(_i32(volume_idx)), ctypes.c_int(vol_stride),
       _p(_u16(grad_in_h)), _p(out))
    return out


def sh_encode(dirs, degree=4):
    dirs = _f32(dirs)
    out = np.empty((dirs.shape[0], degree * degree), np.float32)
    rc = lib().oracle_sh_encode(ctypes.c_int(dirs.shape[0]), ctypes.c_int(degree), _p(dirs), _p(out))
    assert rc == 0
    return out


def count_valid(se, mask):
    se = _i32(se)
    out = np.empty(se.shape[0], np.int32)
    lib().oracle_count_valid(ctypes.c_int(se.shape[0]), _p(se), _p(_i32(mask)), _p(out))
    return out


def filter_idx_bounds(se, mask):
    se = _i32(se)
    out = np.empty_like(se)
    lib().oracle_filter_idx_bounds(ctypes.c_int(se.shape[0]), _p(se), _p(_i32(mask)), _p(out))
    return out


def flex_sum(val, se):
    val, se = _f32(val), _i32(se)
    c = 1 if val.ndim == 1 else val.shape[1]
    out = np.empty((se.shape[0],) if val.ndim == 1 else (se.shape[0], c), np.float32)
    lib().oracle_flex_sum_fwd(ctypes.c_int(se.shape[0]), ctypes.c_int(c), _p(val), _p(se), _p(out))
    return out


def flex_sum_bwd(dsum, se, n_all):
    dsum, se = _f32(dsum), _i32(se)
    c = 1 if dsum.ndim == 1 else dsum.shape[1]
    out = np.zeros((n_all,) if dsum.ndim == 1 else (n_all, c), np.float32)
    lib().oracle_flex_sum_bwd(ctypes.c_int(se.shape[0]), ctypes.c_int(c), _p(dsum), _p(se), _p(out))
    return out


def flex_acc(val, se, include_this):
    val, se = _f32(val), _i32(se)
    out = np.zeros_like(val)
    lib().oracle_flex_acc_fwd(ctypes.c_int(se.shape[0]), ctypes.c_int(int(include_this)), _p(val), _p(se), _p(out))
    return out


def flex_acc_bwd(dsum, se, include_this):
    dsum, se = _f32(dsum), _i32(se)
    out = np.zeros_like(dsum)
    lib().oracle_flex_acc_bwd(ctypes.c_int(se.shape[0]), ctypes.c_int(int(include_this)), _p(dsum), _p(se), _p(out))
    return out


def weight_var(weights, se):
    weights, se = _f32(weights), _i32(se)
    out = np.empty(se.shape[0], np.float32)
    lib().oracle_weight_var_fwd(ctypes.c_int(se.shape[0]), _p(weights), _p(se), _p(out))
    return out


def weight_var_bwd(weights, se, dvar):
    weights, se = _f32(weights), _i32(se)
    out = np.zeros_like(weights)
    lib().oracle_weight_var_bwd(ctypes.c_int(se.shape[0]), _p(weights), _p(se), _p(_f32(dvar)), _p(out))
    return out


def grad_scaling_bwd(vals, se, progress):
    vals, se = _f32(vals).copy(), _i32(se)
    c = 1 if vals.ndim == 1 else vals.shape[1]
    lib().oracle_grad_scaling_bwd(ctypes.c_int(se.shape[0]), ctypes.c_int(c), ctypes.c_float(progress), _p(se),
                                  _p(vals))
    return vals


def scatter_idx(n_all, se, emb_idx):
    se = _i32(se)
    out = np.zeros(n_all, np.int32)
    lib().oracle_scatter_idx(ctypes.c_int(se.shape[0]), _p(se), _p(_i32(emb_idx)), _p(out))
    return out


def scatter_add(emb, idx, to_add):
    emb, out = _f32(emb), _f32(to_add).copy()
    lib().oracle_scatter_add_fwd(ctypes.c_int(out.shape[0]), ctypes.c_int(emb.shape[1]), _p(emb), _p(_i32(idx)),
                                 _p(out))
    return out


def scatter_add_bwd(n_emb, idx, dsum):
    dsum = _f32(dsum)
    out = np.zeros((n_emb, dsum.shape[1]), np.float32)
    lib().oracle_scatter_add_bwd(ctypes.c_int(n_emb), ctypes.c_int(dsum.shape[0]), ctypes.c_int(dsum.shape[1]),
                                 _p(_i32(idx)), _p(dsum), _p(out))
    return out


def mlp_n_params(d_in, d_hidden, n_hidden):
    return lib().oracle_mlp_n_params(ctypes.c_int(d_in), ctypes.c_int(d_hidden), ctypes.c_int(n_hidden))


class mlp_accumulator:
    """Context manager: the accumulator model of the MLP forward (0 = fp32 in k order, the parity comparator; 1 = a binary16
    accumulator rounded after every 16-wide k-block, the other plausible reading of a WMMA fully-fused kernel)."""

    def __init__(self, mode):
        self.mode = int(mode)

    def __enter__(self):
        self.prev = lib().oracle_get_mlp_accumulator()
        lib().oracle_set_mlp_accumulator(ctypes.c_int(self.mode))
        return self

    def __exit__(self, *exc):
        lib().oracle_set_mlp_accumulator(ctypes.c_int(self.prev))
        return False


def mlp_fwd(params, x, d_hidden, n_hidden, want_acts=False):
    x = _f32(x)
    n, d_in = x.shape
    out = np.empty((n, 16), np.uint16)
    acts = np.empty((n, n_hidden, d_hidden), np.uint16) if want_acts else None
    rc = lib().oracle_mlp_fwd(ctypes.c_int(n), ctypes.c_int(d_in), ctypes.c_int(d_hidden), ctypes.c_int(n_hidden),
                              _p(_f32(params)), _p(x), _p(out), _p(acts))
    assert rc == 0
    return (out, acts) if want_acts else out


def mlp_bwd(params, x, acts, dy, d_hidden, n_hidden, loss_scale=128.0):
    x, dy = _f32(x), _f32(dy)
    n, d_in = x.shape
    dparams = np.zeros(mlp_n_params(d_in, d_hidden, n_hidden), np.float32)
    dx = np.empty((n, d_in), np.float32)
    dxh = np.empty((n, d_in), np.uint16)
    rc = lib().oracle_mlp_bwd(ctypes.c_int(n), ctypes.c_int(d_in), ctypes.c_int(d_hidden), ctypes.c_int(n_hidden),
                              ctypes.c_float(loss_scale), _p(_f32(params)), _p(x), _p(_u16(acts)), _p(dy), _p(dparams),
                              _p(dx), _p(dxh))
    assert rc == 0
    return dparams, dx, dxh


def num_threads():
    return lib().oracle_num_threads()


# ---------------------------------------------------------------- lane-local pieces on their own (tests/test_lane_code_cpu.py)
def slab(o, d, c, side, near, far):
    o, d, c, side = _f32(o), _f32(d), _f32(c), _f32(side)
    nf = np.ascontiguousarray(np.stack([np.full(len(o), near, np.float32), np.full(len(o), far, np.float32)], -1))
    lib().oracle_slab(ctypes.c_int(len(o)), _p(o), _p(d), _p(c), _p(side), _p(nf))
    return nf


def warp(transes, trans_idx, pts):
    transes, trans_idx, pts = _u8(transes), _i32(trans_idx), _f32(pts)
    out = np.empty((len(pts), 3), np.float32)
    jac = np.empty((len(pts), 3, 3), np.float32)
    lib().oracle_warp(ctypes.c_int(len(pts)), _p(transes), _p(trans_idx), _p(pts), _p(out), _p(jac))
    return out, jac


def hash_cell(pt01, mul, prim, bias, local_size):
    pt01, mul, prim, bias = _f32(pt01), _f32(mul), _i32(prim), _f32(bias)
    ls = np.ascontiguousarray(local_size, np.uint32)
    pos = np.empty((len(pt01), 8), np.uint32)
    w = np.empty((len(pt01), 8), np.float32)
    lib().oracle_hash_cell(ctypes.c_int(len(pt01)), _p(pt01), _p(mul), _p(prim), _p(bias), _p(ls), _p(pos), _p(w))
    return pos, w


def undistort(k4, uv):
    k4, uv = _f32(k4), _f32(uv).copy()
    lib().oracle_undistort(ctypes.c_int(len(uv)), _p(k4), _p(uv))
    return uv
