"""ctypes binding of libf2n_hip.so (include/f2n_abi.h) for torch tensors on a HIP device.

Thin by design: every function checks dtype/contiguity/device, passes raw device pointers and the current
torch HIP stream, and raises on a non-zero status.  No computation happens here and nothing falls back to
torch ops or to the CPU oracle."""
import ctypes
import os

import torch

from . import build

_lib = None
ABI_VERSION = 13  # include/f2n_abi.h


class F2nError(RuntimeError):
    pass


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(build.LIB):
            raise F2nError("native library %s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(there is no CPU fallback)" % build.LIB)
        _lib = ctypes.CDLL(build.LIB)
        _lib.f2n_build_info.restype = ctypes.c_char_p
        if _lib.f2n_abi_version() != ABI_VERSION:  # a stale library next to a newer binding: argument lists would not match
            raise F2nError("%s reports ABI version %d, this binding is written for %d: rebuild (__graft_entry__.build())"
                           % (build.LIB, _lib.f2n_abi_version(), ABI_VERSION))
    return _lib


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


_DT = {"f32": torch.float32, "i32": torch.int32, "u8": torch.uint8, "h16": torch.float16}


def _p(t, kind=None, allow_none=False):
    if t is None:
        if allow_none:
            return ctypes.c_void_p(0)
        raise F2nError("required tensor is None")
    if not t.is_cuda:
        raise F2nError("tensor must live on the HIP device (no CPU path)")
    if not t.is_contiguous():
        raise F2nError("tensor must be contiguous")
    if kind is not None and t.dtype != _DT[kind]:
        raise F2nError("expected dtype %s, got %s" % (kind, t.dtype))
    return ctypes.c_void_p(t.data_ptr())


def _ck(rc, name):
    if rc != 0:
        raise F2nError("%s failed with status %d" % (name, rc))


_i = ctypes.c_int
_f = ctypes.c_float
_d = ctypes.c_double  # (the Adam betas: doubles as in torch::optim::AdamOptions, include/f2n_abi.h)


def build_info():
    return lib().f2n_build_info().decode()


def debug_counters(reset=False):
    """Diagnostic event counters of the library (f2n_debug_counters): [0] = scatter records applied by the atomic fallback,
    [1] = table slices whose owner summed in fp64 instead of its packed fixed-point image (same bits, slower), [2] = (debug variant) the
    largest per-slice sum of |addend| as float bits, [3] = records that travelled through an overflow list (exact, order-free)."""
    out = (ctypes.c_int32 * 8)()
    torch.cuda.synchronize()
    _ck(lib().f2n_debug_counters(out, _i(1 if reset else 0)), "f2n_debug_counters")
    return [int(v) for v in out]


def _debug_only(name):
    if not hasattr(lib(), name):
        raise RuntimeError("%s exists in the debug variant of the library only (include/f2n_debug.h): set F2N_DEBUG_BUILD=1 before "
                           "importing the package" % name)


def debug_pollute(value):
    """Garbage derived from `value` in 64 KB of LDS and ~100 vector registers of every CU, on the current stream (f2n_debug_pollute)."""
    _debug_only("f2n_debug_pollute")
    _ck(lib().f2n_debug_pollute(_stream(), ctypes.c_uint(int(value) & 0xFFFFFFFF)), "f2n_debug_pollute")


def debug_spin(microseconds):
    """One wave spinning for that long on the current stream (f2n_debug_spin)."""
    _debug_only("f2n_debug_spin")
    _ck(lib().f2n_debug_spin(_stream(), _i(microseconds)), "f2n_debug_spin")


# ---------------------------------------------------------------- sampler
def normalize_dirs(n, dirs, out):
    _ck(lib().f2n_normalize_dirs(_stream(), _i(n), _p(dirs, "f32"), _p(out, "f32")), "f2n_normalize_dirs")


def sampler_prologue(n, dirs, out, zero=None, u=None, fineness=0.0, noise_out=None):
    _ck(lib().f2n_sampler_prologue(_stream(), _i(n), _p(dirs, "f32"), _p(out, "f32"), _p(zero, "i32", True),
                                   _i(0 if zero is None else zero.numel()), _i(0 if u is None else u.numel()), _p(u, "f32", True),
                                   _f(fineness), _p(noise_out, "f32", True)), "f2n_sampler_prologue")


def sampler_prologue_keyed(n, dirs, out, zero, n_noise, key, seq, fineness, noise_out):
    """f2n_sampler_prologue with the march noise drawn by the launch (Philox4x32-10 keyed by (key, seq): include/f2n_abi.h)."""
    _ck(lib().f2n_sampler_prologue_keyed(_stream(), _i(n), _p(dirs, "f32"), _p(out, "f32"), _p(zero, "i32", True),
                                         _i(0 if zero is None else zero.numel()), _i(n_noise), ctypes.c_uint64(int(key) & (2 ** 64 - 1)),
                                         ctypes.c_uint64(int(seq) & (2 ** 64 - 1)), _f(fineness), _p(noise_out, "f32")), "f2n_sampler_prologue_keyed")


def edge_samples_ex(n, edge_pool, n_edges, transes, edge_idx, edge_coords, u01, out_pts, out_idx, idx_stride, out_pts2=None,
                    out_idx2=None, idx_stride2=1):
    _ck(lib().f2n_edge_samples_ex(_stream(), _i(n), _p(edge_pool, "u8"), _i(n_edges), _p(transes, "u8"), _p(edge_idx, "i32", True),
                                  _p(edge_coords, "f32", True), _p(u01, "f32", True), _p(out_pts, "f32"), _p(out_idx, "i32"),
                                  _i(idx_stride), _p(out_pts2, "f32", True), _p(out_idx2, "i32", True), _i(idx_stride2)),
        "f2n_edge_samples_ex")


def oct_intersect_count(n_rays, max_hits, search_order, rays_o, rays_d, near, far, tree_nodes, hit_counts, child_blocks=None):
    _ck(lib().f2n_oct_intersect_count(_stream(), _i(n_rays), _i(max_hits), _p(search_order, "u8"), _p(rays_o, "f32"),
                                      _p(rays_d, "f32"), _f(near), _f(far), _p(tree_nodes, "u8"), _p(hit_counts, "i32"),
                                      _p(child_blocks, "u8", True)), "f2n_oct_intersect_count")


def oct_intersect_strided(n_rays, max_hits, search_order, rays_o, rays_d, near, far, tree_nodes, oct_se, oct_idx, oct_nf, total,
                          oct_trans=None, child_blocks=None):
    _ck(lib().f2n_oct_intersect_strided(_stream(), _i(n_rays), _i(max_hits), _p(search_order, "u8"), _p(rays_o, "f32"),
                                        _p(rays_d, "f32"), _f(near), _f(far), _p(tree_nodes, "u8"), _p(oct_se, "i32"),
                                        _p(oct_idx, "i32"), _p(oct_nf, "f32"), _p(total, "i32"), _p(oct_trans, "i32", True),
                                        _p(child_blocks, "u8", True)), "f2n_oct_intersect_strided")


def oct_lds_max_interior():
    return int(lib().f2n_oct_lds_max_interior())


def oct_intersect_strided_lds(n_rays, max_hits, search_order, rays_o, rays_d, near, far, tree_nodes, oct_se, oct_idx, oct_nf, total,
                              oct_trans, child_blocks, interior_nodes, rank_of):
    """The walk of oct_intersect_strided out of LDS-resident child records (trees with few interior nodes)."""
    _ck(lib().f2n_oct_intersect_strided_lds(_stream(), _i(n_rays), _i(max_hits), _p(search_order, "u8"), _p(rays_o, "f32"),
                                            _p(rays_d, "f32"), _f(near), _f(far), _p(tree_nodes, "u8"), _p(oct_se, "i32"),
                                            _p(oct_idx, "i32"), _p(oct_nf, "f32"), _p(total, "i32"), _p(oct_trans, "i32", True),
                                            _p(child_blocks, "u8"), _p(interior_nodes, "i32"), _p(rank_of, "i32"),
                                            _i(int(interior_nodes.numel()))), "f2n_oct_intersect_strided_lds")


def segment_scan(n, counts, start_end, total):
    _ck(lib().f2n_segment_scan(_stream(), _i(n), _p(counts, "i32"), _p(start_end, "i32"), _p(total, "i32")),
        "f2n_segment_scan")


def _mapped(host_tensor):
    """Device address of a PINNED host int32 tensor (hipHostMalloc memory is mapped into the device's address space at the
    same address), or NULL."""
    if host_tensor is None:
        return ctypes.c_void_p(0)
    if host_tensor.is_cuda or not host_tensor.is_pinned() or host_tensor.dtype != _DT["i32"] or not host_tensor.is_contiguous():
        raise F2nError("mirror must be a contiguous pinned host int32 tensor")
    return ctypes.c_void_p(host_tensor.data_ptr())


def segment_scan_ex(n, counts, start_end, total, mirror=None, also=None):
    """mirror: pinned host int32 tensor of len(also) + 1 words that the scan kernel itself fills (read it after a sync)."""
    n_also = 0 if also is None else int(also.numel())
    _ck(lib().f2n_segment_scan_ex(_stream(), _i(n), _p(counts, "i32"), _p(start_end, "i32"), _p(total, "i32"), _mapped(mirror),
                                  _p(also, "i32", True), _i(n_also)), "f2n_segment_scan_ex")


def oct_intersect_fill(n_rays, search_order, rays_o, rays_d, near, far, tree_nodes, oct_se, oct_idx, oct_nf, child_blocks=None):
    _ck(lib().f2n_oct_intersect_fill(_stream(), _i(n_rays), _p(search_order, "u8"), _p(rays_o, "f32"), _p(rays_d, "f32"),
                                     _f(near), _f(far), _p(tree_nodes, "u8"), _p(oct_se, "i32"), _p(oct_idx, "i32"),
                                     _p(oct_nf, "f32"), _p(child_blocks, "u8", True)), "f2n_oct_intersect_fill")


def ray_march_count(n_rays, sample_l, scale_by_dis, rays_o, rays_d, noise, oct_se, oct_idx, oct_nf, tree_nodes, transes,
                    counts):
    _ck(lib().f2n_ray_march_count(_stream(), _i(n_rays), _f(sample_l), _i(int(scale_by_dis)), _p(rays_o, "f32"),
                                  _p(rays_d, "f32"), _p(noise, "f32"), _p(oct_se, "i32"), _p(oct_idx, "i32"),
                                  _p(oct_nf, "f32"), _p(tree_nodes, "u8"), _p(transes, "u8"), _p(counts, "i32")),
        "f2n_ray_march_count")


def ray_march_fill(n_rays, sample_l, scale_by_dis, rays_o, rays_d, noise, oct_se, oct_idx, oct_nf, tree_nodes, transes,
                   pts_se, pts, dirs, dt, t, anchors, first_oct_dis):
    _ck(lib().f2n_ray_march_fill(_stream(), _i(n_rays), _f(sample_l), _i(int(scale_by_dis)), _p(rays_o, "f32"),
                                 _p(rays_d, "f32"), _p(noise, "f32"), _p(oct_se, "i32"), _p(oct_idx, "i32"),
                                 _p(oct_nf, "f32"), _p(tree_nodes, "u8"), _p(transes, "u8"), _p(pts_se, "i32"),
                                 _p(pts, "f32"), _p(dirs, "f32"), _p(dt, "f32"), _p(t, "f32"), _p(anchors, "i32"),
                                 _p(first_oct_dis, "f32")), "f2n_ray_march_fill")


def ray_march_strided(n_rays, sample_l, scale_by_dis, rays_o, rays_d, noise, oct_se, oct_idx, oct_nf, tree_nodes, transes, counts,
                      s_pts, s_dt, s_t, s_anchors, first_oct_dis, oct_trans=None):
    _ck(lib().f2n_ray_march_strided(_stream(), _i(n_rays), _f(sample_l), _i(int(scale_by_dis)), _p(rays_o, "f32"), _p(rays_d, "f32"),
                                    _p(noise, "f32"), _p(oct_se, "i32"), _p(oct_idx, "i32"), _p(oct_nf, "f32"), _p(tree_nodes, "u8"),
                                    _p(transes, "u8"), _p(counts, "i32"), _p(s_pts, "f32", True), _p(s_dt, "f32"), _p(s_t, "f32"),
                                    _p(s_anchors, "i32"), _p(first_oct_dis, "f32"), _p(oct_trans, "i32", True)), "f2n_ray_march_strided")


def pack_samples(n_rays, pts_se, rays_o, rays_d, transes, s_pts, s_dt, s_t, s_anchors, pts, dirs, dt, t, anchors):
    _ck(lib().f2n_pack_samples(_stream(), _i(n_rays), _p(pts_se, "i32"), _p(rays_o, "f32", True), _p(rays_d, "f32"),
                               _p(transes, "u8", True), _p(s_pts, "f32", True), _p(s_dt, "f32"),
                               _p(s_t, "f32"), _p(s_anchors, "i32"), _p(pts, "f32"), _p(dirs, "f32"), _p(dt, "f32"), _p(t, "f32"),
                               _p(anchors, "i32")), "f2n_pack_samples")


def pack_samples_repair(n_rays, pts_se, rays_o, rays_d, transes, s_pts, s_dt, s_t, s_anchors, pts, dirs, dt, t, anchors, death_epoch,
                        spec_epoch):
    """pack_samples again, on the device only if a leaf died in an epoch >= spec_epoch (see f2n_abi.h, speculative sampling)."""
    _ck(lib().f2n_pack_samples_repair(_stream(), _i(n_rays), _p(pts_se, "i32"), _p(rays_o, "f32", True), _p(rays_d, "f32"),
                                      _p(transes, "u8", True), _p(s_pts, "f32", True), _p(s_dt, "f32"), _p(s_t, "f32"),
                                      _p(s_anchors, "i32"), _p(pts, "f32"), _p(dirs, "f32"), _p(dt, "f32"), _p(t, "f32"),
                                      _p(anchors, "i32"), _p(death_epoch, "i32"), _i(spec_epoch)), "f2n_pack_samples_repair")


def edge_samples(n, edge_pool, transes, edge_idx, edge_coords, out_pts, out_idx):
    _ck(lib().f2n_edge_samples(_stream(), _i(n), _p(edge_pool, "u8"), _p(transes, "u8"), _p(edge_idx, "i32"),
                               _p(edge_coords, "f32"), _p(out_pts, "f32"), _p(out_idx, "i32")), "f2n_edge_samples")


def early_stop_votes(n_rays, pts_se, f0, f0_stride, dt, weights, alphas, mask, kept, anchors, anchor_stride, w_adder, a_adder, mark,
                     visit_cnt):
    _ck(lib().f2n_early_stop_votes(_stream(), _i(n_rays), _p(pts_se, "i32"), _p(f0, "f32"), _i(f0_stride), _p(dt, "f32"),
                                   _p(weights, "f32"), _p(alphas, "f32"), _p(mask, "i32"), _p(kept, "i32"), _i(w_adder.numel()),
                                   _p(anchors, "i32"), _i(anchor_stride), _p(w_adder, "i32"), _p(a_adder, "i32"), _p(mark, "i32"),
                                   _p(visit_cnt, "i32")), "f2n_early_stop_votes")


def oct_mark_visit(n_rays, pts_se, anchors, anchor_stride, weights, alphas, w_adder, a_adder, mark, visit_cnt):
    _ck(lib().f2n_oct_mark_visit(_stream(), _i(n_rays), _i(w_adder.numel()), _p(pts_se, "i32"), _p(anchors, "i32"), _i(anchor_stride),
                                 _p(weights, "f32"), _p(alphas, "f32"), _p(w_adder, "i32"), _p(a_adder, "i32"),
                                 _p(mark, "i32"), _p(visit_cnt, "i32")), "f2n_oct_mark_visit")


def oct_update_stats(n_nodes, w_adder, a_adder, mark, w_stats, a_stats, tree_nodes, child_blocks=None, reset_votes=False):
    _ck(lib().f2n_oct_update_stats(_stream(), _i(n_nodes), _p(w_adder, "i32"), _p(a_adder, "i32"), _p(mark, "i32"),
                                   _p(w_stats, "i32"), _p(a_stats, "i32"), _p(tree_nodes, "u8"),
                                   _p(child_blocks, "u8", True), _i(1 if reset_votes else 0)), "f2n_oct_update_stats")


def oct_update_stats_ex(n_nodes, w_adder, a_adder, mark, w_stats, a_stats, tree_nodes, child_blocks, reset_votes, died_at, epoch,
                        death_epoch):
    _ck(lib().f2n_oct_update_stats_ex(_stream(), _i(n_nodes), _p(w_adder, "i32"), _p(a_adder, "i32"), _p(mark, "i32"),
                                      _p(w_stats, "i32"), _p(a_stats, "i32"), _p(tree_nodes, "u8"), _p(child_blocks, "u8", True),
                                      _i(int(reset_votes)), _p(died_at, "i32", True), _i(epoch), _p(death_epoch, "i32", True),
                                      ctypes.c_void_p(0)), "f2n_oct_update_stats_ex")


def oct_intersect_repair(n_rays, max_hits, search_order, rays_o, rays_d, near, far, tree_nodes, oct_se, oct_idx, oct_nf, total,
                         oct_trans, child_blocks, died_at, spec_epoch, death_epoch, repair_flags, n_repaired=None):
    _ck(lib().f2n_oct_intersect_repair(_stream(), _i(n_rays), _i(max_hits), _p(search_order, "u8"), _p(rays_o, "f32"),
                                       _p(rays_d, "f32"), _f(near), _f(far), _p(tree_nodes, "u8"), _p(oct_se, "i32"),
                                       _p(oct_idx, "i32"), _p(oct_nf, "f32"), _p(total, "i32"), _p(oct_trans, "i32", True),
                                       _p(child_blocks, "u8", True), _p(died_at, "i32"), _i(spec_epoch), _p(death_epoch, "i32"),
                                       _p(repair_flags, "i32"), _p(n_repaired, "i32", True)), "f2n_oct_intersect_repair")


def ray_march_repair(n_rays, sample_l, scale_by_dis, rays_o, rays_d, noise, oct_se, oct_idx, oct_nf, tree_nodes, transes, counts,
                     s_pts, s_dt, s_t, s_anchors, first_oct_dis, oct_trans, repair_flags, death_epoch, spec_epoch):
    _ck(lib().f2n_ray_march_repair(_stream(), _i(n_rays), _f(sample_l), _i(int(scale_by_dis)), _p(rays_o, "f32"), _p(rays_d, "f32"),
                                   _p(noise, "f32"), _p(oct_se, "i32"), _p(oct_idx, "i32"), _p(oct_nf, "f32"), _p(tree_nodes, "u8"),
                                   _p(transes, "u8"), _p(counts, "i32"), _p(s_pts, "f32", True), _p(s_dt, "f32"), _p(s_t, "f32"),
                                   _p(s_anchors, "i32"), _p(first_oct_dis, "f32"), _p(oct_trans, "i32", True),
                                   _p(repair_flags, "i32"), _p(death_epoch, "i32"), _i(spec_epoch)), "f2n_ray_march_repair")


def ray_march_strided_rec(n_rays, max_hits, sample_l, scale_by_dis, rays_o, rays_d, noise, oct_se, oct_idx, oct_nf, tree_nodes, transes,
                          counts, s_pts, s_dt, s_t, s_anchors, first_oct_dis, oct_trans, leaf_state, reached):
    """ray_march_strided that records resumable states per leaf-list entry (tail repair of speculative batches)."""
    _ck(lib().f2n_ray_march_strided_rec(_stream(), _i(n_rays), _i(max_hits), _f(sample_l), _i(int(scale_by_dis)), _p(rays_o, "f32"),
                                        _p(rays_d, "f32"), _p(noise, "f32"), _p(oct_se, "i32"), _p(oct_idx, "i32"), _p(oct_nf, "f32"),
                                        _p(tree_nodes, "u8"), _p(transes, "u8"), _p(counts, "i32"), _p(s_pts, "f32", True),
                                        _p(s_dt, "f32"), _p(s_t, "f32"), _p(s_anchors, "i32"), _p(first_oct_dis, "f32"),
                                        _p(oct_trans, "i32", True), _p(leaf_state, "i32"), _p(reached, "i32")),
        "f2n_ray_march_strided_rec")


def ray_march_persistent(n_rays, max_hits, n_blocks, sample_l, scale_by_dis, rays_o, rays_d, noise, oct_se, oct_idx, oct_nf, tree_nodes,
                         transes, counts, s_pts, s_dt, s_t, s_anchors, first_oct_dis, oct_trans, leaf_state, reached, order, counter, block_waves=1):
    """The strided march on n_blocks persistent waves (workgroups of block_waves), rays sorted by leaf count (see f2n_abi.h)."""
    _ck(lib().f2n_ray_march_persistent(_stream(), _i(n_rays), _i(max_hits), _i(n_blocks), _i(block_waves), _f(sample_l), _i(int(scale_by_dis)),
                                       _p(rays_o, "f32"), _p(rays_d, "f32"), _p(noise, "f32"), _p(oct_se, "i32"), _p(oct_idx, "i32"),
                                       _p(oct_nf, "f32"), _p(tree_nodes, "u8"), _p(transes, "u8"), _p(counts, "i32"),
                                       _p(s_pts, "f32", True), _p(s_dt, "f32"), _p(s_t, "f32"), _p(s_anchors, "i32"),
                                       _p(first_oct_dis, "f32"), _p(oct_trans, "i32", True), _p(leaf_state, "i32", True),
                                       _p(reached, "i32", True), _p(order, "i32"), _p(counter, "i32")), "f2n_ray_march_persistent")


def oct_list_repair(n_rays, max_hits, oct_se, oct_idx, oct_nf, oct_trans, total, died_at, spec_epoch, death_epoch, reached, repair_from,
                    n_repaired, n_full):
    _ck(lib().f2n_oct_list_repair(_stream(), _i(n_rays), _i(max_hits), _p(oct_se, "i32"), _p(oct_idx, "i32"), _p(oct_nf, "f32"),
                                  _p(oct_trans, "i32", True), _p(total, "i32"), _p(died_at, "i32"), _i(spec_epoch),
                                  _p(death_epoch, "i32"), _p(reached, "i32"), _p(repair_from, "i32"), _p(n_repaired, "i32", True),
                                  _p(n_full, "i32")), "f2n_oct_list_repair")


def oct_intersect_repair_flagged(n_rays, max_hits, search_order, rays_o, rays_d, near, far, tree_nodes, oct_se, oct_idx, oct_nf, total,
                                 oct_trans, child_blocks, death_epoch, spec_epoch, repair_from, n_full):
    _ck(lib().f2n_oct_intersect_repair_flagged(_stream(), _i(n_rays), _i(max_hits), _p(search_order, "u8"), _p(rays_o, "f32"),
                                               _p(rays_d, "f32"), _f(near), _f(far), _p(tree_nodes, "u8"), _p(oct_se, "i32"),
                                               _p(oct_idx, "i32"), _p(oct_nf, "f32"), _p(total, "i32"), _p(oct_trans, "i32", True),
                                               _p(child_blocks, "u8", True), _p(death_epoch, "i32"), _i(spec_epoch),
                                               _p(repair_from, "i32"), _p(n_full, "i32")), "f2n_oct_intersect_repair_flagged")


def ray_march_repair_tail(n_rays, max_hits, sample_l, scale_by_dis, rays_o, rays_d, noise, oct_se, oct_idx, oct_nf, tree_nodes, transes,
                          counts, s_pts, s_dt, s_t, s_anchors, first_oct_dis, oct_trans, leaf_state, reached, repair_from, death_epoch,
                          spec_epoch):
    _ck(lib().f2n_ray_march_repair_tail(_stream(), _i(n_rays), _i(max_hits), _f(sample_l), _i(int(scale_by_dis)), _p(rays_o, "f32"),
                                        _p(rays_d, "f32"), _p(noise, "f32"), _p(oct_se, "i32"), _p(oct_idx, "i32"), _p(oct_nf, "f32"),
                                        _p(tree_nodes, "u8"), _p(transes, "u8"), _p(counts, "i32"), _p(s_pts, "f32", True),
                                        _p(s_dt, "f32"), _p(s_t, "f32"), _p(s_anchors, "i32"), _p(first_oct_dis, "f32"),
                                        _p(oct_trans, "i32", True), _p(leaf_state, "i32"), _p(reached, "i32"), _p(repair_from, "i32"),
                                        _p(death_epoch, "i32"), _i(spec_epoch)), "f2n_ray_march_repair_tail")


def oct_build_child_blocks(n_nodes, tree_nodes, child_blocks):
    _ck(lib().f2n_oct_build_child_blocks(_stream(), _i(n_nodes), _p(tree_nodes, "u8"), _p(child_blocks, "u8")),
        "f2n_oct_build_child_blocks")


def oct_mark_invisible(n_nodes, n_cams, tree_nodes, intris, w2cs, bounds):
    _ck(lib().f2n_oct_mark_invisible(_stream(), _i(n_nodes), _i(n_cams), _p(tree_nodes, "u8"), _p(intris, "f32"),
                                     _p(w2cs, "f32"), _p(bounds, "f32")), "f2n_oct_mark_invisible")


# ---------------------------------------------------------------- hash grid / MLP / field
def hash_fwd(n, n_volumes, table_h, prim_pool, local_idx, local_size, bias_pool, level_scale, pts, pts_are_warped,
             volume_idx, vol_stride, out_h):
    _ck(lib().f2n_hash_fwd(_stream(), _i(n), _i(n_volumes), _p(table_h, "h16"), _p(prim_pool, "i32"), _p(local_idx, "i32"),
                           _p(local_size, "i32"), _p(bias_pool, "f32"), _p(level_scale, "f32"), _p(pts, "f32"),
                           _i(int(pts_are_warped)), _p(volume_idx, "i32"), _i(vol_stride), _p(out_h, "h16")), "f2n_hash_fwd")


def hash_bwd(n, n_volumes, prim_pool, local_idx, local_size, bias_pool, level_scale, pts, pts_are_warped, volume_idx,
             vol_stride, grad_in_h, grad_table_h, level_entries=0):
    _ck(lib().f2n_hash_bwd(_stream(), _i(n), _i(n_volumes), _p(prim_pool, "i32"), _p(local_idx, "i32"),
                           _p(local_size, "i32"), _p(bias_pool, "f32"), _p(level_scale, "f32"), _p(pts, "f32"),
                           _i(int(pts_are_warped)), _p(volume_idx, "i32"), _i(vol_stride), _p(grad_in_h, "h16"),
                           _p(grad_table_h, "h16"), _i(level_entries)), "f2n_hash_bwd")


def mlp_n_params(d_in, d_hidden, n_hidden):
    return lib().f2n_mlp_n_params(_i(d_in), _i(d_hidden), _i(n_hidden))


def mlp_init_params(seed, d_in, d_hidden, n_hidden, params_f32):
    _ck(lib().f2n_mlp_init_params(_stream(), ctypes.c_uint64(seed), _i(d_in), _i(d_hidden), _i(n_hidden),
                                  _p(params_f32, "f32")), "f2n_mlp_init_params")


def params_to_h16(n, params_f32, params_h):
    _ck(lib().f2n_params_to_h16(_stream(), _i(n), _p(params_f32, "f32"), _p(params_h, "h16")), "f2n_params_to_h16")


def mlp_fwd(n, d_in, d_hidden, n_hidden, params_h, x, out_h):
    _ck(lib().f2n_mlp_fwd(_stream(), _i(n), _i(d_in), _i(d_hidden), _i(n_hidden), _p(params_h, "h16"), _p(x, "f32"),
                          _p(out_h, "h16")), "f2n_mlp_fwd")


def mlp_bwd(n, d_in, d_hidden, n_hidden, loss_scale, params_h, x, dy, dparams_scaled, dx):
    _ck(lib().f2n_mlp_bwd(_stream(), _i(n), _i(d_in), _i(d_hidden), _i(n_hidden), _f(loss_scale), _p(params_h, "h16"),
                          _p(x, "f32"), _p(dy, "f32"), _p(dparams_scaled, "f32"), _p(dx, "f32", True)), "f2n_mlp_bwd")


def field_fwd(n, n_volumes, table_h, prim_pool, local_idx, local_size, bias_pool, level_scale, pts_warped, volume_idx,
              vol_stride, mlp_params_h, out_feat, out_f0, save_x_h):
    _ck(lib().f2n_field_fwd(_stream(), _i(n), _i(n_volumes), _p(table_h, "h16"), _p(prim_pool, "i32"), _p(local_idx, "i32"),
                            _p(local_size, "i32"), _p(bias_pool, "f32"), _p(level_scale, "f32"), _p(pts_warped, "f32"),
                            _p(volume_idx, "i32"), _i(vol_stride), _p(mlp_params_h, "h16"), _p(out_feat, "f32", True),
                            _p(out_f0, "f32", True), _p(save_x_h, "h16", True)), "f2n_field_fwd")


def hash_gather_planes(n, n_volumes, table_h, prim_pool, local_idx, local_size, bias_pool, level_scale, pts, pts_are_warped,
                       volume_idx, vol_stride, planes_h):
    _ck(lib().f2n_hash_gather_planes(_stream(), _i(n), _i(n_volumes), _p(table_h, "h16"), _p(prim_pool, "i32"),
                                     _p(local_idx, "i32"), _p(local_size, "i32"), _p(bias_pool, "f32"), _p(level_scale, "f32"),
                                     _p(pts, "f32"), _i(int(pts_are_warped)), _p(volume_idx, "i32"), _i(vol_stride),
                                     _p(planes_h, "h16")), "f2n_hash_gather_planes")


def hash_gather_planes_binned(n, n_volumes, table_h, prim_pool, local_idx, local_size, bias_pool, level_scale, pts, pts_are_warped,
                              volume_idx, vol_stride, planes_h, level_entries, first_binned_pair):
    """hash_gather_planes for tables beyond the L2s: level pairs >= first_binned_pair through the slice-binned pipeline."""
    _ck(lib().f2n_hash_gather_planes_binned(_stream(), _i(n), _i(n_volumes), _p(table_h, "h16"), _p(prim_pool, "i32"),
                                            _p(local_idx, "i32"), _p(local_size, "i32"), _p(bias_pool, "f32"), _p(level_scale, "f32"),
                                            _p(pts, "f32"), _i(int(pts_are_warped)), _p(volume_idx, "i32"), _i(vol_stride),
                                            _p(planes_h, "h16"), _i(level_entries), _i(first_binned_pair)),
        "f2n_hash_gather_planes_binned")


def hash_gather_planes_balanced(n, n_volumes, table_h, prim_pool, local_idx, local_size, bias_pool, level_scale, pts,
                                pts_are_warped, volume_idx, vol_stride, planes_h, step01, level_scale_host):
    """level_scale_host: 16 floats on the HOST (numpy array / list); step01 <= 0 -> the plain one-pair-per-XCD gather."""
    import ctypes
    arr = (ctypes.c_float * 16)(*[float(v) for v in level_scale_host])
    _ck(lib().f2n_hash_gather_planes_balanced(_stream(), _i(n), _i(n_volumes), _p(table_h, "h16"), _p(prim_pool, "i32"),
                                              _p(local_idx, "i32"), _p(local_size, "i32"), _p(bias_pool, "f32"),
                                              _p(level_scale, "f32"), _p(pts, "f32"), _i(int(pts_are_warped)),
                                              _p(volume_idx, "i32"), _i(vol_stride), _p(planes_h, "h16"),
                                              ctypes.c_float(step01), arr), "f2n_hash_gather_planes_balanced")


def hash_gather_variant(n, n_volumes, step01, level_scale_host):
    """bit 0: staged hash constants, bit 1: run combining + balanced split (what f2n_hash_gather_planes_balanced would launch)."""
    import ctypes
    arr = (ctypes.c_float * 16)(*[float(v) for v in level_scale_host])
    rc = lib().f2n_hash_gather_variant(_i(n), _i(n_volumes), ctypes.c_float(step01), arr)
    if rc < 0:
        _ck(rc, "f2n_hash_gather_variant")
    return rc


def field_mlp_planes(n, planes_h, mlp_params_h, out_feat, out_f0, save_x_h):
    _ck(lib().f2n_field_mlp_planes(_stream(), _i(n), _p(planes_h, "h16"), _p(mlp_params_h, "h16"), _p(out_feat, "f32", True),
                                   _p(out_f0, "f32", True), _p(save_x_h, "h16", True)), "f2n_field_mlp_planes")


def field_fwd_cached(n, n_cache, src_rows, x_cache_h, mlp_params_h, out_feat, out_f0, save_x_h):
    _ck(lib().f2n_field_fwd_cached(_stream(), _i(n), _i(n_cache), _p(src_rows, "i32", True), _p(x_cache_h, "h16"),
                                   _p(mlp_params_h, "h16"), _p(out_feat, "f32", True), _p(out_f0, "f32", True),
                                   _p(save_x_h, "h16", True)), "f2n_field_fwd_cached")


def field_bwd(n, n_volumes, prim_pool, local_idx, local_size, bias_pool, level_scale, pts_warped, volume_idx, vol_stride,
              mlp_params_h, saved_x_h, dfeat, loss_scale, dparams_scaled, grad_table_h, level_entries=0):
    _ck(lib().f2n_field_bwd(_stream(), _i(n), _i(n_volumes), _p(prim_pool, "i32"), _p(local_idx, "i32"),
                            _p(local_size, "i32"), _p(bias_pool, "f32"), _p(level_scale, "f32"), _p(pts_warped, "f32"),
                            _p(volume_idx, "i32"), _i(vol_stride), _p(mlp_params_h, "h16"), _p(saved_x_h, "h16"),
                            _p(dfeat, "f32"), _f(loss_scale), _p(dparams_scaled, "f32"), _p(grad_table_h, "h16"),
                            _i(level_entries)), "f2n_field_bwd")


# ---------------------------------------------------------------- shader
def sh_encode(n, degree, dirs, out):
    _ck(lib().f2n_sh_encode(_stream(), _i(n), _i(degree), _p(dirs, "f32"), _p(out, "f32")), "f2n_sh_encode")


def scatter_idx(n_rays, start_end, ray_val, out):
    _ck(lib().f2n_scatter_idx(_stream(), _i(n_rays), _p(start_end, "i32"), _p(ray_val, "i32"), _p(out, "i32")),
        "f2n_scatter_idx")


def shade_fwd(n, feat, dirs, app_emb, sample_emb_idx, mlp_params_h, rgb, save_x_h):
    _ck(lib().f2n_shade_fwd(_stream(), _i(n), _p(feat, "f32"), _p(dirs, "f32"), _p(app_emb, "f32", True),
                            _p(sample_emb_idx, "i32", True), _p(mlp_params_h, "h16"), _p(rgb, "f32"),
                            _p(save_x_h, "h16", True)), "f2n_shade_fwd")


def field_shade_fwd(n, src_rows, x_cache_h, field_params_h, dirs, app_emb, sample_emb_idx, color_params_h, out_f0, save_field_x_h,
                    save_shade_x_h, rgb, n_dev=None, x_extra_h=None, feat_extra=None, save_x_extra_h=None):
    n_extra = 0 if x_extra_h is None else x_extra_h.shape[0]
    _ck(lib().f2n_field_shade_fwd_extra(_stream(), _i(n), _p(n_dev, "i32", True), _p(src_rows, "i32", True), _p(x_cache_h, "h16"),
                                        _p(field_params_h, "h16"), _p(dirs, "f32"), _p(app_emb, "f32", True),
                                        _p(sample_emb_idx, "i32", True), _p(color_params_h, "h16"), _p(out_f0, "f32", True),
                                        _p(save_field_x_h, "h16", True), _p(save_shade_x_h, "h16", True), _p(rgb, "f32"), _i(n_extra),
                                        _p(x_extra_h, "h16", True), _p(feat_extra, "f32", True), _p(save_x_extra_h, "h16", True)),
        "f2n_field_shade_fwd_extra")


def shade_bwd(n, drgb, sample_emb_idx, mlp_params_h, saved_x_h, loss_scale, dfeat, dparams_scaled, dapp_emb, df0=None):
    _ck(lib().f2n_shade_bwd(_stream(), _i(n), _p(drgb, "f32"), _p(sample_emb_idx, "i32", True), _p(mlp_params_h, "h16"),
                            _p(saved_x_h, "h16"), _f(loss_scale), _p(dfeat, "f32"), _p(dparams_scaled, "f32"),
                            _p(dapp_emb, "f32", True), _i(0 if dapp_emb is None else dapp_emb.shape[0]), _p(df0, "f32", True)),
        "f2n_shade_bwd")


# ---------------------------------------------------------------- ray generation
def img2world_rays(n, poses, intri, dist_params, cam_idx, ij, rays_o, rays_d):
    _ck(lib().f2n_img2world_rays(_stream(), _i(n), _p(poses, "f32"), _p(intri, "f32"), _p(dist_params, "f32"), _p(cam_idx, "i32"),
                                 _p(ij, "i32"), _p(rays_o, "f32"), _p(rays_d, "f32")), "f2n_img2world_rays")


def draw_ray_batch(n_rays, u01, image_set, height, width, poses, intri, dist_params, images, cam_bounds, cam_indices, ij, rays_o,
                   rays_d, gt_colors, bounds):
    _ck(lib().f2n_draw_ray_batch(_stream(), _i(n_rays), _p(u01, "f32"), _p(image_set, "i32"), _i(int(image_set.numel())), _i(height),
                                 _i(width), _p(poses, "f32"), _p(intri, "f32"), _p(dist_params, "f32"), _p(images, "f32", True),
                                 _p(cam_bounds, "f32"), _p(cam_indices, "i32"), _p(ij, "i32"), _p(rays_o, "f32"), _p(rays_d, "f32"),
                                 _p(gt_colors, "f32", True), _p(bounds, "f32")), "f2n_draw_ray_batch")


def gather_pixels(n, height, width, images, cam_bounds, cam_idx, ij, gt_colors, bounds):
    _ck(lib().f2n_gather_pixels(_stream(), _i(n), _i(height), _i(width), _p(images, "f32", True), _p(cam_bounds, "f32", True),
                                _p(cam_idx, "i32"), _p(ij, "i32"), _p(gt_colors, "f32", True), _p(bounds, "f32", True)),
        "f2n_gather_pixels")


# ---------------------------------------------------------------- renderer
def early_stop(n_rays, pts_se, f0, f0_stride, dt, weights, alphas, mask, kept):
    _ck(lib().f2n_early_stop(_stream(), _i(n_rays), _p(pts_se, "i32"), _p(f0, "f32"), _i(f0_stride), _p(dt, "f32"),
                             _p(weights, "f32"), _p(alphas, "f32"), _p(mask, "i32"), _p(kept, "i32")), "f2n_early_stop")


def compact_samples(n_rays, old_se, new_se, mask, pts, dirs, dt, t, anchors, o_pts, o_dirs, o_dt, o_t, o_anchors):
    _ck(lib().f2n_compact_samples(_stream(), _i(n_rays), _p(old_se, "i32"), _p(new_se, "i32"), _p(mask, "i32"),
                                  _p(pts, "f32"), _p(dirs, "f32"), _p(dt, "f32"), _p(t, "f32"), _p(anchors, "i32"),
                                  _p(o_pts, "f32"), _p(o_dirs, "f32"), _p(o_dt, "f32"), _p(o_t, "f32"),
                                  _p(o_anchors, "i32")), "f2n_compact_samples")


def compact_samples_src(n_rays, old_se, new_se, mask, pts, dirs, dt, t, anchors, o_pts, o_dirs, o_dt, o_t, o_anchors, o_src,
                        o_vol=None, ray_val=None, o_ray_val=None):
    _ck(lib().f2n_compact_samples_src(_stream(), _i(n_rays), _p(old_se, "i32"), _p(new_se, "i32"), _p(mask, "i32"),
                                      _p(pts, "f32"), _p(dirs, "f32"), _p(dt, "f32"), _p(t, "f32"), _p(anchors, "i32"),
                                      _p(o_pts, "f32"), _p(o_dirs, "f32"), _p(o_dt, "f32"), _p(o_t, "f32"),
                                      _p(o_anchors, "i32"), _p(o_src, "i32"), _p(o_vol, "i32", True), _p(ray_val, "i32", True),
                                      _p(o_ray_val, "i32", True)), "f2n_compact_samples_src")


def oct_visible_cams(n_boxes, n_cams, boxes, c2w, bounds, fx, fy, cx, cy, res_h, res_w, pix_i, pix_j, visible):
    _ck(lib().f2n_oct_visible_cams(_stream(), _i(n_boxes), _i(n_cams), _p(boxes, "f32"), _p(c2w, "f32"), _p(bounds, "f32"), _f(fx),
                                   _f(fy), _f(cx), _f(cy), _i(res_h), _i(res_w), _p(pix_i, "f32"), _p(pix_j, "f32"),
                                   _p(visible, "u8")), "f2n_oct_visible_cams")


def march_noise(n, u, fineness, out):
    _ck(lib().f2n_march_noise(_stream(), _i(n), _p(u, "f32"), _f(fineness), _p(out, "f32")), "f2n_march_noise")


def composite_fwd(n_rays, pts_se, feat, dt, t, rgb, bg, colors, disparity, depth, weights, f0_stride=16, out_vars=None):
    """feat: the field output [M,16] (f0_stride 16, column 0 is read) or a compact density array [M] (f0_stride 1)."""
    _ck(lib().f2n_composite_fwd(_stream(), _i(n_rays), _p(pts_se, "i32"), _p(feat, "f32"), _i(f0_stride), _p(dt, "f32"),
                                _p(t, "f32"), _p(rgb, "f32"), _p(bg, "f32"), _p(colors, "f32"), _p(disparity, "f32"),
                                _p(depth, "f32"), _p(weights, "f32"), _p(out_vars, "f32", True)), "f2n_composite_fwd")


def composite_bwd(n_rays, pts_se, feat, dt, t, rgb, bg, dcolors, ddisp, ddepth, dweights, gs_progress, drgb, dfeat, f0_stride=16,
                  df0_stride=16, var_weights=None, dvars=None):
    _ck(lib().f2n_composite_bwd(_stream(), _i(n_rays), _p(pts_se, "i32"), _p(feat, "f32"), _i(f0_stride), _p(dt, "f32"),
                                _p(t, "f32"), _p(rgb, "f32"), _p(bg, "f32"), _p(dcolors, "f32", True), _p(ddisp, "f32", True),
                                _p(ddepth, "f32", True), _p(dweights, "f32", True), _f(gs_progress), _p(drgb, "f32"),
                                _p(dfeat, "f32"), _i(df0_stride), _p(var_weights, "f32", True), _p(dvars, "f32", True)),
        "f2n_composite_bwd")


def weight_var_fwd(n_rays, weights, pts_se, out):
    _ck(lib().f2n_weight_var_fwd(_stream(), _i(n_rays), _p(weights, "f32"), _p(pts_se, "i32"), _p(out, "f32")),
        "f2n_weight_var_fwd")


def weight_var_bwd(n_rays, weights, pts_se, dvars, dweights):
    _ck(lib().f2n_weight_var_bwd(_stream(), _i(n_rays), _p(weights, "f32"), _p(pts_se, "i32"), _p(dvars, "f32"),
                                 _p(dweights, "f32")), "f2n_weight_var_bwd")


def flex_sum_fwd(n_rays, vec, val, se, out):
    _ck(lib().f2n_flex_sum_fwd(_stream(), _i(n_rays), _i(vec), _p(val, "f32"), _p(se, "i32"), _p(out, "f32")), "f2n_flex_sum_fwd")


def flex_sum_bwd(n_rays, vec, dsum, se, out):
    _ck(lib().f2n_flex_sum_bwd(_stream(), _i(n_rays), _i(vec), _p(dsum, "f32"), _p(se, "i32"), _p(out, "f32")), "f2n_flex_sum_bwd")


def flex_acc_fwd(n_rays, include_this, val, se, out):
    _ck(lib().f2n_flex_acc_fwd(_stream(), _i(n_rays), _i(int(include_this)), _p(val, "f32"), _p(se, "i32"), _p(out, "f32")),
        "f2n_flex_acc_fwd")


def flex_acc_bwd(n_rays, include_this, dsum, se, out):
    _ck(lib().f2n_flex_acc_bwd(_stream(), _i(n_rays), _i(int(include_this)), _p(dsum, "f32"), _p(se, "i32"), _p(out, "f32")),
        "f2n_flex_acc_bwd")


# ---------------------------------------------------------------- optimiser
def adam_coefficients(step, lr, beta1=0.9, beta2=0.99, eps=1e-15, weight_decay=0.0, grad_scale=1.0):
    """The nine float scalars the Adam kernels are launched with (host function, no device): [lr / (1 - beta1^step),
    sqrt(1 - beta2^step), beta1, beta2, 1 - beta1, 1 - beta2, eps, weight_decay, grad_scale]."""
    out = (ctypes.c_float * 9)()
    _ck(lib().f2n_adam_coefficients(_i(step), _f(lr), _d(beta1), _d(beta2), _f(eps), _f(weight_decay), _f(grad_scale), out),
        "f2n_adam_coefficients")
    return [float(v) for v in out]


def adam_step(n, param, grad, grad_scale, grad_round_h16, exp_avg, exp_avg_sq, step, lr, beta1, beta2, eps, wd, param_h,
              skip_flag=None, zero_grad=False):
    _ck(lib().f2n_adam_step(_stream(), _i(n), _p(param, "f32"), _p(grad, "f32"), _f(grad_scale), _i(int(grad_round_h16)),
                            _p(exp_avg, "f32"), _p(exp_avg_sq, "f32"), _i(step), _f(lr), _d(beta1), _d(beta2), _f(eps),
                            _f(wd), _p(param_h, "h16", True), _i(int(zero_grad)), _p(skip_flag, "i32", True)), "f2n_adam_step")


class _AdamGroup(ctypes.Structure):
    _fields_ = [("param", ctypes.c_void_p), ("grad", ctypes.c_void_p), ("exp_avg", ctypes.c_void_p), ("exp_avg_sq", ctypes.c_void_p),
                ("param_h", ctypes.c_void_p), ("n", ctypes.c_int), ("grad_scale", ctypes.c_float), ("weight_decay", ctypes.c_float),
                ("grad_round_h16", ctypes.c_int), ("check_finite", ctypes.c_int)]


def adam_small_groups(groups, step, lr, beta1, beta2, eps, zero_grad, flags=None, skip_flag=None):
    """groups: list of dicts {param, grad, exp_avg, exp_avg_sq, param_h (or None), grad_scale, weight_decay, grad_round_h16,
    check_finite}; one launch (f2n_adam_small_groups)."""
    arr = (_AdamGroup * len(groups))()
    for a, g in zip(arr, groups):
        a.param = _p(g["param"], "f32").value
        a.grad = _p(g["grad"], "f32").value
        a.exp_avg = _p(g["exp_avg"], "f32").value
        a.exp_avg_sq = _p(g["exp_avg_sq"], "f32").value
        a.param_h = _p(g.get("param_h"), "h16", True).value
        a.n = int(g["param"].numel())
        a.grad_scale = float(g["grad_scale"])
        a.weight_decay = float(g["weight_decay"])
        a.grad_round_h16 = int(bool(g.get("grad_round_h16", False)))
        a.check_finite = int(bool(g.get("check_finite", False)))
    _ck(lib().f2n_adam_small_groups(_stream(), _i(len(groups)), arr, _i(step), _f(lr), _d(beta1), _d(beta2), _f(eps),
                                    _i(int(zero_grad)), _p(flags, "i32", True), _p(skip_flag, "i32", True)), "f2n_adam_small_groups")


def adam_fused(groups, table, step, lr, beta1, beta2, eps, zero_grad, skip_flag=None):
    """groups as in adam_small_groups (check_finite ignored); table: dict {param, grad_h, exp_avg, exp_avg_sq, param_h, grad_scale, n}
    or None; ONE launch (f2n_adam_fused)."""
    arr = (_AdamGroup * max(len(groups), 1))()
    for a, g in zip(arr, groups):
        a.param = _p(g["param"], "f32").value
        a.grad = _p(g["grad"], "f32").value
        a.exp_avg = _p(g["exp_avg"], "f32").value
        a.exp_avg_sq = _p(g["exp_avg_sq"], "f32").value
        a.param_h = _p(g.get("param_h"), "h16", True).value
        a.n = int(g["param"].numel())
        a.grad_scale = float(g["grad_scale"])
        a.weight_decay = float(g["weight_decay"])
        a.grad_round_h16 = int(bool(g.get("grad_round_h16", False)))
        a.check_finite = 0
    t = table or {}
    _ck(lib().f2n_adam_fused(_stream(), _i(len(groups)), arr, _i(int(t.get("n", 0))), _p(t.get("param"), "f32", True),
                             _p(t.get("grad_h"), "h16", True), _f(float(t.get("grad_scale", 1.0))), _p(t.get("exp_avg"), "f32", True),
                             _p(t.get("exp_avg_sq"), "f32", True), _p(t.get("param_h"), "h16", True), _i(step), _f(lr), _d(beta1),
                             _d(beta2), _f(eps), _i(int(zero_grad)), _p(skip_flag, "i32", True)), "f2n_adam_fused")


def field_bwd_dyn(n_max, n_dev, n_off, n_volumes, prim_pool, local_idx, local_size, bias_pool, level_scale, pts_warped, volume_idx,
                  vol_stride, mlp_params_h, saved_x_h, dfeat, loss_scale, dparams_scaled, grad_table_h, level_entries, defer_reduce):
    _ck(lib().f2n_field_bwd_dyn(_stream(), _i(n_max), _p(n_dev, "i32", True), _i(n_off), _i(n_volumes), _p(prim_pool, "i32"),
                                _p(local_idx, "i32"), _p(local_size, "i32"), _p(bias_pool, "f32"), _p(level_scale, "f32"),
                                _p(pts_warped, "f32"), _p(volume_idx, "i32"), _i(vol_stride), _p(mlp_params_h, "h16"),
                                _p(saved_x_h, "h16"), _p(dfeat, "f32"), _f(loss_scale), _p(dparams_scaled, "f32"),
                                _p(grad_table_h, "h16"), _i(level_entries), _i(int(defer_reduce))), "f2n_field_bwd_dyn")


class _StepTail(ctypes.Structure):
    _fields_ = [("n_flags_a", ctypes.c_int), ("flags_grad_a", ctypes.c_void_p), ("n_flags_b", ctypes.c_int), ("flags_grad_b", ctypes.c_void_p),
                ("flags", ctypes.c_void_p), ("flags_mirror", ctypes.c_void_p), ("n_groups", ctypes.c_int), ("groups", ctypes.c_void_p),
                ("n_table", ctypes.c_int), ("table_param", ctypes.c_void_p), ("table_exp_avg", ctypes.c_void_p),
                ("table_exp_avg_sq", ctypes.c_void_p), ("table_param_h", ctypes.c_void_p), ("table_grad_scale", ctypes.c_float),
                ("step", ctypes.c_int), ("lr", ctypes.c_float), ("beta1", ctypes.c_double), ("beta2", ctypes.c_double), ("eps", ctypes.c_float),
                ("after_reduce", ctypes.c_void_p), ("after_reduce_user", ctypes.c_void_p), ("leave_table_to_caller", ctypes.c_int)]


def field_bwd_step_tail(n_max, n_dev, n_off, n_volumes, prim_pool, local_idx, local_size, bias_pool, level_scale, pts_warped, volume_idx,
                        vol_stride, mlp_params_h, saved_x_h, dfeat, loss_scale, dparams_scaled, grad_table_h, level_entries, flags_a,
                        flags_b, flags, groups, table, step, lr, beta1, beta2, eps, tail_stream=None, flags_mirror=None, leave_table_to_caller=False,
                        after_reduce=None):
    """f2n_field_bwd_step_tail: the field backward + the rest of the training step (deferred reductions, finiteness flags, Adam of
    `groups` (as adam_fused) and of `table` = {param, exp_avg, exp_avg_sq, param_h, grad_scale, n}) re-ordered around the scatter.
    Returns 1 when the scatter's owners stepped the table."""
    arr = (_AdamGroup * max(len(groups), 1))()
    for a, g in zip(arr, groups):
        a.param = _p(g["param"], "f32").value
        a.grad = _p(g["grad"], "f32").value
        a.exp_avg = _p(g["exp_avg"], "f32").value
        a.exp_avg_sq = _p(g["exp_avg_sq"], "f32").value
        a.param_h = _p(g.get("param_h"), "h16", True).value
        a.n = int(g["param"].numel())
        a.grad_scale = float(g["grad_scale"])
        a.weight_decay = float(g["weight_decay"])
        a.grad_round_h16 = int(bool(g.get("grad_round_h16", False)))
        a.check_finite = 0
    t = _StepTail()
    t.n_flags_a, t.flags_grad_a = int(flags_a.numel()), _p(flags_a, "f32").value
    t.n_flags_b, t.flags_grad_b = int(flags_b.numel()), _p(flags_b, "f32").value
    t.flags, t.flags_mirror = _p(flags, "i32").value, _mapped(flags_mirror).value
    t.n_groups, t.groups = len(groups), ctypes.cast(arr, ctypes.c_void_p).value
    t.n_table = int(table["n"])
    t.table_param, t.table_exp_avg, t.table_exp_avg_sq = _p(table["param"], "f32").value, _p(table["exp_avg"], "f32").value, _p(table["exp_avg_sq"], "f32").value
    t.table_param_h, t.table_grad_scale = _p(table["param_h"], "h16").value, float(table["grad_scale"])
    t.step, t.lr, t.beta1, t.beta2, t.eps = int(step), float(lr), float(beta1), float(beta2), float(eps)
    t.leave_table_to_caller = int(bool(leave_table_to_caller))
    cb = None
    if after_reduce is not None:  # (python callable(chain_stream_handle): what a data-parallel host does between the reductions and the flags)
        cb = ctypes.CFUNCTYPE(None, ctypes.c_void_p, ctypes.c_void_p)(lambda user, chain: after_reduce(chain))
        t.after_reduce = ctypes.cast(cb, ctypes.c_void_p).value
    by_owners = ctypes.c_int(0)
    ts = ctypes.c_void_p(tail_stream.cuda_stream) if tail_stream is not None else ctypes.c_void_p(0)
    _ck(lib().f2n_field_bwd_step_tail(_stream(), ts, _i(n_max), _p(n_dev, "i32", True), _i(n_off), _i(n_volumes), _p(prim_pool, "i32"),
                                      _p(local_idx, "i32"), _p(local_size, "i32"), _p(bias_pool, "f32"), _p(level_scale, "f32"),
                                      _p(pts_warped, "f32"), _p(volume_idx, "i32"), _i(vol_stride), _p(mlp_params_h, "h16"),
                                      _p(saved_x_h, "h16"), _p(dfeat, "f32"), _f(loss_scale), _p(dparams_scaled, "f32"),
                                      _p(grad_table_h, "h16"), _i(level_entries), ctypes.byref(t), ctypes.byref(by_owners)),
        "f2n_field_bwd_step_tail")
    return by_owners.value


def reduce_deferred():
    _ck(lib().f2n_reduce_deferred(_stream()), "f2n_reduce_deferred")


def deferred_reset():
    _ck(lib().f2n_deferred_reset(), "f2n_deferred_reset")


def adam_step_h16grad(n, param, grad_h, grad_scale, exp_avg, exp_avg_sq, step, lr, beta1, beta2, eps, wd, param_h,
                      zero_grad, skip_flag=None):
    _ck(lib().f2n_adam_step_h16grad(_stream(), _i(n), _p(param, "f32"), _p(grad_h, "h16"), _f(grad_scale),
                                    _p(exp_avg, "f32"), _p(exp_avg_sq, "f32"), _i(step), _f(lr), _d(beta1), _d(beta2),
                                    _f(eps), _f(wd), _p(param_h, "h16"), _i(int(zero_grad)),
                                    _p(skip_flag, "i32", True)), "f2n_adam_step_h16grad")


def train_loss(n_rays, pred, gt, disparity, sampled_var, n_edge, feat_dim, edge_feats, var_w, disp_w, tv_w, out_losses,
               dcolors, ddisparity, dvar, dedge_feats):
    _ck(lib().f2n_train_loss(_stream(), _i(n_rays), _p(pred, "f32"), _p(gt, "f32"), _p(disparity, "f32", True),
                             _p(sampled_var, "f32", True), _i(n_edge), _i(feat_dim), _p(edge_feats, "f32", True), _f(var_w),
                             _f(disp_w), _f(tv_w), _p(out_losses, "f32"), _p(dcolors, "f32", True), _p(ddisparity, "f32", True),
                             _p(dvar, "f32", True), _p(dedge_feats, "f32", True)), "f2n_train_loss")


def composite_train(n_rays, se, f0, f0_stride, dt, t, rgb, bg, gt, var_w, disp_w, tv_w, gs_progress, n_edge, feat_dim, edge_feats,
                    dedge_feats, colors, weights, drgb, df0, df0_stride, out_losses, defer_reduce=False):
    """composite_fwd + train_loss + composite_bwd in one launch (include/f2n_abi.h)."""
    _ck(lib().f2n_composite_train(_stream(), _i(n_rays), _p(se, "i32"), _p(f0, "f32"), _i(f0_stride), _p(dt, "f32"), _p(t, "f32"),
                                  _p(rgb, "f32"), _p(bg, "f32"), _p(gt, "f32"), _f(var_w), _f(disp_w), _f(tv_w), _f(gs_progress),
                                  _i(n_edge), _i(feat_dim), _p(edge_feats, "f32", True), _p(dedge_feats, "f32", True),
                                  _p(colors, "f32"), _p(weights, "f32"), _p(drgb, "f32"), _p(df0, "f32"), _i(df0_stride),
                                  _p(out_losses, "f32"), _i(1 if defer_reduce else 0)), "f2n_composite_train")


def nonfinite_flags(n_a, a, n_b, b, flags, mirror=None):
    _ck(lib().f2n_nonfinite_flags_ex(_stream(), _i(n_a), _p(a, "f32", True), _i(n_b), _p(b, "f32", True), _p(flags, "i32"),
                                     _mapped(mirror)), "f2n_nonfinite_flags_ex")
