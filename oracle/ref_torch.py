"""TEST INFRASTRUCTURE ONLY -- ctypes binding of oracle/_ref/libf2n_ref_torch.so: the reference's own host code for
perspective-warp construction (DistanceSummary, GetVisiCams, PCA, PersOctree::ConstructTrans) compiled against this image's
libtorch on the CPU by oracle/build_ref_torch.py.  Import torch before the first call (the library resolves its libtorch
symbols against the copy the Python process has loaded, and draws from the same global generator as torch.randint)."""
import ctypes
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "_ref", "libf2n_ref_torch.so")
_lib = None


def available():
    return os.path.exists(SO)


def lib():
    global _lib
    if _lib is None:
        import torch  # noqa: F401  (libtorch first)
        _lib = ctypes.CDLL(SO)
        _lib.ref_distance_summary.restype = ctypes.c_float
    return _lib


def _p(a):
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(ctypes.c_void_p)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def check_failures():
    return int(lib().ref_torch_check_failures())


def distance_summary(dis):
    dis = _f32(dis).reshape(-1)
    return float(lib().ref_distance_summary(ctypes.c_int(len(dis)), _p(dis)))


def get_visi_cams(side_len, center, c2w, intri, bound):
    c2w, intri, bound, center = _f32(c2w), _f32(intri), _f32(bound), _f32(center)
    n = c2w.shape[0]
    out = np.zeros(n, np.int32)
    k = lib().ref_get_visi_cams(ctypes.c_float(side_len), _p(center), ctypes.c_int(n), _p(c2w), _p(intri), _p(bound), _p(out))
    return out[:k].tolist()


def construct_trans(rand_pts, c2w, intri33, center):
    """One TransInfo (544 raw bytes).  Draws its first camera with torch::randint on the process-wide CPU generator."""
    rand_pts, c2w, intri33, center = _f32(rand_pts), _f32(c2w), _f32(intri33), _f32(center)
    out = np.zeros(544, np.uint8)
    lib().ref_construct_trans(ctypes.c_int(rand_pts.shape[0]), _p(rand_pts), ctypes.c_int(c2w.shape[0]), _p(c2w), _p(intri33),
                              _p(center), _p(out))
    return out


def build_octree(max_depth, bbox_side_len, split_dist_thres, c2w, intri, bound, cap_nodes=1 << 16, cap_trans=1 << 14):
    """PersOctree::PersOctree's node / warp construction (PersSampler.cpp:70-82, 359-612) -> (TreeNode bytes, TransInfo bytes).
    Draws (rand points of every visited node, first camera of every warp) come from the process-wide CPU generator."""
    c2w, intri, bound = _f32(c2w), _f32(intri), _f32(bound)
    nodes = np.zeros(cap_nodes * 64, np.uint8)
    trans = np.zeros(cap_trans * 544, np.uint8)
    counts = np.zeros(2, np.int32)
    rc = lib().ref_build_octree(ctypes.c_int(max_depth), ctypes.c_float(bbox_side_len), ctypes.c_float(split_dist_thres),
                                ctypes.c_int(c2w.shape[0]), _p(c2w), _p(intri), _p(bound), _p(nodes), ctypes.c_int(cap_nodes),
                                _p(trans), ctypes.c_int(cap_trans), _p(counts))
    assert rc >= 0, "capacity"
    return nodes[:int(counts[0]) * 64].copy(), trans[:int(counts[1]) * 544].copy()


def normalize_scene(poses, bounds, bounds_factor):
    """Dataset::NormalizeScene (Dataset.cpp:127-146) + bounds relaxation / clamp (:73-76) -> dict like rigs.prepare_scene's."""
    poses, bounds = _f32(poses).copy(), _f32(bounds).copy()
    n = poses.shape[0]
    w2c = np.zeros((n, 3, 4), np.float32)
    center = np.zeros(3, np.float32)
    radius = np.zeros(1, np.float32)
    lib().ref_normalize_scene(ctypes.c_int(n), _p(poses), _p(bounds), ctypes.c_float(bounds_factor[0]), ctypes.c_float(bounds_factor[1]),
                              _p(w2c), _p(center), _p(radius))
    return dict(poses=poses, bounds=bounds, w2c=w2c, center=center, radius=np.float32(radius[0]))


def pose_interpolate(a, b, alpha):
    """PoseInterpolate (Utils/CameraUtils.cpp:11-44) on two [3,4] poses."""
    a, b = _f32(a), _f32(b)
    out = np.zeros((3, 4), np.float32)
    lib().ref_pose_interpolate(_p(a), _p(b), ctypes.c_float(alpha), _p(out))
    return out
