"""Python-side glue for the C++ host layer: loading the pybind module, building an ExpRunner from a serialised
scene state (the reference's checkpoint order), and synthetic ray batches for throughput runs.

Everything numeric happens in the native libraries; this file only prepares inputs."""
import importlib.util
import os
import sys

import numpy as np
import torch

from . import build, config

_host = None
N_LEVELS = 16


def host():
    """The pybind11 module `_f2n_host` (C++/LibTorch plugin classes).  Fails loudly if it was not built."""
    global _host
    if _host is None:
        path = build.host_module_path()
        if not os.path.exists(path) or not os.path.exists(build.LIB):
            raise RuntimeError("native extensions missing (%s / %s): run __graft_entry__.build(); there is no "
                               "Python/CPU fallback" % (path, build.LIB))
        spec = importlib.util.spec_from_file_location("f2_nerf_amd._f2n_host", path)
        _host = importlib.util.module_from_spec(spec)
        sys.modules["f2_nerf_amd._f2n_host"] = _host
        spec.loader.exec_module(_host)
    return _host


def xavier_mlp_params(rng, n_hidden, d_in=32, d_hidden=64):
    """Flat fp32 parameter vector in the exchange layout (layers first->last, each [out,in] row-major, last layer
    padded to 16 rows).  tcnn's pcg32 init stream cannot be reproduced, weights are exchanged as arrays."""
    parts = []
    for rows, cols in [(d_hidden, d_in)] + [(d_hidden, d_hidden)] * (n_hidden - 1) + [(16, d_hidden)]:
        s = np.sqrt(6.0 / (rows + cols))
        parts.append(rng.uniform(-s, s, rows * cols).astype(np.float32))
    return np.concatenate(parts)


def initial_states(state, cfg, seed=2022, table_init="reference", n_images=None):
    """The renderer state vector in the reference's checkpoint order for a fresh run on the
    serialised scene `state` (tests/golden/fox_state.npz): PersSampler [nodes, warps, visit_cnt, milestones] ->
    Hash3DAnchored [feat_pool, prim_pool, bias_pool, n_volumes, field MLP] -> SHShader [colour MLP] -> app_emb."""
    rng = np.random.default_rng(seed)
    n_nodes = state["tree_nodes"].size // 64
    log2 = int(cfg["field"]["log2_table_size"])
    pool = (1 << log2) * N_LEVELS
    if table_init == "reference":  # Hash3DAnchored.cpp:33
        table = ((rng.random((pool, 2), dtype=np.float32) * np.float32(.2) - np.float32(1.)) * np.float32(1e-4))
    else:  # a "trained-looking" table for tests that want non-trivial features
        table = rng.standard_normal((pool, 2)).astype(np.float32) * np.float32(table_init)
    n_images = int(n_images if n_images is not None else len(state["poses"]))
    milestones = np.array(list(reversed(cfg["pts_sampler"]["sub_div_milestones"])), np.int32)
    arrays = [state["tree_nodes"], state["pers_trans"], np.zeros(n_nodes, np.int32), milestones,
              table, state["prim_pool"], state["bias_pool"], np.array([int(state["n_volumes"])], np.int32),
              xavier_mlp_params(rng, int(cfg["field"]["n_hidden_layers"]), 32, int(cfg["field"]["mlp_hidden_dim"])),
              xavier_mlp_params(rng, int(cfg["shader"]["n_hiddens"]), int(cfg["shader"]["d_in"]), int(cfg["shader"]["d_hidden"])),
              (rng.standard_normal((n_images, 16)) * 0.1).astype(np.float32)]
    return arrays


def make_runner(state, preset="wanjinyou", overrides=None, seed=2022, table_init="reference", device="cuda:0", cfg=None):
    """ExpRunner on `device` for the serialised scene `state` with one of the reference's experiment presets."""
    if not torch.cuda.is_available():
        raise RuntimeError("no HIP device: the hot path has no CPU implementation")
    torch.cuda.set_device(device)
    cfg = cfg if cfg is not None else config.preset(preset, overrides)
    flat = config.flatten(cfg)
    flat["runtime.n_volumes"] = str(int(state["n_volumes"]))
    n_images = len(state["poses"])
    runner = host().ExpRunner(flat, n_images)
    arrays = initial_states(state, cfg, seed, table_init, n_images)
    runner.load_states([torch.from_numpy(np.ascontiguousarray(a)) for a in arrays])
    runner.set_edge_pool(torch.from_numpy(np.ascontiguousarray(state["edge_pool"])))
    ts = state["train_set"]
    runner.set_train_cameras(torch.from_numpy(state["w2c"][ts]), torch.from_numpy(state["intri"][ts]),
                             torch.from_numpy(state["bounds"][ts]))
    return runner, cfg, arrays


def make_runner_from_cameras(poses, intri, bounds, train_set, preset="wanjinyou", overrides=None, device="cuda:0", cfg=None):
    """A fresh ExpRunner for a NEW scene: octree, perspective warps and edge pool are constructed from the training
    cameras on the device (host().build_octree), table / primes / biases / MLPs are initialised as the
    reference's constructors do.  poses [C,3,4] normalised c2w, intri [C,3,3], bounds [C,2] (already relaxed)."""
    if not torch.cuda.is_available():
        raise RuntimeError("no HIP device: the hot path has no CPU implementation")
    torch.cuda.set_device(device)
    cfg = cfg if cfg is not None else config.preset(preset, overrides)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32))
    ts = np.asarray(train_set)
    ps = cfg["pts_sampler"]
    built = host().build_octree(t(poses[ts]), t(intri[ts]), t(bounds[ts]), int(ps["max_level"]),
                                float(1 << (int(ps["bbox_levels"]) - 1)), float(ps["split_dist_thres"]))
    flat = config.flatten(cfg)
    flat["runtime.n_volumes"] = str(int(built["n_volumes"]))
    runner = host().ExpRunner(flat, len(poses))
    runner.install_octree(built["tree_nodes"], built["pers_trans"], built["edge_pool"])
    w2c = np.linalg.inv(np.concatenate([poses, np.tile(np.array([[[0, 0, 0, 1]]], np.float32), (len(poses), 1, 1))], 1))[:, :3]
    runner.set_train_cameras(t(w2c[ts]), t(intri[ts]), t(bounds[ts]))
    return runner, cfg, built


def make_dataset(state, images=None):
    """Device-resident ray source (host C++ `Dataset`) for the serialised scene; `images` fp32 [C,H,W,3] or None."""
    H, W = [int(v) for v in state["image_hw"]]
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    img = torch.empty(0) if images is None else (images if torch.is_tensor(images) else t(np.asarray(images, np.float32)))
    return host().Dataset(t(state["poses"]), t(state["intri"]), t(state["dist_params"]), t(state["bounds"]), img, H, W,
                          [int(v) for v in state["train_set"]], [int(v) for v in state["test_set"]], [])


# ---------------------------------------------------------------------------------------------------------
# synthetic rays (throughput runs): Dataset::RandRaysWholeSpace semantics (Dataset/Dataset.cpp:245-255)
# ---------------------------------------------------------------------------------------------------------
def _quat_from_rot(R):
    t = np.trace(R)
    if t > 0:
        s = np.sqrt(t + 1.0) * 2
        q = [0.25 * s, (R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s]
    else:
        i = int(np.argmax(np.diag(R)))
        j, k = (i + 1) % 3, (i + 2) % 3
        s = np.sqrt(R[i, i] - R[j, j] - R[k, k] + 1.0) * 2
        q = [0, 0, 0, 0]
        q[0] = (R[k, j] - R[j, k]) / s
        q[1 + i] = 0.25 * s
        q[1 + j] = (R[j, i] + R[i, j]) / s
        q[1 + k] = (R[k, i] + R[i, k]) / s
    return np.array(q, np.float64)


def _rot_from_quat(q):
    w, x, y, z = q / np.linalg.norm(q)
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def pose_interpolate(a, b, alpha):
    """CameraUtils.cpp:11-41: quaternion slerp of the rotations, lerp of the positions."""
    qa, qb = _quat_from_rot(a[:3, :3].astype(np.float64)), _quat_from_rot(b[:3, :3].astype(np.float64))
    d = float(np.dot(qa, qb))
    if d < 0:
        qb, d = -qb, -d
    if d > 1 - 1e-6:
        q = qa * (1 - alpha) + qb * alpha
    else:
        th = np.arccos(d)
        q = (np.sin((1 - alpha) * th) * qa + np.sin(alpha * th) * qb) / np.sin(th)
    out = np.zeros((3, 4), np.float32)
    out[:3, :3] = _rot_from_quat(q)
    out[:3, 3] = a[:3, 3] * (1 - alpha) + b[:3, 3] * alpha
    return out


def synthetic_ray_batch(state, n_rays, rng, n_poses=16):
    """Random-pose rays: each of `n_poses` poses blends 3 cameras of a random 10-camera window; uniform pixels.
    Returns numpy (rays_o, rays_d [unnormalised, as the reference hands them to GetSamples], bounds, gt, emb_idx)."""
    poses, K = state["poses"], state["intri"][0]
    n_img = len(poses)
    H, W = [int(v) for v in state["image_hw"]]
    per = (n_rays + n_poses - 1) // n_poses
    ro, rd = [], []
    for _ in range(n_poses):
        w = rng.random(3) + 1e-7
        idx = rng.integers(0, 10, 3) + rng.integers(0, n_img - 10)
        pose = pose_interpolate(poses[idx[0]], poses[idx[1]], w[1] / (w[1] + w[0]))
        pose = pose_interpolate(pose, poses[idx[2]], w[2] / w.sum())
        i = rng.integers(0, H, per).astype(np.float32) + np.float32(.5)
        j = rng.integers(0, W, per).astype(np.float32) + np.float32(.5)
        d_cam = np.stack([(j - K[0, 2]) / K[0, 0], -(i - K[1, 2]) / K[1, 1], -np.ones(per, np.float32)], -1)
        rd.append((d_cam @ pose[:3, :3].T).astype(np.float32))
        ro.append(np.repeat(pose[None, :3, 3], per, 0).astype(np.float32))
    ro, rd = np.concatenate(ro)[:n_rays], np.concatenate(rd)[:n_rays]
    bounds = np.tile(np.array([[state["bounds"][:, 0].min(), state["bounds"][:, 1].max()]], np.float32), (n_rays, 1))
    gt = rng.random((n_rays, 3), dtype=np.float32)
    emb = rng.integers(0, n_img, n_rays).astype(np.int32)
    return ro, rd, bounds, gt, emb


def to_dev(*arrays, device="cuda"):
    return [torch.from_numpy(np.ascontiguousarray(a)).to(device) for a in arrays]
