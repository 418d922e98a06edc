"""Synthetic camera rigs for the BASELINE workloads whose data sets cannot be reached from here:

  llff      forward-facing: ~60 cameras on a jittered 2-D grid in a plane, all looking down -z at a scene 2..12 units away
            (LLFF "horns": 1008 x 756 at dataset.factor 4, near / far from the scene's depth range) -- confs/llff.yaml
  nerf-360  inward ring: ~185 cameras at radius ~1 around the origin with +-20 degrees of elevation, looking at the centre
            (mip-NeRF-360 "garden": 1297 x 840 at factor 4) -- confs/nerf-360.yaml

A rig is emitted in the reference's own on-disk convention -- cams_meta rows [pose 3x4 | K 3x3 | k1 k2 p1 p2 | near far]
(Dataset.cpp:35-52; OpenGL camera axes: x right, y up, looking along -z) -- and then goes through the reference's scene
preparation: NormalizeScene (camera centroid to the origin, farthest camera at radius 1, Dataset.cpp:127-146), bounds
relaxation by dataset.bounds_factor and clamping to [1e-2, 1e9] (:73-76), every-8th-image test split (:105-109).  The
octree / perspective warps / edge pool are then BUILT from those cameras on the device (host().build_octree), exactly as
for a real capture.  Pixels are noise: these rigs exist to measure throughput and to check sampler parity on other
geometries than the fox, not image quality.
"""
import numpy as np

F32 = np.float32


def _look_at(pos, target, up=(0., 1., 0.)):
    """c2w 3x4 with OpenGL axes: the camera looks along its -z."""
    pos, target, up = np.asarray(pos, np.float64), np.asarray(target, np.float64), np.asarray(up, np.float64)
    z = pos - target
    z /= np.linalg.norm(z)
    x = np.cross(up, z)
    x /= np.linalg.norm(x)
    y = np.cross(z, x)
    return np.concatenate([np.stack([x, y, z], 1), pos[:, None]], 1)


def forward_facing(rng, n_side=(8, 8), extent=(2.0, 1.5), jitter=0.08, depth=(2.0, 12.0), hw=(756, 1008), focal=840.0):
    """LLFF-like: n_side[0] x n_side[1] cameras in the z = 0 plane, looking down -z with a few degrees of wobble."""
    poses, bounds = [], []
    for iy in range(n_side[1]):
        for ix in range(n_side[0]):
            p = np.array([(ix / (n_side[0] - 1) - .5) * extent[0], (iy / (n_side[1] - 1) - .5) * extent[1], 0.])
            p[:2] += rng.normal(0, jitter, 2)
            p[2] += rng.normal(0, jitter * .5)
            target = np.array([p[0] * .3, p[1] * .3, -0.5 * (depth[0] + depth[1])]) + rng.normal(0, .15, 3)
            poses.append(_look_at(p, target))
            bounds.append([depth[0] * (1 + rng.normal(0, .03)), depth[1] * (1 + rng.normal(0, .03))])
    return _cams_meta(np.array(poses), hw, focal, np.array(bounds))


def inward_ring(rng, n_cams=185, radius=4.0, elevation_deg=20.0, hw=(840, 1297), focal=960.0, depth=(1.2, 14.0)):
    """360-like: cameras on a wobbly ring around the origin, elevations in +-elevation_deg, all looking at the centre."""
    poses, bounds = [], []
    for k in range(n_cams):
        az = 2 * np.pi * k / n_cams + rng.normal(0, .01)
        el = np.deg2rad(elevation_deg) * np.sin(3.1 * az + 0.4) + rng.normal(0, .02)
        r = radius * (1 + rng.normal(0, .04))
        p = np.array([r * np.cos(el) * np.cos(az), r * np.sin(el) + .6, r * np.cos(el) * np.sin(az)])
        poses.append(_look_at(p, rng.normal(0, .1, 3)))
        bounds.append([depth[0] * (1 + rng.normal(0, .03)), depth[1] * (1 + rng.normal(0, .03))])
    return _cams_meta(np.array(poses), hw, focal, np.array(bounds))


def _cams_meta(poses, hw, focal, bounds):
    n = len(poses)
    K = np.zeros((n, 3, 3))
    K[:, 0, 0] = K[:, 1, 1] = focal
    K[:, 0, 2], K[:, 1, 2], K[:, 2, 2] = hw[1] * .5, hw[0] * .5, 1.
    meta = np.concatenate([poses.reshape(n, 12), K.reshape(n, 9), np.zeros((n, 4)), bounds.reshape(n, 2)], 1)
    return np.ascontiguousarray(meta, np.float64), (int(hw[0]), int(hw[1]))


def prepare_scene(cams_meta, hw, factor=1.0, bounds_factor=(0.5, 4.0), intrinsics_already_scaled=True):
    """The reference's Dataset constructor on a cams_meta array (Dataset.cpp:35-146) -> the dict layout of
    tests/golden/fox_state.npz minus the octree (poses, intri, dist_params, bounds, w2c, center, radius, splits, image_hw).
    `hw` is the image size the intrinsics refer to (the rigs above emit intrinsics at the final resolution)."""
    import torch
    cam = np.asarray(cams_meta, np.float64).astype(F32).reshape(-1, 27)
    n = len(cam)
    poses = cam[:, 0:12].reshape(n, 3, 4).copy()
    intri = cam[:, 12:21].reshape(n, 3, 3).copy()
    if not intrinsics_already_scaled:
        intri[:, 0:2, 0:3] /= F32(factor)
    dist = cam[:, 21:25].copy()
    bounds = cam[:, 25:27].copy()
    # NormalizeScene (Dataset.cpp:127-146) with the very ATen ops the reference calls (mean / linalg_norm / linalg_inv on
    # float32 tensors): numpy's pairwise float32 mean lands one ulp away from ATen's, and the scene centre moves every pose.
    # Pinned against the reference's own function: tests/test_oracle_vs_ref.py.
    with torch.no_grad():
        tp = torch.from_numpy(poses)
        cam_pos = tp[:, :3, 3].clone()
        center_t = cam_pos.mean(0, False)
        radius_t = torch.linalg.norm(cam_pos - center_t.unsqueeze(0), 2, -1, False).max()
        radius = F32(radius_t.item())
        tp[:, :3, 3] = (cam_pos - center_t.unsqueeze(0)) / float(radius)
        w2c4 = torch.eye(4, dtype=torch.float32).unsqueeze(0).repeat(n, 1, 1).contiguous()
        w2c4[:, :3, :] = tp.clone()
        w2c = torch.linalg.inv(w2c4)[:, :3, :].contiguous().numpy()
        tb = (torch.from_numpy(bounds) / float(radius)).contiguous()
        tb = torch.stack([tb[..., 0] * float(bounds_factor[0]), tb[..., 1] * float(bounds_factor[1])], -1).contiguous()  # :73-75
        tb.clamp_(1e-2, 1e9)
        bounds = tb.numpy().astype(F32)
        center = center_t.numpy().astype(F32)
    test = np.array([i for i in range(n) if i % 8 == 0], np.int32)
    train = np.array([i for i in range(n) if i % 8 != 0], np.int32)
    return dict(poses=poses.astype(F32), intri=intri.astype(F32), dist_params=dist.astype(F32), bounds=bounds, w2c=w2c,
                center=center, radius=radius, train_set=train, test_set=test, image_hw=np.array(hw, np.int32))


PRESET_RIG = {"llff": "forward_facing", "nerf-360": "inward_ring"}


def make_scene(preset, cfg, seed=2022):
    """Scene dict for a preset: the fox capture for the wanjinyou family / free, a synthetic rig for llff and nerf-360."""
    rng = np.random.default_rng(seed)
    bf = tuple(float(v) for v in cfg["dataset"]["bounds_factor"])
    if preset == "llff":
        meta, hw = forward_facing(rng)
    elif preset == "nerf-360":
        meta, hw = inward_ring(rng)
    else:
        raise KeyError("no synthetic rig for preset %r" % preset)
    return prepare_scene(meta, hw, float(cfg["dataset"]["factor"]), bf)


def build_runner(preset, overrides=None, seed=2022, device="cuda:0"):
    """(runner, cfg, scene dict incl. the octree built on the device) for a rig preset."""
    from . import config, runtime
    import torch
    cfg = config.preset(preset, overrides)
    sc = make_scene(preset, cfg, seed)
    torch.manual_seed(seed)
    runner, cfg, built = runtime.make_runner_from_cameras(sc["poses"], sc["intri"], sc["bounds"], sc["train_set"], preset, overrides,
                                                          device=device)
    sc = dict(sc)
    sc["tree_nodes"] = built["tree_nodes"].numpy().copy()
    sc["pers_trans"] = built["pers_trans"].numpy().copy()
    sc["edge_pool"] = built["edge_pool"].numpy().copy()
    sc["n_volumes"] = np.int32(built["n_volumes"])
    return runner, cfg, sc
