"""Config composition for the hot path -- replaces the reference's hydra launcher (scripts/run.py:38-71) with a
PyYAML `defaults:` composer, and flattens the result to the "a.b.c" -> string map GlobalDataPool consumes
(yaml-cpp is not available to the C++ host).

Two ways to obtain a config:
  * compose_yaml(conf_dir, name, overrides): reads a hydra-style directory (e.g. the reference's own `confs/`)
    in place: `<name>.yaml` + its `defaults:` groups, `_self_` last, then dotted overrides;
  * preset(name): the same key/value content for the reference's shipped experiment files
    (confs/wanjinyou.yaml, wanjinyou_big.yaml, llff.yaml, nerf-360.yaml, free.yaml), kept here as data so that the
    GPU box (which has no /root/reference) can run them.  Keys are the reference's, verbatim.
"""
import copy
import os

GROUP_DEFAULTS = {
    "train": {  # confs/train/20k.yaml
        "pts_batch_size": 262144, "end_iter": 20000, "report_freq": 50, "vis_freq": 2500, "stats_freq": 5000,
        "save_freq": 20000, "validate_freq": 100000, "tv_loss_weight": 1e-1, "ray_march_init_fineness": 4,
        "ray_march_fineness_decay_end_iter": 10000, "disp_loss_weight": 0., "learning_rate": 1e-2,
        "learning_rate_alpha": 1e-1, "learning_rate_warm_up_end_iter": 1000, "var_loss_weight": 1e-2,
        "var_loss_start": 5000, "var_loss_end": 10000, "gradient_scaling_start": 0, "gradient_scaling_end": 0},
    "dataset": {"factor": 1.0, "ray_sample_mode": "all_images", "data_at_gpu": True, "bounds_factor": [0.5, 2.0]},
    "renderer": {"bg_color": "rand_noise", "use_app_emb": False},
    "pts_sampler": {"type": "PersSampler", "bbox_min": [-1.0, -1.0, -1.0], "bbox_max": [1.0, 1.0, 1.0],
                    "sub_div_milestones": [2000, 4000, 6000, 8000, 10000], "compact_freq": 1000,
                    "max_oct_intersect_per_ray": 1024, "bbox_levels": 10, "max_level": 16, "split_dist_thres": 1.5,
                    "sample_l": 3.90625e-3, "scale_by_dis": False, "near": 0.05},
    "field": {"type": "Hash3DAnchored", "log2_table_size": 19, "rand_bias": True, "mlp_hidden_dim": 64,
              "mlp_out_dim": 16, "n_hidden_layers": 1},
    "shader": {"type": "SHShader", "d_in": 32, "d_out": 3, "d_hidden": 64, "n_hiddens": 2, "degree": 4},
}

_TRAIN_50K = {"end_iter": 50000, "save_freq": 25000}  # confs/train/50k.yaml differs from 20k only here

PRESETS = {
    "wanjinyou": {"renderer": {"use_app_emb": True}, "pts_sampler": {"near": 0.01, "scale_by_dis": True},
                  "dataset": {"factor": 2, "bounds_factor": [0.5, 4.0]},
                  "train": {"ray_march_init_fineness": 16, "gradient_scaling_start": 1000, "gradient_scaling_end": 5000}},
    "wanjinyou_big": {"field": {"log2_table_size": 20}, "renderer": {"use_app_emb": True},
                      "pts_sampler": {"split_dist_thres": 1.5, "near": 0.01, "scale_by_dis": True},
                      "dataset": {"factor": 2, "bounds_factor": [0.5, 4.0]},
                      "train": dict(_TRAIN_50K, ray_march_init_fineness=16, gradient_scaling_start=1000,
                                    gradient_scaling_end=5000)},
    "llff": {"pts_sampler": {"sub_div_milestones": [1000, 2000, 4000, 6000, 8000, 10000], "sample_l": 1.953125e-3},
             "dataset": {"factor": 4, "bounds_factor": [0.25, 4.0]}, "train": {"disp_loss_weight": 5e-2}},
    "nerf-360": {"dataset": {"factor": 2, "bounds_factor": [0.5, 4.0]}},
    "free": {"dataset": {"factor": 2, "bounds_factor": [0.5, 4.0]}},
}


def _merge(dst, src):
    for k, v in src.items():
        if isinstance(v, dict) and isinstance(dst.get(k), dict):
            _merge(dst[k], v)
        else:
            dst[k] = copy.deepcopy(v)
    return dst


def _apply_overrides(cfg, overrides):
    for ov in overrides or []:
        key, val = ov.split("=", 1)
        import yaml
        node = cfg
        parts = key.split(".")
        for p in parts[:-1]:
            node = node.setdefault(p, {})
        node[parts[-1]] = yaml.safe_load(val)
    return cfg


def preset(name, overrides=None):
    if name not in PRESETS:
        raise KeyError("unknown preset %r (have %s)" % (name, sorted(PRESETS)))
    cfg = copy.deepcopy(GROUP_DEFAULTS)
    _merge(cfg, PRESETS[name])
    cfg.update({"case_name": "tmp", "exp_name": "test", "is_continue": False, "mode": "train"})
    return _apply_overrides(cfg, overrides)


def compose_yaml(conf_dir, name, overrides=None):
    """Hydra-style composition: `defaults:` entries `- group: option` load conf_dir/group/option.yaml under key
    `group`; `_self_` marks where the file's own body merges (last if absent)."""
    import yaml
    with open(os.path.join(conf_dir, name + ".yaml")) as f:
        root = yaml.safe_load(f)
    defaults = root.pop("defaults", [])
    cfg = {}
    self_done = False
    for d in defaults:
        if d == "_self_":
            _merge(cfg, root)
            self_done = True
        else:
            (group, option), = d.items()
            with open(os.path.join(conf_dir, group, str(option) + ".yaml")) as f:
                _merge(cfg.setdefault(group, {}), yaml.safe_load(f) or {})
    if not self_done:
        _merge(cfg, root)
    return _apply_overrides(cfg, overrides)


def flatten(cfg, prefix=""):
    """Nested dict -> {"a.b": "value"}; lists become comma-separated; bools 'true'/'false' (GlobalDataPool.h)."""
    out = {}
    for k, v in cfg.items():
        key = prefix + str(k)
        if isinstance(v, dict):
            out.update(flatten(v, key + "."))
        elif isinstance(v, (list, tuple)):
            out[key] = ",".join(repr(float(x)) if isinstance(x, float) else str(x) for x in v)
        elif isinstance(v, bool):
            out[key] = "true" if v else "false"
        elif isinstance(v, float):
            out[key] = repr(v)
        else:
            out[key] = str(v)
    return out
