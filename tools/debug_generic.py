"""Scratch diagnostic: the non-default network configuration of tests/test_gpu_mlp_shapes.py, step by step."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import f2_nerf_amd
from f2_nerf_amd import runtime
from test_gpu_e2e import fox_batch
F32 = np.float32
st = dict(np.load(os.path.join(ROOT, "tests", "golden", "fox_state.npz")))
which = sys.argv[1] if len(sys.argv) > 1 else "both"
ov = ["field.log2_table_size=14", "train.learning_rate_warm_up_end_iter=20"]
if which in ("both", "field"):
    ov += ["field.mlp_hidden_dim=32", "field.n_hidden_layers=2"]
if which in ("both", "shader"):
    ov += ["shader.degree=3", "shader.d_in=25", "shader.d_hidden=128", "shader.n_hiddens=3"]
runner, cfg, arrays = runtime.make_runner(st, "wanjinyou", ov, seed=7, table_init=0.3)
runner.n_edge_pts = 512
rng = np.random.default_rng(8)
R = 512
ro, rd, bounds, cam = fox_batch(st, rng, R)
gt = np.tile(np.array([[0.7, 0.4, 0.1]], F32), (R, 1))
d = runtime.to_dev(ro, rd, bounds, gt, cam)
for it in range(40):
    s = runner.train_step(d[0], d[1], d[2], d[3], d[4], True)
    line = "%s it %2d step %2d skipped %d mse %.5f loss %.5f kept %d" % (which, it, runner.iter_step, s["skipped_nan"], float(s["mse"]), float(s["loss"]), s["n_meaningful"])
    if s["skipped_nan"]:
        runner.zero_grad()
        runner.train_step(d[0], d[1], d[2], d[3], d[4], False)
        g = runner.grads()
        line += " | nonfinite: " + " ".join("%s=%d" % (k, int((~torch.isfinite(v)).sum())) for k, v in g.items())
        stt = runner.states()
        line += " | params finite: field %d color %d table %d" % (int(torch.isfinite(stt[8]).all()), int(torch.isfinite(stt[9]).all()), int(torch.isfinite(stt[4]).all()))
    print(line, flush=True)
