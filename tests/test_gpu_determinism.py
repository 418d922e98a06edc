"""Training results must not depend on the sampling schedule (round-4 verdict, weak 1; north star: "bit-exact sample indices").

Every random draw of a training run is keyed by what it is for -- (seed, purpose, batch / step sequence number), host/KeyedDraws.h --
and a batch is sized from the meaningful-samples average at one fixed lag (ExpRunner::BatchSizeFor), so WHEN a batch is sampled
(behind the stat update, one step ahead, two steps ahead, speculatively and repaired, dropped and sampled again) and HOW the
iterations are split over ExpRunner::Train calls cannot move a single draw.  The reference samples batch k + 1 behind
UpdateOctNodes of batch k (src/ExpRunner.cpp:86-93, src/PtsSampler/PersSampler.cu:536-615); every mode below must equal that.
"""
import numpy as np
import pytest
import torch

import f2_nerf_amd  # noqa: F401

pytestmark = pytest.mark.gpu

ITERS = 2400   # compactions at 0 / 1000 / 2000, the first subdivision + MarkInvisibleNodes at 2000, leaves dying from ~1000 on

# (speculative_sampling, speculation_depth, tail_repair, iterations per Train call)
MODES = {
    "default":            (2, 2, True, ITERS),
    "after_the_update":   (0, 1, True, ITERS),    # the reference's order: batch k + 1 is sampled behind the stat update of batch k
    "always_one_ahead":   (1, 1, True, ITERS),
    "always_two_ahead":   (1, 3, True, ITERS),
    "auto_two_ahead_full_repair_chunks_of_37": (2, 3, False, 37),
    "default_chunks_of_100": (2, 2, True, 100),
    # (round 6) ExpRunner::Train draws its batches on the tail stream; here on the main queue as in rounds 4-5, and on the tail
    # stream with the spec_start event kept
    "default_draws_on_the_main_queue": (2, 2, True, ITERS, False, True),
    "two_ahead_draws_off_main_with_the_start_event": (1, 3, True, 61, True, False),
}


def _csum(t):
    return int(t.detach().contiguous().view(torch.int32).to(torch.int64).sum().item())


@pytest.fixture(scope="module")
def fox_scene():
    from f2_nerf_amd import fox_data
    st = fox_data.load_state()
    sc, images = fox_data.scene(2)
    return st, sc, images.cuda()


def _train(fox_scene, mode):
    from f2_nerf_amd import runtime
    st, sc, images = fox_scene
    spec, depth, tail, chunk = MODES[mode][:4]
    draws_off_main, no_start_event = (MODES[mode][4:] + (True, True))[:2]
    ds = runtime.make_dataset(sc, images)
    runner, cfg, _ = runtime.make_runner(st, "wanjinyou", ["train.end_iter=20000"], seed=2022)
    runner.speculative_sampling = spec
    runner.speculation_depth = depth
    runner.tail_repair = tail
    runner.draws_off_main = draws_off_main
    runner.spec_start_without_event = no_start_event
    runner.digest_table = True
    torch.manual_seed(2022)
    it = 0
    while it < ITERS:
        it = min(ITERS, it + chunk)
        runner.train(ds, it, 1)
    s = runner.states()  # [0] nodes [2] visit counts [4] table [8] field MLP [9] colour MLP [10] app_emb
    c = runner.counters()
    out = dict(table=_csum(s[4]), field_mlp=_csum(s[8]), color_mlp=_csum(s[9]), app_emb=_csum(s[-1]),
               nodes=s[0].cpu().numpy().copy(), visit=s[2].cpu().numpy().copy(),
               stats=[t.cpu().numpy().copy() for t in runner.occupancy_buffers()[:2]],
               marched=c["total_marched"], meaningful=c["total_meaningful"], iter_step=runner.iter_step, step_seq=runner.step_seq,
               digest=[tuple(int(v) for v in row) for row in runner.step_digest()], spec=dict(runner.speculation_counters()))
    del runner, ds
    return out


def test_training_is_independent_of_the_sampling_schedule(fox_scene):
    runs = {m: _train(fox_scene, m) for m in MODES}
    ref = runs["after_the_update"]
    assert ref["iter_step"] == ITERS and ref["step_seq"] >= ITERS
    assert ref["spec"]["speculative"] == 0
    # the modes did differ in what they DID: batches were begun ahead, rays repaired, batches dropped at the ProcOctree iterations
    assert runs["always_two_ahead"]["spec"]["speculative"] > ITERS // 2 and runs["always_one_ahead"]["spec"]["speculative"] > ITERS // 2
    assert runs["always_one_ahead"]["spec"]["rays_repaired"] > 0 or runs["always_two_ahead"]["spec"]["rays_repaired"] > 0
    n_nodes = ref["nodes"].size // 64
    assert n_nodes > 897, n_nodes  # the subdivision at iteration 2000 ran
    for m, r in runs.items():
        print("MODE %-42s nodes %6d marched %d meaningful %d table %d spec %s" % (m, r["nodes"].size // 64, r["marched"], r["meaningful"], r["table"], r["spec"]))
    for m, r in runs.items():
        if m == "after_the_update":
            continue
        # per-step digest first: it names the FIRST step and quantity that differ (seq, iter, rays, marched, kept, table checksum)
        tail = ref["digest"][-min(len(ref["digest"]), len(r["digest"])):]
        tail_r = r["digest"][-len(tail):]
        first = next((i for i, (a, b) in enumerate(zip(tail, tail_r)) if a != b), None)
        assert first is None, "%s parts from sampling-after-the-update at step %s: %s vs %s" % (m, tail[first][0], tail[first], tail_r[first])
        for k in ("table", "field_mlp", "color_mlp", "app_emb", "marched", "meaningful", "iter_step", "step_seq"):
            assert r[k] == ref[k], (m, k, r[k], ref[k])
        assert r["nodes"].shape == ref["nodes"].shape and (r["nodes"] == ref["nodes"]).all(), m
        assert (r["visit"] == ref["visit"]).all(), m
        assert all((a == b).all() for a, b in zip(r["stats"], ref["stats"])), m


def test_keyed_draws_do_not_depend_on_when_or_how_often_they_are_made(fox_scene):
    """The draws themselves: batch k's rays are a function of (seed, k, ray count) -- drawn early, late, twice, or by another
    data set object --, and a different k or seed gives other rays."""
    from f2_nerf_amd import runtime
    st, sc, images = fox_scene
    torch.manual_seed(77)
    a, b = runtime.make_dataset(sc, images), runtime.make_dataset(sc, images)
    x5 = a.rand_rays_data(4096, 1, 5)
    a.rand_rays_data(1000, 1, 9)            # (something else drawn in between)
    torch.rand(123, device="cuda")           # (the default generator's position is not an input either)
    y5 = b.rand_rays_data(4096, 1, 5)
    z5 = a.rand_rays_data(4096, 1, 5)
    for u, v, w in zip(x5, y5, z5):
        assert torch.equal(u, v) and torch.equal(u, w)
    x6 = a.rand_rays_data(4096, 1, 6)
    assert not torch.equal(x5[1], x6[1])
    torch.manual_seed(78)
    w5 = a.rand_rays_data(4096, 1, 5)
    assert not torch.equal(x5[1], w5[1])
    # unkeyed calls walk the object's own sequence: consecutive batches differ
    p, q = a.rand_rays_data(512), a.rand_rays_data(512)
    assert not torch.equal(p[1], q[1])


def test_replicas_of_a_data_parallel_run_draw_their_own_streams(fox_scene):
    """With one seed on every rank (the replicas' parameters and octree come from it), what makes rank r's batches its own is the
    salt of the keyed draws (host/KeyedDraws.h: key = seed ^ purpose ^ salt(rank); DataParallel::Attach sets it): another replica
    draws other rays for the same (seed, batch number), the same replica the same ones again, and rank 0 the single-GPU ones."""
    from f2_nerf_amd import runtime
    st, sc, images = fox_scene
    torch.manual_seed(77)
    host = runtime.host()
    a = runtime.make_dataset(sc, images)
    x0 = a.rand_rays_data(4096, 1, 5)
    try:
        host.dp_set_replica(3)
        x3 = a.rand_rays_data(4096, 1, 5)
        y3 = a.rand_rays_data(4096, 1, 5)
    finally:
        host.dp_set_replica(0)
    z0 = a.rand_rays_data(4096, 1, 5)
    assert not torch.equal(x0[1], x3[1])
    assert torch.equal(x3[1], y3[1]) and torch.equal(x0[1], z0[1])


def test_in_kernel_draws_are_philox_of_their_key():
    """The march noise a prologue launch draws for itself is Philox4x32-10 of (key, seq, element) bit for bit (integer work: exact),
    mapped like PersSampler.cu:372-381: ((u - .5) + 1) * fineness."""
    from f2_nerf_amd import capi
    n_rays, n_noise, key, seq, fin = 256, 1024 + 256 + 10, 0x1234567890ABCDEF, (1 << 40) + 7, np.float32(2.5)
    dirs = torch.randn(n_rays, 3, device="cuda")
    out, zero, noise = torch.empty_like(dirs), torch.ones(3, dtype=torch.int32, device="cuda"), torch.empty(n_noise, device="cuda")
    capi.sampler_prologue_keyed(n_rays, dirs, out, zero, n_noise, key, seq, float(fin), noise)
    i = np.arange(n_noise)
    ctr = np.stack([i >> 2, np.zeros_like(i), np.full_like(i, seq & 0xFFFFFFFF), np.full_like(i, seq >> 32)], 1).astype(np.uint32)
    from philox_ref import philox4x32_10
    x = philox4x32_10(ctr, (key & 0xFFFFFFFF, key >> 32))[i, i & 3]
    u = (x >> np.uint32(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)
    want = ((u - np.float32(.5)) + np.float32(1.)) * fin
    got = noise.cpu().numpy()
    assert (got.view(np.uint32) == want.view(np.uint32)).all()
    assert int(zero.sum()) == 0 and torch.allclose(out.norm(dim=1), torch.ones(n_rays, device="cuda"), atol=1e-5)
    assert 0.45 < float(u.mean()) < 0.55 and u.min() >= 0 and u.max() < 1
