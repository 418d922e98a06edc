"""Test infrastructure.  What the kernels of tests/wave_emul/selftest.hip must produce -- the CDNA ISA's semantics of the cross-lane
operations the product uses.  `run(name, *args)` launches entry point `name` of a library built from selftest.hip with numpy arrays
for its buffers and returns when the results are in them: the emulation (tests/test_wave_emul_cpu.py) passes host pointers straight
through; tools/wave_selftest_on_gpu.py stages the arrays through device memory for the hipcc build on an MI355X."""
import numpy as np


def check_permutes(run):
    out = np.full((18, 64), 12345, np.int32)
    run("st_permutes", out)
    l = np.arange(64)
    row, c = l & ~15, l & 15
    want = [
        l ^ 1, l ^ 2,                                                                                         # quad_perm
        np.where(c >= 1, l - 1, -7), np.where(c >= 3, l - 3, 0), np.where(c + 2 <= 15, l + 2, -7),            # row_shr:1 / :3 bound_ctrl / row_shl:2
        row + 15 - c, (l & ~7) | (7 - (l & 7)),                                                               # row_mirror, row_half_mirror
        np.where(((l >> 4) & 1) == 0, l ^ 1, -7), np.where(c < 8, l ^ 1, -7), row + ((c - 1) & 15),           # row_mask, bank_mask, row_ror:1
        row + 15, np.full(64, 70 & 63), l ^ 32, l ^ 3, np.where(c >= 1, l - 1, l), np.where((l & 7) + 2 < 8, l + 2, l),  # __shfl family
        (((l & ~7) + 7) * 0.5).astype(np.int32), l ^ 1,                                                       # float and 8-byte values
    ]
    for k, w in enumerate(want):
        assert (out[k] == w).all(), (k, out[k], w)


def check_masks(run):
    out = np.zeros((8, 64), np.uint64)
    run("st_masks", out)
    l = np.arange(64)
    mask = lambda pred: int(sum(1 << int(i) for i in l[pred]))  # noqa: E731
    untouched = 0xDEAD
    assert (out[0] == mask(l % 3 == 0)).all()
    assert (out[1][l % 2 == 1] == mask((l % 2 == 1) & (l < 40))).all() and (out[1][l % 2 == 0] == untouched).all()   # only odd lanes vote
    assert (out[2][l % 2 == 0] == mask((l % 2 == 0) & (l >= 60))).all() and (out[2][l % 2 == 1] == untouched).all()  # the other branch
    assert (out[3][l >= 13] == 13).all() and (out[3][l < 13] == untouched).all()                                     # readfirstlane
    assert (out[4][l != 5] == 0).all() and out[4][5] == untouched              # ds_bpermute from an inactive lane: 0
    want5 = np.where(l >= 2, np.where((l & 15) >= 1, l - 1, -7), 0).astype(np.int64)
    want5[2] = -7                                                              # DPP from an inactive lane, no bound_ctrl: old
    assert (out[5][2:].astype(np.uint32) == want5[2:].astype(np.uint32)).all() and (out[5][:2] == untouched).all()
    left = lambda i: 64 - 8 * (i + 1)  # noqa: E731  (lanes still in the loop at trip i)
    assert (out[6] == np.array([sum(left(i) for i in range(int(x) >> 3)) for x in l])).all()
    assert (out[7][:50] == (1 << 50) - 1).all() and (out[7][50:] == untouched).all()   # lanes 50..63 have left the kernel


def check_block(run):
    nb = 3
    out = np.zeros(nb * 256 + 1, np.int32)
    run("st_block", out, nb)
    for b in range(nb):
        t = np.arange(256)
        a = b * 1000 + (255 - t)
        got = out[b * 256:(b + 1) * 256]
        assert (got[:64] == a[:64].sum()).all()
        assert (got[64:] == (a + (255 - t) ** 2 + 1000000 * (b == 1))[64:]).all()
    assert out[-1] == nb


def check_mfma(run):
    rng = np.random.default_rng(5)
    A = rng.integers(-8, 9, (16, 32)).astype(np.float16)
    B = rng.integers(-8, 9, (32, 16)).astype(np.float16)
    C = rng.integers(-100, 100, (16, 16)).astype(np.float32)
    D16, D32 = np.zeros((16, 16), np.float32), np.zeros((16, 16), np.float32)
    run("st_mfma", A, B, C, D16, D32)
    Af, Bf = A.astype(np.float32), B.astype(np.float32)
    assert (D16 == Af[:, :16] @ Bf[:16] + C).all()   # (small integers: every sum is exact)
    assert (D32 == Af @ Bf + C).all()


def check_persistent(run):
    n_groups, n_blocks = 23, 3
    out = np.zeros((n_groups, 64), np.int32)
    counter = np.zeros(1, np.int32)
    run("st_persistent", out, counter, n_groups, n_blocks)
    lane = np.arange(64)
    for g in range(n_groups):
        trips = (g * 7 + (lane >> 4) * 5) % 11
        acc = np.array([sum(i + (int(x) ^ 1) for i in range(int(t))) if t != 3 else 0 for x, t in zip(lane, trips)])
        assert (out[g] == acc + 1).all(), (g, out[g], acc + 1)
    assert counter[0] == n_groups + n_blocks  # (every block's last fetch comes back empty)


CASES = ("permutes", "masks", "block", "mfma", "persistent")
