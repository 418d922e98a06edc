"""numpy restatement of Philox4x32-10 (test infrastructure: the checker of the kernels' keyed draws, csrc/f2n_dev.h f2n_philox4x32)."""
import numpy as np


def philox4x32_10(c, k):
    """numpy restatement of Philox4x32-10 (Salmon et al., SC'11) for uint32 counter columns c [n,4] and key k (2,)."""
    c = c.astype(np.uint64).copy()
    k0, k1 = np.uint64(k[0]), np.uint64(k[1])
    M0, M1, MASK = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57), np.uint64(0xFFFFFFFF)
    for _ in range(10):
        p0, p1 = M0 * c[:, 0], M1 * c[:, 2]
        hi0, lo0, hi1, lo1 = p0 >> np.uint64(32), p0 & MASK, p1 >> np.uint64(32), p1 & MASK
        c = np.stack([(hi1 ^ c[:, 1] ^ k0) & MASK, lo1, (hi0 ^ c[:, 3] ^ k1) & MASK, lo0], 1)
        k0, k1 = (k0 + np.uint64(0x9E3779B9)) & MASK, (k1 + np.uint64(0xBB67AE85)) & MASK
    return c.astype(np.uint32)
