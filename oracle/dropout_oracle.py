"""numpy restatement of the dropout mask of ``mv_dropout`` (TEST INFRASTRUCTURE).

nn.Dropout (reference vit.py:50,52,75,311) draws its Bernoulli mask from torch's generator; the HIP path uses
Philox4x32-10 (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3", SC'11: the generator torch's own CUDA dropout
is built on) with counter = (i >> 2 as two 32-bit words, offset as two 32-bit words), key = seed as two 32-bit words, and
element i takes word ``i & 3``; it is kept when that word >= p * 2^32.  The reference's RNG STREAM cannot be reproduced on
another device, so parity with the reference is statistical (keep rate, scaling); this file pins the kernel's own definition
bit for bit.  PARITY UNPINNED against the reference.
"""
import numpy as np

M0, M1, W0, W1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57), 0x9E3779B9, 0xBB67AE85
MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Vectorised over the counter words (uint64 arrays holding 32-bit values); k0, k1 python ints."""
    c0, c1, c2, c3 = (np.asarray(c, dtype=np.uint64) for c in (c0, c1, c2, c3))
    for _ in range(10):
        p0, p1 = M0 * c0, M1 * c2
        hi0, lo0, hi1, lo1 = p0 >> np.uint64(32), p0 & MASK, p1 >> np.uint64(32), p1 & MASK
        c0, c1, c2, c3 = (hi1 ^ c1 ^ np.uint64(k0)), lo1, (hi0 ^ c3 ^ np.uint64(k1)), lo0
        k0, k1 = (k0 + W0) & 0xFFFFFFFF, (k1 + W1) & 0xFFFFFFFF
    return c0, c1, c2, c3


def dropout_keep_mask(n, p, seed, offset):
    """bool [n]: True where element i survives nn.Dropout(p) under (seed, offset)."""
    n4 = (n + 3) // 4
    i = np.arange(n4, dtype=np.uint64)
    z = np.zeros(n4, dtype=np.uint64)
    r = philox4x32_10(i & MASK, i >> np.uint64(32), z + np.uint64(offset & 0xFFFFFFFF), z + np.uint64((offset >> 32) & 0xFFFFFFFF),
                      seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    words = np.stack(r, axis=1).reshape(-1)[:n]
    return words >= np.uint64(int(p * 4294967296.0))
