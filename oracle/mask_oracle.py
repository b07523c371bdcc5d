"""Oracle: the Dropout2d factors of seeded MC passes (include/rcu.h, rcu_dropout_masks).  TEST INFRASTRUCTURE ONLY.

The reference draws its masks from torch's global generator (common/model/unet.py:16, nn.Dropout2d): there is nothing to be bit-identical with
-- only the law, Bernoulli(1 - p) / (1 - p) per (sample, channel), matters.  The build pins the draw to a counter-based generator so that a pass's
mask is a function of its seed alone; this file restates that definition in numpy.  Philox4x32-10 is Salmon et al., "Parallel Random Numbers:
As Easy as 1, 2, 3" (SC'11); the restatement is pinned by the known-answer vectors of the Random123 distribution (kat_vectors, philox4x32 10
rounds), see tests/test_oracle_golden.py.
"""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = np.uint32(0x9E3779B9), np.uint32(0xBB67AE85)


def philox4x32_10(counter, key):
    """counter: uint32 array [..., 4]; key: (k0, k1) uint32 -> uint32 array [..., 4] (ten rounds)."""
    c = np.array(counter, dtype=np.uint32, copy=True)
    k0, k1 = np.uint32(key[0]), np.uint32(key[1])
    for _ in range(10):
        p0 = M0 * c[..., 0].astype(np.uint64)
        p1 = M1 * c[..., 2].astype(np.uint64)
        hi0, lo0 = (p0 >> np.uint64(32)).astype(np.uint32), p0.astype(np.uint32)
        hi1, lo1 = (p1 >> np.uint64(32)).astype(np.uint32), p1.astype(np.uint32)
        c = np.stack([hi1 ^ c[..., 1] ^ k0, lo1, hi0 ^ c[..., 3] ^ k1, lo0], axis=-1)
        with np.errstate(over='ignore'):
            k0, k1 = np.uint32(k0 + W0), np.uint32(k1 + W1)
    return c


def pass_mask(seed, n, site_channels, site_keep, first_sample=0):
    """The mask of ONE pass in its own layout [site][n][C_site], flattened.  The factor of (image i, site s, channel c) is 1 / keep where the 24-bit
    uniform from word e & 3 of Philox4x32-10(key = (seed low, seed high), counter = (e >> 2 low, e >> 2 high, 0, 0)) is below keep, else 0, with
    e = (first_sample + i) * sum(site_channels) + sum(site_channels[:s]) + c -- a function of the image's global index, not of its batch;
    keep < 0: the site is inactive (ones)."""
    channels = [int(c) for c in site_channels]
    per_sample = sum(channels)
    seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    chunks, off = [], 0
    for ch, keep in zip(channels, site_keep):
        g = np.uint64(first_sample) + np.arange(n, dtype=np.uint64)
        e = (g[:, None] * np.uint64(per_sample) + np.uint64(off) + np.arange(ch, dtype=np.uint64)[None, :]).reshape(-1)
        q = e >> np.uint64(2)
        counter = np.zeros((e.size, 4), dtype=np.uint32)
        counter[:, 0] = (q & np.uint64(0xFFFFFFFF)).astype(np.uint32)
        counter[:, 1] = (q >> np.uint64(32)).astype(np.uint32)
        words = philox4x32_10(counter, (seed & 0xFFFFFFFF, seed >> 32))
        word = words[np.arange(e.size), (e & np.uint64(3)).astype(np.int64)]
        u = (word >> np.uint32(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)
        keep = np.float32(keep)
        if keep < 0:
            chunks.append(np.ones(e.size, dtype=np.float32))
        elif keep == 0:
            chunks.append(np.zeros(e.size, dtype=np.float32))
        else:
            chunks.append(np.where(u < keep, np.float32(1.0) / keep, np.float32(0.0)).astype(np.float32))
        off += ch
    return np.concatenate(chunks)


def group_masks(seeds, n, site_channels, site_keep, first_sample=0):
    """The passes of a launch in its layout [site][pass * n + i][C_site], flattened (what rcu_dropout_masks writes)."""
    per_pass = [pass_mask(s, n, site_channels, site_keep, first_sample) for s in seeds]
    chunks, at = [], 0
    for c in site_channels:
        length = n * int(c)
        chunks.extend(m[at:at + length] for m in per_pass)
        at += length
    return np.concatenate(chunks)
