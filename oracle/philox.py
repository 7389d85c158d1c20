"""numpy restatement of the engine's counter-based noise stream.

TEST INFRASTRUCTURE (see oracle/__init__.py).

The reference draws eps with torch's generator (``new_zeros(...).normal_()``,
model.py:1087, :671-699), which a fused kernel cannot reproduce, so parity of
the layers is tested with eps *injected*.  Production noise is this stream:
Philox4x32-10 (Salmon et al., SC'11; Random123 v1.14 ``philox4x32_R(10,..)``,
checked below against the Random123 known-answer vectors) keyed

    key     = (seed_lo, seed_hi)
    counter = (block_lo, block_hi, stream, step)      block = element_index >> 2

and turned into four N(0,1) values per counter block by two Box-Muller pairs.
Nothing in the key depends on the rank, so backward and every data-parallel
rank regenerate identical eps with no storage.
"""
import numpy as np

M0 = np.uint64(0xD2511F53)
M1 = np.uint64(0xCD9E8D57)
W0 = np.uint32(0x9E3779B9)
W1 = np.uint32(0xBB67AE85)
MASK = np.uint64(0xFFFFFFFF)

# stream ids (must match include/bayeslm.h BLM_STREAM_*)
STREAM_WEIGHT = 0x10000000   # class in the top 4 bits + tensor_id (28 bits)
STREAM_DROPOUT = 0x20000000  # + site_id


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Vectorised over numpy uint32 arrays.  Returns 4 uint32 arrays."""
    c0 = np.asarray(c0, np.uint32).copy()
    c1 = np.asarray(c1, np.uint32).copy()
    c2 = np.asarray(c2, np.uint32).copy()
    c3 = np.asarray(c3, np.uint32).copy()
    k0 = np.uint32(k0)
    k1 = np.uint32(k1)
    with np.errstate(over="ignore"):
        for _ in range(10):
            p0 = M0 * c0.astype(np.uint64)
            p1 = M1 * c2.astype(np.uint64)
            hi0 = (p0 >> np.uint64(32)).astype(np.uint32)
            lo0 = (p0 & MASK).astype(np.uint32)
            hi1 = (p1 >> np.uint64(32)).astype(np.uint32)
            lo1 = (p1 & MASK).astype(np.uint32)
            c0, c1, c2, c3 = hi1 ^ c1 ^ k0, lo1, hi0 ^ c3 ^ k1, lo0
            k0 = np.uint32((int(k0) + int(W0)) & 0xFFFFFFFF)
            k1 = np.uint32((int(k1) + int(W1)) & 0xFFFFFFFF)
    return c0, c1, c2, c3


def _box_muller(ra, rb):
    """u1 = ((ra>>9)+1) * 2^-23 in (0,1];  u2 = (rb>>8) * 2^-24 in [0,1)."""
    u1 = ((ra >> np.uint32(9)).astype(np.float32) + np.float32(1.0)) * np.float32(2.0 ** -23)
    u2 = (rb >> np.uint32(8)).astype(np.float32) * np.float32(2.0 ** -24)
    r = np.sqrt(np.float32(-2.0) * np.log(u1)).astype(np.float32)
    ang = (np.float32(2.0 * np.pi) * u2).astype(np.float32)
    return (r * np.cos(ang)).astype(np.float32), (r * np.sin(ang)).astype(np.float32)


def normal(n, seed, stream, step):
    """First n elements of the N(0,1) stream (row-major element index)."""
    nblk = (n + 3) // 4
    blk = np.arange(nblk, dtype=np.uint64)
    r0, r1, r2, r3 = philox4x32_10((blk & MASK).astype(np.uint32), (blk >> np.uint64(32)).astype(np.uint32),
                                   np.full(nblk, stream, np.uint32), np.full(nblk, step, np.uint32),
                                   seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    z0, z1 = _box_muller(r0, r1)
    z2, z3 = _box_muller(r2, r3)
    return np.stack([z0, z1, z2, z3], axis=1).reshape(-1)[:n]


def keep_mask(n, p, seed, stream, step):
    """Dropout keep mask: element kept iff its 32-bit draw >= floor(p * 2^32)."""
    nblk = (n + 3) // 4
    blk = np.arange(nblk, dtype=np.uint64)
    r = philox4x32_10((blk & MASK).astype(np.uint32), (blk >> np.uint64(32)).astype(np.uint32),
                      np.full(nblk, stream, np.uint32), np.full(nblk, step, np.uint32),
                      seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    thr = np.uint32(min(int(p * 4294967296.0), 0xFFFFFFFF))
    return (np.stack(r, axis=1).reshape(-1)[:n] >= thr)


# Random123 v1.14 kat_vectors, philox4x32 10 rounds: (counter, key) -> output
KAT = [
    ((0x00000000, 0x00000000, 0x00000000, 0x00000000), (0x00000000, 0x00000000),
     (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
    ((0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff), (0xffffffff, 0xffffffff),
     (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
    ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
     (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)),
]
