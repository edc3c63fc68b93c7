"""numpy replica of csrc/common.h:o2_hash64 -- TEST INFRASTRUCTURE: lets the oracle apply the exact dropout
mask the HIP kernels generate (keep <=> byte >= thr, thr = round(p*256), scale 256/(256-thr))."""
import numpy as np

M32 = np.uint64(0xFFFFFFFF)


def o2_hash64(seed: int, idx: np.ndarray) -> np.ndarray:
    idx = idx.astype(np.uint64)
    seed = np.uint64(seed & 0xFFFFFFFFFFFFFFFF)
    s_lo, s_hi = seed & M32, (seed >> np.uint64(32)) & M32
    lo, hi = idx & M32, (idx >> np.uint64(32)) & M32
    t = (hi ^ s_hi) & M32
    rot = ((t << np.uint64(16)) | (t >> np.uint64(16))) & M32
    h = (lo ^ s_lo ^ rot ^ ((t + (t << np.uint64(3))) & M32)) & M32
    h ^= h >> np.uint64(16)
    h = (h * np.uint64(0x7FEB352D)) & M32
    h ^= h >> np.uint64(15)
    h = (h * np.uint64(0x846CA68B)) & M32
    h ^= h >> np.uint64(16)
    return h


def keep_mask(seed: int, n_elems: int, p: float):
    """mask over a flat element range [0, n_elems): element i uses byte (i & 3) of hash(i >> 2)."""
    thr = int(p * 256.0 + 0.5)
    if thr == 0:
        return np.ones(n_elems, dtype=np.float32), 1.0
    i = np.arange(n_elems, dtype=np.uint64)
    h = o2_hash64(seed, i >> np.uint64(2))
    byte = (h >> ((i & np.uint64(3)) * np.uint64(8))) & np.uint64(0xFF)
    return (byte >= np.uint64(thr)).astype(np.float32), 256.0 / (256.0 - thr)


ATTN_KEY_SALT = 0x85EBCA6B9E3779B9


def attn_keep_mask(seed: int, BH: int, L: int, p: float):
    """[BH, L(query), L(key)] keep mask of the attention-probability dropout (csrc/common.h: o2_attn_*):
    byte (key & 3) of mix(R(row) ^ K(key >> 2)) >= thr with R = o2_hash64(seed, bh*L + q),
    K = o2_hash64(seed ^ SALT, key >> 2), mix(r, k) = x ^ (x >> 16), x = ((r ^ k) * 0x9E3779B1) mod 2^32."""
    thr = int(p * 256.0 + 0.5)
    if thr == 0:
        return np.ones((BH, L, L), dtype=np.float32), 1.0
    R = o2_hash64(seed, np.arange(BH * L, dtype=np.uint64)).reshape(BH, L, 1)
    K = o2_hash64((seed ^ ATTN_KEY_SALT) & 0xFFFFFFFFFFFFFFFF, np.arange((L + 3) // 4, dtype=np.uint64))
    key = np.arange(L, dtype=np.uint64)
    x = ((R ^ K[(key >> np.uint64(2)).astype(np.int64)][None, None, :]) * np.uint64(0x9E3779B1)) & M32
    x ^= x >> np.uint64(16)
    byte = (x >> ((key & np.uint64(3)) * np.uint64(8))[None, None, :]) & np.uint64(0xFF)
    return (byte >= np.uint64(thr)).astype(np.float32), 256.0 / (256.0 - thr)
