"""Host restatements of the two counter-based random streams of the HIP path, so that tests can feed the SAME random
draws to the CPU oracle:

* ``dropout_keep``  -- csrc/train_kernels.h ``dropout_keep(seed, idx, p)``: the keep mask of the inter-layer GRU dropout
  (reference models/models.py:26-33, ``nn.GRU(dropout=0.5)``), a pure function of (seed, element index);
* ``gauss_noise``   -- csrc/features.hip ``gauss_pair(seed, b, i >> 1)``: the N(0,1) sample added to sample i of utterance b
  by the fused ``add_noise`` (reference scripts/augment.py:82-96).

Integer parts are bit-exact (wrap-around arithmetic); the Box-Muller transform uses the device's hardware log2 / sqrt /
sin / cos there and numpy here, so a noise value may differ from the device's by ~1e-6 absolute (times sigma <= 1e-2)."""
import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _u64(x):
    return np.uint64(int(x) & 0xFFFFFFFFFFFFFFFF)


def dropout_keep(seed, n, p):
    """bool [n]: True where element idx is KEPT (then scaled by 1/(1-p))."""
    with np.errstate(over="ignore"):
        idx = np.arange(n, dtype=np.uint64)
        x = _u64(seed) ^ (idx * np.uint64(0x9E3779B97F4A7C15))
        x ^= x >> np.uint64(33)
        x *= np.uint64(0xFF51AFD7ED558CCD)
        x ^= x >> np.uint64(33)
        x *= np.uint64(0xC4CEB9FE1A85EC53)
        x ^= x >> np.uint64(33)
    u = (x >> np.uint64(40)).astype(np.uint32).astype(np.float32) * np.float32(1.0 / 16777216.0)
    return u >= np.float32(p)


def _fmix32(x):
    x = x.astype(np.uint32)
    with np.errstate(over="ignore"):
        x ^= x >> np.uint32(16)
        x *= np.uint32(0x85EBCA6B)
        x ^= x >> np.uint32(13)
        x *= np.uint32(0xC2B2AE35)
        x ^= x >> np.uint32(16)
    return x


def gauss_noise(seed, b, n):
    """float32 [n]: the standard-normal stream of utterance ``b`` (sample indices 0..n-1).  One Box-Muller draw per
    sample PAIR p = i >> 1: sample 2p takes r*cos, sample 2p+1 takes r*sin (features.hip ``gauss_pair``)."""
    seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    lo, hi = np.uint32(seed & 0xFFFFFFFF), np.uint32(seed >> 32)
    npair = (n + 1) // 2
    p = np.arange(npair, dtype=np.uint32)
    with np.errstate(over="ignore"):
        k = _fmix32(lo ^ (p * np.uint32(0x9E3779B1)) ^ np.uint32((int(b) * 0x85EBCA77) & 0xFFFFFFFF))
        a = _fmix32(k ^ hi)
        c = _fmix32(a + np.uint32(0x632BE5AB) + p)
    u1 = ((a >> np.uint32(8)) + np.uint32(1)).astype(np.float32) * np.float32(1.0 / 16777216.0)
    u2 = (c >> np.uint32(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)
    r = np.sqrt(np.float32(-1.38629436111989061883) * np.log2(u1)).astype(np.float32)
    ang = (np.float64(2.0 * np.pi) * u2.astype(np.float64))
    out = np.empty(2 * npair, dtype=np.float32)
    out[0::2] = r * np.cos(ang).astype(np.float32)
    out[1::2] = r * np.sin(ang).astype(np.float32)
    return out[:n]
