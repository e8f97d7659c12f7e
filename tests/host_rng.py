"""Host restatements of the two counter-based random streams of the HIP path, so that tests can feed the SAME random
draws to the CPU oracle:

* ``dropout_keep``  -- csrc/train_kernels.h ``dropout_keep(seed, idx, p)``: the keep mask of the inter-layer GRU dropout
  (reference models/models.py:26-33, ``nn.GRU(dropout=0.5)``), a pure function of (seed, element index);
* ``gauss_noise``   -- csrc/features.hip ``gauss_at(seed, b, i)``: the N(0,1) sample added to sample i of utterance b by the
  fused ``add_noise`` (reference scripts/augment.py:82-96).

Integer parts are bit-exact (uint64 wrap-around arithmetic); the Box-Muller transform is evaluated in float32 like the
device does, so a noise value may differ from the device's by an ulp of logf / cosf (~1e-7 relative)."""
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


def _splitmix64(x):
    with np.errstate(over="ignore"):
        x = x + np.uint64(0x9E3779B97F4A7C15)
        x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return x ^ (x >> np.uint64(31))


def gauss_noise(seed, b, n):
    """float32 [n]: the standard-normal stream of utterance ``b`` (sample indices 0..n-1)."""
    i = np.arange(n, dtype=np.uint64)
    key = (np.uint64(b) << np.uint64(32)) | i
    r = _splitmix64(_u64(seed) ^ _splitmix64(key))
    u1 = ((r >> np.uint64(40)).astype(np.uint32) + np.uint32(1)).astype(np.float32) * np.float32(1.0 / 16777216.0)
    u2 = (r & np.uint64(0xFFFFFF)).astype(np.uint32).astype(np.float32) * np.float32(1.0 / 16777216.0)
    return (np.sqrt(np.float32(-2.0) * np.log(u1)) * np.cos(np.float32(6.28318530717958647692) * u2)).astype(np.float32)
