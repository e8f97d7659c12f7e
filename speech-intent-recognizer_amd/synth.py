"""Deterministic synthetic inputs and weights.

No dataset or checkpoint ships with the reference (data/ is git-ignored,
checkpoints/best_model.pt is a missing blob), so tests and benchmarks use:

* clips  : ``0.1*randn(L) + A*sin(2*pi*f*n/16000)`` clamped to [-1, 1]
* weights: a numpy-PCG64 recipe that mimics PyTorch's default init bounds of
  ``models/models.py:10-39`` (kaiming-uniform convs, U(-1/sqrt(H), 1/sqrt(H)) GRU,
  Linear) but is independent of torch's RNG stream, so the same bits are
  regenerated on any machine.  BN affine/buffers are randomised (not 1/0) so
  that BN folding is actually exercised.
"""
import math
from collections import OrderedDict

import numpy as np
import torch

SAMPLE_RATE = 16000


def synth_clips(n, length=48000, seed=1234, dtype=torch.float32):
    """n seeded clips [n, length] float32 in [-1, 1] (CPU)."""
    g = torch.Generator().manual_seed(seed)
    noise = 0.1 * torch.randn(n, length, generator=g)
    f = 100.0 + 3900.0 * torch.rand(n, 1, generator=g)
    a = 0.05 + 0.45 * torch.rand(n, 1, generator=g)
    t = torch.arange(length, dtype=torch.float32).unsqueeze(0)
    x = noise + a * torch.sin(2.0 * math.pi * f * t / SAMPLE_RATE)
    return x.clamp_(-1.0, 1.0).to(dtype)


def synth_labels(n, num_classes=31, seed=1235):
    g = torch.Generator().manual_seed(seed)
    return torch.randint(0, num_classes, (n,), generator=g, dtype=torch.int64)


def to_int16(x):
    return torch.round(x * 32767.0).clamp_(-32768, 32767).to(torch.int16)


def _uniform(rng, shape, bound):
    return torch.from_numpy(rng.uniform(-bound, bound, size=shape).astype(np.float32))


def synth_state_dict(num_classes=31, seed=0, hidden=256, n_mels=64):
    """State dict with the reference's 38 keys (models/models.py:10-39)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    sd = OrderedDict()
    chans = [(1, 32), (32, 64), (64, 128)]
    for i, (ci, co) in enumerate(chans, start=1):
        sd[f"conv{i}.weight"] = _uniform(rng, (co, ci, 3, 3), 1.0 / math.sqrt(ci * 9))
        sd[f"bn{i}.weight"] = torch.from_numpy(rng.uniform(0.5, 1.5, co).astype(np.float32))
        sd[f"bn{i}.bias"] = _uniform(rng, (co,), 0.2)
        sd[f"bn{i}.running_mean"] = _uniform(rng, (co,), 0.1)
        sd[f"bn{i}.running_var"] = torch.from_numpy(rng.uniform(0.5, 1.5, co).astype(np.float32))
        sd[f"bn{i}.num_batches_tracked"] = torch.zeros((), dtype=torch.int64)
    gru_in = 128 * (n_mels // 8)
    k = 1.0 / math.sqrt(hidden)
    for layer, in_sz in ((0, gru_in), (1, 2 * hidden)):
        for suf in ("", "_reverse"):
            sd[f"gru.weight_ih_l{layer}{suf}"] = _uniform(rng, (3 * hidden, in_sz), k)
            sd[f"gru.weight_hh_l{layer}{suf}"] = _uniform(rng, (3 * hidden, hidden), k)
            sd[f"gru.bias_ih_l{layer}{suf}"] = _uniform(rng, (3 * hidden,), k)
            sd[f"gru.bias_hh_l{layer}{suf}"] = _uniform(rng, (3 * hidden,), k)
    sd["attention.weight"] = _uniform(rng, (1, 2 * hidden), 1.0 / math.sqrt(2 * hidden))
    sd["attention.bias"] = _uniform(rng, (1,), 1.0 / math.sqrt(2 * hidden))
    sd["fc.weight"] = _uniform(rng, (num_classes, 2 * hidden), 1.0 / math.sqrt(2 * hidden))
    sd["fc.bias"] = _uniform(rng, (num_classes,), 1.0 / math.sqrt(2 * hidden))
    return sd


def synth_features(n, t=200, seed=7, n_mels=64):
    """Seeded stand-in for normalised log-mel features [n, n_mels, t] (unit-variance noise
    with a smooth spectral tilt), for model-only tests that do not run the feature stage."""
    rng = np.random.Generator(np.random.PCG64(seed))
    x = rng.standard_normal((n, n_mels, t)).astype(np.float32)
    tilt = np.linspace(0.8, -0.8, n_mels, dtype=np.float32)[None, :, None]
    return torch.from_numpy(x + tilt)
