"""Seeded input recipes shared by ``make_golden.py`` and the tests (data, not reference code)."""
import math
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from sir_amd import synth  # noqa: E402

GOLDEN_DIR = os.path.dirname(os.path.abspath(__file__))

# Adam hyper-parameters of the shipped config (configs/config.yaml:14-15)
LR = 5e-5
WEIGHT_DECAY = 1e-4
N_SAMPLES = 64          # sampled elements per tensor stored in the golden file


def feature_cases():
    """name -> float32 waveform [L].  Covers SURVEY 8(c): plain clips, silence, half-silent,
    pure tone, L not multiple of hop, L < n_fft, L > 5 s (truncated)."""
    clips = synth.synth_clips(4, 48000, seed=1234)
    t = torch.arange(48000, dtype=torch.float32)
    tone = 0.5 * torch.sin(2 * math.pi * 1000.0 * t / 16000.0)
    half = clips[2].clone()
    half[24000:] = 0.0
    long = synth.synth_clips(1, 90000, seed=77)[0]
    return {
        "clip0": clips[0],
        "clip1": clips[1],
        "silence": torch.zeros(48000),
        "half_silent": half,
        "tone_1k": tone,
        "tone_noise": (tone + 1e-3 * synth.synth_clips(1, 48000, seed=5)[0]).clamp(-1, 1),
        "len_47999": clips[3][:47999],
        "len_700": clips[3][:700],
        "len_90000": long,
    }


def sample_indices(key, numel, n=N_SAMPLES):
    seed = int.from_bytes(key.encode()[:8].ljust(8, b"\0"), "little") % (2 ** 32)
    rng = np.random.Generator(np.random.PCG64(seed))
    return rng.integers(0, numel, size=min(n, numel))


def varied_features(n, t=200, seed=3):
    """Inputs with per-sample scale / spectral tilt / temporal modulation, so that the
    (centred, sharpened) classifier head of ``sharp_head`` predicts many different classes."""
    rng = np.random.Generator(np.random.PCG64(seed))
    x = rng.standard_normal((n, 64, t)).astype(np.float32)
    a = rng.uniform(0.2, 3.0, (n, 1, 1)).astype(np.float32)
    sl = rng.uniform(-2, 2, (n, 1, 1)).astype(np.float32)
    tilt = np.linspace(1, -1, 64, dtype=np.float32)[None, :, None]
    tm = np.sin(np.linspace(0, 1, t, dtype=np.float32)[None, None, :]
                * rng.uniform(1, 20, (n, 1, 1)).astype(np.float32))
    return torch.from_numpy(a * x + sl * tilt + tm)


HEAD_GAIN = 40.0


def sharp_head(sd, fc_bias):
    """Weight set whose argmax depends on the input: fc.weight * HEAD_GAIN and the stored,
    batch-centring fc.bias from the golden file."""
    sd = dict(sd)
    sd["fc.weight"] = sd["fc.weight"] * HEAD_GAIN
    sd["fc.bias"] = torch.as_tensor(fc_bias).clone()
    return sd


def model_inputs():
    return {
        "x_sharp64": varied_features(64, 200, seed=3),
        "x_eval8": synth.synth_features(8, 200, seed=7),
        "x_eval1_t94": synth.synth_features(1, 94, seed=8).unsqueeze(1),   # un-padded 4-D
        "x_train8": synth.synth_features(8, 200, seed=9),
        "y_train8": synth.synth_labels(8, 31, seed=1235),
    }
