"""Waveform augmentation of /root/reference/scripts/augment.py, fused into the HIP feature kernel.

The reference's ``time_shift`` (:6-28) and ``add_noise`` (:82-96) operate on one ``[1, L]`` CPU
tensor at a time (and are not called from anywhere in the reference).  Here they are parameters of
``sir_features_fwd``: the shift moves the read index and the noise is a counter-based N(0,1) keyed
by (seed, utterance, sample), both applied while the samples are loaded for the FFT, so an
augmented batch costs no extra pass over HBM.  ``pitch_shift`` / ``speed_change`` need libsox and are
out of scope.
"""
import random

import torch

from sir_amd.featurizer import get_featurizer


def draw_time_shift(length, shift_limit=0.1, rng=random):
    """int(U(-limit, limit) * length), as augment.py:18-19."""
    return int(rng.uniform(-shift_limit, shift_limit) * length)


def draw_noise_level(noise_level_range=(0.001, 0.01), rng=random):
    """U(lo, hi), as augment.py:93."""
    return float(rng.uniform(*noise_level_range))


def draw_batch_params(lengths, augment_prob=0.7, rng=random):
    """Per-utterance (shift, sigma) with the reference's gating (augment.py:119-133): the whole
    augmentation with probability ``augment_prob``, then shift and noise each with probability 0.5."""
    shifts, sigmas = [], []
    for n in lengths:
        s, g = 0, 0.0
        if rng.random() < augment_prob:
            if rng.random() < 0.5:
                s = draw_time_shift(int(n), rng=rng)
            if rng.random() < 0.5:
                g = draw_noise_level(rng=rng)
        shifts.append(s)
        sigmas.append(g)
    return torch.tensor(shifts, dtype=torch.int32), torch.tensor(sigmas, dtype=torch.float32)


def time_shift(waveform, shift_limit=0.1):
    """[1, L] -> shifted [1, L] (zero fill), same semantics as augment.py:6-28 (host tensor op,
    kept for API compatibility; the training path uses the fused form below)."""
    length = waveform.shape[1]
    shift = draw_time_shift(length, shift_limit)
    out = torch.zeros_like(waveform)
    if shift > 0:
        out[:, shift:] = waveform[:, : length - shift]
    elif shift < 0:
        out[:, : length + shift] = waveform[:, -shift:]
    else:
        out = waveform.clone()
    return out


def augmented_features(wave, lengths=None, augment_prob=0.7, seed=0, step=0, t_pad=200, rng=random,
                       spec_masks=None):
    """Features of a GPU waveform batch [B, L] with time-shift + noise (and optional SpecAugment
    masks) fused into the feature kernel.  ``seed``/``step`` key the noise stream."""
    bsz, length = wave.shape
    host_lengths = [length] * bsz if lengths is None else [int(v) for v in lengths.tolist()]
    shift, sigma = draw_batch_params(host_lengths, augment_prob, rng)
    tm = fm = None
    if spec_masks is not None:
        tm, fm = spec_masks
    return get_featurizer()(wave, lengths, t_pad=t_pad, shift=shift, noise_sigma=sigma,
                            noise_seed=(int(seed) << 32) ^ int(step), time_mask=tm, freq_mask=fm)
