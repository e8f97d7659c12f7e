"""Waveform augmentation of /root/reference/scripts/augment.py, fused into the HIP feature kernel.

The reference's ``time_shift`` (:6-28) and ``add_noise`` (:82-96) operate on one ``[1, L]`` CPU
tensor at a time (and are not called from anywhere in the reference).  Here they are parameters of
``sir_features_fwd``: the shift moves the read index and the noise is a counter-based N(0,1) keyed
by (seed, utterance, sample), both applied while the samples are loaded for the FFT, so an
augmented batch costs no extra pass over HBM.  ``pitch_shift`` / ``speed_change`` need libsox and are
out of scope: calling them raises, and ``apply_augmentation`` skips them
(drawing their random numbers, so the shift / noise draws stay on the reference's RNG stream).

Host forms with the reference's signatures (``time_shift``, ``add_noise``, ``apply_augmentation``,
``apply_spec_augmentation``) are kept for API compatibility; the training path uses ``draw_batch_params`` /
``draw_spec_masks`` + the fused kernel arguments (``augmented_features``, ``scripts.train`` with
``waveform_augment: true``).
"""
import logging
import random

import torch

logger = logging.getLogger(__name__)
_warned = set()


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


def draw_spec_masks(frames, augment_prob=0.5, time_mask_param=20, freq_mask_param=10, n_mels=64, rng=random):
    """Per-utterance SpecAugment bands as ``FSCIntentDataset`` applies them (dataset.py:105-106 gate with
    ``augment_prob``, :160-176: p = 0.5 time mask up to 20 frames, p = 0.5 frequency mask up to 10 mels; torchaudio
    ``mask_along_axis`` draw: v = U*param, s = U*(size - v), band [floor(s), floor(s) + floor(v))), as the
    ``time_mask`` / ``freq_mask`` arguments of the fused kernel: int32 [B, 2] = (start, width), width 0 = none.
    ``frames[b]`` is the un-padded frame count of utterance b (the size the reference masks along)."""
    tm, fm = [], []
    for t in frames:
        t0 = tw = f0 = fw = 0
        if rng.random() < augment_prob:
            if rng.random() < 0.5:
                v = rng.random() * time_mask_param
                s = rng.random() * (int(t) - v)
                t0, tw = max(int(s), 0), int(v)
            if rng.random() < 0.5:
                v = rng.random() * freq_mask_param
                s = rng.random() * (n_mels - v)
                f0, fw = max(int(s), 0), int(v)
        tm.append((t0, tw))
        fm.append((f0, fw))
    return torch.tensor(tm, dtype=torch.int32), torch.tensor(fm, dtype=torch.int32)


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


def add_noise(waveform, noise_level_range=(0.001, 0.01)):
    """``waveform + randn_like(waveform) * U(lo, hi)`` (augment.py:82-96; host tensor op on any device)."""
    noise_level = draw_noise_level(noise_level_range)
    return waveform + torch.randn_like(waveform) * noise_level


def pitch_shift(waveform, sample_rate, pitch_factor_range=(-2.0, 2.0)):
    """augment.py:30-54 runs libsox's ``pitch`` effect; sox is not part of this build."""
    raise NotImplementedError("pitch_shift needs torchaudio.sox_effects (libsox); it is out of scope of the MI355X build")


def speed_change(waveform, sample_rate, speed_factor_range=(0.85, 1.15)):
    """augment.py:56-80 runs libsox's ``tempo`` effect; sox is not part of this build."""
    raise NotImplementedError("speed_change needs torchaudio.sox_effects (libsox); it is out of scope of the MI355X build")


def _skip_sox(name, lo, hi):
    random.uniform(lo, hi)                       # the draw the reference's effect would have made
    if name not in _warned:
        _warned.add(name)
        logger.warning(f"apply_augmentation: {name} needs libsox and is skipped in this build")


def apply_augmentation(waveform, sample_rate, augment_prob=0.7):
    """augment.py:98-135: with probability ``augment_prob``, each of time shift / pitch shift / speed change / noise
    with probability 0.5, in that order and with the same sequence of ``random`` draws.  The two sox effects are
    skipped (see the module docstring); shift and noise are applied exactly as the reference does."""
    if not isinstance(waveform, torch.Tensor):
        waveform = torch.tensor(waveform).float()
        if waveform.dim() == 1:
            waveform = waveform.unsqueeze(0)
    if random.random() < augment_prob:
        if random.random() < 0.5:
            waveform = time_shift(waveform)
        if random.random() < 0.5:
            _skip_sox("pitch_shift", -2.0, 2.0)
        if random.random() < 0.5:
            _skip_sox("speed_change", 0.85, 1.15)
        if random.random() < 0.5:
            waveform = add_noise(waveform)
    return waveform


def apply_spec_augmentation(mel_spec, time_mask_param=20, freq_mask_param=10):
    """augment.py:137-164 on one ``[freq, time]`` spectrogram (host tensor op): p = 0.5 time mask, p = 0.5 frequency mask."""
    from sir_amd.scripts.dataset import mask_along_axis
    if random.random() < 0.5:
        mel_spec = mask_along_axis(mel_spec, time_mask_param, axis=1)
    if random.random() < 0.5:
        mel_spec = mask_along_axis(mel_spec, freq_mask_param, axis=0)
    return mel_spec


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
    from sir_amd.featurizer import get_featurizer
    return get_featurizer()(wave, lengths, t_pad=t_pad, shift=shift, noise_sigma=sigma,
                            noise_seed=(int(seed) << 32) ^ int(step), time_mask=tm, freq_mask=fm)
