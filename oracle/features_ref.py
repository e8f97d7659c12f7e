"""Oracle: waveform -> normalised, padded log-mel features (CPU).

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).  PARITY UNPINNED by the
reference: torchaudio is absent, so this restates the documented semantics of
the torchaudio calls the reference makes.

Follows, step by step:
  /root/reference/scripts/precompute_features.py:21-36  (transform parameters)
  /root/reference/scripts/precompute_features.py:59-73  (truncate, mel, dB, z-norm)
  /root/reference/scripts/dataset.py:109-113            (trim / zero-pad to 200)
cross-checked against the duplicate statement at scripts/dataset.py:137-152.

Two independent implementations:
  * ``*_f32``  : what the reference's CPU path executes -- ``torch.stft`` (the
    very call torchaudio's ``Spectrogram`` makes) in float32, dense fbank matmul.
  * ``*_f64``  : numpy float64 from first principles (np.fft.rfft), used to tell
    which of two float32 answers is closer to the truth.
"""
import math

import numpy as np
import torch

SAMPLE_RATE = 16000
N_FFT = 1024
HOP = 512
N_MELS = 64
N_FREQS = N_FFT // 2 + 1
AMIN = 1e-10
NORM_EPS = 1e-5
MAX_DURATION_S = 5.0
MEL_SPEC_LENGTH = 200


def num_frames(length, hop=HOP):
    """torch.stft(center=True): 1 + L // hop frames."""
    return 1 + length // hop


def hann_window_f32(n_fft=N_FFT):
    # torchaudio.transforms.Spectrogram default window_fn=torch.hann_window (periodic)
    return torch.hann_window(n_fft, periodic=True, dtype=torch.float32)


def mel_fbank_f32(n_freqs=N_FREQS, f_min=0.0, f_max=None, n_mels=N_MELS, sample_rate=SAMPLE_RATE):
    """torchaudio.functional.melscale_fbanks(norm=None, mel_scale='htk') in float32.

    MelSpectrogram defaults: f_min=0, f_max=sample_rate//2, norm=None, mel_scale='htk'.
    Returns [n_freqs, n_mels].
    """
    if f_max is None:
        f_max = float(sample_rate // 2)
    all_freqs = torch.linspace(0, sample_rate // 2, n_freqs)
    m_min = 2595.0 * math.log10(1.0 + (f_min / 700.0))
    m_max = 2595.0 * math.log10(1.0 + (f_max / 700.0))
    m_pts = torch.linspace(m_min, m_max, n_mels + 2)
    f_pts = 700.0 * (10 ** (m_pts / 2595.0) - 1.0)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)
    zero = torch.zeros(1)
    down_slopes = (-1.0 * slopes[:, :-2]) / f_diff[:-1]
    up_slopes = slopes[:, 2:] / f_diff[1:]
    fb = torch.max(zero, torch.min(down_slopes, up_slopes))
    return fb


def power_spectrogram_f32(wave):
    """wave: float32 [L] -> [513, T] power (|stft|^2), torchaudio Spectrogram(power=2)."""
    wave = torch.as_tensor(wave, dtype=torch.float32)
    spec = torch.stft(
        wave, n_fft=N_FFT, hop_length=HOP, win_length=N_FFT, window=hann_window_f32(),
        center=True, pad_mode="reflect", normalized=False, onesided=True, return_complex=True,
    )
    return spec.abs().pow(2.0)


def mel_power_f32(wave, fb=None):
    if fb is None:
        fb = mel_fbank_f32()
    spec = power_spectrogram_f32(wave)                       # [513, T]
    return torch.matmul(spec.transpose(-1, -2), fb).transpose(-1, -2)   # [64, T]


def power_to_db_f32(mel):
    # AmplitudeToDB(): stype='power' -> multiplier 10, amin 1e-10, ref 1.0, top_db None
    db_multiplier = math.log10(max(AMIN, 1.0))
    x_db = 10.0 * torch.log10(torch.clamp(mel, min=AMIN))
    x_db = x_db - 10.0 * db_multiplier
    return x_db


def normalise_f32(x_db):
    # precompute_features.py:73 -- whole-utterance mean, unbiased std, eps added to std
    return (x_db - x_db.mean()) / (x_db.std() + NORM_EPS)


def pad_or_trim(feat, length=MEL_SPEC_LENGTH):
    # dataset.py:109-113
    if feat.shape[1] > length:
        return feat[:, :length]
    if feat.shape[1] < length:
        return torch.nn.functional.pad(feat, (0, length - feat.shape[1]))
    return feat


def truncate(wave, max_duration=MAX_DURATION_S, sample_rate=SAMPLE_RATE):
    max_samples = int(max_duration * sample_rate)
    return wave[:max_samples]


def extract_features_f32(wave, max_duration=MAX_DURATION_S, stages=False):
    """Mono 16 kHz float32 waveform [L] -> normalised [64, T] (un-padded).

    Returns None when the reference would (torch.stft reflect pad needs L > 512;
    the reference swallows the RuntimeError and returns None,
    precompute_features.py:77-79).
    """
    wave = truncate(torch.as_tensor(wave, dtype=torch.float32), max_duration)
    if wave.numel() <= N_FFT // 2:
        return None
    mel = mel_power_f32(wave)
    db = power_to_db_f32(mel)
    norm = normalise_f32(db)
    if stages:
        return {"mel_power": mel, "db": db, "norm": norm}
    return norm


def batch_features_f32(waves, lengths, t_pad=MEL_SPEC_LENGTH):
    """waves [B, Lmax] float32, lengths [B] -> [B, 64, t_pad], one clip at a time
    exactly like precompute_features.py:124-130 followed by dataset.py:109-113."""
    out = []
    for w, n in zip(waves, lengths):
        f = extract_features_f32(w[: int(n)])
        out.append(pad_or_trim(f, t_pad))
    return torch.stack(out)


# ----------------------------------------------------------------------------
# independent float64 implementation (numpy only)
# ----------------------------------------------------------------------------

def mel_fbank_f64(n_freqs=N_FREQS, f_min=0.0, f_max=8000.0, n_mels=N_MELS, sample_rate=SAMPLE_RATE):
    all_freqs = np.linspace(0.0, sample_rate // 2, n_freqs)
    m_min = 2595.0 * np.log10(1.0 + f_min / 700.0)
    m_max = 2595.0 * np.log10(1.0 + f_max / 700.0)
    m_pts = np.linspace(m_min, m_max, n_mels + 2)
    f_pts = 700.0 * (10.0 ** (m_pts / 2595.0) - 1.0)
    fb = np.zeros((n_freqs, n_mels))
    for j in range(n_mels):
        lo, ce, hi = f_pts[j], f_pts[j + 1], f_pts[j + 2]
        up = (all_freqs - lo) / (ce - lo)
        down = (hi - all_freqs) / (hi - ce)
        fb[:, j] = np.maximum(0.0, np.minimum(up, down))
    return fb


def extract_features_f64(wave, stages=False):
    x = np.asarray(wave, dtype=np.float64)[: int(MAX_DURATION_S * SAMPLE_RATE)]
    if x.size <= N_FFT // 2:
        return None
    pad = N_FFT // 2
    xp = np.pad(x, (pad, pad), mode="reflect")
    n = np.arange(N_FFT)
    win = 0.5 - 0.5 * np.cos(2.0 * np.pi * n / N_FFT)
    t = 1 + x.size // HOP
    frames = np.stack([xp[i * HOP: i * HOP + N_FFT] * win for i in range(t)])      # [T, 1024]
    spec = np.abs(np.fft.rfft(frames, axis=1)) ** 2                                 # [T, 513]
    mel = (spec @ mel_fbank_f64()).T                                                # [64, T]
    db = 10.0 * np.log10(np.maximum(mel, AMIN))
    norm = (db - db.mean()) / (db.std(ddof=1) + NORM_EPS)
    if stages:
        return {"mel_power": mel, "db": db, "norm": norm}
    return norm
