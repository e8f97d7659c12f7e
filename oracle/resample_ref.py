"""CPU restatement of the sample-rate conversion the reference applies before the feature path
(TEST INFRASTRUCTURE: imported only by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline).

Reference call sites: ``torchaudio.transforms.Resample(sr, 16000)`` at scripts/precompute_features.py:54-56,
scripts/dataset.py:132-135, scripts/test_model.py:68-72, preceded by the channel mean of
precompute_features.py:50-51.  torchaudio is not installed here and is unpinned in requirements.txt:6, so
its published algorithm (``sinc_interp_hann``, ``lowpass_filter_width=6``, ``rolloff=0.99``: a polyphase
windowed-sinc FIR evaluated as a strided conv1d) is restated below with the same torch primitives in the
same order -- including the float32 phase term that an int64 ``arange`` divided by an int produces and the
float64 -> float32 cast of the finished kernel.  PARITY UNPINNED by the reference (it ships no vectors for
this step); ``resample_f64`` is an independent direct evaluation used as a cross-check.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

LOWPASS_FILTER_WIDTH = 6
ROLLOFF = 0.99


def sinc_resample_kernel(orig_freq, new_freq):
    """-> (kernel float32 [new, 1, 2*width + orig], width) for the gcd-reduced rates."""
    g = math.gcd(int(orig_freq), int(new_freq))
    orig, new = int(orig_freq) // g, int(new_freq) // g
    base_freq = min(orig, new) * ROLLOFF
    width = math.ceil(LOWPASS_FILTER_WIDTH * orig / base_freq)
    idx = torch.arange(-width, width + orig, dtype=torch.float64)[None, None] / orig
    t = torch.arange(0, -new, -1)[:, None, None] / new + idx          # int64 / int -> float32, then promoted
    t = t * base_freq
    t = t.clamp(-LOWPASS_FILTER_WIDTH, LOWPASS_FILTER_WIDTH)
    window = torch.cos(t * math.pi / LOWPASS_FILTER_WIDTH / 2) ** 2
    t = t * math.pi
    scale = base_freq / orig
    kernels = torch.where(t == 0, torch.tensor(1.0, dtype=t.dtype), t.sin() / t)
    kernels = kernels * (window * scale)
    return kernels.to(torch.float32), width


def resample(waveform, orig_freq, new_freq):
    """waveform float32 [..., L] -> [..., ceil(new * L / orig)] (identity when the rates are equal)."""
    if int(orig_freq) == int(new_freq):
        return waveform
    g = math.gcd(int(orig_freq), int(new_freq))
    orig, new = int(orig_freq) // g, int(new_freq) // g
    kernel, width = sinc_resample_kernel(orig_freq, new_freq)
    shape = waveform.shape
    x = waveform.reshape(-1, shape[-1])
    n, length = x.shape
    x = F.pad(x, (width, width + orig))
    y = F.conv1d(x[:, None], kernel, stride=orig)
    y = y.transpose(1, 2).reshape(n, -1)
    target = int(math.ceil(new * length / orig))
    return y[..., :target].reshape(shape[:-1] + (target,))


def output_length(length, orig_freq, new_freq):
    g = math.gcd(int(orig_freq), int(new_freq))
    return int(math.ceil((int(new_freq) // g) * length / (int(orig_freq) // g)))


def resample_f64(waveform, orig_freq, new_freq):
    """Independent float64 evaluation of the same filter, one output sample at a time (numpy):
    y[j] = sum_i x[i] * h(j/new - i/orig), h(t) = scale * sinc(base t) * hann(base t / 6), |base t| < 6."""
    g = math.gcd(int(orig_freq), int(new_freq))
    orig, new = int(orig_freq) // g, int(new_freq) // g
    x = np.asarray(waveform, dtype=np.float64)
    base = min(orig, new) * ROLLOFF
    scale = base / orig
    n_out = int(math.ceil(new * len(x) / orig))
    half = LOWPASS_FILTER_WIDTH / base                                 # support half-width in units of 1/orig... seconds*g
    out = np.zeros(n_out)
    for j in range(n_out):
        tj = j / new
        lo = max(0, int(math.floor((tj - half) * orig)) - 1)
        hi = min(len(x) - 1, int(math.ceil((tj + half) * orig)) + 1)
        i = np.arange(lo, hi + 1)
        t = (i / orig - tj) * base
        t = np.clip(t, -LOWPASS_FILTER_WIDTH, LOWPASS_FILTER_WIDTH)
        w = np.cos(t * math.pi / LOWPASS_FILTER_WIDTH / 2) ** 2
        tp = t * math.pi
        s = np.where(tp == 0, 1.0, np.sin(tp) / np.where(tp == 0, 1.0, tp))
        out[j] = np.sum(x[i] * s * w * scale)
    return out


def to_mono(waveform):
    """[channels, L] float32 -> [1, L] (precompute_features.py:50-51)."""
    if waveform.shape[0] > 1:
        return torch.mean(waveform, dim=0, keepdim=True)
    return waveform
