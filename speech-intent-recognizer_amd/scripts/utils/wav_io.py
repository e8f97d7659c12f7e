"""Minimal RIFF/WAVE reader and writer (host I/O; stands in for ``torchaudio.load`` /
``torchaudio.save`` at scripts/precompute_features.py:47 and scripts/dataset.py:126, which are
not installable here).  PCM 8/16/24/32-bit and IEEE float32; returns float32 in [-1, 1) with the
torchaudio normalisation (int16 / 32768) or the raw int16 samples for the PCM16 fast path."""
import struct

import numpy as np
import torch


class WavError(ValueError):
    pass


def _chunks(buf):
    pos = 12
    n = len(buf)
    while pos + 8 <= n:
        cid = buf[pos:pos + 4]
        size = struct.unpack_from("<I", buf, pos + 4)[0]
        yield cid, pos + 8, min(size, n - pos - 8)
        pos += 8 + size + (size & 1)


def read_wav(path, prefer_int16=False):
    """-> (samples [channels, frames], sample_rate).  ``prefer_int16`` keeps PCM16 data as int16
    (the GPU kernel dequantises with the same 1/32768) instead of converting on the host."""
    with open(path, "rb") as f:
        buf = f.read()
    if len(buf) < 12 or buf[:4] != b"RIFF" or buf[8:12] != b"WAVE":
        raise WavError(f"{path}: not a RIFF/WAVE file")
    fmt = None
    data = None
    for cid, off, size in _chunks(buf):
        if cid == b"fmt ":
            tag, ch, sr, _, _, bits = struct.unpack_from("<HHIIHH", buf, off)
            if tag == 0xFFFE and size >= 26:                     # WAVE_FORMAT_EXTENSIBLE
                tag = struct.unpack_from("<H", buf, off + 24)[0]
            fmt = (tag, ch, sr, bits)
        elif cid == b"data":
            data = (off, size)
    if fmt is None or data is None:
        raise WavError(f"{path}: missing fmt/data chunk")
    tag, ch, sr, bits = fmt
    off, size = data
    if ch < 1:
        raise WavError(f"{path}: no channels")
    raw = memoryview(buf)[off:off + size]
    if tag == 1 and bits == 16:
        x = np.frombuffer(raw, dtype="<i2", count=(size // 2 // ch) * ch).reshape(-1, ch).T
        if prefer_int16:
            return torch.from_numpy(np.array(x, dtype=np.int16, order="C")), sr
        return torch.from_numpy(x.astype(np.float32) / 32768.0), sr
    if tag == 1 and bits == 8:
        x = np.frombuffer(raw, dtype=np.uint8, count=(size // ch) * ch).reshape(-1, ch).T
        return torch.from_numpy((x.astype(np.float32) - 128.0) / 128.0), sr
    if tag == 1 and bits == 24:
        n = (size // 3 // ch) * ch
        b = np.frombuffer(raw, dtype=np.uint8, count=n * 3).reshape(-1, 3).astype(np.int32)
        v = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
        v = np.where(v >= 1 << 23, v - (1 << 24), v)
        return torch.from_numpy((v.astype(np.float32) / float(1 << 23)).reshape(-1, ch).T.copy()), sr
    if tag == 1 and bits == 32:
        x = np.frombuffer(raw, dtype="<i4", count=(size // 4 // ch) * ch).reshape(-1, ch).T
        return torch.from_numpy(x.astype(np.float32) / float(1 << 31)), sr
    if tag == 3 and bits == 32:
        x = np.frombuffer(raw, dtype="<f4", count=(size // 4 // ch) * ch).reshape(-1, ch).T
        return torch.from_numpy(np.array(x, dtype=np.float32, order="C")), sr
    raise WavError(f"{path}: unsupported WAVE format tag={tag} bits={bits}")


def read_wav_interleaved(path):
    """-> (samples 1-D tensor, frame-major interleaved, int16 for PCM16 files else float32; channels;
    sample_rate).  The GPU front-end (sir_mix_to_mono / sir_resample) dequantises and mixes channels."""
    with open(path, "rb") as f:
        head = f.read(12)
    if len(head) < 12 or head[:4] != b"RIFF" or head[8:12] != b"WAVE":
        raise WavError(f"{path}: not a RIFF/WAVE file")
    x, sr = read_wav(path, prefer_int16=True)                 # [channels, frames]
    ch = x.shape[0]
    if ch == 1:
        return x[0].contiguous(), 1, sr
    return x.t().contiguous().reshape(-1), ch, sr


def write_wav_pcm16(path, samples, sample_rate):
    """samples: float tensor/array [frames] or [channels, frames] in [-1, 1] (or int16)."""
    x = samples.detach().cpu().numpy() if isinstance(samples, torch.Tensor) else np.asarray(samples)
    if x.ndim == 1:
        x = x[None, :]
    if x.dtype != np.int16:
        x = np.clip(np.round(x * 32768.0), -32768, 32767).astype(np.int16)
    ch, n = x.shape
    payload = np.ascontiguousarray(x.T).astype("<i2").tobytes()
    with open(path, "wb") as f:
        f.write(b"RIFF" + struct.pack("<I", 36 + len(payload)) + b"WAVE")
        f.write(b"fmt " + struct.pack("<IHHIIHH", 16, 1, ch, sample_rate, sample_rate * ch * 2, ch * 2, 16))
        f.write(b"data" + struct.pack("<I", len(payload)) + payload)
