"""Batched log-mel features on the GPU (host side of ``sir_features_fwd``).

One call turns a whole batch of waveforms resident in HBM into normalised, padded
``[B, n_mels, t_pad]`` features -- what the reference does one file at a time on the CPU in
``AudioFeatureExtractor.extract_features`` (scripts/precompute_features.py:59-73) followed by the
pad/trim of ``FSCIntentDataset.__getitem__`` (scripts/dataset.py:109-113).
"""
import ctypes as C
import math

import torch

from . import _native

N_FFT = 1024
HOP = 512
MAX_DURATION_S = 5.0


def htk_mel_fbanks(n_freqs, f_min, f_max, n_mels, sample_rate):
    """float32 filterbank [n_freqs, n_mels] with the arithmetic torchaudio's MelScale uses
    (HTK scale, norm=None), so the kernel gets the reference's own table bit for bit."""
    all_freqs = torch.linspace(0, sample_rate // 2, n_freqs)
    m_min = 2595.0 * math.log10(1.0 + f_min / 700.0)
    m_max = 2595.0 * math.log10(1.0 + f_max / 700.0)
    m_pts = torch.linspace(m_min, m_max, n_mels + 2)
    f_pts = 700.0 * (10 ** (m_pts / 2595.0) - 1.0)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)
    down = (-1.0 * slopes[:, :-2]) / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    return torch.clamp(torch.min(down, up), min=0.0).contiguous()


class HipFeaturizer:
    """Owns one ``sir_handle`` on the current HIP device."""

    def __init__(self, sample_rate=16000, n_mels=64, n_fft=N_FFT, hop_length=HOP):
        _native.require_hip()
        self.sample_rate, self.n_mels, self.n_fft, self.hop_length = sample_rate, n_mels, n_fft, hop_length
        self.device = torch.device("cuda", torch.cuda.current_device())
        window = torch.hann_window(n_fft, periodic=True, dtype=torch.float32).contiguous()
        fb = htk_mel_fbanks(n_fft // 2 + 1, 0.0, float(sample_rate // 2), n_mels, sample_rate).float().contiguous()
        cfg = _native.FeatureConfig(sample_rate, n_fft, hop_length, n_mels, 0.0, float(sample_rate // 2),
                                    window.data_ptr(), fb.data_ptr())
        self._h = C.c_void_p()
        _native.check(_native.lib().sir_create(C.byref(cfg), C.byref(self._h)), "sir_create")
        self._ws = None
        self._pins = 0          # library objects (sir_pipeline) created from this handle that are still alive

    def pin(self):
        """Called by whoever creates a longer-lived library object from this handle (``sir_pipeline``, kept for the life of
        the process: sir_amd/pipeline.py): the handle must then outlive it, so ``__del__`` leaves it to process exit."""
        self._pins += 1

    def __del__(self):
        try:
            if getattr(self, "_pins", 0) > 0:
                return
            if getattr(self, "_h", None):
                _native.lib().sir_destroy(self._h)
                self._h = None
        except Exception:
            pass

    @property
    def handle(self):
        return self._h

    def num_frames(self, length):
        return 1 + length // self.hop_length

    def __call__(self, wave, lengths=None, t_pad=200, shift=None, noise_sigma=None, noise_seed=0,
                 time_mask=None, freq_mask=None, out=None, db_out=None):
        """wave: [B, L] float32 or int16 on the GPU; lengths: int32 [B] on the GPU (default: all L).
        ``db_out`` (optional, same shape as the result) receives the un-normalised dB values.
        Returns [B, n_mels, t_pad] float32."""
        _native.require_hip(wave, lengths)
        if wave.dim() != 2 or wave.stride(1) != 1:
            raise _native.SirError("wave must be [B, L] with unit inner stride")
        if wave.dtype == torch.float32:
            dt = _native.WAVE_F32
        elif wave.dtype == torch.int16:
            dt = _native.WAVE_I16
        else:
            raise _native.SirError(f"unsupported waveform dtype {wave.dtype}")
        bsz, max_len = wave.shape
        if lengths is None:
            lengths = torch.full((bsz,), max_len, dtype=torch.int32, device=wave.device)
        lengths = lengths.to(torch.int32).contiguous()
        if out is None:
            out = torch.empty((bsz, self.n_mels, t_pad), dtype=torch.float32, device=wave.device)
        lib = _native.lib()
        need = lib.sir_features_workspace_bytes(self._h, bsz, max_len)
        if self._ws is None or self._ws.numel() < need or self._ws.device != wave.device:
            self._ws = torch.empty(need, dtype=torch.uint8, device=wave.device)
        aug = None
        keep = []
        if any(a is not None for a in (shift, noise_sigma, time_mask, freq_mask)):
            def ptr(t, dtype):
                if t is None:
                    return None
                t = t.to(device=wave.device, dtype=dtype).contiguous()
                keep.append(t)
                return t.data_ptr()
            aug = _native.Augment(ptr(shift, torch.int32), ptr(noise_sigma, torch.float32), int(noise_seed),
                                  ptr(time_mask, torch.int32), ptr(freq_mask, torch.int32))
        rc = lib.sir_features_fwd(self._h, wave.data_ptr(), dt, wave.stride(0), lengths.data_ptr(), bsz, max_len,
                                  out.data_ptr(), t_pad, db_out.data_ptr() if db_out is not None else None,
                                  self._ws.data_ptr(), self._ws.numel(),
                                  C.byref(aug) if aug is not None else None, _native.current_stream_ptr())
        _native.check(rc, "sir_features_fwd")
        return out


    # ---- waveform front-end (channel mix-down, sample-rate conversion) -----------------------------
    def mix_to_mono(self, pcm, channels, frames=None):
        """pcm: [B, frames * channels] interleaved int16 or float32 on the GPU -> [B, frames] float32
        (``torch.mean(waveform, dim=0)`` of precompute_features.py:50-51 after torchaudio.load's /32768)."""
        _native.require_hip(pcm, frames)
        if pcm.dim() != 2 or pcm.stride(1) != 1 or pcm.shape[1] % channels:
            raise _native.SirError("pcm must be [B, frames * channels] with unit inner stride")
        dt = {torch.float32: _native.WAVE_F32, torch.int16: _native.WAVE_I16}.get(pcm.dtype)
        if dt is None:
            raise _native.SirError(f"unsupported sample dtype {pcm.dtype}")
        bsz, max_frames = pcm.shape[0], pcm.shape[1] // channels
        if frames is not None:
            frames = frames.to(torch.int32).contiguous()
        out = torch.empty((bsz, max_frames), dtype=torch.float32, device=pcm.device)
        rc = _native.lib().sir_mix_to_mono(self._h, pcm.data_ptr(), dt, channels, pcm.stride(0),
                                           frames.data_ptr() if frames is not None else None, bsz, max_frames,
                                           out.data_ptr(), out.stride(0), _native.current_stream_ptr())
        _native.check(rc, "sir_mix_to_mono")
        return out

    def resample(self, wave, orig_freq, new_freq=None, lengths=None):
        """``torchaudio.transforms.Resample(orig_freq, new_freq)`` (precompute_features.py:54-56) for a batch:
        wave [B, L] float32/int16 on the GPU, lengths int32 [B] -> ([B, ceil(new * L / orig)] float32, out lengths)."""
        new_freq = self.sample_rate if new_freq is None else new_freq
        _native.require_hip(wave, lengths)
        if wave.dim() != 2 or wave.stride(1) != 1:
            raise _native.SirError("wave must be [B, L] with unit inner stride")
        dt = {torch.float32: _native.WAVE_F32, torch.int16: _native.WAVE_I16}.get(wave.dtype)
        if dt is None:
            raise _native.SirError(f"unsupported waveform dtype {wave.dtype}")
        bsz, max_len = wave.shape
        if lengths is None:
            lengths = torch.full((bsz,), max_len, dtype=torch.int32, device=wave.device)
        lengths = lengths.to(torch.int32).contiguous()
        if int(orig_freq) == int(new_freq):
            return (wave if wave.dtype == torch.float32 else wave.float() / 32768.0), lengths
        lib = _native.lib()
        max_out = lib.sir_resample_out_len(max_len, int(orig_freq), int(new_freq))
        out = torch.empty((bsz, max_out), dtype=torch.float32, device=wave.device)
        out_len = torch.empty((bsz,), dtype=torch.int32, device=wave.device)
        rc = lib.sir_resample(self._h, wave.data_ptr(), dt, wave.stride(0), lengths.data_ptr(), bsz, max_len,
                              int(orig_freq), int(new_freq), out.data_ptr(), out.stride(0), max_out,
                              out_len.data_ptr(), _native.current_stream_ptr())
        _native.check(rc, "sir_resample")
        return out, out_len


_featurizers = {}


def get_featurizer(sample_rate=16000, n_mels=64, n_fft=N_FFT, hop_length=HOP):
    """Per-device cached featurizer."""
    _native.require_hip()
    key = (torch.cuda.current_device(), sample_rate, n_mels, n_fft, hop_length)
    if key not in _featurizers:
        _featurizers[key] = HipFeaturizer(sample_rate, n_mels, n_fft, hop_length)
    return _featurizers[key]
