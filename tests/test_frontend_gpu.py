"""GPU parity of the waveform front-end (sir_mix_to_mono, sir_resample, C ABI via sir_amd.featurizer)
against oracle/resample_ref.py, and of the whole file -> features chain for a WAV that needs both.

Tolerance: the device kernel sums the same float32 taps as the oracle's conv1d but only the in-window
ones and in its own order: |a-b| <= 2e-6 * max|x| per sample (measured ~2e-7); features 1e-4 as elsewhere."""
import numpy as np
import pytest
import torch

from oracle import features_ref, resample_ref
from sir_amd.featurizer import get_featurizer
from sir_amd.scripts.precompute_features import AudioFeatureExtractor
from sir_amd.scripts.utils import wav_io

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _clips(n, length, seed):
    g = torch.Generator().manual_seed(seed)
    t = torch.arange(length, dtype=torch.float32)
    f = 100.0 + 3000.0 * torch.rand(n, 1, generator=g)
    return (0.1 * torch.randn(n, length, generator=g) + 0.4 * torch.sin(2 * torch.pi * f * t / 22050.0)).clamp(-1, 1)


@pytest.mark.parametrize("orig,new", [(22050, 16000), (24000, 16000), (8000, 16000), (44100, 16000), (48000, 16000)])
def test_resample_matches_oracle(orig, new):
    fz = get_featurizer()
    x = _clips(5, 7001, seed=orig)
    lengths = torch.tensor([7001, 7000, 3511, 1, 640], dtype=torch.int32)
    y, out_len = fz.resample(x.to(DEV), orig, new, lengths.to(DEV))
    y, out_len = y.cpu(), out_len.cpu()
    for b in range(5):
        n = int(lengths[b])
        ref = resample_ref.resample(x[b:b + 1, :n], orig, new)[0]
        assert int(out_len[b]) == ref.numel() == resample_ref.output_length(n, orig, new)
        assert (y[b, : ref.numel()] - ref).abs().max() <= 2e-6 * max(1.0, float(x[b, :n].abs().max()))
        assert (y[b, ref.numel():] == 0).all()                      # zero beyond the clip's own output


def test_resample_int16_input_and_identity():
    fz = get_featurizer()
    x = _clips(3, 5000, seed=3)
    xi = torch.round(x * 32767.0).to(torch.int16)
    y, n = fz.resample(xi.to(DEV), 22050, 16000)
    ref = resample_ref.resample(xi.float() / 32768.0, 22050, 16000)
    assert (y.cpu() - ref).abs().max() <= 2e-6
    same, _ = fz.resample(x.to(DEV), 16000, 16000)
    assert torch.equal(same.cpu(), x)


@pytest.mark.parametrize("channels", [1, 2, 5])
def test_mix_to_mono_matches_torch_mean(channels):
    fz = get_featurizer()
    g = torch.Generator().manual_seed(channels)
    frames = torch.tensor([1000, 777, 1], dtype=torch.int32)
    pcm = torch.randint(-32768, 32767, (3, 1000 * channels), generator=g, dtype=torch.int16)
    out = fz.mix_to_mono(pcm.to(DEV), channels, frames.to(DEV)).cpu()
    for b in range(3):
        n = int(frames[b])
        planar = (pcm[b, : n * channels].float() / 32768.0).reshape(n, channels).t()     # what torchaudio.load returns
        ref = resample_ref.to_mono(planar)[0]
        assert (out[b, :n] - ref).abs().max() <= 1e-7
        assert (out[b, n:] == 0).all()
    f32 = torch.randn(2, 300 * channels, generator=g)
    ref = f32.reshape(2, 300, channels).mean(dim=2)
    assert (fz.mix_to_mono(f32.to(DEV), channels).cpu() - ref).abs().max() <= 2e-7


def test_extract_features_of_stereo_22050_wav(tmp_path):
    """File -> features through the drop-in AudioFeatureExtractor for a clip that needs decode, mix-down,
    22.05 -> 16 kHz and truncation; oracle: wav_io + resample_ref + features_ref in the reference's order."""
    sr = 22050
    g = torch.Generator().manual_seed(9)
    stereo = (0.2 * torch.randn(2, int(2.5 * sr), generator=g)).clamp(-1, 1)
    stereo[1] += 0.3 * torch.sin(2 * torch.pi * 700.0 * torch.arange(stereo.shape[1]) / sr)
    path = str(tmp_path / "stereo22k.wav")
    wav_io.write_wav_pcm16(path, stereo, sr)
    long_path = str(tmp_path / "long24k.wav")                        # > 5 s at 24 kHz: exercises the truncation
    wav_io.write_wav_pcm16(long_path, 0.3 * torch.randn(int(6.2 * 24000), generator=g).clamp(-1, 1), 24000)
    ex = AudioFeatureExtractor()
    got = ex.extract_batch([path, long_path, str(tmp_path / "missing.wav")])
    assert got[2] is None
    for p, f in ((path, got[0]), (long_path, got[1])):
        wave, rate = wav_io.read_wav(p)
        wave = resample_ref.resample(resample_ref.to_mono(wave), rate, 16000)[:, :80000]
        ref = features_ref.extract_features_f32(wave[0])
        assert f.shape == ref.shape
        assert (f - ref).abs().max() <= 1e-4 * max(1.0, float(ref.abs().max()))
    assert got[1].shape[1] == 157                                    # 80000 samples -> 157 frames
