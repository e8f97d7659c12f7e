"""CPU checks of the front-end oracle (oracle/resample_ref.py): the torch restatement of torchaudio's
sinc_interp_hann resampler against an independent float64 evaluation, output lengths, mono mix."""
import math

import numpy as np
import pytest
import torch

from oracle import resample_ref as R

RATES = [(22050, 16000), (24000, 16000), (8000, 16000), (44100, 16000), (48000, 16000), (11025, 16000)]


@pytest.mark.parametrize("orig,new", RATES)
def test_resample_matches_float64_evaluation(orig, new):
    g = torch.Generator().manual_seed(orig)
    n = 3000 + orig % 7
    x = 0.3 * torch.randn(n, generator=g) + 0.2 * torch.sin(2 * math.pi * 440.0 * torch.arange(n) / orig)
    y = R.resample(x[None], orig, new)[0].numpy()
    y64 = R.resample_f64(x.numpy(), orig, new)
    assert y.shape == y64.shape == (R.output_length(n, orig, new),)
    # float32 taps and accumulation + torchaudio's float32 phase term: 1e-5 of the signal scale
    assert np.abs(y - y64).max() < 2e-5


def test_kernel_shape_and_dc_gain():
    k, width = R.sinc_resample_kernel(22050, 16000)
    assert k.shape == (320, 1, 2 * width + 441) and k.dtype == torch.float32 and width == 9
    # every polyphase branch passes DC with gain ~1 (the filter is a low-pass normalised by base/orig)
    assert (k.sum(dim=2).squeeze() - 1.0).abs().max() < 2e-3
    # identity when the rates agree
    x = torch.randn(2, 100)
    assert R.resample(x, 16000, 16000) is x


def test_output_length_and_mono():
    assert R.output_length(22050, 22050, 16000) == 16000
    assert R.output_length(22051, 22050, 16000) == 16001
    assert R.output_length(1, 48000, 16000) == 1
    x = torch.tensor([[1.0, 2.0, 3.0], [3.0, 2.0, -1.0]])
    assert torch.equal(R.to_mono(x), torch.tensor([[2.0, 2.0, 1.0]]))
    assert R.to_mono(x[:1]) is not None and R.to_mono(x[:1]).shape == (1, 3)
