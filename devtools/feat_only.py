"""Runs only the feature kernel (batch 256, 3 s clips) -- for rocprofv3 passes and quick timings.
usage: python3 devtools/feat_only.py [iters] [i16|f32] [aug]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from sir_amd.featurizer import get_featurizer  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
i16 = len(sys.argv) > 2 and sys.argv[2] == "i16"
aug = len(sys.argv) > 3 and sys.argv[3] == "aug"
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(1)
pool = [(0.1 * torch.randn(256, 48000, generator=g, device=dev)).clamp_(-1, 1) for _ in range(8)]
if i16:
    pool = [(p * 32767).round().to(torch.int16) for p in pool]
lengths = torch.full((256,), 48000, dtype=torch.int32, device=dev)
out = torch.empty(256, 64, 200, device=dev)
fz = get_featurizer()
kw = {}
if aug:
    kw = dict(shift=torch.randint(-4800, 4800, (256,), dtype=torch.int32, device=dev),
              noise_sigma=torch.full((256,), 0.005, device=dev), noise_seed=5)
for i in range(3):
    fz(pool[i % 8], lengths, t_pad=200, out=out, **kw)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(iters):
    fz(pool[i % 8], lengths, t_pad=200, out=out, **kw)
torch.cuda.synchronize()
print(f"feature stage: {(time.perf_counter() - t0) / iters * 1e6:.1f} us per batch of 256 ({'i16' if i16 else 'f32'}{', aug' if aug else ''})")
