"""How far ahead of the GPU does the host run in a training step?  Enqueue time of K steps (host only) vs their wall time.
usage: python3 devtools/host_overhead.py [steps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from sir_amd import synth, train_ops  # noqa: E402
from sir_amd.featurizer import get_featurizer  # noqa: E402
from sir_amd.models.models import CNNAudioGRU  # noqa: E402
from sir_amd.optim import FusedAdam  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dev = torch.device("cuda", 0)
m = CNNAudioGRU(31)
m.load_state_dict(synth.synth_state_dict(31, seed=0))
m = m.to(dev).train()
opt = FusedAdam(m.parameters(), lr=5e-5, weight_decay=1e-4)
fz = get_featurizer()
wave = (0.1 * torch.randn(256, 48000, device=dev)).clamp_(-1, 1)
lengths = torch.full((256,), 48000, dtype=torch.int32, device=dev)
labels = torch.randint(0, 31, (256,), device=dev)
feats = torch.empty(256, 64, 200, device=dev)


def step():
    x = fz(wave, lengths, t_pad=200, out=feats)
    opt.zero_grad(set_to_none=True)
    loss = train_ops.fused_cross_entropy(m(x), labels)
    loss.backward()
    opt.step()


for _ in range(5):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(K):
    step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"{K} training steps: host enqueue {1e3 * (t1 - t0) / K:.3f} ms/step, wall {1e3 * (t2 - t0) / K:.3f} ms/step")
import cProfile, pstats
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    step()
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
