#!/usr/bin/env python3
"""Training step alone (features -> forward/backward -> Adam at batch 256, dropout 0.5), for A/B runs of environment switches
inside ONE box session: median ms / step over `--repeats` regions and the per-kernel HIP-event averages of `sir_profile_*`.

    SIR_BPTT=1 python devtools/train_only.py --steps 30 --tag bptt1      # prints one JSON line

Developer tool (run through gpurun); the product path only, nothing from oracle/."""
import argparse
import ctypes as C
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--repeats", type=int, default=5)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--tag", default="")
    ap.add_argument("--kernels", default="", help="comma-separated substrings: only these kernel averages are printed")
    args = ap.parse_args()
    import bench
    from sir_amd import _native, ops, synth, train_ops
    from sir_amd.featurizer import get_featurizer
    from sir_amd.models.models import CNNAudioGRU
    from sir_amd.optim import FusedAdam
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    lib = _native.lib()
    model = CNNAudioGRU(bench.NUM_CLASSES)
    model.load_state_dict(synth.synth_state_dict(bench.NUM_CLASSES, seed=0))
    model = model.to(dev).train()
    fz = get_featurizer()
    pool = [bench.device_clips(args.batch, bench.CLIP_LEN, 1234 + i, dev) for i in range(4)]
    lengths = torch.full((args.batch,), bench.CLIP_LEN, dtype=torch.int32, device=dev)
    labels = torch.randint(0, bench.NUM_CLASSES, (args.batch,), device=dev)
    feats = torch.empty(args.batch, 64, bench.T_PAD, device=dev)
    opt = FusedAdam(model.parameters(), lr=5e-5, weight_decay=1e-4)

    def step(i):
        x = fz(pool[i % 4], lengths, t_pad=bench.T_PAD, out=feats)
        opt.zero_grad(set_to_none=True)
        loss = train_ops.fused_cross_entropy(model(x), labels)
        loss.backward()
        opt.step()
        return loss

    for i in range(8):
        step(i)
    torch.cuda.synchronize()
    nk = lib.sir_profile_kernel_count()
    names = [lib.sir_profile_kernel_name(i).decode() for i in range(nk)]
    lib.sir_profile_enable(fz.handle, 1, -1)
    for i in range(5):
        step(i)
        torch.cuda.synchronize()
    ms = (C.c_double * nk)()
    cnt = (C.c_int64 * nk)()
    _native.check(lib.sir_profile_collect(fz.handle, ms, cnt, nk), "sir_profile_collect")
    lib.sir_profile_enable(fz.handle, 0, -1)
    kern = {names[i]: round(ms[i] / cnt[i] * 1e3, 1) for i in range(nk) if cnt[i]}
    if args.kernels:
        keys = args.kernels.split(",")
        kern = {k: v for k, v in kern.items() if any(s in k for s in keys)}
    times = []
    for r in range(args.repeats):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.steps):
            loss = step(i)
        torch.cuda.synchronize()
        times.append((time.perf_counter() - t0) / args.steps * 1e3)
    ops.check_status()
    sw = {k: v for k, v in os.environ.items() if k.startswith("SIR_")}
    print(json.dumps({"tag": args.tag, "env": sw, "ms_per_step_median": round(statistics.median(times), 4),
                      "ms_per_step_min": round(min(times), 4), "ms_per_step_max": round(max(times), 4),
                      "loss": round(float(loss), 6), "kernels_us": kern}), flush=True)


if __name__ == "__main__":
    main()
