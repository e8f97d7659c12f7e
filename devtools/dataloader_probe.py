#!/usr/bin/env python3
"""Why is train_epoch() over FSCIntentDataset + DataLoader(batch 256, num_workers 8, pin_memory) two orders of magnitude below the step
rate (bench.py `dropin_epoch.dataloader`)?  One synthetic cache, several loader configurations x several consumers; prints, per
combination, utterances/s and the share of the time the consumer spent waiting for the loader.  Developer tool (run through gpurun)."""
import json
import os
import shutil
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def main():
    import pandas as pd
    from torch.utils.data import DataLoader
    import bench
    from sir_amd import synth, train_ops
    from sir_amd.models.models import CNNAudioGRU
    from sir_amd.optim import FusedAdam
    from sir_amd.scripts import train as tr
    from sir_amd.scripts.dataset import FSCIntentDataset
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    if os.environ.get("SIR_PROBE_LIMIT_THREADS", "1") == "1":
        from sir_amd.dist_utils import limit_host_threads
        print(json.dumps({"torch_threads_before": torch.get_num_threads(), "after": limit_host_threads(reserve=8)}), flush=True)
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    tmp = tempfile.mkdtemp(prefix="sir_probe_")
    try:
        feats = torch.randn(n, 64, 94)
        names = [f"intent_{i:02d}" for i in range(31)]
        labels = torch.randint(0, 31, (n,))
        paths = [os.path.join(tmp, f"c{i}.wav") for i in range(n)]
        csv, lm, cache = os.path.join(tmp, "train_data.csv"), os.path.join(tmp, "lm.json"), os.path.join(tmp, "cache")
        os.makedirs(cache)
        pd.DataFrame({"path": paths, "label": [names[int(v)] for v in labels]}).to_csv(csv, index=False)
        json.dump({k: i for i, k in enumerate(names)}, open(lm, "w"))
        torch.save({p: {"features": feats[i], "label": names[int(labels[i])]} for i, p in enumerate(paths)}, os.path.join(cache, "train_data_features.pt"))
        ds = FSCIntentDataset(csv, lm, is_training=True, augment_prob=0.7, cache_dir=cache)
        model = CNNAudioGRU(31)
        model.load_state_dict(synth.synth_state_dict(31, seed=0))
        model = model.to(dev).train()
        opt = FusedAdam(model.parameters(), lr=5e-5, weight_decay=1e-4)
        crit = torch.nn.CrossEntropyLoss()
        big = torch.randn(4096, 4096, device=dev)

        def consume_none(mel, lab):
            mel.to(dev, non_blocking=True)

        def consume_train(mel, lab):
            mel, lab = mel.to(dev, non_blocking=True), lab.to(dev, non_blocking=True)
            opt.zero_grad(set_to_none=True)
            loss = train_ops.fused_cross_entropy(model(mel), lab)
            loss.backward()
            opt.step()

        def consume_gpu_only(mel, lab):          # one ~2 ms GPU kernel per batch, launched with a single Python call
            mel.to(dev, non_blocking=True)
            torch.mm(big, big)

        def consume_cpu_busy(mel, lab):          # 2 ms of pure-Python work on the main thread (holds the GIL), no GPU
            mel.to(dev, non_blocking=True)
            t = time.perf_counter()
            while time.perf_counter() - t < 0.002:
                pass

        def consume_sleep(mel, lab):             # 2 ms asleep (GIL released)
            mel.to(dev, non_blocking=True)
            time.sleep(0.002)

        # HostStager taken apart: where do the milliseconds of a staged step go?
        stage_t = {"sync": 0.0, "cpu_copy": 0.0, "h2d_call": 0.0, "step": 0.0, "n": 0}
        ring = [torch.empty(256, 64, 200).pin_memory() for _ in range(3)]
        evs = [torch.cuda.Event() for _ in range(3)]
        side = torch.cuda.Stream()

        def make_staged(train, use_side):
            def fn(mel, lab):
                i = stage_t["n"] % 3
                stage_t["n"] += 1
                t0 = time.perf_counter()
                evs[i].synchronize()
                t1 = time.perf_counter()
                if mel.shape == ring[i].shape:
                    ring[i].copy_(mel)
                    src = ring[i]
                else:
                    src = mel
                t2 = time.perf_counter()
                if use_side:
                    with torch.cuda.stream(side):
                        x = src.to(dev, non_blocking=True)
                        evs[i].record()
                    torch.cuda.current_stream().wait_stream(side)
                else:
                    x = src.to(dev, non_blocking=True)
                    evs[i].record()
                t3 = time.perf_counter()
                if train:
                    lab_d = lab.to(dev)
                    opt.zero_grad(set_to_none=True)
                    loss = train_ops.fused_cross_entropy(model(x), lab_d)
                    loss.backward()
                    opt.step()
                t4 = time.perf_counter()
                stage_t["sync"] += t1 - t0; stage_t["cpu_copy"] += t2 - t1; stage_t["h2d_call"] += t3 - t2; stage_t["step"] += t4 - t3
            return fn

        consumers = [("staged_only", make_staged(False, False)), ("staged_train", make_staged(True, False)),
                     ("staged_train_side_stream", make_staged(True, True)),
                     ("none", consume_none), ("train_step", consume_train), ("gpu_kernel_only", consume_gpu_only),
                     ("python_busy_2ms", consume_cpu_busy), ("sleep_2ms", consume_sleep)]
        configs = [dict(num_workers=8, pin_memory=False), dict(num_workers=0, pin_memory=False), dict(num_workers=8, pin_memory=True)]
        if os.environ.get("SIR_PROBE_SPAWN", "0") == "1":     # workers that do NOT inherit the parent's HIP / KFD state (fresh interpreters)
            configs = [dict(num_workers=8, pin_memory=False, multiprocessing_context="spawn"),
                       dict(num_workers=8, pin_memory=False, multiprocessing_context="forkserver")]
        if len(sys.argv) > 2:
            consumers = [c for c in consumers if c[0] in sys.argv[2].split(",")]
        for cfg in configs:
            for cname, fn in consumers:
                loader = DataLoader(ds, batch_size=256, shuffle=True, collate_fn=tr.collate_fn, **cfg)
                it = iter(loader)
                wait = 0.0
                count = 0
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                while True:
                    tw = time.perf_counter()
                    try:
                        mel, lab = next(it)
                    except StopIteration:
                        break
                    wait += time.perf_counter() - tw
                    fn(mel, lab)
                    count += mel.size(0)
                torch.cuda.synchronize()
                el = time.perf_counter() - t0
                rec = {"loader": cfg, "consumer": cname, "utts_per_s": round(count / el, 1), "epoch_s": round(el, 3),
                       "waiting_share": round(wait / el, 3)}
                if cname.startswith("staged") and stage_t["n"]:
                    rec["ms_per_batch"] = {k: round(v / stage_t["n"] * 1e3, 3) for k, v in stage_t.items() if k != "n"}
                    for k in stage_t:
                        stage_t[k] = 0 if k == "n" else 0.0
                print(json.dumps(rec), flush=True)
                del it, loader
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
