"""GPU end-to-end through the reference's own surface: WAV files -> precompute_dataset_features
(cache file format) -> FSCIntentDataset + DataLoader workers + collate_fn -> train() (best_model.pt)
-> evaluate() (report files), plus a 2-rank data-parallel step (gloo, both ranks on the one GPU)."""
import json
import os
import types

import numpy as np
import pandas as pd
import pytest
import torch

from oracle import features_ref, resample_ref
from sir_amd import synth
from sir_amd.scripts.utils import wav_io

pytestmark = pytest.mark.gpu

LABELS = ["activate_lights", "deactivate_lights", "increase_volume", "decrease_volume"]


def _make_corpus(root, n=24):
    os.makedirs(root, exist_ok=True)
    clips = synth.synth_clips(n, 48000, seed=4321)
    rng = np.random.Generator(np.random.PCG64(1))
    rows = []
    for i in range(n):
        length = int(rng.integers(16000, 48000))
        path = os.path.join(root, f"utt{i:03d}.wav")
        if i == 3:                                   # stereo file: channel mean (precompute_features.py:50-51)
            wav_io.write_wav_pcm16(path, torch.stack([clips[i, :length], clips[(i + 1) % n, :length]]), 16000)
        elif i == 5:                                 # shorter than the reflect pad: reference returns None
            wav_io.write_wav_pcm16(path, clips[i, :300], 16000)
        elif i == 7:                                 # other sample rate: resampled on the GPU (precompute_features.py:54-56)
            wav_io.write_wav_pcm16(path, clips[i, :length], 22050)
        elif i == 9:
            path = os.path.join(root, "missing.wav")  # never written
        else:
            wav_io.write_wav_pcm16(path, clips[i, :length], 16000)
        rows.append({"path": path, "label": LABELS[i % 4]})
    return rows


def test_precompute_dataset_train_evaluate(tmp_path):
    from sir_amd.scripts import evaluate as ev
    from sir_amd.scripts import precompute_features as pf
    from sir_amd.scripts import train as tr
    from sir_amd.scripts.dataset import FSCIntentDataset

    rows = _make_corpus(str(tmp_path / "wav"))
    csvs = {}
    for split, sl in (("train", slice(0, 16)), ("valid", slice(16, 20)), ("test", slice(20, 24))):
        p = tmp_path / f"{split}_data.csv"
        pd.DataFrame(rows[sl]).to_csv(p, index=False)
        csvs[split] = str(p)
    lm = tmp_path / "label_map.json"
    lm.write_text(json.dumps({l: i for i, l in enumerate(sorted(LABELS))}))
    cache_dir = str(tmp_path / "cache")

    cache = pf.precompute_dataset_features(csvs["train"], cache_dir)
    for split in ("valid", "test"):
        pf.precompute_dataset_features(csvs[split], cache_dir)
    feats = torch.load(cache)
    assert os.path.basename(cache) == "train_data_features.pt"
    bad = {rows[i]["path"] for i in (5, 9)}
    assert set(feats) == {r["path"] for r in rows[:16]} - bad       # failures are skipped, not raised
    for path, item in feats.items():
        wave, sr = wav_io.read_wav(path)
        ref = features_ref.extract_features_f32(resample_ref.resample(wave.mean(0, keepdim=True), sr, 16000)[0])
        got = item["features"]
        assert got.shape == ref.shape and isinstance(item["label"], str)
        assert ((got - ref).abs() <= 1e-4 * ref.abs().clamp(min=1.0)).all(), path

    ds = FSCIntentDataset(csvs["train"], str(lm), is_training=True, augment_prob=0.7, cache_dir=cache_dir)
    mel, lab = ds[0]
    assert mel.shape == (64, 200) and isinstance(lab, int)
    mel5, _ = ds[5]                      # not in the cache, file too short -> zeros (dataset.py:156-158)
    assert (mel5 == 0).all()

    cfg = {"batch_size": 8, "num_workers": 2, "num_labels": 31, "lr": 1e-3, "weight_decay": 1e-4, "epochs": 2,
           "early_stop_patience": 5, "use_amp": True, "augment_prob": 0.7, "cache_dir": cache_dir,
           "use_feature_cache": True, "save_path": str(tmp_path / "ckpt")}
    args = types.SimpleNamespace(train_csv=csvs["train"], val_csv=csvs["valid"], label_map=str(lm))
    best = tr.train(args, cfg)
    assert 0.0 <= best <= 1.0
    # the DataLoader route (fork-server workers + the pinned staging ring) instead of the HBM feature store
    best_dl = tr.train(args, dict(cfg, hbm_feature_cache=False, epochs=1, save_path=str(tmp_path / "ckpt_dl")))
    assert 0.0 <= best_dl <= 1.0
    ckpt = os.path.join(cfg["save_path"], "best_model.pt")
    if best > 0:
        sd = torch.load(ckpt)
        assert list(sd.keys()) == list(synth.synth_state_dict(31).keys())
    else:                                 # the reference only saves on improvement over 0 (train.py:281)
        from sir_amd.models.models import CNNAudioGRU
        if not os.path.exists(ckpt):
            os.makedirs(cfg["save_path"], exist_ok=True)
            torch.save(CNNAudioGRU(31).state_dict(), ckpt)

    eargs = types.SimpleNamespace(test_csv=csvs["test"], label_map=str(lm), model_path=ckpt)
    acc = ev.evaluate(eargs, cfg)
    assert 0.0 <= acc <= 1.0
    assert os.path.exists(os.path.join(cfg["save_path"], "evaluation_results", "classification_report.txt"))

    # ---- parity THROUGH the entry points (evaluate.py:79-86, train.py:120-155): the oracle on the same checkpoint and the same
    # cached features must give the same predicted indices -- hence identical correct / total counts -- and the same loss
    from oracle import model_ref
    from sir_amd.models.models import CNNAudioGRU
    from torch.utils.data import DataLoader
    sd_ck = {k: v.cpu() for k, v in torch.load(ckpt).items()}
    allp = tmp_path / "all_data.csv"
    pd.DataFrame(rows).to_csv(allp, index=False)            # all 24 clips (two of them fail -> zero features, dataset.py:156-158)
    pf.precompute_dataset_features(str(allp), cache_dir)

    def oracle_on(csv_path, batch):
        ds_ = FSCIntentDataset(csv_path, str(lm), is_training=False, cache_dir=cache_dir)
        items = [ds_[i] for i in range(len(ds_))]
        mel = torch.stack([m for m, _ in items])
        lab = torch.tensor([l for _, l in items])
        with torch.no_grad():
            logits = model_ref.forward(sd_ck, mel)
        losses = [torch.nn.functional.cross_entropy(logits[i:i + batch], lab[i:i + batch]).item() for i in range(0, len(lab), batch)]
        return logits.argmax(1), lab, float(np.mean(losses)), logits

    for csv_path in (csvs["test"], str(allp)):
        pred_o, lab_o, _, logits_o = oracle_on(csv_path, cfg["batch_size"])
        acc_e = ev.evaluate(types.SimpleNamespace(test_csv=csv_path, label_map=str(lm), model_path=ckpt), cfg)
        n = len(lab_o)
        assert round(acc_e * n) == int((pred_o == lab_o).sum()), (csv_path, acc_e)      # identical counts
        model = CNNAudioGRU(31)
        model.load_state_dict(sd_ck)
        model = model.cuda()
        ds_ = FSCIntentDataset(csv_path, str(lm), is_training=False, cache_dir=cache_dir)
        loader = DataLoader(ds_, batch_size=cfg["batch_size"], shuffle=False, num_workers=0, collate_fn=tr.collate_fn)
        preds, labels = ev.predict_loader(model, loader, torch.device("cuda"))
        assert (torch.as_tensor(preds) == pred_o).all() and (torch.as_tensor(labels) == lab_o).all()      # identical indices
        vloader = DataLoader(ds_, batch_size=2 * cfg["batch_size"], shuffle=False, num_workers=0, collate_fn=tr.collate_fn)
        _, _, loss_o, _ = oracle_on(csv_path, 2 * cfg["batch_size"])
        vloss, vacc = tr.validate(model, vloader, torch.nn.CrossEntropyLoss(), torch.device("cuda"))
        assert round(vacc * n) == int((pred_o == lab_o).sum()) and abs(vacc - acc_e) < 1e-12
        assert abs(vloss - loss_o) <= 1e-5, (vloss, loss_o)
        top2 = logits_o.sort(1).values
        print(f"{os.path.basename(csv_path)}: {n} clips, {len(set(pred_o.tolist()))} predicted classes, min top-2 margin "
              f"{(top2[:, -1] - top2[:, -2]).min():.3e}, accuracy {acc_e:.4f}, val loss {vloss:.6f} (oracle {loss_o:.6f})")


def _ddp_worker(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import cases
    from sir_amd import dist_utils, train_ops
    from sir_amd.models.models import CNNAudioGRU
    dist_utils.init_distributed("gloo")               # both ranks share cuda:0; gloo moves the GPU buffers
    torch.cuda.set_device(0)
    sd = synth.synth_state_dict(31, seed=0)
    m = CNNAudioGRU(31)
    m.load_state_dict(sd)
    m = m.cuda().train()
    m.gru.dropout = 0.0
    x = cases.varied_features(16, 200, seed=21)[rank::world].cuda()
    y = synth.synth_labels(16, 31, seed=9)[rank::world].cuda()
    loss = train_ops.fused_cross_entropy(m(x), y)
    loss.backward()
    torch.cuda.synchronize()
    torch.save({n: p.grad.cpu() for n, p in m.named_parameters()}, os.path.join(out_dir, f"g{rank}.pt"))
    dist_utils.shutdown_distributed()


def test_two_rank_gradient_mean(tmp_path):
    """world_size 2: the all-reduced gradients equal the mean of the two shards' gradients computed
    alone (BatchNorm statistics stay per rank, as in the build plan -- no SyncBN in the reference)."""
    import torch.multiprocessing as mp
    import cases
    from sir_amd import train_ops
    from sir_amd.models.models import CNNAudioGRU
    mp.spawn(_ddp_worker, args=(2, 29633, str(tmp_path)), nprocs=2, join=True)
    g0, g1 = torch.load(tmp_path / "g0.pt"), torch.load(tmp_path / "g1.pt")
    for k in g0:
        assert torch.equal(g0[k], g1[k]), k                      # every rank holds the same mean
    sd = synth.synth_state_dict(31, seed=0)
    shard_grads = []
    for r in range(2):
        m = CNNAudioGRU(31)
        m.load_state_dict(sd)
        m = m.cuda().train()
        m.gru.dropout = 0.0
        x = cases.varied_features(16, 200, seed=21)[r::2].cuda()
        y = synth.synth_labels(16, 31, seed=9)[r::2].cuda()
        train_ops.fused_cross_entropy(m(x), y).backward()
        shard_grads.append({n: p.grad.cpu() for n, p in m.named_parameters()})
    for k in g0:
        mean = 0.5 * (shard_grads[0][k] + shard_grads[1][k])
        assert (g0[k] - mean).abs().max() <= 1e-6 * max(1.0, mean.abs().max().item()), k


def _write_splits(tmp_path, rows):
    csvs = {}
    for split, sl in (("train", slice(0, 16)), ("valid", slice(16, 20)), ("test", slice(20, 24))):
        p = tmp_path / f"{split}_data.csv"
        pd.DataFrame(rows[sl]).to_csv(p, index=False)
        csvs[split] = str(p)
    lm = tmp_path / "label_map.json"
    lm.write_text(json.dumps({l: i for i, l in enumerate(sorted(LABELS))}))
    return csvs, str(lm)


def _oracle_features(path):
    wave, sr = wav_io.read_wav(path)
    return features_ref.extract_features_f32(resample_ref.resample(wave.mean(0, keepdim=True), sr, 16000)[0])


def test_dataset_without_cache_serves_real_features_to_workers(tmp_path):
    """with use_cache=False (or a missing cache file) and DataLoader workers the reference computes the
    features in the worker (dataset.py:97-98, :117-158); the drop-in must not train on zeros.  Every uncached clip is
    extracted on the GPU by the constructor (main process); the workers then serve oracle-matching features, and only the
    two clips that genuinely fail (too short, missing file) are the reference's zero spectrogram."""
    from torch.utils.data import DataLoader
    from sir_amd.scripts import train as tr
    from sir_amd.scripts.dataset import FSCIntentDataset
    rows = _make_corpus(str(tmp_path / "wav"))
    csvs, lm = _write_splits(tmp_path, rows)
    ds = FSCIntentDataset(csvs["train"], lm, is_training=False, use_cache=False, cache_dir=str(tmp_path / "nocache"))
    assert ds.missing_paths() == []                                  # everything was extracted up front
    loader = DataLoader(ds, batch_size=4, shuffle=False, num_workers=2, collate_fn=tr.collate_fn)
    got = torch.cat([mel for mel, _ in loader])
    assert got.shape == (16, 64, 200)
    for i in range(16):
        if i in (5, 9):
            assert (got[i] == 0).all(), i
            continue
        ref = features_ref.pad_or_trim(_oracle_features(rows[i]["path"]))
        assert ((got[i] - ref).abs() <= 1e-4 * ref.abs().clamp(min=1.0)).all(), i
        assert got[i].abs().max() > 0.1
    # a partial cache: only the misses are extracted, cached entries are served as they are
    from sir_amd.scripts import precompute_features as pf
    cache_dir = str(tmp_path / "cache")
    cache = pf.precompute_dataset_features(csvs["train"], cache_dir)
    full = torch.load(cache)
    keep = {k: v for j, (k, v) in enumerate(full.items()) if j % 2 == 0}
    torch.save(keep, cache)
    ds2 = FSCIntentDataset(csvs["train"], lm, is_training=False, use_cache=True, cache_dir=cache_dir)
    assert set(ds2.in_memory_cache) == {r["path"] for r in rows[:16]} - set(keep)
    for i in (0, 1, 2, 4):
        assert torch.equal(ds2[i][0], got[i]), i


def test_train_from_waveforms_with_fused_augmentation(tmp_path):
    """BASELINE configs[2] / [4] from the reference's entry point: `fused_features` + `waveform_augment` make train() stage
    the raw split in HBM (WaveformStore) and run train_epoch_waveforms with shift / noise / SpecAugment drawn per batch."""
    from sir_amd.featurizer import get_featurizer
    from sir_amd.scripts import train as tr
    from sir_amd.waveform_store import WaveformStore
    rows = _make_corpus(str(tmp_path / "wav"))
    csvs, lm = _write_splits(tmp_path, rows)
    store = WaveformStore(csvs["train"], lm, "cuda")
    assert len(store) == 16 and store.wave.dtype == torch.float32      # the corpus has a stereo and a 22.05 kHz clip
    host_len = store.host_lengths.tolist()
    assert host_len[9] == 0 and host_len[5] == 300                      # missing file / short clip keep their rows
    fz = get_featurizer()
    feats = fz(store.wave, store.lengths, t_pad=200).cpu()
    for i in range(16):
        if i in (5, 9):
            assert (feats[i] == 0).all(), i                             # the reference's zero spectrogram
            continue
        ref = features_ref.pad_or_trim(_oracle_features(rows[i]["path"]))
        assert ((feats[i] - ref).abs() <= 1e-4 * ref.abs().clamp(min=1.0)).all(), i
    seen = []
    for wave, lens, labels, hl in store.epoch_batches(6, rank=1, world=2, shuffle=True, seed=3, epoch=1):
        assert wave.shape[0] == lens.shape[0] == labels.shape[0] == len(hl) and lens.cpu().tolist() == hl
        seen += hl
    assert len(seen) == 8
    # a PCM16 / 16 kHz / mono-only split stays int16 (lossless, half the bytes)
    mono = [r for i, r in enumerate(rows) if i not in (3, 7)][:8]
    pd.DataFrame(mono).to_csv(tmp_path / "mono.csv", index=False)
    assert WaveformStore(str(tmp_path / "mono.csv"), lm, "cuda").wave.dtype == torch.int16

    cfg = {"batch_size": 8, "num_workers": 0, "num_labels": 31, "lr": 1e-3, "weight_decay": 1e-4, "epochs": 2,
           "early_stop_patience": 5, "augment_prob": 0.7, "use_feature_cache": False, "cache_dir": str(tmp_path / "nocache"),
           "save_path": str(tmp_path / "ckpt"), "fused_features": True, "waveform_augment": True, "seed": 1}
    args = types.SimpleNamespace(train_csv=csvs["train"], val_csv=csvs["valid"], label_map=lm)
    best = tr.train(args, cfg)
    assert 0.0 <= best <= 1.0
    # the DataLoader route (fork-server workers + the pinned staging ring) instead of the HBM feature store
    best_dl = tr.train(args, dict(cfg, hbm_feature_cache=False, epochs=1, save_path=str(tmp_path / "ckpt_dl")))
    assert 0.0 <= best_dl <= 1.0
    aug = tr.make_waveform_augment(cfg, seed=1, epoch=0)
    kw = aug(0, 4, [48000, 30000, 16000, 700])
    assert set(kw) == {"shift", "noise_sigma", "noise_seed", "time_mask", "freq_mask"} and kw["shift"].shape == (4,)
    assert (kw["shift"].abs() <= torch.tensor([4800, 3000, 1600, 70])).all() and (kw["noise_sigma"] <= 0.01).all()
    assert set(tr.make_waveform_augment({"fused_features": True, "augment_prob": 0.0}, seed=1)(0, 2, [48000, 48000])) == set()
