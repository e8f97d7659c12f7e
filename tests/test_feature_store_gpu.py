"""GPU: the HBM-resident cached-feature route (sir_amd/feature_store.py, ``sir_gather_features``) against the reference-shaped
route it stands in for -- ``FSCIntentDataset.__getitem__`` + ``collate_fn`` (scripts/dataset.py:78-115, scripts/train.py:49-70):
same files in, bit-identical batches out (gather, pad / trim, SpecAugment bands), shards as ``ShardSampler``."""
import json
import os

import numpy as np
import pandas as pd
import pytest
import torch

from sir_amd import _native, ops
from sir_amd.feature_store import FeatureStore
from sir_amd.scripts import train as tr
from sir_amd.scripts.dataset import FSCIntentDataset

pytestmark = pytest.mark.gpu
LABELS = ["activate_lights", "deactivate_lights", "increase_volume"]


def _make_cache(tmp_path, frames):
    g = torch.Generator().manual_seed(5)
    rows, feats = [], {}
    for i, t in enumerate(frames):
        path = str(tmp_path / f"clip{i:03d}.wav")                       # never opened: every item is in the cache
        rows.append({"path": path, "label": LABELS[i % 3] if i != 4 else "not_in_the_map"})
        feats[path] = {"features": torch.randn(64, t, generator=g), "label": rows[-1]["label"]}
    csv = tmp_path / "train_data.csv"
    pd.DataFrame(rows).to_csv(csv, index=False)
    lm = tmp_path / "label_map.json"
    lm.write_text(json.dumps({l: i + 1 for i, l in enumerate(LABELS)}))   # ids 1..3: the fallback id 0 is distinguishable
    cache_dir = tmp_path / "cache"
    os.makedirs(cache_dir)
    torch.save(feats, cache_dir / "train_data_features.pt")
    return str(csv), str(lm), str(cache_dir)


def test_store_rows_equal_dataset_items(tmp_path):
    frames = [94, 157, 30, 200, 230, 8, 1, 199, 201, 94, 120, 64]
    csv, lm, cache = _make_cache(tmp_path, frames)
    store = FeatureStore(csv, lm, "cuda", cache_dir=cache)
    ds = FSCIntentDataset(csv, lm, is_training=False, cache_dir=cache)
    assert len(store) == len(ds) == len(frames) and store.frames == frames
    for i in range(len(ds)):
        mel, lab = ds[i]
        assert torch.equal(store.store[i].cpu(), mel), i                # pad (zeros) / trim to 200 as dataset.py:109-113
        assert int(store.labels[i]) == lab
    assert int(store.labels[4]) == 0                                    # unknown label -> id 0 (dataset.py:84)


def test_gather_with_bands_is_bit_exact_and_checks_indices(tmp_path):
    frames = [94] * 40
    csv, lm, cache = _make_cache(tmp_path, frames)
    store = FeatureStore(csv, lm, "cuda", cache_dir=cache)
    rng = np.random.Generator(np.random.PCG64(2))
    idx = torch.from_numpy(rng.integers(0, 40, 33)).cuda()
    tm = torch.tensor([[int(rng.integers(0, 90)), int(rng.integers(0, 21))] for _ in range(33)], dtype=torch.int32)
    fm = torch.tensor([[int(rng.integers(0, 60)), int(rng.integers(0, 11))] for _ in range(33)], dtype=torch.int32)
    tm[0] = torch.tensor([0, 0]); fm[1] = torch.tensor([63, 5]); tm[2] = torch.tensor([198, 20])      # none / clipped at the edges
    got = store.gather(idx, tm, fm).cpu()
    ref = store.store.cpu()[idx.cpu()].clone()
    for b in range(33):
        ref[b, :, tm[b, 0]: tm[b, 0] + tm[b, 1]] = 0.0
        ref[b, fm[b, 0]: fm[b, 0] + fm[b, 1], :] = 0.0
    assert torch.equal(got, ref)
    assert torch.equal(store.gather(idx).cpu(), store.store.cpu()[idx.cpu()])
    bad = idx.clone()
    bad[3] = 40
    out = store.gather(bad)
    with pytest.raises(_native.SirError):
        ops.check_status()
    assert (out[3] == 0).all()


def test_epoch_batches_cover_the_split_and_shard(tmp_path):
    frames = [94] * 37
    csv, lm, cache = _make_cache(tmp_path, frames)
    store = FeatureStore(csv, lm, "cuda", cache_dir=cache)
    rows = {tuple(store.store[i, 0, :4].cpu().tolist()): i for i in range(37)}

    def ids(batches):
        out = []
        for mel, lab in batches:
            assert mel.shape[1:] == (64, 200) and mel.is_cuda and lab.dtype == torch.int64
            out += [rows[tuple(r[0, :4].cpu().tolist())] for r in mel]
        return out

    assert sorted(ids(store.epoch_batches(8, shuffle=True, seed=3, epoch=1))) == list(range(37))
    a = ids(store.epoch_batches(8, rank=0, world=2, shuffle=True, seed=3, epoch=1))
    b = ids(store.epoch_batches(8, rank=1, world=2, shuffle=True, seed=3, epoch=1))
    assert len(a) == len(b) == 19 and set(a) | set(b) == set(range(37))          # padded by wrap-around: equal step counts
    v = ids(store.epoch_batches(16, rank=1, world=2, shuffle=False, pad=False))
    assert v == list(range(1, 37, 2))
    # augmentation on: rows differ from the store only by zeroed bands
    for mel, _ in store.epoch_batches(37, shuffle=False, augment_prob=1.0):
        src = store.store
        changed = mel != src
        assert (mel[changed] == 0).all()


def test_train_epoch_over_the_store_equals_the_dataloader_route(tmp_path):
    """One epoch of ``train_epoch`` (train.py:72-118) fed by the store == the same epoch fed by FSCIntentDataset + DataLoader +
    collate_fn in the same order (no augmentation, dropout off): identical batches -> bit-identical weights."""
    from torch.utils.data import DataLoader
    from sir_amd import synth
    from sir_amd.dist_utils import ShardSampler
    from sir_amd.models.models import CNNAudioGRU
    from sir_amd.optim import FusedAdam
    frames = [94, 120, 157, 60] * 8
    csv, lm, cache = _make_cache(tmp_path, frames)
    store = FeatureStore(csv, lm, "cuda", cache_dir=cache)
    ds = FSCIntentDataset(csv, lm, is_training=True, augment_prob=0.0, cache_dir=cache)
    results = []
    for route in ("store", "loader"):
        m = CNNAudioGRU(31)
        m.load_state_dict(synth.synth_state_dict(31, seed=0))
        m = m.cuda()
        m.gru.dropout = 0.0
        opt = FusedAdam(m.parameters(), lr=1e-3, weight_decay=1e-4)
        if route == "store":
            batches = store.epoch_batches(8, shuffle=True, seed=11, epoch=2)
        else:
            sampler = ShardSampler(len(ds), 0, 1, shuffle=True, seed=11)
            sampler.set_epoch(2)
            batches = DataLoader(ds, batch_size=8, sampler=sampler, num_workers=2, collate_fn=tr.collate_fn, pin_memory=True)
        loss = tr.train_epoch(m, batches, opt, torch.nn.CrossEntropyLoss(), torch.device("cuda"))
        results.append((loss, {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}))
    assert results[0][0] == results[1][0]
    for k in results[0][1]:
        assert torch.equal(results[0][1][k], results[1][1][k]), k
