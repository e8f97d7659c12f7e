"""A split's CACHED FEATURES resident in HBM (the cached-feature training route without worker processes or PCIe).

The reference keeps the feature cache in host RAM and serves it item by item through ``DataLoader`` worker processes
(scripts/dataset.py:44-56, :78-115), ``collate_fn`` (scripts/train.py:49-70) and a pinned host -> device copy per step
(scripts/train.py:86): 13 MB per batch of 256, i.e. 5 GB/s of Python-assembled batches at the rate the training step
consumes them on one MI355X -- more than eight worker processes deliver (bench.py ``dropin_epoch``).  The whole cache of
a split is small against 288 GB of HBM (Fluent Speech Commands: 23 k clips x 64 x 200 x 4 B = 1.2 GB), so this store
stages it ONCE, already padded to ``mel_spec_length`` frames, and a step's batch is one ``sir_gather_features`` launch:
gather by index + the SpecAugment bands of ``FSCIntentDataset.augment_features`` zeroed on the way.

Same inputs as the reference's route (the CSV, ``label_map.json``, ``<cache_dir>/<csv-stem>_features.pt``; clips missing
from the cache are extracted on the GPU by ``FSCIntentDataset``'s constructor), same item semantics (label fallback id 0,
zero spectrogram for a failed clip, augmentation gate ``augment_prob``, bands drawn as torchaudio's ``mask_along_axis``
along the UN-padded frame count); the RNG stream differs, as it does between any two runs of the reference (it seeds nothing).
"""
import logging
import random

import torch

from . import _native
from .dist_utils import ShardSampler
from .featurizer import get_featurizer

logger = logging.getLogger(__name__)


class FeatureStore:
    """``len(store)`` items of one CSV split on ``device``; ``epoch_batches`` yields ``(mel [B, 64, T] float32, label int64 [B])``
    device tensors in ``ShardSampler`` order (rank r takes i = r mod world of the shuffled epoch)."""

    def __init__(self, csv_path, label_map_path, device, use_cache=True, cache_dir="data/cached_features",
                 mel_spec_length=200, stage_items=4096):
        _native.require_hip()
        from .scripts.dataset import FSCIntentDataset
        if mel_spec_length % 4 != 0:
            raise ValueError("mel_spec_length must be a multiple of 4 (16-byte rows)")
        self.device = torch.device(device)
        self.t_pad = int(mel_spec_length)
        ds = FSCIntentDataset(csv_path, label_map_path, is_training=False, use_cache=use_cache, cache_dir=cache_dir,
                              mel_spec_length=mel_spec_length)
        n = len(ds)
        self.n_mels = ds.n_mels
        self.store = torch.zeros((n, self.n_mels, self.t_pad), dtype=torch.float32, device=self.device)
        frames, labels = [], []
        for start in range(0, n, stage_items):
            stop = min(start + stage_items, n)
            chunk = torch.zeros((stop - start, self.n_mels, self.t_pad), dtype=torch.float32).pin_memory()
            for i in range(start, stop):
                path = ds._paths[i]
                if path in ds.in_memory_cache:
                    mel = ds.in_memory_cache[path]
                elif path in ds.features_dict:
                    mel = ds.features_dict[path]["features"]
                else:
                    mel = ds.extract_features(path)
                t = min(int(mel.shape[1]), self.t_pad)
                chunk[i - start, :, :t] = mel[:, :t]
                frames.append(int(mel.shape[1]))
                labels.append(ds.label_map.get(ds._labels[i], 0))
            self.store[start:stop].copy_(chunk, non_blocking=True)
        torch.cuda.synchronize(self.device)
        self.frames = frames                          # un-padded frame counts (host): the size the time mask is drawn along
        self.labels = torch.tensor(labels, dtype=torch.int64, device=self.device)
        logger.info(f"staged {n} cached feature items as float32 [{n}, {self.n_mels}, {self.t_pad}] on {self.device} "
                    f"({self.store.numel() * 4 / 2 ** 20:.1f} MiB)")

    @classmethod
    def from_tensors(cls, store, frames, labels):
        """A store over features that already sit on the device (bench.py: synthetic clips featurised on the GPU)."""
        self = cls.__new__(cls)
        _native.require_hip(store, labels)
        self.device = store.device
        self.store = store.contiguous()
        self.n_mels, self.t_pad = int(store.shape[1]), int(store.shape[2])
        self.frames = [int(f) for f in frames]
        self.labels = labels.to(torch.int64)
        return self

    def __len__(self):
        return int(self.store.shape[0])

    def gather(self, index, time_mask=None, freq_mask=None, out=None):
        """``out[b] = store[index[b]]`` with the given bands zeroed (``sir_gather_features``); ``index`` int64 on the device,
        masks int32 [B, 2] (host or device) or None."""
        lib = _native.lib()
        bsz = int(index.numel())
        if out is None:
            out = torch.empty((bsz, self.n_mels, self.t_pad), dtype=torch.float32, device=self.device)
        keep = []

        def ptr(t):
            if t is None:
                return None
            t = t.to(device=self.device, dtype=torch.int32, non_blocking=True).contiguous()
            keep.append(t)
            return t.data_ptr()

        index = index.to(device=self.device, dtype=torch.int64).contiguous()
        rc = lib.sir_gather_features(get_featurizer().handle, self.store.data_ptr(), len(self), index.data_ptr(), bsz, self.n_mels,
                                     self.t_pad, ptr(time_mask), ptr(freq_mask), out.data_ptr(), _native.current_stream_ptr())
        _native.check(rc, "sir_gather_features")
        return out

    def epoch_batches(self, batch_size, rank=0, world=1, shuffle=True, seed=0, epoch=0, augment_prob=0.0, rng=None, pad=True):
        """One epoch of this rank's shard.  ``augment_prob`` > 0: SpecAugment as ``FSCIntentDataset.__getitem__`` applies it
        (dataset.py:105-106, :160-176), drawn on the host for the whole batch, applied inside the gather launch."""
        from .scripts import augment as aug
        sampler = ShardSampler(len(self), rank, world, shuffle=shuffle, seed=seed, pad=pad)
        sampler.set_epoch(epoch)
        host_order = list(sampler)
        order = torch.tensor(host_order, dtype=torch.int64).to(self.device)
        rng = rng or random.Random((int(seed) << 20) ^ (int(epoch) << 4) ^ int(rank))
        for start in range(0, len(host_order), batch_size):
            idx = order[start:start + batch_size]
            tm = fm = None
            if augment_prob > 0.0:
                tm, fm = aug.draw_spec_masks([self.frames[i] for i in host_order[start:start + batch_size]], augment_prob,
                                             n_mels=self.n_mels, rng=rng)
            yield self.gather(idx, tm, fm), self.labels.index_select(0, idx)
