"""Drop-in for /root/reference/scripts/dataset.py::FSCIntentDataset.

Same constructor, ``__len__`` and ``__getitem__(idx) -> (FloatTensor[64, 200], int)``, same cache
file format and the same in-band error conventions (label fallback id 0, zero spectrogram on a
failed clip).  Items are CPU tensors served from the feature cache that
``scripts.precompute_features`` writes with the HIP kernels, so the object is fork/pickle-safe for
``DataLoader`` worker processes and never owns a HIP context itself.

Cache misses (``use_cache=False``, a missing or partial cache file): the reference computes them with torchaudio
inside the worker that hits them (dataset.py:97-98, :117-158).  A forked worker must not touch the HIP context, so
here EVERY path of the CSV that is not in the cache is extracted up front, in the constructor, in the main process,
through the batched GPU path (``AudioFeatureExtractor.extract_batch``, a few hundred clips per launch) into the
in-memory cache the workers inherit.  Only a clip that genuinely fails (missing file, undecodable, too short) becomes
the reference's zero spectrogram.  A miss that still reaches a worker (the file appeared after construction, or the
dataset was built without a GPU) raises instead of silently training on zeros.
"""
import json
import logging
import os

import numpy as np
import pandas as pd
import torch
from torch.utils.data import Dataset, get_worker_info

logger = logging.getLogger(__name__)


class WorkerCacheMiss(RuntimeError):
    """A DataLoader worker was asked for a clip that is neither cached nor pre-extracted: the GPU cannot be used from a
    worker process.  The only exception ``FSCIntentDataset.extract_features`` lets through (everything else is logged and
    becomes a zero spectrogram, dataset.py:156-158)."""


def mask_along_axis(spec, mask_param, axis, mask_value=0.0):
    """Band mask with torchaudio's ``mask_along_axis`` draw (TimeMasking / FrequencyMasking,
    dataset.py:69-71): v = rand*param, s = rand*(size - v), mask [floor(s), floor(s)+floor(v))."""
    size = spec.shape[axis]
    value = torch.rand(1) * mask_param
    min_value = torch.rand(1) * (size - value)
    start = int(min_value.long())
    end = start + int(value.long())
    out = spec.clone()
    if axis == 0:
        out[start:end, :] = mask_value
    else:
        out[:, start:end] = mask_value
    return out


class FSCIntentDataset(Dataset):
    """Fluent Speech Commands items from the feature cache (dataset.py:12-176)."""

    def __init__(self, csv_path, label_map_path, is_training=True, augment_prob=0.5,
                 use_cache=True, cache_dir="data/cached_features", mel_spec_length=200):
        self.data = pd.read_csv(csv_path)
        self.sample_rate = 16000
        self.is_training = is_training
        self.augment_prob = augment_prob if is_training else 0.0
        self.n_mels = 64
        self.mel_spec_length = mel_spec_length
        self.use_cache = use_cache
        with open(label_map_path, "r") as f:
            self.label_map = json.load(f)
        self.in_memory_cache = {}
        self.features_dict = {}
        if use_cache:
            dataset_name = os.path.basename(csv_path).replace(".csv", "")
            self.cache_file = os.path.join(cache_dir, f"{dataset_name}_features.pt")
            if os.path.exists(self.cache_file):
                logger.info(f"Loading cached features from {self.cache_file}")
                self.features_dict = torch.load(self.cache_file)
                logger.info(f"Loaded {len(self.features_dict)} cached features")
            else:
                logger.info(f"No cached features found at {self.cache_file}")
        self.time_mask_param = 20
        self.freq_mask_param = 10
        self._paths = self.data["path"].tolist()
        self._labels = self.data["label"].tolist()
        self.prefetch_missing()
        logger.info(f"Initialized dataset with {len(self.data)} samples, {len(self.label_map)} classes")

    def missing_paths(self):
        """CSV paths (unique, in order) that neither the disk cache nor the in-memory cache holds."""
        seen, out = set(), []
        for p in self._paths:
            if p not in seen and p not in self.features_dict and p not in self.in_memory_cache:
                seen.add(p)
                out.append(p)
        return out

    def prefetch_missing(self, batch_size=256):
        """Main process only: extract every uncached clip on the GPU in batches (what the reference does one clip at a
        time inside its workers).  Failed clips get the reference's zero spectrogram (dataset.py:121-123, :156-158).
        Without a HIP device nothing is done here and a later miss follows ``extract_features``.  Returns the number of
        clips extracted."""
        if get_worker_info() is not None:
            return 0
        missing = self.missing_paths()
        if not missing or not torch.cuda.is_available():
            return 0
        from sir_amd.scripts.precompute_features import AudioFeatureExtractor
        extractor = AudioFeatureExtractor(self.sample_rate, self.n_mels, 1024, 512)
        logger.info(f"{len(missing)} clips are not in the feature cache: extracting them on the GPU")
        failed = 0
        for start in range(0, len(missing), batch_size):
            chunk = missing[start:start + batch_size]
            try:
                feats = extractor.extract_batch(chunk, max_duration=5.0)
            except Exception as e:                      # in-band error convention: log, zeros
                logger.error(f"Error processing a batch of {len(chunk)} clips: {str(e)}")
                feats = [None] * len(chunk)
            for p, f in zip(chunk, feats):
                if f is None:
                    failed += 1
                    f = torch.zeros((self.n_mels, self.mel_spec_length))
                self.in_memory_cache[p] = f
        if failed:
            logger.error(f"{failed} of {len(missing)} uncached clips could not be processed (zero spectrograms)")
        return len(missing)

    def __len__(self):
        return len(self.data)

    def __getitem__(self, idx):
        audio_path = self._paths[idx]
        label_id = self.label_map.get(self._labels[idx], 0)
        if audio_path in self.in_memory_cache:
            mel_spec = self.in_memory_cache[audio_path]
        elif audio_path in self.features_dict:
            mel_spec = self.features_dict[audio_path]["features"]
            self.in_memory_cache[audio_path] = mel_spec
        else:
            mel_spec = self.extract_features(audio_path)
            if mel_spec is not None:
                self.in_memory_cache[audio_path] = mel_spec
        if self.is_training and np.random.random() < self.augment_prob:
            mel_spec = self.augment_features(mel_spec)
        if mel_spec.size(1) > self.mel_spec_length:
            mel_spec = mel_spec[:, : self.mel_spec_length]
        elif mel_spec.size(1) < self.mel_spec_length:
            mel_spec = torch.nn.functional.pad(mel_spec, (0, self.mel_spec_length - mel_spec.size(1)))
        return mel_spec, label_id

    def extract_features(self, audio_path):
        """Cache-miss path (dataset.py:117-158): zeros(64, 200) on any failure, never raises."""
        zeros = torch.zeros((self.n_mels, self.mel_spec_length))
        try:
            if not os.path.exists(audio_path):
                logger.error(f"File not found: {audio_path}")
                return zeros
            if get_worker_info() is not None:
                # (not reachable when the dataset was built on a GPU box: the constructor extracted every miss)
                raise WorkerCacheMiss(f"{audio_path}: not in the feature cache and the GPU cannot be used from a DataLoader "
                                   "worker process; call dataset.prefetch_missing() in the main process, run "
                                   "scripts.precompute_features, or use num_workers=0")
            from sir_amd.scripts.precompute_features import AudioFeatureExtractor
            if not hasattr(self, "_extractor"):
                self._extractor = AudioFeatureExtractor(self.sample_rate, self.n_mels, 1024, 512)
            feat = self._extractor.extract_features(audio_path, max_duration=5.0)
            return feat if feat is not None else zeros
        except WorkerCacheMiss:             # a set-up error of the job, not a bad clip: must not turn into silent zeros
            raise
        except Exception as e:              # everything else (corrupt / undecodable file, HIP error): log + zeros (dataset.py:156-158)
            logger.error(f"Error processing {audio_path}: {str(e)}")
            return zeros

    def augment_features(self, mel_spec):
        """SpecAugment as the reference applies it (dataset.py:160-176): p=0.5 time mask (<=20
        frames), p=0.5 frequency mask (<=10 mels), fill 0."""
        if np.random.random() < 0.5:
            mel_spec = mask_along_axis(mel_spec, self.time_mask_param, axis=1)
        if np.random.random() < 0.5:
            mel_spec = mask_along_axis(mel_spec, self.freq_mask_param, axis=0)
        return mel_spec

    def __getstate__(self):
        state = self.__dict__.copy()
        state.pop("_extractor", None)       # holds a HIP handle: never crosses a process boundary
        return state
