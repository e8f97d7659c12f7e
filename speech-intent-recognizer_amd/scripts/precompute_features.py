"""Drop-in for /root/reference/scripts/precompute_features.py with the feature arithmetic on MI355X.

Same surface: ``AudioFeatureExtractor(sample_rate, n_mels, n_fft, hop_length).extract_features(path,
max_duration)`` -> ``FloatTensor[64, T]`` or ``None``; ``precompute_dataset_features(csv_path,
output_dir, label_map_path, max_duration)`` -> cache path; the same CLI flags; the same
``<csv-stem>_features.pt`` / ``cache_info.json`` formats.  What differs is how the work is done:
files are decoded on the host and pushed through ``sir_features_fwd`` in batches of several hundred
clips (one launch pair per batch) instead of one torchaudio call chain per file
(reference :124-130).  Errors follow the reference's convention: they are logged and the clip is
skipped (``None``), never raised (:42-44, :77-79).
"""
import argparse
import json
import logging
import os

import pandas as pd
import torch

from sir_amd import _native
from sir_amd.featurizer import get_featurizer
from sir_amd.scripts.utils import wav_io

logging.basicConfig(level=logging.INFO, format="%(asctime)s - %(levelname)s - %(message)s")
logger = logging.getLogger(__name__)

DEFAULT_BATCH = 256


class AudioFeatureExtractor:
    """Log-mel features with the reference's parameters (precompute_features.py:21-36)."""

    def __init__(self, sample_rate=16000, n_mels=64, n_fft=1024, hop_length=512):
        self.sample_rate = sample_rate
        self.n_mels = n_mels
        self.n_fft = n_fft
        self.hop_length = hop_length
        self._fz = None

    def _featurizer(self):
        if self._fz is None:
            self._fz = get_featurizer(self.sample_rate, self.n_mels, self.n_fft, self.hop_length)
        return self._fz

    # -- host side: decode, mono, truncate (precompute_features.py:47-61) ------------------------
    def _load(self, audio_path, max_duration):
        if not os.path.exists(audio_path):
            logger.error(f"File not found: {audio_path}")
            return None
        waveform, sr = wav_io.read_wav(audio_path)
        if waveform.shape[0] > 1:
            waveform = torch.mean(waveform, dim=0, keepdim=True)
        if sr != self.sample_rate:
            # the reference resamples with torchaudio (sinc_interp_hann); not built yet on this path
            raise wav_io.WavError(f"sample rate {sr} != {self.sample_rate}: resampling is not available")
        max_samples = int(max_duration * self.sample_rate)
        wave = waveform[0, :max_samples]
        if wave.numel() <= self.n_fft // 2:
            # torch.stft's reflect padding fails here in the reference, which then returns None
            raise wav_io.WavError(f"clip too short for reflect padding ({wave.numel()} samples)")
        return wave.contiguous()

    def extract_batch(self, audio_paths, max_duration=5.0):
        """Features for many files with one GPU pass -> list of ``FloatTensor[64, T]`` / ``None``."""
        waves, slots = [], []
        for i, p in enumerate(audio_paths):
            try:
                w = self._load(p, max_duration)
            except Exception as e:  # swallow-and-continue, as the reference does
                logger.error(f"Error processing {p}: {str(e)}")
                w = None
            if w is not None:
                waves.append(w)
                slots.append(i)
        out = [None] * len(audio_paths)
        if not waves:
            return out
        try:
            feats, frames = self.features_from_waveforms(waves)
        except Exception as e:
            logger.error(f"Error processing batch of {len(waves)} clips: {str(e)}")
            return out
        for k, i in enumerate(slots):
            out[i] = feats[k, :, : frames[k]].clone()
        return out

    def features_from_waveforms(self, waves):
        """list of float32 [L_i] (CPU) -> (features [N, 64, Tmax] on the CPU, frame counts)."""
        fz = self._featurizer()
        lengths = [int(w.numel()) for w in waves]
        lmax = max(lengths)
        host = torch.zeros((len(waves), lmax), dtype=torch.float32).pin_memory()
        for i, w in enumerate(waves):
            host[i, : lengths[i]] = w
        dev = fz.device
        frames = [fz.num_frames(n) for n in lengths]
        feats = fz(host.to(dev, non_blocking=True), torch.tensor(lengths, dtype=torch.int32, device=dev),
                   t_pad=max(frames))
        return feats.cpu(), frames

    def extract_features(self, audio_path, max_duration=5.0):
        """Single-file form of the reference API (precompute_features.py:38-79)."""
        return self.extract_batch([audio_path], max_duration)[0]


def precompute_dataset_features(csv_path, output_dir, label_map_path=None, max_duration=5.0,
                                batch_size=DEFAULT_BATCH):
    """CSV -> ``{path: {'features': FloatTensor[64,T], 'label': str}}`` saved with ``torch.save`` to
    ``<output_dir>/<csv-stem>_features.pt`` (precompute_features.py:81-147)."""
    df = pd.read_csv(csv_path)
    logger.info(f"Loaded {len(df)} samples from {csv_path}")
    extractor = AudioFeatureExtractor()
    os.makedirs(output_dir, exist_ok=True)
    dataset_name = os.path.basename(csv_path).replace(".csv", "")
    cache_file = os.path.join(output_dir, f"{dataset_name}_features.pt")

    # label column fallbacks of precompute_features.py:108-120
    if "label" in df.columns:
        label_column = "label"
    elif "intent" in df.columns:
        label_column = "intent"
    elif "action" in df.columns and "object" in df.columns:
        df["label"] = df["action"] + "_" + df["object"]
        label_column = "label"
    else:
        df["label"] = "unknown"
        label_column = "label"
        logger.warning("Could not find label column, using 'unknown' as label")
    logger.info(f"Using '{label_column}' column for labels")

    features_dict = {}
    error_count = 0
    paths = df["path"].tolist()
    labels = df[label_column].tolist()
    for start in range(0, len(paths), batch_size):
        chunk = paths[start:start + batch_size]
        feats = extractor.extract_batch(chunk, max_duration)
        for p, lab, f in zip(chunk, labels[start:start + batch_size], feats):
            if f is not None:
                features_dict[p] = {"features": f, "label": lab}
            else:
                error_count += 1
    torch.save(features_dict, cache_file)
    logger.info(f"Saved {len(features_dict)} features to {cache_file}")
    logger.info(f"Failed to process {error_count} files")
    return cache_file


def main():
    parser = argparse.ArgumentParser(description="Precompute audio features on MI355X")
    parser.add_argument("--train_csv", type=str, required=True, help="Path to training CSV file")
    parser.add_argument("--valid_csv", type=str, required=True, help="Path to validation CSV file")
    parser.add_argument("--test_csv", type=str, required=True, help="Path to test CSV file")
    parser.add_argument("--output_dir", type=str, default="data/cached_features",
                        help="Output directory for cached features")
    parser.add_argument("--label_map", type=str, default=None, help="Path to label map JSON file")
    args = parser.parse_args()
    _native.require_hip()
    os.makedirs(args.output_dir, exist_ok=True)
    logger.info("Starting feature precomputation...")
    cache_info = {
        "train_features": precompute_dataset_features(args.train_csv, args.output_dir, args.label_map),
        "valid_features": precompute_dataset_features(args.valid_csv, args.output_dir, args.label_map),
        "test_features": precompute_dataset_features(args.test_csv, args.output_dir, args.label_map),
    }
    with open(os.path.join(args.output_dir, "cache_info.json"), "w") as f:
        json.dump(cache_info, f, indent=2)
    logger.info("Feature precomputation complete!")


if __name__ == "__main__":
    main()
