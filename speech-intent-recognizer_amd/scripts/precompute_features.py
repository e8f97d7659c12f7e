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

    # -- host side: decode only (precompute_features.py:47); mono / resample / truncate run on the GPU ----
    def _load(self, audio_path, max_duration):
        """-> (interleaved samples 1-D int16|float32, channels, sample_rate), cut to what max_duration needs."""
        if not os.path.exists(audio_path):
            logger.error(f"File not found: {audio_path}")
            return None
        data, ch, sr = wav_io.read_wav_interleaved(audio_path)
        keep = int(max_duration * sr) + 256                      # + the resampling filter's support
        if data.numel() > keep * ch:
            data = data[: keep * ch]
        return data, ch, sr

    def extract_batch(self, audio_paths, max_duration=5.0):
        """Features for many files -> list of ``FloatTensor[64, T]`` / ``None``.  Clips are grouped by
        (channels, sample rate, sample type); each group is one mix-down launch (if multi-channel), one
        resampling launch (if the rate is not ``self.sample_rate``) and one feature launch pair."""
        out = [None] * len(audio_paths)
        groups = {}
        for i, p in enumerate(audio_paths):
            try:
                item = self._load(p, max_duration)
            except Exception as e:  # swallow-and-continue, as the reference does (:77-79)
                logger.error(f"Error processing {p}: {str(e)}")
                item = None
            if item is not None:
                data, ch, sr = item
                groups.setdefault((ch, sr, data.dtype), []).append((i, data))
        for (ch, sr, _), items in groups.items():
            try:
                feats, frames, ok = self._features_of_group([d for _, d in items], ch, sr, max_duration)
            except Exception as e:
                logger.error(f"Error processing batch of {len(items)} clips: {str(e)}")
                continue
            for k, (i, _) in enumerate(items):
                if ok[k]:
                    out[i] = feats[k, :, : frames[k]].clone()
                else:
                    # torch.stft's reflect padding fails on such a clip in the reference, which then returns None
                    logger.error(f"Error processing {audio_paths[i]}: clip too short for reflect padding")
        return out

    def waveforms_of_group(self, datas, channels, sr, max_duration):
        """Decoded clips of one (channels, rate, sample type) group -> (wave [N, L] on the GPU, lengths int32 [N] on the
        GPU): mono mix-down, resampling to ``self.sample_rate`` and truncation to ``max_duration`` as
        precompute_features.py:50-61 does per file.  PCM16 mono clips already at the target rate stay int16 (the
        feature kernel dequantises with torchaudio.load's 1/32768)."""
        fz = self._featurizer()
        dev = fz.device
        nfr = [int(d.numel()) // channels for d in datas]
        host = torch.zeros((len(datas), max(max(nfr), 1) * channels), dtype=datas[0].dtype).pin_memory()
        for k, d in enumerate(datas):
            host[k, : nfr[k] * channels] = d[: nfr[k] * channels]
        wave = host.to(dev, non_blocking=True)
        lens = torch.tensor(nfr, dtype=torch.int32, device=dev)
        if channels > 1:
            wave = fz.mix_to_mono(wave, channels, lens)                       # precompute_features.py:50-51
        if sr != self.sample_rate:
            wave, lens = fz.resample(wave, sr, self.sample_rate, lens)       # :54-56
        max_samples = int(max_duration * self.sample_rate)
        lens = torch.clamp(lens, max=max_samples)                             # :59-61
        wave = wave[:, :max_samples]
        if wave.stride(1) != 1 or (wave.stride(0) * wave.element_size()) % 16:
            wave = wave.contiguous()
        return wave, lens

    def _features_of_group(self, datas, channels, sr, max_duration):
        fz = self._featurizer()
        wave, lens = self.waveforms_of_group(datas, channels, sr, max_duration)
        host_lens = lens.cpu().tolist()
        ok = [n > self.n_fft // 2 for n in host_lens]
        frames = [fz.num_frames(n) for n in host_lens]
        feats = fz(wave, lens, t_pad=max(frames))
        return feats.cpu(), frames, ok

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
