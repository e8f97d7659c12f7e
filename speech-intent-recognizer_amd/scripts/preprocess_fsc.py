"""Drop-in for /root/reference/scripts/preprocess_fsc.py (host-only CSV step ahead of the hot path).

``preprocess_dataset(train_csv, valid_csv, test_csv, output_dir, label_map_path=None, use_torchaudio=False)``
-> ``{'train_csv', 'valid_csv', 'test_csv', 'label_map'}`` paths (or ``None`` on failure): column
normalisation (``path`` aliases, ``intent``/``class`` -> ``label``, ``action + '_' + object``), audio
validation (exists, decodable, >= 100 samples; :24-54), ``train_data.csv`` / ``valid_data.csv`` /
``test_data.csv`` + ``label_map.json`` (sorted unique training labels -> index; :133-148,186-198).
Validation reads only the RIFF header here (no soundfile / torchaudio in this image): no sample is decoded.
"""
import argparse
import json
import logging
import os
import struct

import pandas as pd

from sir_amd.scripts.utils.path_utils import normalize_audio_path

ROOT_DIR = os.getcwd()
logging.basicConfig(level=logging.INFO, format="%(asctime)s - %(levelname)s - %(message)s")
logger = logging.getLogger(__name__)

PATH_ALIASES = ["file_path", "audio_path", "filepath", "audio_file", "wav_path", "wav_file"]


def _wav_frames(path):
    """Number of sample frames promised by the RIFF header (raises on a non-WAVE file)."""
    with open(path, "rb") as f:
        head = f.read(12)
        if len(head) < 12 or head[:4] != b"RIFF" or head[8:12] != b"WAVE":
            raise ValueError("not a RIFF/WAVE file")
        block = None
        while True:
            hdr = f.read(8)
            if len(hdr) < 8:
                raise ValueError("missing fmt/data chunk")
            cid, size = hdr[:4], struct.unpack("<I", hdr[4:])[0]
            if cid == b"fmt ":
                fmt = f.read(size + (size & 1))
                block = struct.unpack_from("<H", fmt, 12)[0]
            elif cid == b"data":
                if not block:
                    raise ValueError("data chunk before fmt chunk")
                return size // block
            else:
                f.seek(size + (size & 1), 1)


def validate_audio(audio_path, use_torchaudio=False):
    """preprocess_fsc.py:24-54 (``use_torchaudio`` is accepted for signature compatibility)."""
    try:
        if not os.path.exists(audio_path):
            logger.warning(f"File not found: {audio_path}")
            return False
        if _wav_frames(audio_path) < 100:
            logger.warning(f"Audio too short: {audio_path}")
            return False
        return True
    except Exception as e:
        logger.warning(f"Invalid audio file: {audio_path} - {e}")
        return False


def process_dataset(csv_path, base_path, use_torchaudio=False):
    """preprocess_fsc.py:56-131 -> DataFrame with ``path`` and ``label`` columns, or ``None``."""
    logger.info(f"Processing {csv_path}")
    if not os.path.exists(csv_path):
        logger.error(f"CSV file not found: {csv_path}")
        return None
    try:
        df = pd.read_csv(csv_path)
    except Exception as e:
        logger.error(f"Error reading CSV file {csv_path}: {e}")
        return None
    logger.info(f"Loaded {len(df)} examples from {csv_path}")
    required = ["path"] if "path" in df.columns else []
    if not required:
        for col in PATH_ALIASES:
            if col in df.columns:
                df = df.rename(columns={col: "path"})
                required = ["path"]
                break
    if "action" in df.columns and "object" in df.columns:
        required.extend(["action", "object"])
    elif "label" in df.columns or "intent" in df.columns or "class" in df.columns:
        if "intent" in df.columns:
            df = df.rename(columns={"intent": "label"})
        if "class" in df.columns and "label" not in df.columns:
            df = df.rename(columns={"class": "label"})
        required.append("label")
    if "path" not in df.columns or not all(col in df.columns for col in required):
        missing = [col for col in required + ["path"] if col not in df.columns]
        logger.error(f"CSV file {csv_path} missing required columns: {missing}")
        return None
    if "label" not in df.columns and "action" in df.columns and "object" in df.columns:
        df["label"] = df["action"] + "_" + df["object"]
    df["path"] = df["path"].apply(lambda p: normalize_audio_path(p, base_path))
    valid = []
    for idx, p in enumerate(df["path"].tolist()):
        if validate_audio(p, use_torchaudio):
            valid.append(idx)
        else:
            logger.warning(f"Invalid audio file: {p}")
    if not valid:
        logger.error(f"No valid audio files found in {csv_path}")
        return None
    total = len(df)
    df = df.iloc[valid].reset_index(drop=True)
    logger.info(f"Kept {len(df)} valid audio files out of {total}")
    return df


def create_label_map(df):
    """preprocess_fsc.py:133-148."""
    label_column = "label" if "label" in df.columns else "intent"
    if label_column not in df.columns:
        if "action" in df.columns and "object" in df.columns:
            df["label"] = df["action"] + "_" + df["object"]
            label_column = "label"
        else:
            logger.error("Could not find label column in dataframe")
            return {}
    return {label: idx for idx, label in enumerate(sorted(df[label_column].unique()))}


def preprocess_dataset(train_csv, valid_csv, test_csv, output_dir, label_map_path=None, use_torchaudio=False):
    """Pipeline entry (preprocess_fsc.py:150-163)."""
    return main(argparse.Namespace(train_csv=train_csv, valid_csv=valid_csv, test_csv=test_csv, output_dir=output_dir,
                                   label_map_path=label_map_path, use_torchaudio=use_torchaudio))


def main(args):
    """preprocess_fsc.py:165-207."""
    os.makedirs(args.output_dir, exist_ok=True)
    frames = [process_dataset(p, ROOT_DIR, args.use_torchaudio) for p in (args.train_csv, args.valid_csv, args.test_csv)]
    if any(f is None for f in frames):
        logger.error("Failed to process one or more datasets")
        return None
    train_df, valid_df, test_df = frames
    label_map = create_label_map(train_df)
    logger.info(f"Created label map with {len(label_map)} classes")
    outputs = {k: os.path.join(args.output_dir, f"{k}_data.csv") for k in ("train", "valid", "test")}
    for k, df in (("train", train_df), ("valid", valid_df), ("test", test_df)):
        df.to_csv(outputs[k], index=False)
    logger.info(f"Saved processed CSV files to {args.output_dir}")
    label_map_path = args.label_map_path or os.path.join(args.output_dir, "label_map.json")
    os.makedirs(os.path.dirname(os.path.abspath(label_map_path)), exist_ok=True)
    with open(label_map_path, "w") as f:
        json.dump(label_map, f, indent=2)
    logger.info(f"Saved label map to {label_map_path}")
    logger.info(f"Total samples: Train={len(train_df)}, Valid={len(valid_df)}, Test={len(test_df)}")
    return {"train_csv": outputs["train"], "valid_csv": outputs["valid"], "test_csv": outputs["test"], "label_map": label_map_path}


if __name__ == "__main__":
    parser = argparse.ArgumentParser(description="Preprocess FSC dataset")
    parser.add_argument("--train_csv", type=str, required=True, help="Path to train CSV")
    parser.add_argument("--valid_csv", type=str, required=True, help="Path to validation CSV")
    parser.add_argument("--test_csv", type=str, required=True, help="Path to test CSV")
    parser.add_argument("--output_dir", type=str, required=True, help="Output directory")
    parser.add_argument("--label_map_path", type=str, help="Path to save label map")
    parser.add_argument("--use_torchaudio", action="store_true", help="Accepted for compatibility")
    main(parser.parse_args())
