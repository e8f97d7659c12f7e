"""Drop-in for /root/reference/scripts/test_model.py: single-file / directory / interactive inference.

Same surface -- ``load_model(model_path, num_classes, device)``, ``extract_features(audio_path)`` ->
``FloatTensor[1, 64, T]`` or ``None``, ``predict(model, audio_path, label_map, device)`` ->
``{"predicted_label", "confidence", "top_predictions"}`` or ``None``, ``get_top_predictions``, ``batch_test``,
``interactive_test`` and the ``--model --label_map --audio --interactive`` CLI (test_model.py:29-290) -- with the
decode -> mono -> resample -> log-mel chain (:62-94) and the forward (:121-125) on MI355X.  ``batch_test``
pushes the whole directory through ONE feature pass and one forward instead of a file at a time; the
un-padded variable-length form of scripts/test_tts_samples.py:74-114 is ``predict(..., pad_to=None)``.
Errors follow the reference: logged, ``None`` returned, never raised.
"""
import argparse
import json
import logging
import os
import sys

import numpy as np
import torch

from sir_amd import ops
from sir_amd.models.models import CNNAudioGRU
from sir_amd.scripts.precompute_features import AudioFeatureExtractor

logging.basicConfig(level=logging.INFO, format="%(asctime)s - %(levelname)s - %(message)s",
                    handlers=[logging.StreamHandler(sys.stdout)])
logger = logging.getLogger(__name__)

MAX_LENGTH = 200                 # test_model.py:115
_extractor = None


def _get_extractor():
    global _extractor
    if _extractor is None:
        _extractor = AudioFeatureExtractor()
    return _extractor


def load_model(model_path, num_classes, device):
    """test_model.py:29-49."""
    try:
        logger.info(f"Loading model from {model_path}")
        logger.info(f"Using device: {device}")
        model = CNNAudioGRU(num_classes=num_classes).to(device)
        if not os.path.exists(model_path):
            logger.error(f"Model file not found: {model_path}")
            return None
        model.load_state_dict(torch.load(model_path, map_location=device))
        model.eval()
        logger.info("Model loaded successfully")
        return model
    except Exception as e:
        logger.error(f"Failed to load model: {str(e)}")
        return None


def extract_features(audio_path):
    """-> FloatTensor [1, 64, T] (CPU) or None (test_model.py:51-104).  The reference does not truncate
    here, so the clip is featurised at its full length (bounded at 10 minutes)."""
    try:
        logger.info(f"Processing audio file: {audio_path}")
        if not os.path.exists(audio_path):
            logger.error(f"Audio file not found: {audio_path}")
            return None
        feats = _get_extractor().extract_batch([audio_path], max_duration=600.0)[0]
        if feats is None:
            return None
        logger.info(f"Mel spectrogram shape: {tuple(feats.shape)}")
        return feats.unsqueeze(0)
    except Exception as e:
        logger.error(f"Error extracting features: {str(e)}")
        return None


def _pad_or_trim(mel_spec, max_length):
    if mel_spec.size(2) > max_length:
        return mel_spec[:, :, :max_length]
    return torch.nn.functional.pad(mel_spec, (0, max_length - mel_spec.size(2)))


def get_top_predictions(probs, inv_label_map, k=3):
    """test_model.py:142-153."""
    probs = probs.cpu().numpy()[0]
    top_indices = np.argsort(probs)[::-1][:k]
    return [{"label": inv_label_map.get(int(idx), "Unknown"), "probability": float(probs[idx])} for idx in top_indices]


def _result(output_row, inv_label_map):
    probs = torch.nn.functional.softmax(output_row, dim=1)
    pred_class = int(torch.argmax(output_row, dim=1).item())
    return {"predicted_label": inv_label_map.get(pred_class, "Unknown"),
            "confidence": float(probs[0][pred_class].item()),
            "top_predictions": get_top_predictions(probs, inv_label_map, k=3)}


def predict(model, audio_path, label_map, device, pad_to=MAX_LENGTH):
    """test_model.py:106-140.  ``pad_to=None`` feeds the un-padded ``[1, 1, 64, T]`` features
    (test_tts_samples.py:83-87; needs ``T >= 8``)."""
    try:
        mel_spec = extract_features(audio_path)
        if mel_spec is None:
            return None
        if pad_to is not None:
            mel_spec = _pad_or_trim(mel_spec, pad_to)
        mel_spec = mel_spec.to(device)
        with torch.no_grad():
            output = model(mel_spec)
        result = _result(output, {v: k for k, v in label_map.items()})
        ops.check_status()
        return result
    except Exception as e:
        logger.error(f"Error during prediction: {str(e)}")
        return None


def predict_many(model, audio_paths, label_map, device, pad_to=MAX_LENGTH):
    """Batched ``predict``: one feature pass and one forward for all files -> list of results / ``None``."""
    try:
        feats = _get_extractor().extract_batch(list(audio_paths), max_duration=600.0)
        keep = [i for i, f in enumerate(feats) if f is not None]
        results = [None] * len(feats)
        if not keep:
            return results
        batch = torch.stack([_pad_or_trim(feats[i].unsqueeze(0), pad_to)[0] for i in keep]).to(device)
        with torch.no_grad():
            output = model(batch)
        inv = {v: k for k, v in label_map.items()}
        for row, i in enumerate(keep):
            results[i] = _result(output[row:row + 1], inv)
        ops.check_status()
        return results
    except Exception as e:
        logger.error(f"Error during prediction: {str(e)}")
        return [None] * len(audio_paths)


def interactive_test(model, label_map, device):
    """test_model.py:155-186."""
    print("\n===== INTERACTIVE TESTING =====")
    print("Enter the path to an audio file (or 'q' to quit):")
    while True:
        user_input = input("\nAudio file path (or 'q' to quit): ")
        if user_input.lower() == "q":
            break
        if not os.path.exists(user_input):
            print(f"File not found: {user_input}")
            continue
        result = predict(model, user_input, label_map, device)
        if result is None:
            print("Failed to make prediction. Check logs for details.")
            continue
        _print_result(result)


def _print_result(result):
    print("\n----- PREDICTION RESULTS -----")
    print(f"Predicted intent: {result['predicted_label']}")
    print(f"Confidence: {result['confidence'] * 100:.2f}%")
    print("\nTop predictions:")
    for i, pred in enumerate(result["top_predictions"]):
        print(f"  {i + 1}. {pred['label']} ({pred['probability'] * 100:.2f}%)")


def batch_test(model, audio_dir, label_map, device):
    """test_model.py:188-223, batched."""
    print(f"\n===== BATCH TESTING on {audio_dir} =====")
    audio_files = sorted(os.path.join(audio_dir, f) for f in os.listdir(audio_dir) if f.endswith((".wav", ".mp3", ".flac")))
    if not audio_files:
        print(f"No audio files found in {audio_dir}")
        return
    print(f"Found {len(audio_files)} audio files")
    results = []
    for path, result in zip(audio_files, predict_many(model, audio_files, label_map, device)):
        if result is None:
            print(f"Failed to process {path}")
            continue
        result["file"] = os.path.basename(path)
        results.append(result)
    print("\n----- BATCH RESULTS SUMMARY -----")
    for result in results:
        print(f"{result['file']}: {result['predicted_label']} ({result['confidence'] * 100:.2f}%)")
    return results


def main():
    parser = argparse.ArgumentParser(description="Test speech intent recognition model")
    parser.add_argument("--model", type=str, default="checkpoints/best_model.pt", help="Path to the trained model")
    parser.add_argument("--label_map", type=str, default="data/processed/label_map.json", help="Path to the label map")
    parser.add_argument("--audio", type=str, default=None, help="Path to an audio file or directory for testing")
    parser.add_argument("--interactive", action="store_true", help="Run in interactive mode")
    args = parser.parse_args()
    if not os.path.exists(args.model):
        logger.error(f"Model file not found: {args.model}")
        return
    if not os.path.exists(args.label_map):
        logger.error(f"Label map not found: {args.label_map}")
        return
    with open(args.label_map, "r") as f:
        label_map = json.load(f)
    device = torch.device("cuda")
    model = load_model(args.model, num_classes=31, device=device)       # the reference hard-codes 31 (:263)
    if model is None:
        return
    if args.interactive:
        interactive_test(model, label_map, device)
    elif args.audio:
        if os.path.isdir(args.audio):
            batch_test(model, args.audio, label_map, device)
        elif os.path.isfile(args.audio):
            result = predict(model, args.audio, label_map, device)
            if result:
                _print_result(result)
        else:
            logger.error(f"Audio path not found: {args.audio}")
    else:
        interactive_test(model, label_map, device)


if __name__ == "__main__":
    main()
