"""Drop-in for /root/reference/scripts/evaluate.py::evaluate with the forward + argmax loop
(evaluate.py:74-86) on MI355X: one ``sir_model_infer`` launch sequence per batch, predictions kept
on the device until the loop ends (one device->host copy instead of one per batch).  Metrics and the
report files (evaluate.py:89-114) are unchanged stock sklearn / matplotlib."""
import argparse
import json
import logging
import os

import torch
import yaml
from torch.utils.data import DataLoader
from tqdm import tqdm

from sir_amd import _native
from sir_amd.models.models import CNNAudioGRU
from sir_amd.scripts.dataset import FSCIntentDataset
from sir_amd.scripts.train import collate_fn

logging.basicConfig(level=logging.INFO, format="%(asctime)s - %(levelname)s - %(message)s")
logger = logging.getLogger(__name__)


def load_config(config_path):
    with open(config_path, "r") as f:
        return yaml.safe_load(f)


@torch.no_grad()
def predict_loader(model, loader, device):
    """argmax predictions and labels over a loader (the hot loop of evaluate.py:79-86)."""
    from sir_amd.pipeline import BatchPipeline
    model.eval()
    pipe = BatchPipeline(model, n_streams=2)          # consecutive batches alternate over two HIP streams
    preds, labels = [], []
    from sir_amd.scripts.train import HostStager
    stage = HostStager(device, slots=4)               # host batches: persistent pinned ring (see HostStager)
    for i, (mel, label) in enumerate(tqdm(loader, desc="Evaluating")):
        if mel is None or label is None or mel.size(0) == 0:
            continue                      # the reference would crash here (evaluate.py:81); skip instead
        _, pred = pipe.infer(i, stage(mel))
        preds.append(pred)
        labels.append(label)
    if not preds:
        return [], []
    pipe.synchronize()
    from sir_amd import ops
    ops.check_status()                    # a timed-out GRU recurrence would have produced invalid predictions: raise
    return torch.cat(preds).cpu().numpy(), torch.cat(labels).numpy()


def evaluate(args, config):
    _native.require_hip()
    from sir_amd.dist_utils import limit_host_threads
    limit_host_threads(reserve=int(config.get("num_workers", 4)))
    device = torch.device("cuda", torch.cuda.current_device())
    logger.info(f"Using device: {device}")
    with open(args.label_map, "r") as f:
        label_map = json.load(f)
    inv_label_map = {v: k for k, v in label_map.items()}
    num_classes = len(label_map)
    model = CNNAudioGRU(num_classes=31).to(device)          # 31-way head of the shipped model (evaluate.py:45)
    state = torch.load(args.model_path, map_location=device)
    if isinstance(state, dict) and "model_state_dict" in state:
        state = state["model_state_dict"]
    model.load_state_dict(state)
    logger.info(f"Loaded model from {args.model_path}")
    test_dataset = FSCIntentDataset(csv_path=args.test_csv, label_map_path=args.label_map, is_training=False,
                                    use_cache=config.get("use_feature_cache", True),
                                    cache_dir=config.get("cache_dir", "data/cached_features"))
    from sir_amd.scripts.train import loader_kwargs
    test_loader = DataLoader(test_dataset, batch_size=config.get("batch_size", 32), shuffle=False, collate_fn=collate_fn,
                             **loader_kwargs(config.get("num_workers", 4)))
    logger.info("Starting evaluation...")
    all_preds, all_labels = predict_loader(model, test_loader, device)

    from sklearn.metrics import accuracy_score, classification_report, confusion_matrix
    accuracy = accuracy_score(all_labels, all_preds)
    logger.info(f"Test Accuracy: {accuracy:.4f}")
    labels_idx = list(range(num_classes))
    target_names = [inv_label_map[i] for i in labels_idx]
    cls_report = classification_report(all_labels, all_preds, labels=labels_idx, target_names=target_names,
                                       zero_division=0)
    logger.info(f"Classification Report:\n{cls_report}")
    cm = confusion_matrix(all_labels, all_preds, labels=labels_idx)
    results_dir = os.path.join(config["save_path"], "evaluation_results")
    os.makedirs(results_dir, exist_ok=True)
    with open(os.path.join(results_dir, "classification_report.txt"), "w") as f:
        f.write(f"Test Accuracy: {accuracy:.4f}\n\n")
        f.write(cls_report)
    try:
        import matplotlib
        matplotlib.use("Agg")
        import matplotlib.pyplot as plt
        from sklearn.metrics import ConfusionMatrixDisplay
        plt.figure(figsize=(10, 8))
        ConfusionMatrixDisplay(confusion_matrix=cm, display_labels=target_names).plot(xticks_rotation=45)
        plt.tight_layout()
        plt.savefig(os.path.join(results_dir, "confusion_matrix.png"))
        plt.close("all")
    except Exception as e:  # plotting is reporting, not part of the hot path
        logger.error(f"confusion matrix plot skipped: {e}")
    logger.info(f"Evaluation results saved to {results_dir}")
    return accuracy


if __name__ == "__main__":
    parser = argparse.ArgumentParser(description="Evaluate speech intent recognition model")
    parser.add_argument("--config", type=str, required=True, help="Path to config file")
    parser.add_argument("--test_csv", type=str, required=True, help="Path to test CSV file")
    parser.add_argument("--label_map", type=str, required=True, help="Path to label map JSON file")
    parser.add_argument("--model_path", type=str, required=True, help="Path to trained model")
    args = parser.parse_args()
    evaluate(args, load_config(args.config))
