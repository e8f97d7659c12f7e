"""Drop-in for /root/reference/run_pipeline.py: the same YAML keys, stage order and artefact paths
(preprocess -> ``<output_dir>/{train,valid,test}_data.csv`` + ``label_map.json``; precompute ->
``<cache_dir>/<csv-stem>_features.pt`` + ``cache_info.json``; train -> ``<save_path>/best_model.pt``;
evaluate -> ``<save_path>/evaluation_results/``), with the stages that compute running on MI355X.

    python -m sir_amd.run_pipeline --config_path configs/config.yaml [--force_precompute] [--gpus N]

Differences from the reference (run_pipeline.py:39-214): stages are started as ``python -m sir_amd.scripts.<name>``
and, with ``--gpus N`` (or ``gpus: N`` in the YAML) greater than one, training is launched through
``python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1`` (one process per GPU,
utterances sharded, RCCL all-reduce of the gradients); commands are argument lists, not shell strings.
``OMP_NUM_THREADS=1`` is exported as in the reference (:42).
"""
import argparse
import logging
import os
import subprocess
import sys
import tempfile

import yaml

from sir_amd.scripts.preprocess_fsc import preprocess_dataset

logging.basicConfig(level=logging.INFO, format="%(asctime)s - %(levelname)s - %(message)s")
logger = logging.getLogger(__name__)

FALLBACK_CSVS = {                       # run_pipeline.py:69-103
    "train": ["data/processed/train_data.csv", "data/FSC/fluent_speech_commands_dataset/data/train_data.csv", "data/train_data.csv"],
    "valid": ["data/processed/valid_data.csv", "data/FSC/fluent_speech_commands_dataset/data/valid_data.csv", "data/valid_data.csv"],
    "test": ["data/processed/test_data.csv", "data/FSC/fluent_speech_commands_dataset/data/test_data.csv", "data/test_data.csv"],
}


def load_config(config_path):
    with open(config_path, "r") as f:
        return yaml.safe_load(f)


def run_subprocess(cmd, name, env=None):
    """Run one stage; ``cmd`` is an argument list.  -> True on exit code 0 (run_pipeline.py:22-37)."""
    logger.info("Command: " + " ".join(cmd))
    env_vars = os.environ.copy()
    if env:
        env_vars.update(env)
    process = subprocess.run(cmd, env=env_vars)
    if process.returncode != 0:
        logger.error(f"Error in {name}. Return code: {process.returncode}")
        return False
    logger.info(f"{name} completed successfully.")
    return True


def stage_commands(config_path, config, train_csv, valid_csv, test_csv, label_map, gpus=1, master_port=29517):
    """The three compute stages as argument lists (kept separate from ``run_pipeline`` so that the command
    construction can be checked without a GPU)."""
    py = sys.executable
    cache_dir = config.get("cache_dir", "data/cached_features")
    precompute = [py, "-m", "sir_amd.scripts.precompute_features", "--train_csv", train_csv, "--valid_csv", valid_csv,
                  "--test_csv", test_csv, "--output_dir", cache_dir, "--label_map", label_map]
    train_tail = ["-m", "sir_amd.scripts.train", "--config", config_path, "--train_csv", train_csv, "--val_csv", valid_csv,
                  "--label_map", label_map]
    if gpus > 1:
        train = [py, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}", "--master-addr", "127.0.0.1",
                 "--master-port", str(master_port)] + train_tail
    else:
        train = [py] + train_tail
    model_path = os.path.join(config.get("save_path", "checkpoints"), "best_model.pt")
    evaluate = [py, "-m", "sir_amd.scripts.evaluate", "--config", config_path, "--test_csv", test_csv, "--label_map", label_map,
                "--model_path", model_path]
    return {"precompute": precompute, "train": train, "evaluate": evaluate, "model_path": model_path}


def run_pipeline(config_path, gpus=None):
    os.environ["OMP_NUM_THREADS"] = "1"
    logger.info("=== Starting Speech Intent Recognition Pipeline ===")
    config = load_config(config_path)
    gpus = int(gpus if gpus is not None else config.get("gpus", 1))

    csvs = {"train": config.get("train_csv", "data/raw/train.csv"), "valid": config.get("valid_csv", "data/raw/valid.csv"),
            "test": config.get("test_csv", "data/raw/test.csv")}
    for split, path in csvs.items():
        if not os.path.exists(path):
            for alt in FALLBACK_CSVS[split]:
                if os.path.exists(alt):
                    csvs[split] = alt
                    logger.info(f"Using alternative {split} data path: {alt}")
                    break
    if not all(os.path.exists(p) for p in csvs.values()):
        logger.error("Could not find required data files. Please check your data paths.")
        return False

    logger.info("=== STEP 1: DATA PREPROCESSING ===")
    output_dir = config.get("output_dir", "data/processed")
    os.makedirs(output_dir, exist_ok=True)
    pre = preprocess_dataset(train_csv=csvs["train"], valid_csv=csvs["valid"], test_csv=csvs["test"], output_dir=output_dir,
                             label_map_path=config.get("label_map_path", os.path.join(output_dir, "label_map.json")),
                             use_torchaudio=True)
    if not pre:
        logger.error("Preprocessing failed. Stopping pipeline.")
        return False
    train_csv, valid_csv, test_csv, label_map = pre["train_csv"], pre["valid_csv"], pre["test_csv"], pre["label_map"]
    cmds = stage_commands(config_path, config, train_csv, valid_csv, test_csv, label_map, gpus)

    if config.get("use_feature_cache", True):
        logger.info("=== STEP 2: PRECOMPUTING FEATURES ===")
        cache_dir = config.get("cache_dir", "data/cached_features")
        os.makedirs(cache_dir, exist_ok=True)
        train_cache = os.path.join(cache_dir, f"{os.path.basename(train_csv).replace('.csv', '')}_features.pt")
        if config.get("force_precompute", False) or not os.path.exists(train_cache):
            if not run_subprocess(cmds["precompute"], "Feature Precomputation"):
                # the reference carries on without the cache (run_pipeline.py:165-169); so does this pipeline:
                # FSCIntentDataset then extracts on a cache miss
                logger.warning("Feature precomputation failed. Will continue without cached features.")
            else:
                logger.info("Feature precomputation completed successfully")
        else:
            logger.info(f"Using existing cached features in {cache_dir}")

    logger.info("=== STEP 3: TRAINING MODEL ===")
    os.makedirs(config.get("save_path", "checkpoints"), exist_ok=True)
    if not run_subprocess(cmds["train"], "Model Training"):
        logger.error("Training failed. Stopping pipeline.")
        return False

    logger.info("=== STEP 4: EVALUATING MODEL ===")
    if not os.path.exists(cmds["model_path"]):
        logger.error(f"Model file not found: {cmds['model_path']}")
        return False
    if not run_subprocess(cmds["evaluate"], "Model Evaluation"):
        logger.error("Evaluation failed. Stopping pipeline.")
        return False
    logger.info("=== Pipeline Completed Successfully ===")
    return True


def main(argv=None):
    parser = argparse.ArgumentParser(description="Run the full Speech Intent Recognition pipeline")
    parser.add_argument("--config_path", type=str, default="configs/config.yaml", help="Path to config file")
    parser.add_argument("--force_precompute", action="store_true", help="Force precomputation of features even if cache exists")
    parser.add_argument("--gpus", type=int, default=None, help="GPUs of this node to train on (default: `gpus` in the YAML, else 1)")
    args = parser.parse_args(argv)
    if args.force_precompute:
        config = load_config(args.config_path)
        config["force_precompute"] = True
        with tempfile.NamedTemporaryFile("w", suffix=".yaml", delete=False) as f:
            yaml.dump(config, f)
            temp_path = f.name
        try:
            return run_pipeline(temp_path, args.gpus)
        finally:
            if os.path.exists(temp_path):
                os.remove(temp_path)
    return run_pipeline(args.config_path, args.gpus)


if __name__ == "__main__":
    sys.exit(0 if main() else 1)
