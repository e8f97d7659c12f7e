"""GPU tests of the surfaces either side of the hot path:
the single-utterance ``predict`` of scripts/test_model.py and the ``run_pipeline`` orchestration."""
import json
import os
import sys

import pandas as pd
import pytest
import torch
import yaml

from oracle import features_ref, model_ref, resample_ref
from sir_amd import synth
from sir_amd.scripts.utils import wav_io

pytestmark = pytest.mark.gpu
LABELS = [f"intent_{i:02d}" for i in range(31)]


def _label_map():
    return {l: i for i, l in enumerate(LABELS)}


def _oracle_probs(sd, path, pad_to):
    wave, sr = wav_io.read_wav(path)
    wave = resample_ref.resample(resample_ref.to_mono(wave), sr, 16000)
    feats = features_ref.extract_features_f32(wave[0])
    if pad_to is not None:
        feats = features_ref.pad_or_trim(feats, pad_to)
    logits = model_ref.forward(sd, feats[None])
    return torch.softmax(logits, dim=1)[0]


def test_predict_single_file_padded_and_unpadded(tmp_path):
    from sir_amd.scripts import test_model as tm
    sd = synth.synth_state_dict(31, seed=0)
    ckpt = str(tmp_path / "best_model.pt")
    torch.save(sd, ckpt)
    model = tm.load_model(ckpt, 31, torch.device("cuda"))
    assert model is not None and not model.training
    assert tm.load_model(str(tmp_path / "absent.pt"), 31, torch.device("cuda")) is None
    clips = synth.synth_clips(3, 40000, seed=77)
    paths = []
    for i, sr in enumerate((16000, 22050, 16000)):
        p = str(tmp_path / f"c{i}.wav")
        wav_io.write_wav_pcm16(p, clips[i], sr)
        paths.append(p)
    lm = _label_map()
    for p in paths:
        for pad_to in (200, None):                      # test_model.py:113-119 / test_tts_samples.py:83-87
            res = tm.predict(model, p, lm, torch.device("cuda"), pad_to=pad_to)
            probs = _oracle_probs(sd, p, pad_to)
            assert res["predicted_label"] == LABELS[int(probs.argmax())]
            assert abs(res["confidence"] - float(probs.max())) < 1e-4
            top = res["top_predictions"]
            assert [t["label"] for t in top] == [LABELS[int(i)] for i in probs.argsort(descending=True)[:3]]
            assert all(abs(t["probability"] - float(probs[LABELS.index(t["label"])])) < 1e-4 for t in top)
    assert tm.predict(model, str(tmp_path / "missing.wav"), lm, torch.device("cuda")) is None
    many = tm.predict_many(model, paths + [str(tmp_path / "missing.wav")], lm, torch.device("cuda"))
    assert many[3] is None
    for p, r in zip(paths, many):
        assert r["predicted_label"] == tm.predict(model, p, lm, torch.device("cuda"))["predicted_label"]
    out = tm.batch_test(model, str(tmp_path), lm, torch.device("cuda"))
    assert sorted(r["file"] for r in out) == ["c0.wav", "c1.wav", "c2.wav"]


def test_run_pipeline_end_to_end(tmp_path):
    """Same YAML keys, stage order and artefact paths as the reference's run_pipeline.py, on a toy corpus."""
    from sir_amd import run_pipeline as rp
    wav_dir = tmp_path / "wav"
    wav_dir.mkdir()
    clips = synth.synth_clips(24, 20000, seed=5)
    rows = []
    for i in range(24):
        p = str(wav_dir / f"u{i:02d}.wav")
        wav_io.write_wav_pcm16(p, clips[i], 16000 if i % 5 else 24000)
        rows.append({"path": p, "action": ["activate", "deactivate"][i % 2], "object": ["lights", "music"][(i // 2) % 2]})
    for split, sl in (("train", slice(0, 16)), ("valid", slice(16, 20)), ("test", slice(20, 24))):
        pd.DataFrame(rows[sl]).to_csv(tmp_path / f"{split}.csv", index=False)
    cfg = {"train_csv": str(tmp_path / "train.csv"), "valid_csv": str(tmp_path / "valid.csv"), "test_csv": str(tmp_path / "test.csv"),
           "output_dir": str(tmp_path / "processed"), "label_map_path": str(tmp_path / "processed" / "label_map.json"),
           "cache_dir": str(tmp_path / "cache"), "save_path": str(tmp_path / "ckpt"), "use_feature_cache": True,
           "batch_size": 8, "num_workers": 0, "num_labels": 31, "lr": 1e-3, "weight_decay": 1e-4, "epochs": 2,
           "early_stop_patience": 3, "use_amp": True, "augment_prob": 0.5}
    cfg_path = tmp_path / "config.yaml"
    cfg_path.write_text(yaml.safe_dump(cfg))
    env_keep = os.environ.get("PYTHONPATH", "")
    os.environ["PYTHONPATH"] = os.pathsep.join([os.path.dirname(os.path.dirname(os.path.abspath(__file__))), env_keep])
    try:
        ok = rp.main(["--config_path", str(cfg_path)])
    finally:
        os.environ["PYTHONPATH"] = env_keep
    assert (tmp_path / "processed" / "train_data.csv").exists() and (tmp_path / "processed" / "label_map.json").exists()
    assert json.load(open(tmp_path / "processed" / "label_map.json")) == {
        "activate_lights": 0, "activate_music": 1, "deactivate_lights": 2, "deactivate_music": 3}
    cache = torch.load(tmp_path / "cache" / "train_data_features.pt")
    assert len(cache) == 16 and all(v["features"].shape[0] == 64 for v in cache.values())
    assert json.load(open(tmp_path / "cache" / "cache_info.json"))["train_features"].endswith("train_data_features.pt")
    if ok:                                                # training saves only on an accuracy improvement over 0
        assert (tmp_path / "ckpt" / "best_model.pt").exists()
        assert (tmp_path / "ckpt" / "evaluation_results" / "classification_report.txt").exists()
    else:
        assert not (tmp_path / "ckpt" / "best_model.pt").exists()
