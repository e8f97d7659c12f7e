"""Generate the committed golden vectors.  Run HERE only (build container):

    python tests/golden/make_golden.py

* ``model_golden.npz``   : outputs of the REFERENCE's own ``models/models.py::CNNAudioGRU``
  (imported from /root/reference, torch-only) on seeded inputs/weights -- logits, argmax,
  one training step (loss, sampled grads, BN running stats, sampled post-Adam params).
  These pin ``oracle/model_ref.py``.
* ``train_traj_golden.npz``: a K = 10 step Adam trajectory of the REFERENCE model on ``x_train8`` (``gru.dropout = 0``): loss
  and logits norm per step, sampled parameters and BN running statistics after step 10 ("matched
  accuracy" over more than one step).
* ``features_golden.npz``: outputs of ``oracle/features_ref.py`` (float32 torch.stft path and the
  float64 numpy path).  torchaudio is not installed, so these are NOT reference outputs:
  feature parity is "unpinned" (see oracle/__init__.py).

The reference never travels: only these small arrays are committed.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import cases  # noqa: E402
from cases import ROOT, LR, WEIGHT_DECAY  # noqa: E402

from sir_amd import synth  # noqa: E402
from oracle import features_ref  # noqa: E402


def make_model_golden():
    sys.path.insert(0, "/root/reference")
    from models.models import CNNAudioGRU  # the reference itself

    torch.manual_seed(0)
    sd = synth.synth_state_dict(31, seed=0)
    inp = cases.model_inputs()
    out = {}

    model = CNNAudioGRU(31)
    model.load_state_dict(sd)
    model.eval()
    with torch.no_grad():
        lg8 = model(inp["x_eval8"])
        lg1 = model(inp["x_eval1_t94"])
    out["eval8_logits"] = lg8.numpy()
    out["eval8_argmax"] = lg8.argmax(1).numpy()
    out["eval1_logits"] = lg1.numpy()
    out["eval1_argmax"] = lg1.argmax(1).numpy()

    # input-dependent argmax: sharpen and centre the head (see cases.sharp_head)
    ctx_store = []
    hook = model.fc.register_forward_pre_hook(lambda mod, a: ctx_store.append(a[0].detach()))
    with torch.no_grad():
        model(inp["x_sharp64"])
    hook.remove()
    w_sharp = sd["fc.weight"] * cases.HEAD_GAIN
    fc_bias = -(w_sharp @ ctx_store[0].mean(0))
    model.load_state_dict(cases.sharp_head(sd, fc_bias))
    with torch.no_grad():
        lgs = model(inp["x_sharp64"])
    out["sharp_fc_bias"] = fc_bias.numpy()
    out["sharp64_logits"] = lgs.numpy()
    out["sharp64_argmax"] = lgs.argmax(1).numpy()
    srt = np.sort(lgs.numpy(), 1)
    print("sharp64: classes", len(set(out["sharp64_argmax"].tolist())),
          "min top-2 margin", (srt[:, -1] - srt[:, -2]).min())

    # one training step, inter-layer dropout disabled (non-deterministic otherwise)
    model = CNNAudioGRU(31)
    model.load_state_dict(sd)
    model.train()
    model.gru.dropout = 0.0
    opt = torch.optim.Adam(model.parameters(), lr=LR, weight_decay=WEIGHT_DECAY)
    crit = torch.nn.CrossEntropyLoss()
    opt.zero_grad(set_to_none=True)
    logits = model(inp["x_train8"])
    loss = crit(logits, inp["y_train8"])
    loss.backward()
    out["train8_loss"] = np.float32(loss.item())
    out["train8_logits"] = logits.detach().numpy()
    for name, p in model.named_parameters():
        g = p.grad.detach().flatten()
        idx = cases.sample_indices(name, g.numel())
        out[f"grad_norm/{name}"] = np.float32(g.double().norm().item())
        out[f"grad_samp/{name}"] = g[idx].numpy()
    opt.step()
    for name, p in model.named_parameters():
        flat = p.detach().flatten()
        idx = cases.sample_indices(name, flat.numel())
        out[f"adam_samp/{name}"] = flat[idx].numpy()
        out[f"adam_delta_norm/{name}"] = np.float32((p.detach() - sd[name]).double().norm().item())
    for i in (1, 2, 3):
        out[f"bn{i}.running_mean"] = getattr(model, f"bn{i}").running_mean.numpy()
        out[f"bn{i}.running_var"] = getattr(model, f"bn{i}").running_var.numpy()
    np.savez_compressed(os.path.join(HERE, "model_golden.npz"), **out)
    print("model_golden.npz:", len(out), "arrays; eval8 argmax", out["eval8_argmax"], "loss", out["train8_loss"])


TRAJ_STEPS = 10


def make_trajectory_golden():
    """K Adam steps (train.py:90-107: zero_grad, forward, CrossEntropyLoss, backward, step; Adam as train.py:246-250 at the shipped
    lr / weight decay) of the reference's own CNNAudioGRU on the SAME batch ``x_train8``; one torch thread so that the file
    regenerates bit for bit whatever the host's core count."""
    sys.path.insert(0, "/root/reference")
    from models.models import CNNAudioGRU  # the reference itself

    nthr = torch.get_num_threads()
    torch.set_num_threads(1)
    try:
        sd = synth.synth_state_dict(31, seed=0)
        inp = cases.model_inputs()
        model = CNNAudioGRU(31)
        model.load_state_dict(sd)
        model.train()
        model.gru.dropout = 0.0
        opt = torch.optim.Adam(model.parameters(), lr=LR, weight_decay=WEIGHT_DECAY)
        crit = torch.nn.CrossEntropyLoss()
        losses, lnorms = [], []
        for _ in range(TRAJ_STEPS):
            opt.zero_grad(set_to_none=True)
            logits = model(inp["x_train8"])
            loss = crit(logits, inp["y_train8"])
            loss.backward()
            opt.step()
            losses.append(loss.item())
            lnorms.append(logits.detach().double().norm().item())
        out = {"steps": np.int32(TRAJ_STEPS), "loss": np.asarray(losses, np.float32), "logits_norm": np.asarray(lnorms, np.float32),
               "final_logits": logits.detach().numpy()}
        for name, p in model.named_parameters():
            flat = p.detach().flatten()
            idx = cases.sample_indices(name, flat.numel())
            out[f"param_samp/{name}"] = flat[idx].numpy()
            out[f"param_delta_norm/{name}"] = np.float32((p.detach() - sd[name]).double().norm().item())
        for i in (1, 2, 3):
            out[f"bn{i}.running_mean"] = getattr(model, f"bn{i}").running_mean.numpy()
            out[f"bn{i}.running_var"] = getattr(model, f"bn{i}").running_var.numpy()
    finally:
        torch.set_num_threads(nthr)
    np.savez_compressed(os.path.join(HERE, "train_traj_golden.npz"), **out)
    print("train_traj_golden.npz:", len(out), "arrays; losses", [f"{v:.5f}" for v in losses])


def make_features_golden():
    out = {}
    for name, wave in cases.feature_cases().items():
        st = features_ref.extract_features_f32(wave, stages=True)
        out[f"{name}/mel_power"] = st["mel_power"].numpy()
        out[f"{name}/db"] = st["db"].numpy()
        out[f"{name}/padded"] = features_ref.pad_or_trim(st["norm"]).numpy()
        st64 = features_ref.extract_features_f64(wave.numpy(), stages=True)
        out[f"{name}/db_f64"] = st64["db"].astype(np.float32)
        out[f"{name}/norm_f64"] = st64["norm"].astype(np.float32)
    np.savez_compressed(os.path.join(HERE, "features_golden.npz"), **out)
    print("features_golden.npz:", len(out), "arrays")


if __name__ == "__main__":
    make_model_golden()
    make_trajectory_golden()
    make_features_golden()
