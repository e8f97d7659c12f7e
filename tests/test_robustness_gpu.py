"""GPU robustness of the hot path around its parity-tested core:

* the GRU recurrences' exchange granules live in handle-owned buffers (csrc/sir_internal.h, sir_xbuf_acquire): alternating
  batch sizes on ONE workspace, and a workspace scribbled over with bit patterns that look like valid granule tags,
  must leave inference and the training step bit-identical;
* a label outside [0, num_classes) is reported (nn.CrossEntropyLoss raises on it; train.py:242 / :105) instead of read
  out of bounds;
* library objects torn down in any order leave the process exiting cleanly (the r2 core dump at interpreter exit was a
  slot stream destroyed before torch's allocator recorded its free-events).
"""
import os
import subprocess
import sys
import textwrap

import pytest
import torch

import cases
from sir_amd import _native, ops, synth, train_ops
from sir_amd.models.models import CNNAudioGRU

pytestmark = pytest.mark.gpu
DEV = "cuda"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def sd():
    return synth.synth_state_dict(31, seed=0)


def _scribble(buf):
    """Fill a workspace with float bit patterns whose top 16 bits are plausible forward-granule tags ({7-bit epoch,
    9-bit step + 1}: 0x3E01..0x3E19 = 0.126..0.149 as floats), then invalidate the prepared weights kept in it."""
    v = buf.view(torch.int32)
    steps = torch.arange(v.numel(), device=buf.device, dtype=torch.int32) % 25 + 1
    v.copy_(((0x3E00 + steps) << 16) | 0x1234)
    ops.bump_weights_epoch()


def test_inference_alternating_batch_sizes_and_scribbled_workspace(sd):
    m = CNNAudioGRU(31)
    m.load_state_dict(sd)
    m = m.to(DEV).eval()
    xs = {"a": cases.varied_features(24, 200, seed=31).to(DEV), "b": cases.varied_features(256, 200, seed=32).to(DEV),
          "c": synth.synth_features(5, 96, seed=33).to(DEV)}
    first = {}
    for rnd, order in enumerate(("abc", "cab", "bca", "aab")):
        for k in order:
            lg, am = m.predict(xs[k])
            torch.cuda.synchronize()
            if k not in first:
                first[k] = (lg.clone(), am.clone())
            assert torch.equal(lg, first[k][0]) and torch.equal(am, first[k][1]), (rnd, k)
            if rnd >= 1:
                _scribble(m._ws.buf)
    ops.check_status()


def test_training_alternating_batch_sizes_and_scribbled_workspace(sd):
    m = CNNAudioGRU(31)
    m.load_state_dict(sd)
    m = m.to(DEV).train()
    m.gru.dropout = 0.0
    data = {k: (cases.varied_features(b, 200, seed=40 + b).to(DEV), synth.synth_labels(b, 31, seed=b).to(DEV))
            for k, b in (("a", 8), ("b", 64), ("c", 19))}
    first = {}
    for rnd, order in enumerate(("abc", "bac", "cba")):
        for k in order:
            x, y = data[k]
            for p in m.parameters():
                p.grad = None
            # same weights and BN buffers every time: only the workspace history differs between the repetitions
            m.load_state_dict(sd)
            loss = train_ops.fused_cross_entropy(m(x), y)
            loss.backward()
            torch.cuda.synchronize()
            g = torch.cat([p.grad.flatten() for p in m.parameters()]).clone()
            if k not in first:
                first[k] = (loss.detach().clone(), g)
            assert torch.equal(loss.detach(), first[k][0]), (rnd, k)
            assert torch.equal(g, first[k][1]), (rnd, k)
            if rnd >= 1:
                _scribble(m._sir_train["ws"].buf)
    # and the eval path of the same module in between (its own workspace, the same handle-owned exchange buffers)
    m.load_state_dict(sd)               # (the last training step above moved the BN running statistics)
    m.eval()
    lg0 = m(data["b"][0]).clone()
    m.train()
    train_ops.fused_cross_entropy(m(data["a"][0]), data["a"][1]).backward()
    m.load_state_dict(sd)
    m.eval()
    assert torch.equal(m(data["b"][0]), lg0)
    ops.check_status()


def test_out_of_range_label_is_reported_not_read_out_of_bounds(sd):
    logits = torch.randn(6, 31, device=DEV, requires_grad=True)
    labels = torch.tensor([0, 30, 31, 5, -100, 7], device=DEV)
    loss = train_ops.fused_cross_entropy(logits, labels)
    loss.backward()
    torch.cuda.synchronize()
    assert torch.isnan(loss)
    with pytest.raises(_native.SirError, match="label outside"):
        ops.check_status()
    ops.check_status()                                   # cleared by the failing check
    good = train_ops.fused_cross_entropy(logits, torch.tensor([0, 30, 3, 5, 1, 7], device=DEV))
    ref = torch.nn.functional.cross_entropy(logits.detach().cpu(), torch.tensor([0, 30, 3, 5, 1, 7]))
    assert abs(good.item() - ref.item()) < 1e-5
    ops.check_status()


def test_ignore_index_rows_follow_torch():
    """nn.CrossEntropyLoss() (train.py:242) has ignore_index = -100: such rows leave the loss, the gradient and the divisor."""
    torch.manual_seed(3)
    lg = torch.randn(9, 31)
    labels = torch.tensor([0, -100, 30, 5, -100, 7, 1, 2, -100])
    a = lg.clone().to(DEV).requires_grad_(True)
    b = lg.clone().requires_grad_(True)
    la = train_ops.fused_cross_entropy(a, labels.to(DEV))
    lb = torch.nn.functional.cross_entropy(b, labels)
    la.backward()
    lb.backward()
    assert abs(la.item() - lb.item()) < 1e-6
    assert (a.grad.cpu() - b.grad).abs().max() < 1e-7 and (a.grad[1] == 0).all() and (a.grad[8] == 0).all()
    ops.check_status()                                   # -100 is not an error
    allign = train_ops.fused_cross_entropy(a.detach(), torch.full((9,), -100, device=DEV))
    assert torch.isnan(allign)                           # 0 / 0, as torch
    ops.check_status()


_TEARDOWN = """
import os, sys, gc
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests", "golden"))
import torch
torch.cuda.set_device(0)
import torch.distributed as dist
from sir_amd import synth, train_ops, ops
from sir_amd.featurizer import HipFeaturizer, get_featurizer
from sir_amd.models.models import CNNAudioGRU
from sir_amd.pipeline import BatchPipeline, FeaturePrefetcher
order = {order!r}
backend = {backend!r}
if backend:
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT={port!r})
    kw = dict(device_id=torch.device("cuda", 0)) if backend == "nccl" else dict()
    dist.init_process_group(backend, rank=0, world_size=1, **kw)
m = CNNAudioGRU(31); m.load_state_dict(synth.synth_state_dict(31, seed=0)); m = m.to("cuda").eval()
wave = synth.synth_clips(16, 48000, seed=3).to("cuda")
objs = dict()
objs["pipe2"] = BatchPipeline(m, n_streams=2)
objs["pipe3"] = BatchPipeline(m, n_streams=3)
objs["pre"] = FeaturePrefetcher(t_pad=200)
objs["fz"] = HipFeaturizer()                 # a private handle next to the cached one
for name in ("pipe2", "pipe3"):
    p = objs[name]
    for i in range(4):
        f = p.features(i, wave, None, t_pad=200)
        p.infer(i, f)
    p.synchronize()
objs["pre"].submit(wave); x = objs["pre"].get(); objs["pre"].release()
objs["fz"](wave)
if backend:
    train_ops.FORCE_EXCHANGE = True
m.train()
loss = train_ops.fused_cross_entropy(m(x), synth.synth_labels(16, 31, seed=1).to("cuda")); loss.backward()
torch.cuda.synchronize()
for name in order:
    if name == "pg":
        if backend:
            dist.destroy_process_group()
    elif name == "model":
        del m, loss, x, f
    else:
        objs.pop(name)
    gc.collect()
    torch.cuda.synchronize()
ops.check_status(all_ranks=False)
print("TEARDOWN-OK")
"""


@pytest.mark.parametrize("order,backend", [
    (("pipe2", "pipe3", "pre", "fz", "model", "pg"), "nccl"),
    (("pg", "model", "fz", "pre", "pipe3", "pipe2"), "nccl"),
    (("fz", "pg", "pipe2", "model", "pre", "pipe3"), "gloo"),
    (("model", "pre", "pg", "pipe3", "fz", "pipe2"), None),
    ((), "nccl"),                       # nothing dropped explicitly: everything is left to interpreter exit
    ((), None),
])
def test_teardown_in_any_order_exits_cleanly(order, backend, tmp_path):
    """Create pipelines (library-owned slot streams wrapped as ExternalStream), a prefetcher, a private featurizer, a
    model with training state and a one-rank process group; drop them in the given order; the interpreter must exit
    with code 0 and no crash at exit."""
    code = textwrap.dedent(_TEARDOWN).format(root=ROOT, order=tuple(order), backend=backend, port=str(29811 + len(order)))
    script = tmp_path / "teardown.py"
    script.write_text(code)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0 and "TEARDOWN-OK" in r.stdout, (r.returncode, r.stdout[-500:], r.stderr[-3000:])


@pytest.mark.parametrize("mask", ["0", "5"])
def test_fallback_conv_kernels_keep_parity(mask):
    """SIR_WINO2 selects per stage between the producer / consumer Winograd kernel (default: conv2, conv3, conv3 data gradient) and
    the first-generation / direct kernels, which shapes outside the new kernel's range still use.  The switch is read once per
    process, so the reference-golden inference and training-step tests run again in a child with the stages switched off
    (0) or mixed (5: conv3 direct, the others on): both sets of kernels stay correct.  SIR_TN2 does the same for the GRU backward
    GEMMs (producer / consumer kernel by default; 0 = the first kernel everywhere, 2 = only the 128-row dX on the new one) and
    SIR_WGW for the convolution weight gradients (Winograd form by default; 0 = the nine-tap kernel, 1 = conv2 only)."""
    env = dict(os.environ, SIR_WINO2=mask, SIR_TN2={"0": "0", "5": "2"}[mask], SIR_WGW={"0": "0", "5": "1"}[mask])
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu",
                        os.path.join(ROOT, "tests", "test_model_gpu.py::test_eval_golden_from_reference"),
                        os.path.join(ROOT, "tests", "test_model_gpu.py::test_stages_vs_oracle"),
                        os.path.join(ROOT, "tests", "test_train_gpu.py::test_train_step_matches_reference_golden"),
                        os.path.join(ROOT, "tests", "test_train_gpu.py::test_train_forward_backward_stages")],
                       capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-2000:])
