"""GPU parity of the training step (forward with batch-stat BN, CE, backward, Adam) against the
oracle (torch autograd over oracle/model_ref.py) and the reference-generated golden samples
(tests/golden/model_golden.npz: CNNAudioGRU in train() with gru.dropout = 0, Adam lr 5e-5 wd 1e-4).

Tolerance: fp32-level everywhere; gradients are compared per tensor, relative to the tensor's RMS:
max|a-b| <= 2e-3 * rms(b) + 1e-7 (different summation orders over up to 8*25*... terms).

ReLU/max-pool ties: two correct fp32 forwards differ in the last bits (z2 here: 3e-6 abs on rms 0.73),
and a 2x2 pooling window whose two largest entries are closer than that routes its gradient to a
different pixel (train8 has one such window: utterance 7, conv2 channel 35, gap 7e-7).  One window moves the
gradients of its layer by ~1/sqrt(#windows) -- 1e-2 at the small test batches -- and everything upstream of it.
The oracle-based tests therefore evaluate the oracle's BACKWARD at the values the device's ReLU / pooling actually
compared (_device_forward_values: z and y = fma(z, scale, shift) of all three blocks, reproduced bit for bit;
model_ref z_override / y_override; forward stages are still compared without it) and stay at 2e-3; the comparison
with the reference-generated gradient samples cannot do that and allows 2e-2 on the CNN parameters."""
import ctypes as C

import numpy as np
import pytest
import torch

import cases
from oracle import model_ref
from sir_amd import _native, synth, train_ops
from sir_amd.featurizer import get_featurizer
from sir_amd.models.models import CNNAudioGRU
from sir_amd.optim import FusedAdam

pytestmark = pytest.mark.gpu
DEV = "cuda"

TB = {"a1": 0, "z2": 1, "a2": 2, "z3": 3, "x0": 4, "y0": 8, "y1": 10, "ctx": 11,
      "dy1": 21, "dy0": 22, "dgi": 23, "dgh": 24, "dx0": 25, "dz3": 26, "da2": 27, "dz2": 28, "da1": 29}


def _model(sd, dropout=0.0):
    m = CNNAudioGRU(31)
    m.load_state_dict(sd)
    m = m.to(DEV).train()
    m.gru.dropout = dropout
    return m


def _views(m, bsz, t):
    lib = _native.lib()
    offs = (C.c_size_t * 40)()
    n = lib.sir_model_train_workspace_offsets(get_featurizer().handle, bsz, t, offs, 40)
    assert n > 0
    ws = m._sir_train["ws"].buf
    wp1, wp2 = t // 2, t // 4
    s = wp2 // 2
    shp = {"a1": (bsz, 32, wp1, 32), "z2": (bsz, 32, wp1, 64), "a2": (bsz, 16, wp2, 64), "z3": (bsz, 16, wp2, 128),
           "x0": (bsz, s, 1024), "y0": (bsz, s, 512), "y1": (bsz, s, 512), "ctx": (bsz, 512),
           "dy1": (bsz, s, 512), "dy0": (bsz, s, 512), "dx0": (bsz, s, 1024), "da2": (bsz, 16, wp2, 64),
           "da1": (bsz, 32, wp1, 32), "dz3": (bsz, 16, wp2, 128), "dz2": (bsz, 32, wp1, 64)}
    out = {}
    for k, sh in shp.items():
        numel = int(np.prod(sh))
        out[k] = ws[offs[TB[k]]: offs[TB[k]] + 4 * numel].view(torch.float32).view(sh).cpu()
    return out


def _loss_scale(bsz):
    """The backward's internal loss scale (csrc/model_train.hip::sir_bwd_loss_scale): 2^8 x batch rounded up to a power of two.  The
    intermediate gradients in the workspace carry it (the parameter gradients do not)."""
    k = 8
    while (1 << (k - 8)) < bsz and k < 24:
        k += 1
    return float(1 << k)


def _rel(a, b):
    a, b = a.detach().cpu().double().flatten(), b.detach().cpu().double().flatten()
    rms = b.pow(2).mean().sqrt().item()
    return (a - b).abs().max().item() / (rms + 1e-30), rms


def _fma32(a, b, c):
    """float32 fma(a, b, c) emulated through float64 (the product of two floats is exact there)."""
    return (a.double() * b.double() + c.double()).float()


def _device_forward_values(m, sd, x, bsz, t, v):
    """What the device's ReLU / max-pool compared, reproduced bit for bit on the host: z (conv outputs) and
    y = fma(z, scale, shift) of the three blocks, NCHW.  z2 / z3 are read from the workspace; z1 (never stored on the
    device) is the same chain of nine fmas per output that conv1's kernels evaluate; scale / shift are the device's."""
    lib = _native.lib()
    offs = (C.c_size_t * 40)()
    lib.sir_model_train_workspace_offsets(get_featurizer().handle, bsz, t, offs, 40)
    bn = m._sir_train["ws"].buf[offs[12]: offs[12] + 4 * 448].view(torch.float32).cpu()
    scale, shift = bn[:224], bn[224:448]
    xp = torch.nn.functional.pad(x.float(), (1, 1, 1, 1))                     # [B, 66, T + 2]
    w1 = sd["conv1.weight"].float().view(32, 9)
    z1 = torch.zeros(bsz, 32, 64, t)
    for ky in range(3):
        for kx in range(3):
            z1 = _fma32(xp[:, None, ky:ky + 64, kx:kx + t], w1[None, :, ky * 3 + kx, None, None], z1)
    nchw = lambda a: a.permute(0, 3, 1, 2)
    z = {1: z1, 2: nchw(v["z2"]), 3: nchw(v["z3"])}
    y = {i: _fma32(z[i], scale[o:o + c][None, :, None, None], shift[o:o + c][None, :, None, None])
         for i, o, c in ((1, 0, 32), (2, 32, 64), (3, 96, 128))}
    return z, y


@pytest.fixture(scope="module")
def sd():
    return synth.synth_state_dict(31, seed=0)


def _hip_step(sd, x, y):
    m = _model(sd)
    logits = m(x.to(DEV))
    loss = train_ops.fused_cross_entropy(logits, y.to(DEV))
    loss.backward()
    torch.cuda.synchronize()
    return m, logits, loss


def test_train_forward_backward_stages(sd):
    """Stage-by-stage diagnostics against the oracle (printed), then the assertions."""
    inp = cases.model_inputs()
    x, y = inp["x_train8"], inp["y_train8"]
    m, logits, loss = _hip_step(sd, x, y)
    st0, st = {}, {}
    ref_loss, _, ref_stats, ref_logits = model_ref.loss_and_grads(sd, x, y, stages=st0)
    v = _views(m, 8, 200)
    nhwc = lambda t: t.permute(0, 2, 3, 1)
    nchw = lambda t: t.permute(0, 3, 1, 2)
    fwd = {"a1": nhwc(st0["conv1"]), "a2": nhwc(st0["conv2"]), "x0": st0["gru_in"], "y0": st0["gru_l0"],
           "y1": st0["gru_l1"], "ctx": st0["ctx"]}
    # backward: the oracle differentiates at the device's conv2/conv3 outputs (see the module docstring)
    zo, yo = _device_forward_values(m, sd, x, 8, 200, v)
    _, ref_grads, _, _ = model_ref.loss_and_grads(sd, x, y, stages=st, z_override=zo, y_override=yo)
    bwd = {"dy1": st["d_gru_l1"], "dy0": st["d_gru_l0"],
           "dx0": st["d_gru_in"], "da2": nhwc(st["d_conv2"]), "da1": nhwc(st["d_conv1"])}
    report = {}
    for k, r in {**fwd, **bwd}.items():
        report[k] = _rel(v[k] / _loss_scale(8) if k in bwd else v[k], r)[0]
    report["logits"] = _rel(logits, ref_logits)[0]
    print("train stage errors (max|a-b|/rms):", {k: f"{e:.1e}" for k, e in report.items()})
    gerr = {}
    for (name, p) in m.named_parameters():
        gerr[name] = _rel(p.grad, ref_grads[name])[0] if ref_grads[name].abs().max() > 1e-7 else float((p.grad.cpu() - ref_grads[name]).abs().max())
    print("grad errors:", {k: f"{e:.1e}" for k, e in gerr.items()})
    assert abs(loss.item() - ref_loss.item()) < 1e-5
    for k, e in report.items():
        assert e < 2e-3, (k, e)
    for k, e in gerr.items():
        assert e < 2e-3 or k == "attention.bias", (k, e)
    for i in (1, 2, 3):
        bn = getattr(m, f"bn{i}")
        assert torch.allclose(bn.running_mean.cpu(), ref_stats[f"bn{i}.running_mean"], rtol=1e-4, atol=1e-6)
        assert torch.allclose(bn.running_var.cpu(), ref_stats[f"bn{i}.running_var"], rtol=1e-4, atol=1e-6)
        assert int(bn.num_batches_tracked) == 1


def test_train_step_matches_reference_golden(sd, model_golden):
    """loss, logits, sampled gradients, BN running stats and post-Adam parameters of the REFERENCE model."""
    inp = cases.model_inputs()
    m = _model(sd)
    opt = FusedAdam(m.parameters(), lr=cases.LR, weight_decay=cases.WEIGHT_DECAY)
    opt.zero_grad(set_to_none=True)
    logits = m(inp["x_train8"].to(DEV))
    loss = train_ops.fused_cross_entropy(logits, inp["y_train8"].to(DEV))
    loss.backward()
    assert abs(loss.item() - float(model_golden["train8_loss"])) < 1e-5
    np.testing.assert_allclose(logits.detach().cpu().numpy(), model_golden["train8_logits"], rtol=0, atol=2e-5)
    for name, p in m.named_parameters():
        g = p.grad.detach().cpu().flatten()
        idx = cases.sample_indices(name, g.numel())
        norm = float(model_golden[f"grad_norm/{name}"])
        rms = norm / np.sqrt(g.numel())
        tol = 2e-2 if name.startswith(("conv", "bn")) else 2e-3      # pooling ties, see the module docstring
        assert np.abs(g[idx].numpy() - model_golden[f"grad_samp/{name}"]).max() <= tol * rms + 1e-7, name
        assert abs(g.double().norm().item() - norm) <= 0.5 * tol * norm + 1e-7, name
    opt.step()
    torch.cuda.synchronize()
    for name, p in m.named_parameters():
        flat = p.detach().cpu().flatten()
        idx = cases.sample_indices(name, flat.numel())
        # first Adam step moves every element by ~lr regardless of gradient scale; sign flips of
        # near-zero gradients are the only way to differ by more than rounding
        d = np.abs(flat[idx].numpy() - model_golden[f"adam_samp/{name}"])
        assert np.quantile(d, 0.9) <= 2e-6 and d.max() <= 2.1 * cases.LR, (name, d.max())
    for i in (1, 2, 3):
        bn = getattr(m, f"bn{i}")
        np.testing.assert_allclose(bn.running_mean.cpu().numpy(), model_golden[f"bn{i}.running_mean"], rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(bn.running_var.cpu().numpy(), model_golden[f"bn{i}.running_var"], rtol=1e-4, atol=1e-6)


def test_ten_step_trajectory_matches_reference_golden(sd, traj_golden):
    """K = 10 Adam steps on the same batch against the trajectory the REFERENCE's own CNNAudioGRU + torch.optim.Adam produced
    (tests/golden/make_golden.py::make_trajectory_golden; train.py:90-107, :246-250; gru.dropout = 0): the loss of every step
    within 1e-4 abs, the sampled parameters after step 10 within the one-step Adam bound of the test above scaled by K, BN
    running statistics as after one step.  ("matched accuracy" over more than one step: the dataset is absent, so the
    reference's own trajectory on a fixed batch is the available form.)"""
    inp = cases.model_inputs()
    k = int(traj_golden["steps"])
    m = _model(sd)
    opt = FusedAdam(m.parameters(), lr=cases.LR, weight_decay=cases.WEIGHT_DECAY)
    x, y = inp["x_train8"].to(DEV), inp["y_train8"].to(DEV)
    losses = []
    for _ in range(k):
        opt.zero_grad(set_to_none=True)
        logits = m(x)
        loss = train_ops.fused_cross_entropy(logits, y)
        loss.backward()
        opt.step()
        losses.append(loss.item())
    torch.cuda.synchronize()
    dl = np.abs(np.asarray(losses, np.float32) - traj_golden["loss"])
    print("trajectory: max |loss - reference| per step", dl.max(), "losses", [f"{v:.5f}" for v in losses])
    assert dl.max() <= 1e-4, dl
    assert losses[-1] < losses[0] - 0.5                              # (the reference drops 3.50 -> 2.62 on this batch)
    np.testing.assert_allclose(logits.detach().cpu().numpy(), traj_golden["final_logits"], rtol=0, atol=2e-4)
    worst = 0.0
    for name, p in m.named_parameters():
        flat = p.detach().cpu().flatten()
        idx = cases.sample_indices(name, flat.numel())
        d = np.abs(flat[idx].numpy() - traj_golden[f"param_samp/{name}"])
        worst = max(worst, float(d.max()))
        assert np.quantile(d, 0.9) <= 2e-6 * k and d.max() <= 2.1 * cases.LR * k, (name, np.quantile(d, 0.9), d.max())
        dn = (p.detach().cpu() - sd[name]).double().norm().item()
        assert abs(dn - float(traj_golden[f"param_delta_norm/{name}"])) <= 0.02 * float(traj_golden[f"param_delta_norm/{name}"]) + 1e-7, name
    print("trajectory: worst sampled parameter difference after", k, "steps:", worst)
    for i in (1, 2, 3):
        # the running statistics are functions of weights that may differ by 2.1 * lr per step (sign flips of near-zero gradients,
        # the bound above): one-step tolerances scaled by K on the relative part, 5e-6 per step on the absolute one
        bn = getattr(m, f"bn{i}")
        np.testing.assert_allclose(bn.running_mean.cpu().numpy(), traj_golden[f"bn{i}.running_mean"], rtol=1e-4 * k, atol=5e-6 * k)
        np.testing.assert_allclose(bn.running_var.cpu().numpy(), traj_golden[f"bn{i}.running_var"], rtol=1e-4 * k, atol=5e-6 * k)
        assert int(bn.num_batches_tracked) == k


def test_adam_kernel_vs_oracle_over_steps():
    torch.manual_seed(0)
    ps = [torch.randn(n) for n in (5, 4097, 70000)]
    gs = [[torch.randn_like(p) * (10.0 ** (-i)) for p in ps] for i in range(3)]
    dev_ps = [torch.nn.Parameter(p.clone().to(DEV)) for p in ps]
    opt = FusedAdam(dev_ps, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2)
    ref = [(p.clone(), torch.zeros_like(p), torch.zeros_like(p)) for p in ps]
    for step in range(3):
        for p, g in zip(dev_ps, gs[step]):
            p.grad = g.to(DEV)
        opt.step()
        ref = [model_ref.adam_step(p, g, m, v, step + 1, 1e-3, 0.9, 0.999, 1e-8, 1e-2)
               for (p, m, v), g in zip(ref, gs[step])]
    for p, (rp, _, _) in zip(dev_ps, ref):
        assert (p.detach().cpu() - rp).abs().max() < 2e-6


def test_loss_decreases_with_dropout_and_full_batch(sd):
    """A few real steps at batch 64 with the inter-layer dropout on: finite, decreasing loss."""
    m = _model(sd, dropout=0.5)
    opt = FusedAdam(m.parameters(), lr=1e-3, weight_decay=1e-4)
    x = cases.varied_features(64, 200, seed=11).to(DEV)
    y = synth.synth_labels(64, 31, seed=5).to(DEV)
    losses = []
    for _ in range(8):
        opt.zero_grad(set_to_none=True)
        loss = train_ops.fused_cross_entropy(m(x), y)
        loss.backward()
        opt.step()
        losses.append(loss.item())
    print("losses:", [f"{v:.3f}" for v in losses])
    assert all(np.isfinite(losses)) and losses[-1] < losses[0]
    m.eval()
    assert torch.isfinite(m(x)).all()


@pytest.mark.parametrize("bsz,t", [(5, 96), (18, 200), (3, 94), (2, 61)])
def test_ragged_training_step_vs_oracle(sd, bsz, t):
    """Backward parity at batch sizes that leave GRU groups / pairs partly empty and at a shorter sequence."""
    x = cases.varied_features(bsz, t, seed=77 + bsz)
    y = synth.synth_labels(bsz, 31, seed=bsz)
    m, logits, loss = _hip_step(sd, x, y)
    v = _views(m, bsz, t) if t == 200 else None
    if v is not None:                      # oracle backward at the device's forward values: no pooling / ReLU tie ambiguity
        zo, yo = _device_forward_values(m, sd, x, bsz, t, v)
        ref_loss, ref_grads, _, ref_logits = model_ref.loss_and_grads(sd, x, y, z_override=zo, y_override=yo)
        tol = 2e-3
    else:                                  # no stage views for this shape: plain oracle, looser on the CNN (pooling ties)
        ref_loss, ref_grads, _, ref_logits = model_ref.loss_and_grads(sd, x, y)
        tol = 2e-2
    assert abs(loss.item() - ref_loss.item()) < 1e-5
    assert (logits.detach().cpu() - ref_logits).abs().max() < 2e-5
    for name, p in m.named_parameters():
        if ref_grads[name].abs().max() <= 1e-7:
            continue
        e = _rel(p.grad, ref_grads[name])[0]
        assert e < (tol if name.startswith(("conv", "bn")) else 2e-3), (name, e)


def test_backward_in_two_halves_is_bit_identical(sd, monkeypatch):
    """sir_model_train_bwd_part(HEAD_GRU) + (CNN) == sir_model_train_bwd (the split only exists to start the gradient
    exchange early); checked by forcing the data-parallel code path with a one-rank gloo group."""
    import torch.distributed as dist
    inp = cases.model_inputs()
    x, y = inp["x_train8"], inp["y_train8"]
    m, _, _ = _hip_step(sd, x, y)
    ref = {n: p.grad.clone() for n, p in m.named_parameters()}
    import os
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29655")
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        monkeypatch.setattr(train_ops, "world_size", lambda: 2)          # take the two-bucket branch
        monkeypatch.setattr(train_ops, "OVERLAP_GRAD_EXCHANGE", True)
        m2, _, _ = _hip_step(sd, x, y)
        for n, p in m2.named_parameters():
            # sum over a 1-rank group = identity, then * 1/2 from the patched world size
            assert torch.equal(p.grad * 2.0, ref[n]), n
    finally:
        dist.destroy_process_group()


def test_gradient_hand_over_matches_autograd_accumulation(sd, monkeypatch):
    """After zero_grad(set_to_none=True) the flat-buffer views become .grad directly (no AccumulateGrad clones); the values
    must equal the autograd path's, a second backward without zero_grad must ACCUMULATE (not alias the buffer the
    kernels overwrite), and a frozen parameter must stay without gradient."""
    inp = cases.model_inputs()
    x, y = inp["x_train8"], inp["y_train8"]
    monkeypatch.setattr(train_ops, "HAND_OVER_GRADS", False)
    m0, _, _ = _hip_step(sd, x, y)
    ref = {n: p.grad.clone() for n, p in m0.named_parameters()}
    monkeypatch.setattr(train_ops, "HAND_OVER_GRADS", True)
    m1, _, _ = _hip_step(sd, x, y)
    flat = m1._sir_train["grads"].flat
    for n, p in m1.named_parameters():
        assert torch.equal(p.grad, ref[n]), n
        assert flat.data_ptr() <= p.grad.data_ptr() < flat.data_ptr() + 4 * flat.numel(), n     # handed over, not cloned
    # accumulate: same batch again without zero_grad (dropout off in _hip_step's module => identical gradients)
    loss = train_ops.fused_cross_entropy(m1(x.to(DEV)), y.to(DEV))
    loss.backward()
    for n, p in m1.named_parameters():
        assert torch.allclose(p.grad, 2.0 * ref[n], rtol=1e-6, atol=1e-9), n
    # frozen parameter
    for p in m1.parameters():
        p.grad = None
    m1.fc.bias.requires_grad_(False)
    train_ops.fused_cross_entropy(m1(x.to(DEV)), y.to(DEV)).backward()
    assert m1.fc.bias.grad is None
    assert torch.equal(m1.fc.weight.grad, ref["fc.weight"])


def test_small_bn_gamma_channels_vs_oracle(sd):
    """The BatchNorm backward sums come from the pooled activations (xhat = (a - beta) / gamma); channels whose gamma is too
    small for that division -- here exactly zero and 1e-4 in bn2 and bn3 -- must take the z path and still match the oracle."""
    sd2 = {k: v.clone() for k, v in sd.items()}
    sd2["bn2.weight"][:3] = torch.tensor([0.0, 1e-4, -1e-4])
    sd2["bn2.bias"][:3] = torch.tensor([0.3, 0.2, 0.1])
    sd2["bn3.weight"][5:8] = torch.tensor([0.0, 1e-4, -2e-4])
    sd2["bn3.bias"][5:8] = torch.tensor([0.25, 0.5, 0.05])
    bsz, t = 6, 200
    x = cases.varied_features(bsz, t, seed=5)
    y = synth.synth_labels(bsz, 31, seed=6)
    m, logits, loss = _hip_step(sd2, x, y)
    v = _views(m, bsz, t)
    zo, yo = _device_forward_values(m, sd2, x, bsz, t, v)
    ref_loss, ref_grads, _, ref_logits = model_ref.loss_and_grads(sd2, x, y, z_override=zo, y_override=yo)
    assert abs(loss.item() - ref_loss.item()) < 1e-5
    for name in ("bn2.weight", "bn2.bias", "bn3.weight", "bn3.bias", "conv2.weight", "conv3.weight", "conv1.weight"):
        e = _rel(dict(m.named_parameters())[name].grad, ref_grads[name])[0]
        assert e < 2e-3, (name, e)


def test_waveform_epoch_with_prefetch_matches_inline_steps(sd):
    """train_epoch_waveforms (features one batch ahead on a side stream) == features + step in line, bit for bit:
    same kernels, only the stream changes."""
    from sir_amd.scripts.train import train_epoch_waveforms
    nb, bsz = 4, 6
    waves = [(synth.synth_clips(bsz, 30000 + 1000 * i, seed=40 + i) * 32767).round().to(torch.int16) for i in range(nb)]
    lens = [torch.tensor([w.shape[1] - 37 * j for j in range(bsz)], dtype=torch.int32) for w in waves]
    labels = [synth.synth_labels(bsz, 31, seed=50 + i) for i in range(nb)]

    ma = _model(sd)
    oa = FusedAdam(ma.parameters(), lr=1e-3, weight_decay=1e-4)
    fz = get_featurizer()
    ref_losses = []
    for w, l, y in zip(waves, lens, labels):
        x = fz(w.to(DEV), l.to(DEV), t_pad=200)
        oa.zero_grad(set_to_none=True)
        loss = train_ops.fused_cross_entropy(ma(x), y.to(DEV))
        loss.backward()
        oa.step()
        ref_losses.append(loss.detach())
    ref_mean = torch.stack(ref_losses).mean().item()

    mb = _model(sd)
    ob = FusedAdam(mb.parameters(), lr=1e-3, weight_decay=1e-4)
    mean = train_epoch_waveforms(mb, list(zip(waves, lens, labels)), ob, torch.nn.CrossEntropyLoss(), DEV)
    torch.cuda.synchronize()
    assert mean == ref_mean
    for (n, pa), (_, pb) in zip(ma.named_parameters(), mb.named_parameters()):
        assert torch.equal(pa, pb), n


# ---- round 2: dropout-on parity, the bench's own batch size, augmentation-fed step, cache / status hazards -------------
import host_rng  # noqa: E402


def _oracle_f64(sd, x, y, zo, yo, dropout_mask=None, stages=None):
    """The oracle's backward in FLOAT64 at the device's forward values (z / y overrides, themselves fp32 device values).  At B = 256 a
    convolution weight gradient is a sum of 3.3 M products that largely cancel: the fp32 CPU backward carries up to ~1e-2 of
    rms in it, and how much depends on how many threads split the sum (seen: 6.6e-4 with the box's default pool, 8.3e-3 once an
    earlier test had capped the pool at the CPU quota) -- the reference for these two tests is therefore computed in double."""
    d = lambda t: t.double() if torch.is_tensor(t) and t.is_floating_point() else t
    sd64 = {k: d(v) for k, v in sd.items()}
    zo64 = {k: d(v) for k, v in zo.items()}
    yo64 = {k: d(v) for k, v in yo.items()}
    loss, grads, stats, logits = model_ref.loss_and_grads(sd64, d(x), y, dropout_mask=d(dropout_mask) if dropout_mask is not None else None,
                                                          stages=stages, z_override=zo64, y_override=yo64)
    return loss, grads, stats, logits


def _grad_errors(m, ref_grads):
    out = {}
    for name, p in m.named_parameters():
        if ref_grads[name].abs().max() <= 1e-7:
            out[name] = float((p.grad.cpu() - ref_grads[name]).abs().max())
        else:
            out[name] = _rel(p.grad, ref_grads[name])[0]
    return out


def test_dropout_on_training_step_vs_oracle(sd):
    """The reference trains with nn.GRU(dropout=0.5) (models/models.py:26-33).  The device mask is a pure function of
    (seed, element index): rebuilt on the host (tests/host_rng.py), checked bit for bit against what the dropout kernel
    wrote (y0d = keep ? y0 / (1 - p) : 0), fed to the oracle's ``dropout_mask``; all 29 gradients at 2e-3 * rms."""
    inp = cases.model_inputs()
    x, y = inp["x_train8"], inp["y_train8"]
    p = 0.5
    m = _model(sd, dropout=p)
    logits = m(x.to(DEV))
    loss = train_ops.fused_cross_entropy(logits, y.to(DEV))
    loss.backward()
    torch.cuda.synchronize()
    seed, p_used = m._sir_last_dropout
    assert p_used == p
    bsz, t, s = 8, 200, 25
    keep = torch.from_numpy(host_rng.dropout_keep(seed, bsz * s * 512, p)).view(bsz, s, 512)
    assert 0.45 < keep.float().mean().item() < 0.55                      # mask rate
    lib = _native.lib()
    offs = (C.c_size_t * 40)()
    lib.sir_model_train_workspace_offsets(get_featurizer().handle, bsz, t, offs, 40)
    ws = m._sir_train["ws"].buf
    n = bsz * s * 512
    y0 = ws[offs[8]: offs[8] + 4 * n].view(torch.float32).view(bsz, s, 512).cpu()
    y0d = ws[offs[9]: offs[9] + 4 * n].view(torch.float32).view(bsz, s, 512).cpu()
    assert torch.equal(y0d, torch.where(keep, y0 * (1.0 / (1.0 - p)), torch.zeros_like(y0)))    # 1/(1-p) scale, same mask
    mask = keep.float() / (1.0 - p)
    v = _views(m, bsz, t)
    zo, yo = _device_forward_values(m, sd, x, bsz, t, v)
    ref_loss, ref_grads, _, ref_logits = model_ref.loss_and_grads(sd, x, y, dropout_mask=mask, z_override=zo, y_override=yo)
    assert abs(loss.item() - ref_loss.item()) < 1e-5
    assert (logits.detach().cpu() - ref_logits).abs().max() < 2e-5
    gerr = _grad_errors(m, ref_grads)
    print("dropout-on grad errors:", {k: f"{e:.1e}" for k, e in gerr.items()})
    for k, e in gerr.items():
        assert e < 2e-3 or k == "attention.bias", (k, e)
    # a second step draws a different mask, and another rank would too
    m(x.to(DEV))
    assert m._sir_last_dropout[0] != seed
    assert train_ops.dropout_seed(5, rank=0) != train_ops.dropout_seed(5, rank=1)


def test_training_step_at_bench_batch_256_vs_oracle(sd):
    """B = 256, T = 200 -- the shapes bench.py times: M = 6400 tokens takes the 128-row gemm_tn tiles, the split-K slab
    plan, the per-image wgrad slabs and the two-pass slab reduces that the small-batch tests never reach.  Same method
    as above (oracle differentiated at the device's z / y); BN running statistics and one Adam step included."""
    bsz, t = 256, 200
    x = cases.varied_features(bsz, t, seed=256)
    y = synth.synth_labels(bsz, 31, seed=257)
    m = _model(sd)
    opt = FusedAdam(m.parameters(), lr=cases.LR, weight_decay=cases.WEIGHT_DECAY)
    opt.zero_grad(set_to_none=True)
    logits = m(x.to(DEV))
    loss = train_ops.fused_cross_entropy(logits, y.to(DEV))
    loss.backward()
    torch.cuda.synchronize()
    v = _views(m, bsz, t)
    zo, yo = _device_forward_values(m, sd, x, bsz, t, v)
    del v
    ref_loss, ref_grads, ref_stats, ref_logits = _oracle_f64(sd, x, y, zo, yo)
    del zo, yo
    assert abs(loss.item() - ref_loss.item()) < 2e-5
    assert (logits.detach().cpu().double() - ref_logits).abs().max() < 5e-5
    gerr = _grad_errors(m, ref_grads)
    print("B=256 grad errors:", {k: f"{e:.1e}" for k, e in gerr.items()})
    for k, e in gerr.items():
        assert e < 2e-3 or k == "attention.bias", (k, e)
    for name, p in m.named_parameters():                                    # norms (a sampled-element check cannot see a lost slab)
        rn = ref_grads[name].double().norm().item()
        assert abs(p.grad.double().norm().item() - rn) <= 1e-3 * rn + 1e-7, name
    for i in (1, 2, 3):
        bn = getattr(m, f"bn{i}")
        assert torch.allclose(bn.running_mean.cpu().double(), ref_stats[f"bn{i}.running_mean"], rtol=1e-4, atol=1e-6)
        assert torch.allclose(bn.running_var.cpu().double(), ref_stats[f"bn{i}.running_var"], rtol=1e-4, atol=1e-6)
    before = {n: p.detach().cpu().clone() for n, p in m.named_parameters()}
    opt.step()
    torch.cuda.synchronize()
    for name, p in m.named_parameters():
        g = ref_grads[name].float()
        exp, _, _ = model_ref.adam_step(before[name], g, torch.zeros_like(g), torch.zeros_like(g), 1, cases.LR, 0.9, 0.999,
                                        1e-8, cases.WEIGHT_DECAY)
        d = (p.detach().cpu() - exp).abs().flatten()
        # the first Adam step moves every element by ~lr; only a sign flip of a near-zero gradient can differ by more
        assert torch.quantile(d[:: max(1, d.numel() // 100000)], 0.99).item() <= 2e-6 and d.max().item() <= 2.1 * cases.LR, name


def test_bench_configuration_batch_256_with_dropout_vs_oracle(sd):
    """The configuration bench.py times -- B = 256 AND dropout 0.5: layer 1's input gradient then runs as two K halves whose
    ordered add applies the dropout mask (``dx_halves_add_kernel``), a branch the B = 8 / 64 dropout tests never reach
    (ADVICE r3).  Host-rebuilt mask, oracle at the device's z / y, all 29 gradients at 2e-3 * rms."""
    bsz, t, s, p = 256, 200, 25, 0.5
    x = cases.varied_features(bsz, t, seed=258)
    y = synth.synth_labels(bsz, 31, seed=259)
    m = _model(sd, dropout=p)
    logits = m(x.to(DEV))
    loss = train_ops.fused_cross_entropy(logits, y.to(DEV))
    loss.backward()
    torch.cuda.synchronize()
    seed, p_used = m._sir_last_dropout
    assert p_used == p
    keep = torch.from_numpy(host_rng.dropout_keep(seed, bsz * s * 512, p)).view(bsz, s, 512)
    mask = keep.float() / (1.0 - p)
    v = _views(m, bsz, t)
    dy0 = v["dy0"] / _loss_scale(bsz)                                     # gradient wrt the un-dropped layer-0 output
    assert (dy0[~keep] == 0).all() and (dy0[keep] != 0).float().mean() > 0.99     # the mask reached the split-K add
    zo, yo = _device_forward_values(m, sd, x, bsz, t, v)
    del v
    st = {}
    ref_loss, ref_grads, _, ref_logits = _oracle_f64(sd, x, y, zo, yo, dropout_mask=mask, stages=st)
    del zo, yo
    assert abs(loss.item() - ref_loss.item()) < 2e-5
    assert (logits.detach().cpu().double() - ref_logits).abs().max() < 5e-5
    assert _rel(dy0, st["d_gru_l0"])[0] < 2e-3
    gerr = _grad_errors(m, ref_grads)
    print("B=256 dropout-on grad errors:", {k: f"{e:.1e}" for k, e in gerr.items()})
    for k, e in gerr.items():
        assert e < 2e-3 or k == "attention.bias", (k, e)


def _host_augmented(wave, lengths, shifts, sigmas, seed):
    out = []
    for b in range(wave.shape[0]):
        n = lengths[b]
        xx = wave[b, :n].clone()
        yy = torch.zeros_like(xx)
        sft = shifts[b]
        if sft >= 0:
            yy[sft:] = xx[: n - sft]
        else:
            yy[: n + sft] = xx[-sft:]
        if sigmas[b] > 0:
            yy = yy + float(sigmas[b]) * torch.from_numpy(host_rng.gauss_noise(seed, b, n))
        out.append(yy)
    return out


def test_step_on_fused_augmentation_equals_step_on_host_augmented_input(sd):
    """BASELINE configs[4]: time shift + noise inside the feature kernel, SpecAugment masks in the normalise pass.  The
    same augmentation applied on the HOST (shift exact; noise from the restated counter RNG) gives the same features
    (oracle, 1e-4 * max(1, |b|)) and the same training step (loss 1e-5, gradients 2e-3 * rms)."""
    from oracle import features_ref
    bsz, L = 6, 48000
    wave = synth.synth_clips(bsz, L, seed=61)
    lengths = [48000, 47000, 48000, 40001, 48000, 30000]
    shifts = [0, 1600, -2400, 777, -4799, 0]
    sigmas = [0.0, 0.005, 0.0, 0.01, 0.001, 0.0099]
    seed = (7 << 32) ^ 3
    tm = torch.tensor([[10, 15], [0, 0], [80, 19], [0, 0], [5, 1], [0, 0]], dtype=torch.int32)
    fm = torch.tensor([[0, 0], [50, 10], [3, 9], [0, 0], [0, 0], [60, 4]], dtype=torch.int32)
    fz = get_featurizer()
    dl = torch.tensor(lengths, dtype=torch.int32, device=DEV)
    feats_dev = fz(wave.to(DEV), dl, t_pad=200, shift=torch.tensor(shifts, dtype=torch.int32),
                   noise_sigma=torch.tensor(sigmas), noise_seed=seed, time_mask=tm, freq_mask=fm).clone()
    host = _host_augmented(wave, lengths, shifts, sigmas, seed)
    # (1) features: oracle on the host-augmented waveform, masks applied on the host
    for b in range(bsz):
        ref = features_ref.pad_or_trim(features_ref.extract_features_f32(host[b]))
        t = 1 + lengths[b] // 512
        ref[:, int(tm[b, 0]): int(tm[b, 0] + tm[b, 1])] = 0
        ref[int(fm[b, 0]): int(fm[b, 0] + fm[b, 1]), :] = 0
        ref[:, t:] = 0
        err = ((feats_dev[b].cpu() - ref).abs() / ref.abs().clamp(min=1.0)).max().item()
        assert err <= 1e-4, (b, err)
    # (2) the step: device features of the host-augmented waveform (+ masks) vs the fused form
    hw = torch.zeros(bsz, L)
    for b in range(bsz):
        hw[b, : lengths[b]] = host[b]
    feats_host = fz(hw.to(DEV), dl, t_pad=200, time_mask=tm, freq_mask=fm).clone()
    assert (feats_host - feats_dev).abs().max().item() <= 1e-4
    y = synth.synth_labels(bsz, 31, seed=62)
    out = []
    for f in (feats_dev, feats_host):
        m = _model(sd)
        loss = train_ops.fused_cross_entropy(m(f), y.to(DEV))
        loss.backward()
        torch.cuda.synchronize()
        out.append((loss.item(), {n: p.grad.detach().cpu().clone() for n, p in m.named_parameters()}))
    assert abs(out[0][0] - out[1][0]) < 1e-5
    for n in out[0][1]:
        ga, gb = out[0][1][n], out[1][1][n]
        if gb.abs().max() <= 1e-7:
            continue
        # conv / BN gradients see ReLU / pooling ties flip between the two (1e-6-different) inputs: module docstring
        tol = 2e-2 if n.startswith(("conv", "bn")) else 2e-3
        assert _rel(ga, gb)[0] < tol, (n, _rel(ga, gb)[0])


def test_prepared_weight_cache_is_not_shared_between_models(sd):
    """two models built one after the other in the same place (same id(), same workspace address from
    the caching allocator, same B / T) must each get THEIR prepared weight layouts."""
    x = synth.synth_features(4, 200, seed=70)

    def run(seed):
        sdi = synth.synth_state_dict(31, seed=seed)
        mm = CNNAudioGRU(31)
        mm.load_state_dict(sdi)
        mm = mm.to(DEV).eval()
        out = mm(x.to(DEV)).cpu()
        return out, model_ref.forward(sdi, x)

    outs = []
    for seed in (0, 1, 2, 3):
        got, ref = run(seed)
        assert (got - ref).abs().max() < 2e-5, seed
        outs.append(got)
    assert (outs[0] - outs[1]).abs().max() > 1e-3                       # the weight sets really differ
    # writes through .data (broadcast_module_) do not bump torch's version counters: the epoch bump must invalidate
    sdi = synth.synth_state_dict(31, seed=5)
    mm = CNNAudioGRU(31)
    mm.load_state_dict(synth.synth_state_dict(31, seed=4))
    mm = mm.to(DEV).eval()
    mm(x.to(DEV))
    for k, t in list(mm.named_parameters()) + list(mm.named_buffers()):
        if k in sdi:
            t.data.copy_(sdi[k])
    from sir_amd import ops
    ops.bump_weights_epoch()
    assert (mm(x.to(DEV)).cpu() - model_ref.forward(sdi, x)).abs().max() < 2e-5


def test_gru_timeout_is_reported_not_swallowed(tmp_path):
    """a recurrence whose inter-workgroup exchange times out sets the handle's status word; the host
    must see SirError at the next check instead of silently wrong logits.  The timeout is injected with SIR_GRU_DBG
    (bit 3: quarter 3 never publishes, bit 4: short spin limit) in a child process (the switch is read once)."""
    import os
    import subprocess
    import sys
    code = (
        "import sys, torch\n"
        f"sys.path.insert(0, {cases.ROOT!r})\n"
        "from sir_amd import _native, ops, synth\n"
        "from sir_amd.models.models import CNNAudioGRU\n"
        "m = CNNAudioGRU(31); m.load_state_dict(synth.synth_state_dict(31, seed=0)); m = m.cuda().eval()\n"
        "m(synth.synth_features(20, 200, seed=1).cuda())\n"
        "try:\n"
        "    ops.check_status()\n"
        "except _native.SirError as e:\n"
        "    print('RAISED', e); ops.check_status(); print('CLEARED')\n"
    )
    env = dict(os.environ, SIR_GRU_DBG="24")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "RAISED" in r.stdout and "timed out" in r.stdout and "CLEARED" in r.stdout, r.stdout
    env = dict(os.environ, SIR_GRU_DBG="0")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "RAISED" not in r.stdout, (r.stdout, r.stderr[-2000:])


@pytest.mark.parametrize("bsz", [256, 21])
def test_two_stream_backward_is_bit_identical_to_one_stream(sd, bsz):
    """The backward's weight-gradient launches run on a second, library-owned stream (SIR_BWD_STREAMS, csrc/model_train.hip); while every
    kernel is timed (sir_profile_enable mode 1) the same call keeps everything on the caller's stream.  No reduction depends on the
    order in which the two streams finish, so every parameter gradient must be BIT-identical between the two forms, run after run (a
    missing fork / join edge or a shared slab shows up here as a difference)."""
    import ctypes as C
    from sir_amd import _native
    from sir_amd.featurizer import get_featurizer
    lib, h = _native.lib(), get_featurizer().handle
    x = cases.varied_features(bsz, 200, seed=900 + bsz).to(DEV)
    y = synth.synth_labels(bsz, 31, seed=901 + bsz).to(DEV)
    m = _model(sd)                                   # dropout 0: the mask is keyed on a step counter

    def grads():
        m.zero_grad(set_to_none=True)
        loss = train_ops.fused_cross_entropy(m(x), y)
        loss.backward()
        torch.cuda.synchronize()
        return {n: p.grad.clone() for n, p in m.named_parameters()}

    two = [grads() for _ in range(3)]
    _native.check(lib.sir_profile_enable(h, 1, -1), "sir_profile_enable")
    try:
        one = grads()
    finally:
        nk = lib.sir_profile_kernel_count()
        ms, cnt = (C.c_double * nk)(), (C.c_int64 * nk)()
        lib.sir_profile_collect(h, ms, cnt, nk)
        _native.check(lib.sir_profile_enable(h, 0, -1), "sir_profile_enable")
    for n in one:
        for k, g in enumerate(two):
            assert torch.equal(g[n], one[n]), (n, k, (g[n] - one[n]).abs().max().item())
