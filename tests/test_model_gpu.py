"""GPU parity: HIP inference path (sir_model_infer through the C ABI, behind the reference's
CNNAudioGRU surface) vs the CPU oracle and the reference-generated golden vectors.

Tolerance: fp32 throughout; logits |a-b| <= 2e-4 (sums of up to 1024 fp32 products in a different
order than MKL), intermediate stages 1e-4 * max(1,|b|); predicted intent indices identical."""
import numpy as np
import pytest
import torch

import cases
from oracle import model_ref
from sir_amd import _native, synth
from sir_amd.models.models import CNNAudioGRU

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _model(sd, num_classes=31):
    m = CNNAudioGRU(num_classes)
    m.load_state_dict(sd)
    return m.to(DEV).eval()


def _maxerr(a, b):
    a = a.detach().cpu().double()
    b = b.detach().cpu().double()
    return ((a - b).abs() / b.abs().clamp(min=1.0)).max().item()


@pytest.fixture(scope="module")
def sd():
    return synth.synth_state_dict(31, seed=0)


def test_stages_vs_oracle(sd):
    from sir_amd import ops
    x = cases.model_inputs()["x_eval8"]
    m = _model(sd)
    dbg = {}
    logits = ops.model_infer(m, x.to(DEV), m._ws, debug=dbg)
    torch.cuda.synchronize()
    st = {}
    with torch.no_grad():
        ref = model_ref.forward(sd, x, stages=st)
    errs = {
        "conv1": _maxerr(dbg["conv1"], st["conv1"].permute(0, 2, 3, 1)),
        "conv2": _maxerr(dbg["conv2"], st["conv2"].permute(0, 2, 3, 1)),
        "gru_in": _maxerr(dbg["gru_in"], st["gru_in"]),
        "gru_l0": _maxerr(dbg["gru_l0"], st["gru_l0"]),
        "gru_l1": _maxerr(dbg["gru_l1"], st["gru_l1"]),
        "ctx": _maxerr(dbg["ctx"], st["ctx"]),
        "logits": _maxerr(logits, ref),
    }
    print("stage max rel-abs errors:", {k: f"{v:.2e}" for k, v in errs.items()})
    for k, v in errs.items():
        assert v <= 1e-4, (k, v, errs)


def test_eval_golden_from_reference(sd, model_golden):
    inp = cases.model_inputs()
    m = _model(sd)
    lg8, am8 = m.predict(inp["x_eval8"].to(DEV))
    lg1, am1 = m.predict(inp["x_eval1_t94"].to(DEV))          # un-padded 4-D input, T = 94
    np.testing.assert_allclose(lg8.cpu().numpy(), model_golden["eval8_logits"], rtol=0, atol=2e-5)
    np.testing.assert_allclose(lg1.cpu().numpy(), model_golden["eval1_logits"], rtol=0, atol=2e-5)
    assert (am8.cpu().numpy() == model_golden["eval8_argmax"]).all()
    assert (am1.cpu().numpy() == model_golden["eval1_argmax"]).all()


def test_sharp_head_argmax_identical_to_reference(sd, model_golden):
    inp = cases.model_inputs()
    m = _model(cases.sharp_head(sd, model_golden["sharp_fc_bias"]))
    lg, am = m.predict(inp["x_sharp64"].to(DEV))
    np.testing.assert_allclose(lg.cpu().numpy(), model_golden["sharp64_logits"], rtol=0, atol=2e-3)
    assert (am.cpu().numpy() == model_golden["sharp64_argmax"]).all()


def test_full_batch_256_argmax_vs_oracle(sd, model_golden):
    """BASELINE batch (256 x [64,200]): every predicted index equals the oracle's, batch-position
    independence (row i of a 256 batch == the same clip run alone in a batch of 3)."""
    x = torch.cat([cases.varied_features(64, 200, seed=s) for s in (3, 4, 5, 6)])
    sds = cases.sharp_head(sd, model_golden["sharp_fc_bias"])
    m = _model(sds)
    lg, am = m.predict(x.to(DEV))
    with torch.no_grad():
        ref = model_ref.forward(sds, x)
    assert (am.cpu() == ref.argmax(1)).all()
    assert (lg.cpu() - ref).abs().max() < 2e-3
    lg3, _ = m.predict(x[[5, 100, 255]].to(DEV))
    assert torch.equal(lg3.cpu(), lg.cpu()[[5, 100, 255]])


def test_odd_batch_and_short_input(sd):
    m = _model(sd)
    x = synth.synth_features(5, 37, seed=21)       # odd batch (not a multiple of 4), odd frame count
    lg = m(x.to(DEV))
    with torch.no_grad():
        ref = model_ref.forward(sd, x)
    assert (lg.cpu() - ref).abs().max() < 2e-5


def test_cpu_input_is_refused(sd):
    from sir_amd import _native
    m = _model(sd)
    with pytest.raises(_native.SirError):
        m(torch.zeros(2, 64, 200))


def test_batch_pipeline_matches_single_stream(sd):
    """Batches alternating over HIP streams (sir_amd/pipeline.py) give bit-identical logits/argmax."""
    from sir_amd.pipeline import BatchPipeline
    m = CNNAudioGRU(31)
    m.load_state_dict(sd)
    m = m.to(DEV).eval()
    xs = [cases.varied_features(16, 200, seed=100 + i).to(DEV) for i in range(6)]
    ref = [m.predict(x) for x in xs]
    torch.cuda.synchronize()
    pipe = BatchPipeline(m, n_streams=3)
    outs = [pipe.infer(i, x) for i, x in enumerate(xs)]
    pipe.synchronize()
    for (l0, a0), (l1, a1) in zip(ref, outs):
        assert torch.equal(l0, l1) and torch.equal(a0, a1)


def test_library_pipeline_from_one_caller_stream(sd):
    """sir_pipeline (include/sir_hip.h): the LIBRARY owns the slot streams; a caller that stays on one stream, feeds raw
    waveforms and only joins at the end gets bit-identical features / logits / argmax for 1, 2 and 3 slots, and can consume
    the results on its own stream after join() without a host synchronisation."""
    from sir_amd.featurizer import get_featurizer
    from sir_amd.pipeline import BatchPipeline
    m = CNNAudioGRU(31)
    m.load_state_dict(sd)
    m = m.to(DEV).eval()
    fz = get_featurizer()
    waves = [synth.synth_clips(12, 30000 + 512 * i, seed=300 + i).to(DEV) for i in range(7)]
    ref = []
    for w in waves:
        f = fz(w, t_pad=200).clone()
        ref.append((f, *m.predict(f)))
    torch.cuda.synchronize()
    for n in (1, 2, 3):
        pipe = BatchPipeline(m, n_streams=n)
        bufs = [torch.empty(12, 64, 200, device=DEV) for _ in range(n)]
        acc = torch.zeros(12, 31, device=DEV)
        res = []
        for i, w in enumerate(waves):
            k = pipe.slot(i)
            f = pipe.features(i, w, None, t_pad=200, out=bufs[k])
            logits, amax = pipe.infer(i, f)
            res.append((logits, amax))
        pipe.join()                                   # caller's stream now waits for every slot: no host sync
        for logits, _ in res:
            acc += logits                              # consumed on the caller's stream
        torch.cuda.synchronize()
        for (f0, l0, a0), (l1, a1) in zip(ref, res):
            assert torch.equal(l0, l1) and torch.equal(a0, a1), n
        assert torch.equal(acc, sum(l for _, l, _ in ref)), n
    lib = _native.lib()
    import ctypes as C
    bad = C.c_void_p()
    assert lib.sir_pipeline_create(fz.handle, 5, C.byref(bad)) != 0 and b"n_slots" in lib.sir_last_error()


@pytest.mark.parametrize("bsz,t", [(37, 120), (17, 8), (3, 333), (130, 64)])
def test_ragged_shapes_vs_oracle(sd, bsz, t):
    """Batch sizes that do not fill the kernels' utterance groups (16 per GRU cluster, 4 per pair) and frame counts that
    do not fill the pixel tiles: logits within 2e-5 of the oracle, argmax identical."""
    x = cases.varied_features(bsz, t, seed=1000 + bsz)
    m = _model(sd)
    logits, amax = m.predict(x.to(DEV))
    with torch.no_grad():
        ref = model_ref.forward(sd, x)
    assert (logits.cpu() - ref).abs().max() <= 2e-5
    assert torch.equal(amax.cpu(), ref.argmax(1))
