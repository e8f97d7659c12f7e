"""The REAL collective backend (torch.distributed "nccl" = RCCL) on the one GPU of the test box: a one-rank group.

gloo (used by every other multi-rank test) stages device tensors through the host and is synchronous, so it cannot
exercise what the data-parallel training step relies on: RCCL kernels on RCCL's own stream ordered against the HIP
backward kernels on the caller's stream (train_ops._exchange_and_scale: async all-reduce of the GRU / head bucket beside
the conv backward, then the conv bucket, then ``work.wait()`` and one scale).  A one-rank group runs exactly that
machinery -- group init with ``device_id``, both all-reduces, the waits, ``all_gather_object``, ``broadcast`` -- with
the sum over one rank being the identity, so gradients must be BIT-identical to the single-process step
(the step it wraps is scripts/train.py:90-107).
"""
import os

import pytest
import torch
import torch.distributed as dist

import cases
from sir_amd import dist_utils, ops, synth, train_ops
from sir_amd.models.models import CNNAudioGRU
from sir_amd.optim import FusedAdam

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def sd():
    return synth.synth_state_dict(31, seed=0)


@pytest.fixture(scope="module")
def nccl_group():
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29877")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        yield dist.group.WORLD
    finally:
        dist.destroy_process_group()


def _model(sd, dropout=0.0):
    m = CNNAudioGRU(31)
    m.load_state_dict(sd)
    m = m.to(DEV).train()
    m.gru.dropout = dropout
    return m


def _step(sd, x, y):
    m = _model(sd)
    loss = train_ops.fused_cross_entropy(m(x.to(DEV)), y.to(DEV))
    loss.backward()
    torch.cuda.synchronize()
    return m, loss


def test_backend_is_rccl_and_object_collectives_work(nccl_group):
    assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
    out = [None]
    dist.all_gather_object(out, {"rank": 0, "dev": torch.cuda.get_device_name(0)})
    assert out[0]["rank"] == 0
    t = torch.arange(8, dtype=torch.float32, device=DEV)
    dist.all_reduce(t)
    dist.barrier()
    assert torch.equal(t.cpu(), torch.arange(8, dtype=torch.float32))


@pytest.mark.parametrize("overlap", [True, False])
def test_two_bucket_exchange_over_rccl_is_bit_identical(sd, nccl_group, monkeypatch, overlap):
    """The overlapped two-bucket branch (and the single-bucket one) over RCCL == the plain single-process backward."""
    inp = cases.model_inputs()
    x, y = inp["x_train8"], inp["y_train8"]
    monkeypatch.setattr(train_ops, "FORCE_EXCHANGE", False)
    m0, loss0 = _step(sd, x, y)
    ref = {n: p.grad.clone() for n, p in m0.named_parameters()}
    monkeypatch.setattr(train_ops, "FORCE_EXCHANGE", True)
    monkeypatch.setattr(train_ops, "OVERLAP_GRAD_EXCHANGE", overlap)
    for rep in range(3):                                   # repeated: an ordering bug between the streams is a race
        m1, loss1 = _step(sd, x, y)
        assert torch.equal(loss1, loss0)
        for n, p in m1.named_parameters():
            assert torch.equal(p.grad, ref[n]), (n, rep)


def test_full_batch_steps_with_adam_over_rccl_match_plain_steps(sd, nccl_group, monkeypatch):
    """Three optimizer steps at the bench's batch (256 x [64, 200], dropout off) with the exchange on every step: the
    parameters end up bit-identical to the run without a process group in the loop."""
    x = cases.varied_features(256, 200, seed=5).to(DEV)
    y = synth.synth_labels(256, 31, seed=9).to(DEV)

    def run(force):
        monkeypatch.setattr(train_ops, "FORCE_EXCHANGE", force)
        m = _model(sd)
        opt = FusedAdam(m.parameters(), lr=1e-3, weight_decay=1e-4)
        for _ in range(3):
            opt.zero_grad(set_to_none=True)
            train_ops.fused_cross_entropy(m(x), y).backward()
            opt.step()
        torch.cuda.synchronize()
        return m

    a, b = run(False), run(True)
    for (n, p), (_, q) in zip(a.named_parameters(), b.named_parameters()):
        assert torch.equal(p, q), n
    ops.check_status()


def test_zero_contribution_broadcast_and_validate_counts(sd, nccl_group, monkeypatch):
    """The other collectives of the training loop on RCCL: a rank without a batch joins the exchange with zeros
    (train.py:82-83's skipped batch), parameters / BN buffers are broadcast from rank 0, validation counts are summed."""
    monkeypatch.setattr(train_ops, "FORCE_EXCHANGE", True)
    m = _model(sd)
    train_ops.zero_contribution_step(m)
    torch.cuda.synchronize()
    for n, p in m.named_parameters():
        assert p.grad is not None and not p.grad.any(), n
    # broadcast through the patched world size (a 1-rank broadcast from 0 is the identity, the epoch must be bumped)
    monkeypatch.setattr(dist_utils, "world_size", lambda: 2)
    before = {n: t.clone() for n, t in m.state_dict().items()}
    e0 = ops._weights_epoch[0]
    dist_utils.broadcast_module_(m)
    assert ops._weights_epoch[0] == e0 + 1
    for n, t in m.state_dict().items():
        assert torch.equal(t, before[n]), n
    counts = torch.tensor([17, 32], dtype=torch.int64, device=DEV)
    dist_utils.all_reduce_sum_(counts)
    assert counts.tolist() == [17, 32]
    flat = torch.full((1000,), 3.0, device=DEV)
    dist_utils.all_reduce_mean_(flat)                      # sum over one rank, times 1 / (patched world of 2)
    assert torch.equal(flat, torch.full((1000,), 1.5, device=DEV))


def test_status_flag_is_shared_by_all_ranks(sd, nccl_group, monkeypatch):
    """ops.check_status() MAX-reduces the flag before raising (a rank raising alone leaves the others in
    the next collective); with one rank the reduce is the identity and the local error still surfaces."""
    from sir_amd import _native
    monkeypatch.setattr(dist, "get_world_size", lambda *a, **k: 2)
    ops.check_status()                                     # clean: the all-reduce runs and nothing raises
    logits = torch.randn(4, 31, device=DEV)
    train_ops.fused_cross_entropy(logits, torch.tensor([1, 2, 99, 3], device=DEV))
    with pytest.raises(_native.SirError, match="label outside"):
        ops.check_status()
    ops.check_status()
