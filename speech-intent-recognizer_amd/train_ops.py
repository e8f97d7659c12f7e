"""Training-step glue: ``torch.autograd`` nodes around the HIP training kernels.

``forward_train(model, x)`` returns logits whose backward runs ``sir_model_train_bwd`` and hands
autograd the 29 parameter gradients as views of ONE flat fp32 buffer (3 261 184 elements); with
``WORLD_SIZE > 1`` that buffer is averaged over ranks with a single RCCL all-reduce before the
views are returned, so the update equals single-GPU training on the global batch.
``fused_cross_entropy`` is the HIP form of the reference's ``nn.CrossEntropyLoss()`` (train.py:242).
Only pointers move through Python; no arithmetic of the step is done by torch ops.
"""
import ctypes as C
import itertools
import os

import torch

from . import _native, ops
from .dist_utils import (ShardSampler, all_reduce_mean_, all_reduce_sum_, broadcast_module_,  # noqa: F401
                         init_distributed, limit_host_threads, shutdown_distributed, world_size)
from .featurizer import get_featurizer

_seed_counter = itertools.count(1)
OVERLAP_GRAD_EXCHANGE = os.environ.get("SIR_DDP_OVERLAP", "1") != "0"
HAND_OVER_GRADS = os.environ.get("SIR_HAND_OVER_GRADS", "1") != "0"
# take the data-parallel exchange path even in a ONE-rank process group (bench.py's `rccl_world1` leg and the nccl tests
# on a one-GPU box: the collectives, their stream ordering against the backward kernels and the final scale all run;
# the sum over one rank is the identity)
FORCE_EXCHANGE = os.environ.get("SIR_DDP_FORCE", "0") == "1"


def dropout_seed(step_counter, rank=None):
    """64-bit key of one step's inter-layer dropout mask: the process-local step counter and the data-parallel rank
    (every rank must draw a DIFFERENT mask for its shard, as independent ``nn.GRU`` replicas would)."""
    if rank is None:
        rank = torch.distributed.get_rank() if world_size() > 1 else 0
    return (step_counter * 0x9E3779B97F4A7C15 + rank * 0xD1B54A32D192ED03) % (1 << 64)


def _has_grad_hooks(p):
    return bool(getattr(p, "_backward_hooks", None)) or bool(getattr(p, "_post_accumulate_grad_hooks", None))


def param_list(mod):
    """Parameters in ``named_parameters`` order (== the reference's state_dict order minus buffers)."""
    return [p for _, p in mod.named_parameters()]


class GradBuffer:
    """One flat gradient buffer with per-parameter views + the matching ``sir_model_grads`` struct."""

    def __init__(self, mod):
        params = param_list(mod)
        dev = params[0].device
        self.flat = torch.zeros(sum(p.numel() for p in params), dtype=torch.float32, device=dev)
        self.views, off = [], 0
        for p in params:
            self.views.append(self.flat[off: off + p.numel()].view_as(p))
            off += p.numel()
        by_name = {n: v for (n, _), v in zip(mod.named_parameters(), self.views)}
        g = _native.ModelGrads()
        for i in range(3):
            g.conv_w[i] = by_name[f"conv{i + 1}.weight"].data_ptr()
            g.bn_w[i] = by_name[f"bn{i + 1}.weight"].data_ptr()
            g.bn_b[i] = by_name[f"bn{i + 1}.bias"].data_ptr()
        for i, suf in enumerate(ops.GRU_SUFFIXES):
            g.gru_w_ih[i] = by_name["gru.weight_ih" + suf].data_ptr()
            g.gru_w_hh[i] = by_name["gru.weight_hh" + suf].data_ptr()
            g.gru_b_ih[i] = by_name["gru.bias_ih" + suf].data_ptr()
            g.gru_b_hh[i] = by_name["gru.bias_hh" + suf].data_ptr()
        g.attn_w = by_name["attention.weight"].data_ptr()
        g.attn_b = by_name["attention.bias"].data_ptr()
        g.fc_w = by_name["fc.weight"].data_ptr()
        g.fc_b = by_name["fc.bias"].data_ptr()
        self.struct = g
        # the conv / BatchNorm gradients come first in named_parameters order; everything behind them (GRU, attention, fc)
        # is final after the first half of the backward
        self.n_cnn = sum(p.numel() for n, p in mod.named_parameters() if n.startswith(("conv", "bn")))
        names = [n for n, _ in mod.named_parameters()]
        first_other = next(i for i, n in enumerate(names) if not n.startswith(("conv", "bn")))
        assert all(n.startswith(("conv", "bn")) for n in names[:first_other]) and \
            not any(n.startswith(("conv", "bn")) for n in names[first_other:]), "parameter order changed"


def _train_state(mod):
    st = getattr(mod, "_sir_train", None)
    if st is None or st["grads"].flat.device != next(mod.parameters()).device:
        st = {"grads": GradBuffer(mod), "ws": ops.Workspace()}
        mod._sir_train = st
    return st


def _bn_ptr_arrays(mod):
    rm = (C.c_void_p * 3)(*[getattr(mod, f"bn{i}").running_mean.data_ptr() for i in (1, 2, 3)])
    rv = (C.c_void_p * 3)(*[getattr(mod, f"bn{i}").running_var.data_ptr() for i in (1, 2, 3)])
    return rm, rv


class _TrainStep(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, mod, dropout_p, *params):
        lib = _native.lib()
        h = get_featurizer().handle
        bsz, _, t = x.shape
        st = _train_state(mod)
        need = lib.sir_model_workspace_bytes(h, bsz, t, 1)
        if need == 0:
            raise _native.SirError(f"unsupported shape batch={bsz} frames={t}")
        ws = st["ws"].get(need, x.device)
        w, keep = ops.cached_weights(mod)
        rm, rv = _bn_ptr_arrays(mod)
        logits = torch.empty((bsz, w.num_classes), dtype=torch.float32, device=x.device)
        seed = dropout_seed(next(_seed_counter))
        mod._sir_last_dropout = (seed, float(dropout_p))      # lets tests rebuild the mask (tests/dropout_host.py)
        momentum = float(mod.bn1.momentum if mod.bn1.momentum is not None else 0.1)
        rc = lib.sir_model_train_fwd(h, C.byref(w), rm, rv, x.data_ptr(), bsz, t, momentum, float(dropout_p), seed,
                                     logits.data_ptr(), ws.data_ptr(), ws.numel(), _native.current_stream_ptr())
        _native.check(rc, "sir_model_train_fwd")
        ops.bump_weights_epoch()                     # BN running statistics were updated in place
        torch._foreach_add_([getattr(mod, f"bn{i}").num_batches_tracked for i in (1, 2, 3)], 1)
        ctx.mod, ctx.x, ctx.seed, ctx.dropout_p, ctx.ws = mod, x, seed, float(dropout_p), ws
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        lib = _native.lib()
        mod, x = ctx.mod, ctx.x
        h = get_featurizer().handle
        st = _train_state(mod)
        w, keep = ops.cached_weights(mod)
        dlogits = dlogits.contiguous()
        bsz, _, t = x.shape
        grads = st["grads"]
        params = param_list(mod)
        # a .grad left over from the previous step that still aliases the flat buffer (no zero_grad in between:
        # gradient accumulation) must be detached from it before the kernels overwrite the buffer
        for p, v in zip(params, grads.views):
            if p.grad is not None and p.grad.data_ptr() == v.data_ptr():
                p.grad = p.grad.clone()

        def run(part):
            rc = lib.sir_model_train_bwd_part(h, C.byref(w), x.data_ptr(), dlogits.data_ptr(), bsz, t, ctx.dropout_p, ctx.seed,
                                              C.byref(grads.struct), ctx.ws.data_ptr(), ctx.ws.numel(), part,
                                              _native.current_stream_ptr())
            _native.check(rc, "sir_model_train_bwd_part")

        _exchange_and_scale(grads, run)
        need = ctx.needs_input_grad[3:]
        if HAND_OVER_GRADS and all(p.grad is None and not _has_grad_hooks(p) for p in params):
            # zero_grad(set_to_none=True) (train.py:90): the views of the flat buffer BECOME the .grad tensors; returning
            # them through autograd would make AccumulateGrad clone all 29 of them (29 copy launches per step)
            for p, v, n in zip(params, grads.views, need):
                if n:
                    p.grad = v
            return (None, None, None) + (None,) * len(params)
        return (None, None, None) + tuple(v if n else None for v, n in zip(grads.views, need))


def _exchange_and_scale(grads, run):
    """The per-step gradient exchange around the two halves of the backward (``run(part)`` launches one half, or
    nothing for a rank that has no batch)."""
    import torch.distributed as dist
    world = world_size()
    forced = FORCE_EXCHANGE and dist.is_available() and dist.is_initialized()
    if (world > 1 or forced) and OVERLAP_GRAD_EXCHANGE:
        # data parallel: the GRU / attention / fc gradients (96 % of the 13 MB) are final after the first half of the
        # backward; their all-reduce runs beside the conv backward, the small conv / BN bucket follows
        run(_native.BWD_HEAD_GRU)
        tail = grads.flat[grads.n_cnn:]
        work = dist.all_reduce(tail, op=dist.ReduceOp.SUM, async_op=True)
        run(_native.BWD_CNN)
        dist.all_reduce(grads.flat[:grads.n_cnn], op=dist.ReduceOp.SUM)
        work.wait()
        grads.flat.mul_(1.0 / world)
    else:
        run(_native.BWD_ALL)
        if forced:                              # un-overlapped form of the same exchange
            dist.all_reduce(grads.flat, op=dist.ReduceOp.SUM)
            grads.flat.mul_(1.0 / world)
        else:
            all_reduce_mean_(grads.flat)        # the one exchange step of data-parallel training


def zero_contribution_step(mod):
    """Data-parallel step of a rank whose batch is empty (``collate_fn`` dropped every item, train.py:67-68 / :82-83):
    the other ranks are inside the gradient all-reduce, so this rank joins the same collectives with a zero gradient
    and ends up with the same averaged ``.grad`` as they do (the caller then runs ``optimizer.step()`` like everyone
    else, keeping the replicas identical).  Without it the job would hang on the mismatched collective."""
    st = _train_state(mod)
    grads = st["grads"]
    params = param_list(mod)
    for p, v in zip(params, grads.views):
        if p.grad is not None and p.grad.data_ptr() == v.data_ptr():
            p.grad = None
    grads.flat.zero_()
    _exchange_and_scale(grads, lambda part: None)
    for p, v in zip(params, grads.views):
        if p.requires_grad:
            p.grad = v


def forward_train(mod, x):
    """Training-mode forward of ``CNNAudioGRU`` (batch-statistics BN, inter-layer dropout
    ``mod.gru.dropout``), differentiable wrt the module's parameters."""
    _native.require_hip(x)
    x = ops._as_features(x)
    return _TrainStep.apply(x, mod, float(mod.gru.dropout), *param_list(mod))


class _FusedCE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, labels):
        lib = _native.lib()
        logits = logits.contiguous()
        labels = labels.to(torch.int64).contiguous()
        _native.require_hip(logits, labels)
        bsz, ncls = logits.shape
        loss = torch.empty((), dtype=torch.float32, device=logits.device)
        dlogits = torch.empty_like(logits) if ctx.needs_input_grad[0] else None
        rc = lib.sir_ce_loss(get_featurizer().handle, logits.data_ptr(), labels.data_ptr(), bsz, ncls, loss.data_ptr(),
                             dlogits.data_ptr() if dlogits is not None else None, 1.0, _native.current_stream_ptr())
        _native.check(rc, "sir_ce_loss")
        ctx.dlogits = dlogits
        return loss

    @staticmethod
    def backward(ctx, grad_out):
        # loss.backward() passes 1.0; a general scalar is applied by the (tiny) multiply below
        return ctx.dlogits * grad_out, None


def fused_cross_entropy(logits, labels):
    """``nn.CrossEntropyLoss()`` (mean over the batch) computed by ``sir_ce_loss``."""
    return _FusedCE.apply(logits, labels)
