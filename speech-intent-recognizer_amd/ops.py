"""Host glue between ``torch`` tensors and the C ABI for the model path.

Only pointer plumbing lives here: tensors stay owned by torch, the HIP library
gets ``data_ptr()``s plus the current stream.  No arithmetic is done in Python
and nothing falls back to ``torch.nn`` compute.
"""
import ctypes as C
import itertools

import torch

from . import _native
from .featurizer import get_featurizer

GRU_SUFFIXES = ("_l0", "_l0_reverse", "_l1", "_l1_reverse")
WS_NAMES = ("conv1", "conv2", "gru_in", "gi", "gru_l0", "gru_l1", "ctx")


def _f32c(t, name):
    if t.dtype != torch.float32 or not t.is_contiguous():
        raise _native.SirError(f"{name} must be contiguous float32")
    if not t.is_cuda:
        raise _native.SirError(f"{name} is not on a HIP device (no CPU fallback)")
    return t


def weights_struct(mod):
    """sir_model_weights filled with the module's parameter/buffer pointers (reference key names)."""
    w = _native.ModelWeights()
    keep = []

    def p(t, name):
        t = _f32c(t.detach(), name)
        keep.append(t)
        return t.data_ptr()

    for i in range(3):
        conv, bn = getattr(mod, f"conv{i + 1}"), getattr(mod, f"bn{i + 1}")
        w.conv_w[i] = p(conv.weight, f"conv{i + 1}.weight")
        w.bn_w[i] = p(bn.weight, f"bn{i + 1}.weight")
        w.bn_b[i] = p(bn.bias, f"bn{i + 1}.bias")
        w.bn_mean[i] = p(bn.running_mean, f"bn{i + 1}.running_mean")
        w.bn_var[i] = p(bn.running_var, f"bn{i + 1}.running_var")
    for i, suf in enumerate(GRU_SUFFIXES):
        w.gru_w_ih[i] = p(getattr(mod.gru, "weight_ih" + suf), "gru.weight_ih" + suf)
        w.gru_w_hh[i] = p(getattr(mod.gru, "weight_hh" + suf), "gru.weight_hh" + suf)
        w.gru_b_ih[i] = p(getattr(mod.gru, "bias_ih" + suf), "gru.bias_ih" + suf)
        w.gru_b_hh[i] = p(getattr(mod.gru, "bias_hh" + suf), "gru.bias_hh" + suf)
    w.attn_w = p(mod.attention.weight, "attention.weight")
    w.attn_b = p(mod.attention.bias, "attention.bias")
    w.fc_w = p(mod.fc.weight, "fc.weight")
    w.fc_b = p(mod.fc.bias, "fc.bias")
    w.num_classes = mod.fc.weight.shape[0]
    return w, keep


def cached_weights(mod):
    """The pointer struct is rebuilt only when the module's storage may have moved
    (``CNNAudioGRU._apply`` drops the cache); in-place updates keep the pointers valid."""
    cache = getattr(mod, "_sir_wcache", None)
    if cache is None:
        cache = weights_struct(mod)
        mod._sir_wcache = cache
        mod._sir_wptrs = None
    return cache


_weights_epoch = [1]
_model_tokens = itertools.count(1)


def new_model_token():
    """Process-unique id of one set of weights: ``CNNAudioGRU`` takes one in ``__init__`` and a fresh one whenever its
    storage or contents are replaced wholesale (``_apply``, ``load_state_dict``).  ``id(module)`` is NOT such an id: a
    freed module's address is reused by the next one built in the same place."""
    return next(_model_tokens)


def bump_weights_epoch():
    """Called by everything that rewrites parameters/buffers behind torch's back (the HIP Adam step, the
    training forward's running-statistics update, ``broadcast_module_``'s writes through ``.data``), so that
    cached derived weight layouts are rebuilt."""
    _weights_epoch[0] += 1


def weights_version(mod, keep, workspace=None):
    """Non-zero 64-bit fingerprint of (the global epoch, the model's token, torch's in-place version counters, the
    storage addresses of every parameter / buffer, the generation of the workspace that holds the prepared layouts).
    Two different models, a moved or reloaded model, or a re-allocated workspace can therefore never present the
    library with the version under which another set of weights was prepared."""
    token = getattr(mod, "_sir_token", None)
    if token is None:
        token = mod._sir_token = new_model_token()
    ptrs = getattr(mod, "_sir_wptrs", None)
    if ptrs is None or len(ptrs) != len(keep):
        ptrs = mod._sir_wptrs = tuple(t.data_ptr() for t in keep)      # constant while the pointer struct is cached
    key = (_weights_epoch[0], token, workspace.generation if workspace is not None else 0,
           tuple(t._version for t in keep), ptrs)
    return (hash(key) & 0xFFFFFFFFFFFFFFFF) or 1


class Workspace:
    """Grow-only device scratch, 256-byte aligned (torch's caching allocator aligns to 512).  ``generation`` counts the
    allocations: prepared weight layouts live inside the buffer, so a new buffer (even at the old address) invalidates
    them -- ``weights_version`` folds the generation in."""
    _generations = itertools.count(1)

    def __init__(self):
        self.buf = None
        self.generation = 0

    def get(self, nbytes, device):
        if self.buf is None or self.buf.numel() < nbytes or self.buf.device != device:
            self.buf = torch.empty(max(nbytes, 256), dtype=torch.uint8, device=device)
            self.generation = next(Workspace._generations)
        return self.buf


def check_status(all_ranks=True):
    """Raise ``SirError`` if a GRU recurrence launched since the last check timed out in its inter-workgroup exchange
    (``sir_check_status``: its outputs were invalid) or ``sir_ce_loss`` met a label outside ``[0, num_classes)``.
    Host-synchronous on the current stream -- call it where the host waits anyway (end of an epoch, after a batch of
    predictions has been copied back).  In a data-parallel job the flag is MAX-reduced over the ranks first, so that
    EVERY rank raises (a rank that raised alone would leave the others blocked in the next collective until the RCCL
    timeout); every rank must therefore call it at the same point."""
    import torch.distributed as dist
    lib = _native.lib()
    rc = lib.sir_check_status(get_featurizer().handle, _native.current_stream_ptr())
    if all_ranks and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        flag = torch.tensor([1 if rc != _native.SIR_OK else 0], dtype=torch.int32, device="cuda")
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        if rc == _native.SIR_OK and int(flag.item()):
            raise _native.SirError("sir_check_status failed on another rank of this job (a GRU recurrence timed out or a "
                                   "label was out of range there): this rank stops with it")
    _native.check(rc, "sir_check_status")


def _as_features(x):
    if x.dim() == 4:
        if x.shape[1] != 1:
            raise _native.SirError("input_channels must be 1")
        x = x[:, 0]
    if x.dim() != 3 or x.shape[1] != 64:
        raise _native.SirError(f"expected [B,64,T] or [B,1,64,T], got {tuple(x.shape)}")
    if x.shape[2] < 8:
        raise _native.SirError("need at least 8 frames")
    return _f32c(x.contiguous(), "input")


def model_infer(mod, x, workspace, want_argmax=False, debug=None):
    """eval-mode forward of ``CNNAudioGRU`` on the GPU -> logits [B, C] (and argmax int64 [B])."""
    _native.require_hip(x)
    x = _as_features(x)
    bsz, _, t = x.shape
    lib = _native.lib()
    h = get_featurizer().handle
    need = lib.sir_model_workspace_bytes(h, bsz, t, 0)
    if need == 0:
        raise _native.SirError(f"unsupported shape batch={bsz} frames={t}")
    ws = workspace.get(need, x.device)
    w, keep = cached_weights(mod)
    logits = torch.empty((bsz, w.num_classes), dtype=torch.float32, device=x.device)
    amax = torch.empty((bsz,), dtype=torch.int64, device=x.device) if want_argmax else None
    lib.sir_model_set_weights_version(h, weights_version(mod, keep, workspace))
    rc = lib.sir_model_infer(h, C.byref(w), x.data_ptr(), bsz, t, logits.data_ptr(),
                             amax.data_ptr() if amax is not None else None, ws.data_ptr(), ws.numel(),
                             _native.current_stream_ptr())
    _native.check(rc, "sir_model_infer")
    if debug is not None:
        debug.update(stage_views(ws, bsz, t))
    del keep
    return (logits, amax) if want_argmax else logits


def stage_views(ws, bsz, t):
    """Views of the intermediate buffers inside the workspace (for stage-level parity tests)."""
    lib = _native.lib()
    offs = (C.c_size_t * 16)()
    n = lib.sir_model_workspace_offsets(get_featurizer().handle, bsz, t, 0, offs, 16)
    if n <= 0:
        raise _native.SirError("sir_model_workspace_offsets failed")
    wp1, wp2 = t // 2, t // 4
    s = wp2 // 2
    shapes = {"conv1": (bsz, 32, wp1, 32), "conv2": (bsz, 16, wp2, 64), "gru_in": (bsz, s, 1024),
              "gi": (bsz * s, 1536), "gru_l0": (bsz, s, 512), "gru_l1": (bsz, s, 512), "ctx": (bsz, 512)}
    out = {}
    for i, name in enumerate(WS_NAMES):
        shp = shapes[name]
        numel = 1
        for v in shp:
            numel *= v
        out[name] = ws[offs[i]: offs[i] + 4 * numel].view(torch.float32).view(shp)
    return out
