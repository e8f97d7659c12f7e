"""``FusedAdam``: torch.optim.Adam semantics (coupled L2 ``weight_decay``, bias correction, no
AMSGrad -- what the reference builds at scripts/train.py:246-250) with the update of ALL parameter
tensors done by one ``sir_adam_step`` launch (multi-tensor HIP kernel).  State lives in two flat
fp32 buffers (exp_avg, exp_avg_sq) per parameter group."""
import ctypes as C

import torch

from . import _native, ops
from .featurizer import get_featurizer


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        if lr < 0 or eps < 0 or not 0 <= betas[0] < 1 or not 0 <= betas[1] < 1 or weight_decay < 0:
            raise ValueError("invalid Adam hyper-parameter")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))

    def _group_state(self, group):
        gs = self.state.setdefault("_sir_group_%d" % id(group), {})
        if "step" not in gs:
            ps = group["params"]
            n = sum(p.numel() for p in ps)
            gs["step"] = 0
            gs["exp_avg"] = torch.zeros(n, dtype=torch.float32, device=ps[0].device)
            gs["exp_avg_sq"] = torch.zeros(n, dtype=torch.float32, device=ps[0].device)
            offs, off = [], 0
            for p in ps:
                offs.append(off)
                off += p.numel()
            gs["offsets"] = offs
        return gs

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = _native.lib()
        h = get_featurizer().handle
        for group in self.param_groups:
            gs = self._group_state(group)
            gs["step"] += 1
            items = [(p, o) for p, o in zip(group["params"], gs["offsets"]) if p.grad is not None]
            for start in range(0, len(items), 32):
                chunk = items[start:start + 32]
                n = len(chunk)
                P, G, M, V = ((C.c_void_p * n)() for _ in range(4))
                N = (C.c_int64 * n)()
                keep = []
                for i, (p, o) in enumerate(chunk):
                    if p.dtype != torch.float32 or not p.is_contiguous():
                        raise _native.SirError("FusedAdam needs contiguous float32 parameters")
                    _native.require_hip(p)
                    g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                    keep.append(g)
                    P[i], G[i] = p.data_ptr(), g.data_ptr()
                    M[i] = gs["exp_avg"].data_ptr() + 4 * o
                    V[i] = gs["exp_avg_sq"].data_ptr() + 4 * o
                    N[i] = p.numel()
                rc = lib.sir_adam_step(h, n, P, G, M, V, N, gs["step"], float(group["lr"]), float(group["betas"][0]),
                                       float(group["betas"][1]), float(group["eps"]), float(group["weight_decay"]),
                                       _native.current_stream_ptr())
                _native.check(rc, "sir_adam_step")
        ops.bump_weights_epoch()
        return loss
