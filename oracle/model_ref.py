"""Oracle: CNN x3 -> 2-layer BiGRU -> attention pool -> Linear, functional fp32 (CPU).

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).  PINNED against the
reference's own ``CNNAudioGRU`` by ``tests/golden/model_golden.npz``.

A from-scratch restatement over a plain ``state_dict`` -- explicit BatchNorm and an
explicit GRU cell loop instead of ``nn.BatchNorm2d`` / ``nn.GRU`` -- of
  /root/reference/models/models.py:41-68   (forward)
  /root/reference/models/models.py:10-39   (layer shapes / state_dict keys)
  /root/reference/scripts/train.py:104-107 (CE loss, backward, optimiser step)
  /root/reference/scripts/train.py:246-250 (Adam with coupled L2 weight decay)
Gradients come from torch autograd over this functional forward (plain PyTorch
fp32), which is the reference's own backward on CPU.
"""
import math

import torch
import torch.nn.functional as F

BN_EPS = 1e-5
BN_MOMENTUM = 0.1
HIDDEN = 256

PARAM_KEYS = (
    ["conv1.weight", "bn1.weight", "bn1.bias", "conv2.weight", "bn2.weight", "bn2.bias",
     "conv3.weight", "bn3.weight", "bn3.bias"]
    + [f"gru.{n}_l{l}{s}" for l in (0, 1) for s in ("", "_reverse")
       for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")]
    + ["attention.weight", "attention.bias", "fc.weight", "fc.bias"]
)


def _bn(x, sd, i, train, new_stats=None):
    g, b = sd[f"bn{i}.weight"], sd[f"bn{i}.bias"]
    if train:
        mean = x.mean(dim=(0, 2, 3))
        var_b = x.var(dim=(0, 2, 3), unbiased=False)
        if new_stats is not None:
            n = x.numel() // x.shape[1]
            var_u = var_b * (n / (n - 1))
            new_stats[f"bn{i}.running_mean"] = ((1 - BN_MOMENTUM) * sd[f"bn{i}.running_mean"]
                                                + BN_MOMENTUM * mean).detach()
            new_stats[f"bn{i}.running_var"] = ((1 - BN_MOMENTUM) * sd[f"bn{i}.running_var"]
                                               + BN_MOMENTUM * var_u).detach()
    else:
        mean, var_b = sd[f"bn{i}.running_mean"], sd[f"bn{i}.running_var"]
    inv = torch.rsqrt(var_b + BN_EPS)
    return (x - mean[None, :, None, None]) * (inv * g)[None, :, None, None] + b[None, :, None, None]


def cnn_stack(x, sd, train=False, new_stats=None, stages=None, z_override=None, y_override=None):
    """x [B,1,H,W] -> [B,128,H/8,W/8]  (models.py:50-52).

    ``z_override`` {block: [B,C,H,W]} substitutes the VALUES of a block's convolution output
    (straight-through: the gradient still flows through the oracle's own convolution).  ReLU and
    max-pool are not differentiable at ties, so two correct fp32 forwards that differ in the last
    bits can route a gradient to different pixels; evaluating the oracle's backward at the device's
    forward values removes that ambiguity from a backward parity check.
    ``y_override`` {block: [B,C,H,W]} does the same for the BatchNorm OUTPUT (the values ReLU and the
    pooling actually compare): with the device's z alone, the oracle's own rounding of scale / shift can
    still break a near-tie the other way."""
    for i in (1, 2, 3):
        x = F.conv2d(x, sd[f"conv{i}.weight"], bias=None, stride=1, padding=1)
        if z_override is not None and i in z_override:
            x = x + (z_override[i] - x).detach()
        x = _bn(x, sd, i, train, new_stats)
        if y_override is not None and i in y_override:
            x = x + (y_override[i] - x).detach()
        x = F.max_pool2d(torch.relu(x), 2)
        if stages is not None:
            stages[f"conv{i}"] = x
    return x


def gru_direction(x, w_ih, w_hh, b_ih, b_hh, reverse):
    """x [B,T,I] -> [B,T,H]; gate order r,z,n; h0 = 0 (torch.nn.GRU semantics)."""
    bsz, steps, _ = x.shape
    gi = x @ w_ih.t() + b_ih
    h = x.new_zeros(bsz, HIDDEN)
    outs = [None] * steps
    order = range(steps - 1, -1, -1) if reverse else range(steps)
    for t in order:
        gh = h @ w_hh.t() + b_hh
        i_r, i_z, i_n = gi[:, t].chunk(3, dim=1)
        h_r, h_z, h_n = gh.chunk(3, dim=1)
        r = torch.sigmoid(i_r + h_r)
        z = torch.sigmoid(i_z + h_z)
        n = torch.tanh(i_n + r * h_n)
        h = (1.0 - z) * n + z * h
        outs[t] = h
    return torch.stack(outs, dim=1)


def bigru(x, sd, dropout_mask=None, stages=None):
    """2-layer bidirectional GRU, batch_first.  ``dropout_mask`` ([B,T,512], already scaled
    by 1/(1-p)) stands for the inter-layer dropout of models.py:32; None = no dropout."""
    for layer in (0, 1):
        fwd = gru_direction(x, sd[f"gru.weight_ih_l{layer}"], sd[f"gru.weight_hh_l{layer}"],
                            sd[f"gru.bias_ih_l{layer}"], sd[f"gru.bias_hh_l{layer}"], False)
        rev = gru_direction(x, sd[f"gru.weight_ih_l{layer}_reverse"], sd[f"gru.weight_hh_l{layer}_reverse"],
                            sd[f"gru.bias_ih_l{layer}_reverse"], sd[f"gru.bias_hh_l{layer}_reverse"], True)
        x = torch.cat([fwd, rev], dim=2)
        if stages is not None:
            stages[f"gru_l{layer}"] = x
        if layer == 0 and dropout_mask is not None:
            x = x * dropout_mask
    return x


def forward(sd, x, train=False, new_stats=None, dropout_mask=None, stages=None, z_override=None, y_override=None):
    """x [B,64,T] or [B,1,64,T] -> logits [B,C]  (models.py:41-68)."""
    if x.dim() == 3:
        x = x.unsqueeze(1)
    x = cnn_stack(x, sd, train, new_stats, stages, z_override, y_override)
    b, c, h, w = x.shape
    seq = x.permute(0, 3, 1, 2).contiguous().view(b, w, c * h)     # feature = c*h_dim + h
    if stages is not None:
        stages["gru_in"] = seq
    y = bigru(seq, sd, dropout_mask, stages)
    if stages is not None:
        stages["gru_out"] = y
    scores = y @ sd["attention.weight"].t() + sd["attention.bias"]  # [B,T,1]
    attn = torch.softmax(scores, dim=1)
    ctx = (y * attn).sum(dim=1)
    if stages is not None:
        stages["ctx"] = ctx
    return ctx @ sd["fc.weight"].t() + sd["fc.bias"]


def loss_and_grads(sd, x, labels, dropout_mask=None, stages=None, z_override=None, y_override=None):
    """One training-mode forward/backward (CE mean).  Returns loss, grads{key}, new BN stats, logits.
    With ``stages`` (a dict) the intermediate activations are stored under their names and their
    gradients under ``"d_" + name`` (for stage-level checks of the HIP backward)."""
    params = {k: sd[k].detach().clone().requires_grad_(True) for k in PARAM_KEYS}
    full = dict(sd)
    full.update(params)
    new_stats = {}
    logits = forward(full, x, train=True, new_stats=new_stats, dropout_mask=dropout_mask, stages=stages,
                     z_override=z_override, y_override=y_override)
    if stages is not None:
        for t in stages.values():
            t.retain_grad()
    loss = F.cross_entropy(logits, labels)
    grads = torch.autograd.grad(loss, [params[k] for k in PARAM_KEYS], retain_graph=stages is not None)
    if stages is not None:
        names = list(stages)
        inter = torch.autograd.grad(loss, [stages[n] for n in names], allow_unused=True)
        for n, g in zip(names, inter):
            stages["d_" + n] = g
        for n in names:
            stages[n] = stages[n].detach()
    return loss.detach(), dict(zip(PARAM_KEYS, grads)), new_stats, logits.detach()


def adam_step(p, g, m, v, step, lr, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0):
    """torch.optim.Adam (non-AMSGrad, coupled L2) single-tensor update; returns (p, m, v)."""
    if weight_decay != 0.0:
        g = g + weight_decay * p
    m = beta1 * m + (1.0 - beta1) * g
    v = beta2 * v + (1.0 - beta2) * g * g
    bc1 = 1.0 - beta1 ** step
    bc2 = 1.0 - beta2 ** step
    denom = v.sqrt() / math.sqrt(bc2) + eps
    p = p - (lr / bc1) * (m / denom)
    return p, m, v


class FastRef:
    """Same forward through torch's fused CPU kernels (``F.batch_norm`` and a stock ``nn.GRU`` loaded
    with the ``gru.*`` tensors) -- what the reference's module executes on CPU, and therefore the
    fair thing to TIME as the CPU baseline (bench.py).  Checked against ``forward`` above in
    tests/test_oracle_golden.py."""

    def __init__(self, sd):
        self.sd = sd
        self.gru = torch.nn.GRU(input_size=sd["gru.weight_ih_l0"].shape[1], hidden_size=HIDDEN, num_layers=2,
                                batch_first=True, bidirectional=True)
        self.gru.load_state_dict({k[len("gru."):]: v for k, v in sd.items() if k.startswith("gru.")})
        self.gru.eval()

    @torch.no_grad()
    def __call__(self, x):
        sd = self.sd
        if x.dim() == 3:
            x = x.unsqueeze(1)
        for i in (1, 2, 3):
            x = F.conv2d(x, sd[f"conv{i}.weight"], None, 1, 1)
            x = F.batch_norm(x, sd[f"bn{i}.running_mean"], sd[f"bn{i}.running_var"], sd[f"bn{i}.weight"],
                             sd[f"bn{i}.bias"], False, BN_MOMENTUM, BN_EPS)
            x = F.max_pool2d(F.relu(x), 2)
        b, c, h, w = x.shape
        y, _ = self.gru(x.permute(0, 3, 1, 2).reshape(b, w, c * h))
        attn = torch.softmax(F.linear(y, sd["attention.weight"], sd["attention.bias"]), dim=1)
        return F.linear((y * attn).sum(dim=1), sd["fc.weight"], sd["fc.bias"])


class TrainRef(torch.nn.Module):
    """Stock torch.nn layers under the reference's parameter names (models.py:6-39 restated: three
    conv/BN/ReLU/pool blocks, 2-layer bidirectional GRU with inter-layer dropout 0.5, attention pooling,
    linear head) -- what the reference's ``train_epoch`` differentiates on CPU.  Used ONLY to time the CPU
    training baseline in bench.py; the parity checks use ``loss_and_grads`` above."""

    def __init__(self, sd, num_classes):
        super().__init__()
        nn = torch.nn
        self.conv1 = nn.Conv2d(1, 32, 3, padding=1, bias=False)
        self.conv2 = nn.Conv2d(32, 64, 3, padding=1, bias=False)
        self.conv3 = nn.Conv2d(64, 128, 3, padding=1, bias=False)
        self.bn1, self.bn2, self.bn3 = nn.BatchNorm2d(32), nn.BatchNorm2d(64), nn.BatchNorm2d(128)
        self.gru = nn.GRU(1024, HIDDEN, num_layers=2, batch_first=True, bidirectional=True, dropout=0.5)
        self.attention = nn.Linear(2 * HIDDEN, 1)
        self.fc = nn.Linear(2 * HIDDEN, num_classes)
        self.load_state_dict({k: v for k, v in sd.items()}, strict=True)

    def forward(self, x):
        if x.dim() == 3:
            x = x.unsqueeze(1)
        for conv, bn in ((self.conv1, self.bn1), (self.conv2, self.bn2), (self.conv3, self.bn3)):
            x = F.max_pool2d(F.relu(bn(conv(x))), 2)
        b, c, h, w = x.shape
        y, _ = self.gru(x.permute(0, 3, 1, 2).reshape(b, w, c * h))
        attn = torch.softmax(self.attention(y), dim=1)
        return self.fc((y * attn).sum(dim=1))
