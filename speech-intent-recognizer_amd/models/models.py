"""``CNNAudioGRU`` with the reference's constructor, ``forward`` signature and ``state_dict`` keys
(/root/reference/models/models.py:5-68), computed by hand-written HIP kernels on MI355X.

The sub-modules below exist only to own parameters/buffers under the reference's names
(``conv{1,2,3}.weight``, ``bn{1,2,3}.*``, ``gru.weight_ih_l0`` ..., ``attention.*``, ``fc.*``) and
to reproduce PyTorch's default initialisation; their own ``forward`` methods are never called.
``forward`` hands the pointers to ``sir_model_infer`` (eval) or to the fused training step
(train).  There is no CPU path: inputs must live on a HIP device.
"""
import torch
from torch import nn

from sir_amd import _native, ops

CONV_CHANNELS = (32, 64, 128)
GRU_HIDDEN = 256
N_MELS = 64


class CNNAudioGRU(nn.Module):
    def __init__(self, num_classes, input_channels=1):
        super().__init__()
        if input_channels != 1:
            raise ValueError("the HIP path is built for input_channels=1 (log-mel input)")
        cin = input_channels
        for i, cout in enumerate(CONV_CHANNELS, start=1):
            setattr(self, f"conv{i}", nn.Conv2d(cin, cout, kernel_size=3, stride=1, padding=1, bias=False))
            setattr(self, f"bn{i}", nn.BatchNorm2d(cout))
            cin = cout
        # kept for attribute compatibility (models.py:18-20); pooling/activation are fused in the kernels
        self.relu = nn.ReLU(inplace=True)
        self.pool = nn.MaxPool2d(2)
        self.dropout = nn.Dropout(0.5)          # defined but unused by the reference forward
        self.gru_input_size = CONV_CHANNELS[-1] * (N_MELS // 8)
        self.gru = nn.GRU(input_size=self.gru_input_size, hidden_size=GRU_HIDDEN, num_layers=2,
                          batch_first=True, bidirectional=True, dropout=0.5)
        self.attention = nn.Linear(2 * GRU_HIDDEN, 1)
        self.fc = nn.Linear(2 * GRU_HIDDEN, num_classes)
        self._ws = ops.Workspace()
        self._sir_wcache = None
        self._sir_token = ops.new_model_token()

    def _apply(self, fn, *args, **kwargs):
        self._sir_wcache = None             # .to()/.cuda()/.float() may move the storage
        self._sir_token = ops.new_model_token()
        return super()._apply(fn, *args, **kwargs)

    def load_state_dict(self, *args, **kwargs):
        out = super().load_state_dict(*args, **kwargs)
        self._sir_wcache = None             # assign=True swaps the tensors themselves
        self._sir_token = ops.new_model_token()
        return out

    def forward(self, x):
        """x: [B, 64, T] or [B, 1, 64, T] float32 on the GPU -> logits [B, num_classes]."""
        if self.training and torch.is_grad_enabled():
            from sir_amd import train_ops
            return train_ops.forward_train(self, x)
        return ops.model_infer(self, x, self._ws)

    @torch.no_grad()
    def predict(self, x):
        """logits and argmax (scripts/evaluate.py:82-83) in one launch sequence."""
        return ops.model_infer(self, x, self._ws, want_argmax=True)


if __name__ == "__main__":
    model = CNNAudioGRU(num_classes=31).cuda().eval()
    print("Output shape:", model(torch.randn(4, 64, 200, device="cuda")).shape)
