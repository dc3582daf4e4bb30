import torch
import torch.nn as nn

from .. import options

from ._train import batch_norm_rows
from ._core import (PackedWeightsMixin, needs_graph, BLSTMParams, PackedBLSTM, PackedHead, _Workspaces, EPI_RELU, EPI_SIGMOID, _stream, _version_key,
                    heads_take_image, require_device, run_blstm, run_head, use_hip_path)
from ..hip import get_lib


class enhance(PackedWeightsMixin, nn.Module):
    """Drop-in for onssen.nn.enhance (onssen/nn/enhancement.py:5-52): same constructor, parameter names/shapes
    (rnn.*, bn.*, fc_mi, fc_pre, fc_post) and forward contract.

    forward([x (B,T,F), mag_noisy (B,T,F)]) -> [clean (B,T,F)]:
        mask = sigmoid(fc_mi(bn(blstm(x)))),  clean = relu(fc_post(relu(fc_pre(mag_noisy)) * mask)).
    In eval mode without autograd: the BLSTM and the mask head are the kernels of the separation models (BatchNorm
    folded into fc_mi); the two F x F "restoration" layers are exact-fp32 GEMMs whose epilogue applies the ReLU and,
    for fc_pre, the multiplication by the mask (ONSSEN_EPI_RELU)."""

    def __init__(self, input_dim, hidden_dim=300, num_layers=3, dropout=0.3, **hip_options):
        super().__init__()
        options.constructor_options(type(self).__name__, hip_options)      # optional config keys (precision, recurrence, ...)
        self.input_dim, self.hidden_dim, self.num_layers = input_dim, hidden_dim, num_layers
        self.add_module("rnn", BLSTMParams(input_dim, hidden_dim, num_layers, dropout))
        self.add_module("bn", nn.BatchNorm1d(hidden_dim * 2))
        self.add_module("fc_mi", nn.Linear(hidden_dim * 2, input_dim))
        self.add_module("fc_pre", nn.Linear(input_dim, input_dim))
        self.add_module("fc_post", nn.Linear(input_dim, input_dim))
        self._packed = PackedBLSTM(self.rnn)
        self._head = PackedHead(self.fc_mi, self.bn, hidden_dim)
        self._ws = _Workspaces()
        self._init_packed_hooks()
        self._small = None      # (key, w_pre, w_post): F x F weights padded to a multiple of 4 columns

    def _small_weights(self):
        ts = [self.fc_pre.weight, self.fc_post.weight]
        key = _version_key(ts)
        if self._small is None or self._small[0] != key:
            Fq = self.input_dim
            ld = (Fq + 3) // 4 * 4
            ws = []
            for t in ts:
                w = torch.zeros(Fq, ld, device=t.device, dtype=torch.float32)
                w[:, :Fq] = t.detach()
                ws.append(w)
            self._small = (key, ws[0], ws[1], ld)
        return self._small[1:]

    def forward(self, input):
        assert len(input) == 2, "There must be two tensors in the input for the enhance network"
        x, mag_noisy = input[0].float(), input[1].float()
        batch_size, frame, frequency = x.size()
        if not use_hip_path(self) or needs_graph(*input):
            return [self._autograd_forward(x, mag_noisy)]
        require_device(x, "enhance")
        lib = get_lib()
        y = run_blstm(self._packed, self._ws, x, need_y=not heads_take_image(batch_size, self.hidden_dim))
        mask = run_head(self._head, y, batch_size, frame, EPI_SIGMOID)                 # (B, T, F)
        w_pre, w_post, ld = self._small_weights()
        mag = mag_noisy.contiguous()
        M, Fq = batch_size * frame, frequency
        est = torch.empty(batch_size, frame, Fq, device=x.device, dtype=torch.float32)
        clean = torch.empty_like(est)
        # relu(fc_pre(mag_noisy)) * mask, then relu(fc_post(.)): rows are plain (b, t) rows here
        lib.linear(mag.data_ptr(), Fq, 0, 1, M, Fq, w_pre.data_ptr(), ld, self.fc_pre.bias.detach().data_ptr(), Fq, EPI_RELU, 0,
                   0.0, mask.data_ptr(), est.data_ptr(), Fq, 0, _stream())
        lib.linear(est.data_ptr(), Fq, 0, 1, M, Fq, w_post.data_ptr(), ld, self.fc_post.bias.detach().data_ptr(), Fq, EPI_RELU,
                   0, 0.0, None, clean.data_ptr(), Fq, 0, _stream())
        return [clean]

    def _autograd_forward(self, x, mag_noisy):
        r = self.rnn.autograd_forward(x, self.training)
        r = batch_norm_rows(self.bn, r)
        mask = torch.sigmoid(self.fc_mi(r))
        return torch.relu(self.fc_post(torch.relu(self.fc_pre(mag_noisy)) * mask))
