import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import options
from ..hip import get_lib
from ._train import batch_norm_rows
from ._core import (PackedWeightsMixin, needs_graph, BLSTMParams, PackedBLSTM, PackedHead, _Workspaces, EPI_L2NORM, _stream,
                    heads_take_image, require_device, run_blstm, run_head, use_hip_path)
from .chimera import chimera


class phase_net(PackedWeightsMixin, nn.Module):
    """Drop-in for onssen.nn.phase_net (onssen/nn/phase_network.py:7-67).

    Upstream's constructor raises NameError on an undefined ``output_dim``
    (phase_network.py:28); the only shape-consistent value is ``input_dim``
    (fc_phase: 2H -> num_speaker*F, last axis = (re, im)), which is what this
    module uses.

    forward([x_mag (B,T,F), x_phase (B,T,F,2)]) ->
        [embedding, mask_A, mask_B, phase_A (B,T,F,2), phase_B (B,T,F,2)]
    """

    def __init__(self, input_dim, hidden_dim=300, num_layers=3, embedding_dim=20, dropout=0.3, num_speaker=2, **hip_options):
        super().__init__()
        options.constructor_options(type(self).__name__, hip_options)      # optional config keys (precision, recurrence, ...)
        self.input_dim, self.hidden_dim, self.num_speaker = input_dim, hidden_dim, num_speaker
        chimera_net = chimera(input_dim, hidden_dim, num_layers, embedding_dim, dropout, num_speaker)
        self.add_module("rnn", BLSTMParams(input_dim * 3, hidden_dim, num_layers, dropout))
        self.add_module("bn", nn.BatchNorm1d(hidden_dim * 2))
        self.add_module("fc_phase", nn.Linear(hidden_dim * 2, num_speaker * input_dim))
        self.add_module("chimera", chimera_net)
        self._packed = PackedBLSTM(self.rnn)
        self._head = PackedHead(self.fc_phase, self.bn, hidden_dim)
        self._ws = _Workspaces()
        self._init_packed_hooks()

    def forward(self, input):
        assert len(input) == 2, "There must be 2 tensors in the input for phase network"
        [x_mag, x_phase] = input
        if not use_hip_path(self) or needs_graph(*input):
            return self._autograd_forward(x_mag.float(), x_phase.float())
        x_mag, x_phase = x_mag.float().contiguous(), x_phase.float().contiguous()
        require_device(x_mag, "phase_net")
        embedding, masks = self.chimera.embedding_and_masks(x_mag)
        mask_A, mask_B = masks[:, :, :, 0], masks[:, :, :, 1]
        B, T, Fq, C = masks.size()
        # cat(x_mag*mask_s, x_phase.view(B,T,2F)) for both speakers, stacked on the batch axis:
        # the phase BLSTM shares its weights between A and B, so it runs once with batch 2B
        inp = torch.empty(C * B, T, 3 * Fq, device=x_mag.device, dtype=torch.float32)
        get_lib().phase_input(x_mag.data_ptr(), masks.data_ptr(), masks.stride(0), masks.stride(3), masks.stride(1),
                              masks.stride(2), x_phase.data_ptr(), B, C, T, Fq, inp.data_ptr(), _stream())
        # (the head reads the recurrence's x3 output image where there is one: the fp32 rows are then never written)
        y = run_blstm(self._packed, self._ws, inp, tag="phase", need_y=not (C == 2 and heads_take_image(C * B, self.hidden_dim)))
        resid = x_phase.view(B, T, 2 * Fq)
        if getattr(y, "x3_image", None) is not None and C == 2:
            # both speakers in ONE launch of the pre-split-operand GEMM: rows b and b + B add the same mixture phase
            p = run_head(self._head, y, C * B, T, EPI_L2NORM, group=2, eps=1e-12, resid=resid, resid_mod=B)
            return [embedding, mask_A, mask_B, p[:B].view(B, T, Fq, 2), p[B:].view(B, T, Fq, 2)]
        outs = []
        for s in range(2):
            p = run_head(self._head, y, B, T, EPI_L2NORM, group=2, eps=1e-12, resid=resid, b_off=s * B,
                         b_total=C * B)
            outs.append(p.view(B, T, Fq, 2))
        return [embedding, mask_A, mask_B, outs[0], outs[1]]

    def _autograd_forward(self, x_mag, x_phase):
        [embedding, mask_A, mask_B] = self.chimera([x_mag])
        B, T, Fq = mask_A.size()
        outs = []
        for m in (mask_A, mask_B):
            inp = torch.cat((x_mag * m, x_phase.view(B, T, -1)), 2)
            r = self.rnn.autograd_forward(inp, self.training)
            r = batch_norm_rows(self.bn, r)
            p = self.fc_phase(r).reshape(B, T, Fq, -1) + x_phase
            outs.append(F.normalize(p, p=2, dim=-1))
        return [embedding, mask_A, mask_B, outs[0], outs[1]]
