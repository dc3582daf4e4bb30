import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import options
from ._train import batch_norm_rows, head_linear_normalized
from ._core import (PackedWeightsMixin, needs_graph, BLSTMParams, PackedBLSTM, PackedHead, _Workspaces, EPI_L2NORM,
                    as_frames, heads_take_image, require_device, run_blstm, run_head, use_hip_path)


class deep_clustering(PackedWeightsMixin, nn.Module):
    """Drop-in for onssen.nn.deep_clustering (onssen/nn/deep_clustering.py:5-43):
    same constructor, parameter names/shapes and list-in/list-out forward.

    forward([x (B,T,F)]) -> [embedding (B,T,F,D)], unit L2 norm per TF bin.
    In eval mode without autograd the arithmetic runs in libonssen_hip.so
    (input-projection GEMMs, the XCD-local persistent recurrence, BatchNorm folded
    into the fc_dc GEMM whose epilogue normalises each bin).
    """

    def __init__(self, input_dim, hidden_dim=300, num_layers=3, embedding_dim=20, dropout=0.3, **hip_options):
        super().__init__()
        options.constructor_options(type(self).__name__, hip_options)      # optional config keys (precision, recurrence, ...)
        self.input_dim, self.hidden_dim = input_dim, hidden_dim
        self.num_layers, self.embedding_dim = num_layers, embedding_dim
        self.add_module("rnn", BLSTMParams(input_dim, hidden_dim, num_layers, dropout))
        self.add_module("bn", nn.BatchNorm1d(hidden_dim * 2))
        self.add_module("fc_dc", nn.Linear(hidden_dim * 2, embedding_dim * input_dim))
        self._packed = PackedBLSTM(self.rnn)
        self._head = PackedHead(self.fc_dc, self.bn, hidden_dim)
        self._ws = _Workspaces()
        self._init_packed_hooks()

    def forward(self, input, frames=None):
        """``frames`` (extension; inference only): per-row frame counts of a RAGGED batch of whole utterances padded to
        the longest -- every row's embedding at its own frames is bit for bit what a batch-1 forward of that utterance
        returns (the reference's evaluation runs them one at a time, onssen/utils/test.py:29-41)."""
        assert len(input) == 1, "There must be one tensor in the input for the deep clustering model"
        x = input[0].float()
        batch_size, frame, frequency = x.size()
        if not use_hip_path(self) or needs_graph(*input):
            if frames is not None:
                raise RuntimeError("deep_clustering: frames=... (ragged batch) is an inference-path extension")
            return [self._autograd_forward(x)]
        require_device(x, "deep_clustering")
        if frames is not None:
            frames = as_frames(frames, batch_size, frame, x.device)
        y = run_blstm(self._packed, self._ws, x, frames=frames,
                      need_y=not heads_take_image(batch_size, self.hidden_dim, (self.embedding_dim,)))
        emb = run_head(self._head, y, batch_size, frame, EPI_L2NORM, group=self.embedding_dim, eps=1e-12)
        return [emb.view(batch_size, frame, frequency, -1)]

    def fused_loss_dc(self, input, label):
        """``loss_dc(self(input), label)`` (onssen/loss/loss_dc.py:6-44 on onssen/nn/deep_clustering.py:32-43) as ONE autograd node
        behind the BLSTM stack and the BatchNorm -- the fusion a train step can make because it holds the labels when the
        forward runs (``onssen_amd.dist.train_step`` does, for this model with ``onssen_amd.loss.loss_dc``): the embedding is
        written once and read twice, d loss / d embedding is never written.  Returns the (B, B) loss tensor of upstream's
        broadcast, or None when this batch cannot take the fused kernels (the caller then runs model + loss as usual)."""
        from ._train import DcHeadLossFunction, dc_head_loss_applies
        assert len(input) == 1 and len(label) == 2
        x = input[0].float()
        one_hot, mag_mix = label
        if not (self.training and dc_head_loss_applies(self.fc_dc, x, one_hot, self.embedding_dim)):
            return None
        B, T, Fq = x.shape
        r = self.rnn.autograd_forward(x, True)
        r = batch_norm_rows(self.bn, r)
        per_utt, total = DcHeadLossFunction.apply(r, self.fc_dc.weight, self.fc_dc.bias, one_hot.reshape(B, T * Fq, -1),
                                                  mag_mix.reshape(B, T * Fq), self.embedding_dim, 1e-12)
        return per_utt * total.unsqueeze(1)

    def _autograd_forward(self, x):
        B, T, Fq = x.shape
        r = self.rnn.autograd_forward(x, self.training)
        r = batch_norm_rows(self.bn, r)
        e = head_linear_normalized(self.fc_dc, r, self.embedding_dim)
        return e.reshape(B, T, Fq, -1)
