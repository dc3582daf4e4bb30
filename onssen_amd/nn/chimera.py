import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import options
from ._train import head_linear, head_linear_normalized
from ._core import (PackedWeightsMixin, needs_graph, BLSTMParams, PackedBLSTM, PackedHead, _Workspaces, EPI_L2NORM, EPI_SIGMOID,
                    as_frames, heads_take_image, require_device, run_blstm, run_head, run_head_pair, use_hip_path)


class chimera(PackedWeightsMixin, nn.Module):
    """Drop-in for onssen.nn.chimera (onssen/nn/chimera.py:5-46).

    forward([x (B,T,F)]) -> [embedding (B,T,F,D), mask_A (B,T,F), mask_B (B,T,F)];
    the masks are strided views of one (B,T,F,C) buffer exactly as upstream.
    """

    def __init__(self, input_dim, hidden_dim=300, num_layers=3, embedding_dim=20, dropout=0.3, num_speaker=2, **hip_options):
        super().__init__()
        options.constructor_options(type(self).__name__, hip_options)      # optional config keys (precision, recurrence, ...)
        self.input_dim, self.hidden_dim = input_dim, hidden_dim
        self.num_layers, self.embedding_dim, self.num_speaker = num_layers, embedding_dim, num_speaker
        self.add_module("rnn", BLSTMParams(input_dim, hidden_dim, num_layers, dropout))
        self.add_module("fc_dc", nn.Linear(hidden_dim * 2, input_dim * embedding_dim))
        self.add_module("fc_mi", nn.Linear(hidden_dim * 2, input_dim * num_speaker))
        self._packed = PackedBLSTM(self.rnn)
        self._head_dc = PackedHead(self.fc_dc, None, hidden_dim)
        self._head_mi = PackedHead(self.fc_mi, None, hidden_dim)
        self._ws = _Workspaces()
        self._init_packed_hooks()

    def forward(self, input, frames=None):
        """``frames`` (extension; inference only): per-row frame counts of a ragged batch of whole utterances, as in
        ``deep_clustering.forward``."""
        assert len(input) == 1, "There must be one tensor in the input for the chimera network"
        x = input[0].float()
        batch_size, frame, frequency = x.size()
        if not use_hip_path(self) or needs_graph(*input):
            if frames is not None:
                raise RuntimeError("chimera: frames=... (ragged batch) is an inference-path extension")
            return self._autograd_forward(x)
        emb, masks = self.embedding_and_masks(x, frames)
        return [emb, masks[:, :, :, 0], masks[:, :, :, 1]]

    def embedding_and_masks(self, x, frames=None):
        """HIP inference path: x (B,T,F) -> (embedding (B,T,F,D), masks (B,T,F,C)); ``forward``
        returns the per-speaker slices of ``masks`` like upstream (chimera.py:43-45)."""
        x = x.float()
        batch_size, frame, frequency = x.size()
        require_device(x, "chimera")
        if frames is not None:
            frames = as_frames(frames, batch_size, frame, x.device)
        y = run_blstm(self._packed, self._ws, x, frames=frames,
                      need_y=not heads_take_image(batch_size, self.hidden_dim, (self.embedding_dim,)))
        # both heads in one launch where the recurrence left its x3 image: fc_mi's columns ride in fc_dc's last, mostly empty tile
        pair = run_head_pair(self._head_dc, self._head_mi, y, batch_size, frame, self.embedding_dim)
        if pair is not None:
            emb, masks = pair
        else:
            emb = run_head(self._head_dc, y, batch_size, frame, EPI_L2NORM, group=self.embedding_dim, eps=1e-12)
            masks = run_head(self._head_mi, y, batch_size, frame, EPI_SIGMOID)
        return emb.view(batch_size, frame, frequency, -1), masks.view(batch_size, frame, frequency, -1)

    def _autograd_forward(self, x):
        B, T, Fq = x.shape
        r = self.rnn.autograd_forward(x, self.training)
        e = head_linear_normalized(self.fc_dc, r, self.embedding_dim).reshape(B, T, Fq, -1)
        m = torch.sigmoid(head_linear(self.fc_mi, r)).reshape(B, T, Fq, -1)
        return [e, m[:, :, :, 0], m[:, :, :, 1]]
