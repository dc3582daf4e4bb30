"""A/B aid for tests and tools (NOT part of the product path): inside ``aten_lstm_reference()`` the training forward of the
onssen_amd.nn models runs the stock ATen LSTM op (MIOpen's fused RNN on a ROCm device) and the stock ATen ops for the heads /
BatchNorm / normalisation (ONSSEN_TRAIN_HIP=0) instead of the package's HIP kernels -- the comparison point of the
"HIP path vs ATen path" tests and of tools/train_step_bench.py's ATen rows.  The product itself never leaves its HIP kernels
on a GPU (onssen_amd/nn/_core.py: BLSTMParams.autograd_forward)."""
import contextlib
import os

import torch


@contextlib.contextmanager
def aten_lstm_reference():
    from onssen_amd.nn._core import BLSTMParams
    orig = BLSTMParams.autograd_forward
    old_env = os.environ.get("ONSSEN_TRAIN_HIP")

    def aten(self, x, training):
        z = x.new_zeros(2 * self.num_layers, x.shape[0], self.hidden_size)
        out, _, _ = torch._VF.lstm(x, (z, z), self.flat_weights(), True, self.num_layers,
                                   self.dropout if training else 0.0, training, True, True)
        return out
    BLSTMParams.autograd_forward = aten
    os.environ["ONSSEN_TRAIN_HIP"] = "0"
    try:
        yield
    finally:
        BLSTMParams.autograd_forward = orig
        if old_env is None:
            os.environ.pop("ONSSEN_TRAIN_HIP", None)
        else:
            os.environ["ONSSEN_TRAIN_HIP"] = old_env
