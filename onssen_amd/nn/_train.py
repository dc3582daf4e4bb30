"""Training path of the stacked BLSTM on the HIP kernels (second half).

What autograd does for ``self.rnn(x)`` in the reference's training loop (onssen/utils/train.py:70-84 calling
onssen/nn/deep_clustering.py:32-35): here the forward of every layer is the XCD-local persistent recurrence with saved
gate activations / cell states (``onssen_lstm_train_forward_f32``), the backward recurrence is
``onssen_lstm_train_backward_f32`` (XCD-local persistent launch; one launch per time step with ONSSEN_BWD_XCD=0), and the weight / input gradient contractions -- plain
dense GEMMs over T*B rows -- are library GEMMs (rocBLAS through ``torch.mm``) on the pre-activation gradient the
kernel leaves.  The inter-layer dropout of ``nn.LSTM(dropout=0.3)`` (deep_clustering.py:15-22) is applied between the
layers with torch's own generator, as ATen does."""
import os

import torch
import torch.nn.functional as Fn

from .. import _abi
from ..hip import get_lib
from ._core import _XcdStatus


def packed_columns(H, Hp, ug, device=None):
    """LongTensor [4H]: packed gate column (layout of G, onssen_hip.h) of nn.LSTM row n = gate*H + u."""
    n = torch.arange(4 * H, device=device)
    gate, u = n // H, n % H
    return (u // ug) * (4 * ug) + (u % ug) * 4 + gate


def layer_gradients(dP, x_rows, y, w_ih, H, ug):
    """Weight / bias / input gradients of one bidirectional layer from the pre-activation gradient.

    dP (T,B,2,NP) packed columns; x_rows (T*B, In) the layer's input rows in the reference's feature order, time-major;
    y (T,B,2,Hp) the layer's output; w_ih = (W_ih forward, W_ih reverse), each (4H, In).
    Returns dx_rows (T*B, In) and per direction (dW_ih, dW_hh, db)."""
    T, B, _, NP = dP.shape
    Hp = y.shape[3]
    cols = packed_columns(H, Hp, ug, dP.device)
    dx = None
    grads = []
    for d in range(2):
        dPn = dP[:, :, d, :].index_select(2, cols)                 # (T,B,4H) in nn.LSTM row order
        flat = dPn.reshape(T * B, 4 * H)
        dW_ih = flat.t() @ x_rows
        hd = y[:, :, d, :H]
        if d == 0:    # h_{t-1}
            dW_hh = dPn[1:].reshape(-1, 4 * H).t() @ hd[:-1].reshape(-1, H) if T > 1 else flat.new_zeros(4 * H, H)
        else:         # the reverse direction's previous step is t+1
            dW_hh = dPn[:-1].reshape(-1, 4 * H).t() @ hd[1:].reshape(-1, H) if T > 1 else flat.new_zeros(4 * H, H)
        db = flat.sum(0)
        part = flat @ w_ih[d]
        dx = part if dx is None else dx + part
        grads.append((dW_ih, dW_hh, db))
    return dx, grads


class _Workspace:
    """Recurrence workspaces keyed by shape (their header must start out zero and is never zeroed again)."""
    cache = {}

    @classmethod
    def get(cls, key, nbytes, device, zero):
        buf = cls.cache.get(key)
        if buf is None or buf.numel() < nbytes or buf.device != device:
            buf = (torch.zeros if zero else torch.empty)(nbytes, dtype=torch.uint8, device=device)
            cls.cache[key] = buf
        return buf


class BLSTMTrainFunction(torch.autograd.Function):
    """y = BLSTM_stack(x) with HIP forward and backward.  x (B,T,In) -> (B,T,2H)."""

    @staticmethod
    def forward(ctx, x, packed, p_drop, *flat):
        lib = get_lib()
        prm = packed.p
        H, L = prm.hidden_size, prm.num_layers
        B, T, In = x.shape
        ug = 4 * -(-H // 128)
        if H > 640:
            raise RuntimeError("HIP training path: H <= 640 (XCD-local recurrence)")
        _XcdStatus.poll()
        pk = packed.get(ug)
        Hp, NP = pk.Hp, pk.NP
        st = torch.cuda.current_stream().cuda_stream
        dev = x.device
        x = x.contiguous()
        saved = []
        xin, xs_b, xs_t, in_l = x, T * In, In, In
        x_rows = x.transpose(0, 1).reshape(T * B, In)                    # time-major rows for the weight gradients
        for l in range(L):
            nbytes = lib.blstm_workspace_bytes(B, T, in_l, H, 1, ug)
            ws = _Workspace.get(("fwd", B, T, in_l, H), nbytes, dev, zero=True)
            y = torch.empty(T, B, 2, Hp, device=dev, dtype=torch.float32)
            gates = torch.empty(T, B, 2, NP, device=dev, dtype=torch.float32)
            cs = torch.empty(T, B, 2, Hp, device=dev, dtype=torch.float32)
            lib.lstm_train_forward(xin.data_ptr(), xs_b, xs_t, B, T, in_l, H, ug, pk.wih_img[l].data_ptr(),
                                   pk.whh_x3[l].data_ptr(), pk.bias[l].data_ptr(), y.data_ptr(), gates.data_ptr(),
                                   cs.data_ptr(), ws.data_ptr(), ws.numel(), st)
            _XcdStatus.post(ws)           # an aborted exchange is reported at the next poll (never silently)
            mask = None
            if l < L - 1:
                nxt = y.view(T, B, 2 * Hp)
                if p_drop > 0.0:
                    mask = (torch.rand_like(nxt) >= p_drop).to(torch.float32) * (1.0 / (1.0 - p_drop))
                    nxt = nxt * mask
                saved.append((x_rows, y, gates, cs, mask))
                xin, xs_b, xs_t, in_l = nxt, 2 * Hp, B * 2 * Hp, 2 * Hp
                # the reference's feature order [fwd(H) | rev(H)] of the same rows
                x_rows = nxt.view(T, B, 2, Hp)[..., :H].reshape(T * B, 2 * H) if Hp != H else nxt.view(T * B, 2 * H)
            else:
                saved.append((x_rows, y, gates, cs, None))
        ctx.saved_layers = saved
        ctx.packed, ctx.ug, ctx.dims = packed, ug, (B, T, In, H, L, Hp, NP)
        ctx.flat = flat
        return y[..., :H].reshape(T, B, 2 * H).transpose(0, 1).contiguous()

    @staticmethod
    def backward(ctx, dy_bt):
        lib = get_lib()
        B, T, In, H, L, Hp, NP = ctx.dims
        ug, pk = ctx.ug, ctx.packed.get(ctx.ug)
        st = torch.cuda.current_stream().cuda_stream
        dev = dy_bt.device
        flat = ctx.flat
        # (B,T,2H) -> time-major (T,B,2,Hp), padded units zero
        dy = dy_bt.transpose(0, 1).reshape(T, B, 2, H)
        dy = Fn.pad(dy, (0, Hp - H)).contiguous() if Hp != H else dy.contiguous()
        # ONSSEN_BWD_XCD=0: one launch per time step instead of the XCD-local persistent launch
        form = _abi.LSTM_BWD_XCD if os.environ.get("ONSSEN_BWD_XCD", "1") == "1" else _abi.LSTM_BWD_STEPS
        wsb = _Workspace.get(("bwd", B, H, form), lib.lstm_train_backward_workspace_bytes(B, H, ug, form), dev, zero=True)
        whh_img = pk.whh_bwd(form)
        grads = [None] * (8 * L)
        dx_rows = None
        for l in range(L - 1, -1, -1):
            x_rows, y, gates, cs, mask = ctx.saved_layers[l]
            lib.lstm_train_backward(B, T, H, ug, whh_img[l].data_ptr(), dy.data_ptr(), gates.data_ptr(), cs.data_ptr(),
                                    wsb.data_ptr(), wsb.numel(), form, st)
            if form == _abi.LSTM_BWD_XCD:
                _XcdStatus.post(wsb)
            w_ih = (flat[(2 * l) * 4].detach(), flat[(2 * l + 1) * 4].detach())
            dx_rows, g = layer_gradients(gates, x_rows, y, w_ih, H, ug)
            for d in range(2):
                o = (2 * l + d) * 4
                grads[o], grads[o + 1], grads[o + 2], grads[o + 3] = g[d][0], g[d][1], g[d][2], g[d][2]
            if l > 0:
                dyl = dx_rows.view(T, B, 2, H)
                dyl = Fn.pad(dyl, (0, Hp - H)) if Hp != H else dyl
                mprev = ctx.saved_layers[l - 1][4]
                dy = (dyl.reshape(T, B, 2 * Hp) * mprev).view(T, B, 2, Hp) if mprev is not None else dyl
                dy = dy.contiguous()
        dx = dx_rows.view(T, B, In).transpose(0, 1).contiguous() if ctx.needs_input_grad[0] else None
        ctx.saved_layers = None
        return (dx, None, None) + tuple(grads)
