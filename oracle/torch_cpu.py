"""ATen-on-CPU restatement of the network forward (oracle; test infrastructure
and bench.py's ``cpu_baseline`` leg only).

The reference runs nn.LSTM / BatchNorm1d / Linear / F.normalize / sigmoid
(onssen/nn/deep_clustering.py:29-43, chimera.py:30-46, phase_network.py:34-67);
on a CPU those dispatch to the ATen/oneDNN kernels called here through the
functional API.  The reference's Python never travels to the GPU box, so this
is the "port" that is timed on the host cores next to the MI355X numbers
(BASELINE.md section 3).  Pinned to the reference by tests/test_oracle.py via
the committed golden vectors.
"""
import numpy as np
import torch
import torch.nn.functional as F


def _t(sd):
    return {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()}


def _lstm(x, sd, prefix, H):
    L = 0
    flat = []
    while f"{prefix}weight_ih_l{L}" in sd:
        for sfx in ("", "_reverse"):
            flat += [sd[f"{prefix}{n}_l{L}{sfx}"] for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")]
        L += 1
    z = x.new_zeros(2 * L, x.shape[0], H)
    out, _, _ = torch._VF.lstm(x, (z, z), flat, True, L, 0.0, False, True, True)
    return out


def _bn(r, sd, prefix):
    return F.batch_norm(r.permute(0, 2, 1), sd[prefix + "running_mean"], sd[prefix + "running_var"],
                        sd[prefix + "weight"], sd[prefix + "bias"], False, 0.1, 1e-5).permute(0, 2, 1)


@torch.no_grad()
def deep_clustering_forward(sd, x):
    sd = _t(sd)
    x = torch.as_tensor(x).float()
    B, T, Fq = x.shape
    H = sd["rnn.weight_hh_l0"].shape[1]
    r = _bn(_lstm(x, sd, "rnn.", H), sd, "bn.")
    e = F.linear(r, sd["fc_dc.weight"], sd["fc_dc.bias"]).view(B, T * Fq, -1)
    return F.normalize(e, p=2, dim=-1).reshape(B, T, Fq, -1)


@torch.no_grad()
def chimera_forward(sd, x, prefix=""):
    sd = _t(sd) if not isinstance(next(iter(sd.values())), torch.Tensor) else sd
    x = torch.as_tensor(x).float()
    B, T, Fq = x.shape
    H = sd[prefix + "rnn.weight_hh_l0"].shape[1]
    r = _lstm(x, sd, prefix + "rnn.", H)
    e = F.linear(r, sd[prefix + "fc_dc.weight"], sd[prefix + "fc_dc.bias"]).reshape(B, T * Fq, -1)
    e = F.normalize(e, p=2, dim=-1).reshape(B, T, Fq, -1)
    m = torch.sigmoid(F.linear(r, sd[prefix + "fc_mi.weight"], sd[prefix + "fc_mi.bias"])).reshape(B, T, Fq, -1)
    return [e, m[..., 0], m[..., 1]]


@torch.no_grad()
def phase_net_forward(sd, x_mag, x_phase):
    sd = _t(sd)
    x_mag, x_phase = torch.as_tensor(x_mag).float(), torch.as_tensor(x_phase).float()
    B, T, Fq = x_mag.shape
    H = sd["rnn.weight_hh_l0"].shape[1]
    e, mA, mB = chimera_forward(sd, x_mag, prefix="chimera.")
    outs = []
    for m in (mA, mB):
        inp = torch.cat((x_mag * m, x_phase.view(B, T, -1)), 2)
        r = _bn(_lstm(inp, sd, "rnn.", H), sd, "bn.")
        p = F.linear(r, sd["fc_phase.weight"], sd["fc_phase.bias"]).reshape(B, T, Fq, -1) + x_phase
        outs.append(F.normalize(p, p=2, dim=-1))
    return [e, mA, mB, outs[0], outs[1]]
