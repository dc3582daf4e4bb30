#!/usr/bin/env python
"""Generate the golden fixtures under tests/golden/ by running the REFERENCE.

Runs only in the build container (needs /root/reference).  The reference's
``onssen.nn`` package is imported straight from its own files with importlib
(bypassing onssen/__init__.py, which pulls librosa/torchaudio/attrdict that
are not installed), on torch-CPU, fed with this repo's deterministic weights
and inputs.  Only inputs/weight *recipes* and outputs (data) are written;
no reference source travels.

    PYTHONDONTWRITEBYTECODE=1 python tools/gen_golden.py
"""
import importlib.util
import os
import sys

sys.dont_write_bytecode = True
import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from onssen_amd.synthetic import make_state_dict, synth_mixture  # noqa: E402
from oracle import np_oracle  # noqa: E402

REF = "/root/reference/onssen"
OUT = os.path.join(ROOT, "tests", "golden")


def load_ref_pkg(name, sub):
    spec = importlib.util.spec_from_file_location(
        name, f"{REF}/{sub}/__init__.py", submodule_search_locations=[f"{REF}/{sub}"])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def to_torch_sd(sd):
    return {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()}


def run_ref(model, sd, inputs):
    model.load_state_dict(to_torch_sd(sd), strict=True)
    model.eval()
    with torch.no_grad():
        out = model([torch.from_numpy(x) for x in inputs])
    return [o.contiguous().numpy() for o in out]


def logmag_input(seed, B, T, F, n_fft, hop):
    xs = []
    for b in range(B):
        sig = synth_mixture(seed * 100 + b, n_samples=(T - 1) * hop, sr=8000)
        xs.append(np_oracle.log_magnitude(np_oracle.stft(sig, n_fft, hop)))
    x = np.stack(xs).astype(np.float32)
    assert x.shape == (B, T, F), x.shape
    return x


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    os.makedirs(OUT, exist_ok=True)
    ref_nn = load_ref_pkg("ref_nn", "nn")
    F = 129

    # ---- G1: tiny configs, full tensors --------------------------------
    for kind, H, L, B, T in [("deep_clustering", 8, 1, 2, 16),
                             ("deep_clustering", 32, 2, 2, 16),
                             ("chimera", 32, 2, 2, 16),
                             ("phase_net", 16, 2, 2, 12)]:
        seed = 11 + H + L
        sd = make_state_dict(kind, F, H, L, 20, 2, seed=seed, gain=2.0)
        x = logmag_input(seed, B, T, F, 256, 64)
        rec = {"kind": kind, "F": F, "H": H, "L": L, "D": 20, "C": 2, "seed": seed,
               "gain": 2.0, "x": x}
        if kind == "deep_clustering":
            m = ref_nn.deep_clustering(F, H, L, 20)
            outs = run_ref(m, sd, [x])
            names = ["embedding"]
            ins = [x]
        elif kind == "chimera":
            m = ref_nn.chimera(F, H, L, 20)
            outs = run_ref(m, sd, [x])
            names = ["embedding", "mask_A", "mask_B"]
        else:
            sys.modules["ref_nn.phase_network"].output_dim = F   # undefined free variable
            m = ref_nn.phase_net(F, H, L, 20)
            rng = np.random.default_rng(seed)
            xp = rng.normal(0, 1, (B, T, F, 2)).astype(np.float32)
            rec["x_phase"] = xp
            outs = run_ref(m, sd, [x, xp])
            names = ["embedding", "mask_A", "mask_B", "phase_A", "phase_B"]
        for n, o in zip(names, outs):
            rec["out_" + n] = o
        fn = f"{OUT}/g1_{kind}_H{H}_L{L}.npz"
        np.savez_compressed(fn, **rec)
        print("wrote", fn, {n: o.shape for n, o in zip(names, outs)})

    # ---- G2: full-size configs, strided subsample + norms ---------------
    for tag, kind, H, L, B, T, Fq in [("cfg1_dc_L2", "deep_clustering", 600, 2, 2, 400, 129),
                                       ("cfg1_dc_L3", "deep_clustering", 600, 3, 1, 400, 129),
                                       ("cfg3_chimera_L4", "chimera", 600, 4, 1, 400, 129)]:
        sd = make_state_dict(kind, Fq, H, L, 20, 2, seed=0)
        x = logmag_input(7, B, T, Fq, 256, 64)
        if kind == "deep_clustering":
            m = ref_nn.deep_clustering(Fq, H, L, 20)
        else:
            m = ref_nn.chimera(Fq, H, L, 20)
        outs = run_ref(m, sd, [x])
        rec = {"kind": kind, "F": Fq, "H": H, "L": L, "D": 20, "C": 2, "seed": 0, "gain": 1.0,
               "x_seed": 7, "B": B, "T": T,
               "emb_sub": outs[0][:, ::40, ::16, :].copy(),
               "emb_sum_per_frame": outs[0].astype(np.float64).sum(axis=(2, 3)).astype(np.float32)}
        if kind == "chimera":
            rec["mask_A_sub"] = outs[1][:, ::8, :].copy()
            rec["mask_B_sub"] = outs[2][:, ::8, :].copy()
        fn = f"{OUT}/g2_{tag}.npz"
        np.savez_compressed(fn, **rec)
        print("wrote", fn)

    # ---- G4: loss_dc value/grad-norm on a small config (H3) -------------
    ref_loss = load_ref_pkg("ref_loss", "loss")
    H, L, B, T = 32, 2, 3, 20
    sd = make_state_dict("deep_clustering", F, H, L, 20, 2, seed=5)
    x = logmag_input(5, B, T, F, 256, 64)
    rng = np.random.default_rng(5)
    lab = rng.integers(0, 2, (B, T, F))
    one_hot = np.stack([lab, 1 - lab], -1).astype(np.float64)
    one_hot[rng.random((B, T, F)) < 0.2] = 0
    mag = (10 ** x).astype(np.float32)
    m = ref_nn.deep_clustering(F, H, L, 20, dropout=0.0)
    m.load_state_dict(to_torch_sd(sd))
    m.eval()   # fixed BN statistics, no dropout: the gradient parity contract
    out = m([torch.from_numpy(x)])
    loss = ref_loss.loss_dc(out, [torch.from_numpy(one_hot), torch.from_numpy(mag)])
    lavg = torch.mean(loss)
    lavg.backward()
    gn = torch.sqrt(sum((p.grad.double() ** 2).sum() for p in m.parameters() if p.grad is not None))
    np.savez_compressed(f"{OUT}/g4_loss_dc.npz", x=x, one_hot=one_hot, mag=mag, seed=5, H=H, L=L,
                        loss=loss.detach().numpy(), loss_mean=lavg.item(), grad_norm=gn.item(),
                        grad_fc_dc_bias=m.fc_dc.bias.grad.numpy())
    print("wrote g4_loss_dc", loss.shape, lavg.item(), gn.item())


if __name__ == "__main__":
    main()
