#!/usr/bin/env python
"""Same-box bit comparison of library variants (build_variants/libonssen_hip_<name>.so, tools/ab_variants.py build): every variant runs
the same forwards in its own process (ONSSEN_HIP_LIB) and prints SHA-256 digests of the outputs; variants that claim "same bits"
must print the same lines.   gpurun -- 'python tools/ab_bits.py base tg221'"""
import hashlib, os, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child():
    import numpy as np, torch
    sys.path.insert(0, ROOT)
    from onssen_amd import nn as onn
    from onssen_amd.synthetic import make_state_dict
    dev = torch.device("cuda:0")
    out = []
    for kind, H, L, B, T in (("deep_clustering", 600, 2, 32, 400), ("deep_clustering", 600, 2, 16, 100), ("chimera", 600, 2, 64, 60),
                             ("deep_clustering", 300, 2, 32, 50), ("deep_clustering", 600, 1, 5, 37)):
        sd = make_state_dict(kind, 129, H, L, 20, 2, seed=3, gain=1.0)
        m = getattr(onn, kind)(129, H, L, 20)
        m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
        m = m.to(dev).eval()
        x = torch.from_numpy(np.random.default_rng(5).uniform(-6, 1.5, (B, T, 129)).astype(np.float32)).to(dev)
        with torch.no_grad():
            y = m([x])
        torch.cuda.synchronize()
        h = hashlib.sha256(b"".join(t.contiguous().cpu().numpy().tobytes() for t in y)).hexdigest()[:16]
        out.append(f"{kind} H={H} L={L} B={B} T={T}: {h}")
    # training forward + backward (persistent kernels with saved state)
    from onssen_amd import loss as oloss
    sd = make_state_dict("deep_clustering", 129, 600, 2, 20, 2, seed=4, gain=1.0)
    m = onn.deep_clustering(129, 600, 2, 20, dropout=0.0)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    m = m.to(dev).train()
    rng = np.random.default_rng(6)
    x = torch.from_numpy(rng.uniform(-6, 1.5, (16, 100, 129)).astype(np.float32)).to(dev)
    e = m([x])[0]
    (e * torch.from_numpy(rng.standard_normal(tuple(e.shape)).astype(np.float32)).to(dev)).sum().backward()
    torch.cuda.synchronize()
    g = hashlib.sha256(b"".join(p.grad.contiguous().cpu().numpy().tobytes() for p in m.parameters())).hexdigest()[:16]
    out.append(f"train fwd+bwd dc 2x600 B=16 T=100: emb {hashlib.sha256(e.detach().cpu().numpy().tobytes()).hexdigest()[:16]} grads {g}")
    print("\n".join(out))


if __name__ == "__main__":
    if len(sys.argv) == 2 and sys.argv[1] == "--child":
        child()
    else:
        res = {}
        for name in sys.argv[1:]:
            lib = os.path.join(ROOT, "build_variants", f"libonssen_hip_{name}.so")
            env = dict(os.environ, ONSSEN_HIP_LIB=lib)
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=env, capture_output=True, text=True, timeout=900)
            res[name] = r.stdout.strip().splitlines() if r.returncode == 0 else ["FAILED: " + r.stderr.strip()[-400:]]
            print(f"--- {name}\n" + "\n".join(res[name]), flush=True)
        names = list(res)
        for n in names[1:]:
            print(f"{n} vs {names[0]}: {'IDENTICAL' if res[n] == res[names[0]] else 'DIFFERENT'}")
