#!/usr/bin/env python
"""Round 6c soak (GPU box): N steps of separation.DCRaggedPipeline on ragged batches of random lengths (dc_l2, 16 utterances of 1-8 s per
batch), every result compared bit for bit with separate_dc(lengths=) on the same batch; counts mismatches and aborted launches."""
import json, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    from onssen_amd.nn._core import _XcdPolicy, _XcdStatus
    from onssen_amd.separation import DCRaggedPipeline, separate_dc
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    dev = torch.device("cuda", 0)
    wl = bench.build_workload("dc_l2", 16, dev)
    model = wl["model"]
    K, cap = 16, 8 * 8000
    rng = np.random.default_rng(3)
    pool = torch.from_numpy(np.stack([bench_synth(k, cap) for k in range(K)])).to(dev)
    pipe = DCRaggedPipeline(model, K, cap)
    a0 = _XcdPolicy.aborts
    bad, checked = 0, 0
    prev = None
    t0 = time.perf_counter()
    with torch.no_grad():
        for s in range(steps):
            lengths = rng.integers(8000, cap + 1, K)
            n = int(lengths.max())
            perm = torch.from_numpy(rng.permutation(K)).to(dev)
            wav = pool[perm][:, :n].contiguous()
            for b in range(K):
                wav[b, int(lengths[b]):] = 0
            ln = torch.from_numpy(lengths.astype(np.int32)).to(dev)
            out = pipe.push(wav, ln)
            if out is not None and s % 25 == 0:
                got = out.clone()
                ref = separate_dc(model, prev[0], lengths=prev[1])
                checked += 1
                bad += int(not torch.equal(got, ref))
            prev = (wav, ln)
        pipe.flush()
    torch.cuda.synchronize()
    _XcdStatus.poll(wait=True)
    res = {"steps": steps, "utterances": steps * K, "checked_batches": checked, "mismatched_batches": bad, "aborts": _XcdPolicy.aborts - a0,
           "seconds": time.perf_counter() - t0}
    print(json.dumps(res))
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(res, open("gpurun_out/r06c_ragged_soak.json", "w"), indent=1)


def bench_synth(k, n):
    from onssen_amd.synthetic import synth_mixture
    return synth_mixture(900 + k, n)


if __name__ == "__main__":
    main()
