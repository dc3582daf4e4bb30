#!/usr/bin/env python
"""Round 6c (GPU box): bench.ragged_leg by itself -- separate_dc on ragged batches of 16 whole utterances against the ragged pipeline."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
r = bench.ragged_leg(torch.device("cuda", 0))
print(json.dumps(r, indent=1))
os.makedirs("gpurun_out", exist_ok=True)
json.dump(r, open("gpurun_out/r06c_ragged_leg.json", "w"), indent=1)
