"""The pipelined headline step (DCPipeline, 32 x 400 frames, 2 x BLSTM-600) replayed N times -- run under rocprofv3 --kernel-trace --stats."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from onssen_amd import nn as onn
from onssen_amd.synthetic import make_state_dict, synth_batch
from onssen_amd.separation import DCPipeline
dev = torch.device("cuda:0")
H, B, n = 600, 32, 25536
sd = make_state_dict("deep_clustering", 129, H, 2, 20, 2, seed=0)
m = onn.deep_clustering(129, H, 2, 20)
m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
m = m.to(dev).eval()
wav = torch.from_numpy(synth_batch(1, B, n, 8000)).to(dev)
pipe = DCPipeline(m, B, n, graph=os.environ.get('PIPE2_GRAPH', '1') == '1')
pipe.push(wav); pipe.wav[1].copy_(wav)
for _ in range(20):
    pipe.replay()
torch.cuda.synchronize()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    pipe.replay()
e1.record(); torch.cuda.synchronize()
print("pipelined step ms", e0.elapsed_time(e1) / reps)
