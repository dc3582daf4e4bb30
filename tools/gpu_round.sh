#!/bin/bash
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; tail -3 gpurun_out/pytest_gpu.log
python - <<'PY' 2>&1 | grep -v amdgpu.ids | tail -4
import torch, time, numpy as np
from onssen_amd.loss import loss_dc
dev = torch.device("cuda:0")
B, T, F, D = 32, 400, 129, 20
emb = torch.nn.functional.normalize(torch.randn(B, T, F, D, device=dev), dim=-1)
lab = torch.randint(0, 3, (B, T, F), device=dev)
one_hot = torch.stack([lab == 0, lab == 1], -1).double()
mag = torch.rand(B, T, F, device=dev) + 1e-3
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n
oh32 = one_hot.float()
with torch.no_grad():
    t_hip = timeit(lambda: loss_dc([emb], [oh32, mag]))
e2 = emb.clone().requires_grad_(True)
t_aten = timeit(lambda: loss_dc([e2], [oh32, mag]))
print(f"loss_dc value, B=32 x 51600 bins x D=20: HIP {t_hip:.3f} ms ({emb.numel()*4/t_hip/1e6:.0f} GB/s of embedding read) | ATen ops (autograd path, forward only) {t_aten:.3f} ms")
PY
