#!/usr/bin/env python
"""GPU box: the weight-gradient GEMM of one BLSTM-600 layer at the training shape (16 x 400 frames) from row-major images
(onssen_lstm_wgrad_images_f32) and from transposed images (onssen_linear_x3p_batched_split_alt), GEMM launches only.
One JSON line ({"ms_per_step": row-major GEMM ms, ...}: tools/ab_variants.py run prints it per library variant)."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from onssen_amd.hip import get_lib
lib = get_lib()
dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream
H, Kx, T, B = 600, 1200, 400, 16
NP, Hp, K = 4 * H, H, T * B
KB = (K + 31) // 32
torch.manual_seed(0)
dP, y, x = (torch.randn(K, n, device=dev) for n in (2 * NP, 2 * Hp, Kx))
def rows_img(m):
    o = torch.empty(m.shape[0], (m.shape[1] + 31) // 32, 2, 32, device=dev, dtype=torch.int16)
    lib.x3_image(m.data_ptr(), m.shape[1], 0, 1, m.shape[0], m.shape[1], o.data_ptr(), st)
    return o
zero = torch.zeros(Hp + Kx, device=dev)
a_t = torch.empty(2 * NP, KB, 2, 32, device=dev, dtype=torch.int16)
lib.x3_image_t(dP.data_ptr(), 2 * NP, 2 * NP, K, 0, a_t.data_ptr(), st)
w1 = torch.empty(Hp + Kx + Hp, KB, 2, 32, device=dev, dtype=torch.int16)
lib.x3_image_t(y.data_ptr(), 2 * Hp, Hp, K, -B, w1.data_ptr(), st)
lib.x3_image_t(x.data_ptr(), Kx, Kx, K, 0, w1[Hp:].data_ptr(), st)
lib.x3_image_t(y[:, Hp:].data_ptr(), 2 * Hp, Hp, K, B, w1[Hp + Kx:].data_ptr(), st)
dp_img, y_img, x_img = rows_img(dP), rows_img(y), rows_img(x)
ih, hh = torch.empty(2, 4 * H, Kx, device=dev), torch.empty(2, 4 * H, H, device=dev)
def nt():
    lib.linear_x3p_batched_split_alt(a_t.data_ptr(), NP * KB * 64, NP, K, w1.data_ptr(), Hp * KB * 64, zero.data_ptr(), Hp + Kx, 4,
                                     hh.data_ptr(), 4 * H * H, H, H * H, Hp, ih.data_ptr(), 4 * H * Kx, Kx, H * Kx, Kx, 2, st)
def tn():
    lib.lstm_wgrad_images(dp_img.data_ptr(), y_img.data_ptr(), x_img.data_ptr(), K, B, NP, Hp, Kx, zero.data_ptr(), 4,
                          ih.data_ptr(), 4 * H * Kx, Kx, H * Kx, hh.data_ptr(), 4 * H * H, H, H * H, st)
out = {}
for name, fn in (("transposed_images_gemm_us", nt), ("row_major_gemm_us", tn)):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fn()
    e1.record()
    torch.cuda.synchronize()
    out[name] = round(e0.elapsed_time(e1) * 50, 1)
out["ms_per_step"] = out["row_major_gemm_us"] / 1e3
print(json.dumps(out))
