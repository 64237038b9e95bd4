"""head_q_gather_kernel stand-alone: GB/s against the batch size (the (tap, sub-pixel) planes are n*h*w*4 bytes apart: 2 MiB at batch 32)."""
import sys; sys.path.insert(0, str(__import__('pathlib').Path(__file__).resolve().parents[2]))
import torch
from pssr2_amd import ops
h = w = 128
for n in (32, 128):
    q = torch.randn(9, 16, n, h, w, device="cuda")
    out = torch.empty(n, 1, 4 * h, 4 * w, device="cuda")
    bias = torch.zeros(1, device="cuda")
    f = lambda: ops.head_q_gather(q, bias, out, n, h, w, 128.0, 128.0)
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 20 * 1e-3
    by = q.numel() * 4 + out.numel() * 4
    print(f"n={n:4d} {t*1e6:8.1f} us {by/t/1e9:8.1f} GB/s")
