"""head_q_gather kernels stand-alone: GB/s at the batch sizes of the training step (32) and of predict_images (128), back to back on
the same planes and -- as in the step -- right after another kernel has written the planes and a larger tensor."""
import sys; sys.path.insert(0, str(__import__('pathlib').Path(__file__).resolve().parents[2]))
import torch
from pssr2_amd import ops
h = w = 128
for n in (32, 128):
    q = torch.randn(9, 16, n, h, w, device="cuda")
    big = torch.empty(n * 512 * 512 * 64 // 2, device="cuda")            # the size of `pre`'s bf16 activation
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
    ts = []
    for _ in range(10):
        big.fill_(1.0); q.mul_(1.0)
        e0.record(); f(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e-3)
    t2 = sorted(ts)[len(ts) // 2]
    print(f"n={n:4d} back to back {t*1e6:8.1f} us {by/t/1e9:8.1f} GB/s | after a writer {t2*1e6:8.1f} us {by/t2/1e9:8.1f} GB/s")
