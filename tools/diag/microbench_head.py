"""head (3x3 conv 64 -> 1 @512^2, batch 32) kernels: forward and the fused backward"""
import sys; sys.path.insert(0, str(__import__('pathlib').Path(__file__).resolve().parents[2]))
import torch
from pssr2_amd import ops, _lib as L
N, H, W, C, blk = 32, 512, 512, 64, 2
act = torch.relu(torch.randn(N, H, W, C, device="cuda")).to(torch.bfloat16)
dact = torch.empty_like(act)
g = torch.randn(N, 1, H, W, device="cuda")
wt = torch.randn(1, C, 3, 3, device="cuda") * 0.05
b = torch.zeros(1, device="cuda")
dw = torch.zeros_like(wt)
bsum = torch.zeros(16 * C, device="cuda")
out = torch.empty(N, 1, H, W, device="cuda")
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
gb = act.numel() * 2 / 1e9
t = timeit(lambda: ops.head_conv_bwd(g, 1.0, wt, act, dact, blk, dw, bsum, N, H, W, C, 1, L.BF16))
print(f"head_bwd {t:.0f} us  {2 * gb / t * 1e3:.2f} TB/s (activation read + gradient written)")
t = timeit(lambda: ops.head_conv_fwd(act, blk, wt, b, out, N, H, W, C, 1, 1.0, 0.0, L.BF16))
print(f"head_fwd {t:.0f} us  {gb / t * 1e3:.2f} TB/s (activation read)")
