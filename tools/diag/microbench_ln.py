"""LayerNorm2d forward / backward (csrc/rdnet.hip) at RDNet's shapes, microseconds and algorithmic TB/s.  PSSR_LN_DBG /
PSSR_LN_BWD_BLOCKS select diagnostic variants of the backward kernel."""
import sys; sys.path.insert(0, str(__import__('pathlib').Path(__file__).resolve().parents[2]))
import torch
from pssr2_amd import ops, _lib as L
dt = torch.bfloat16; code = L.BF16
N = 32
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for H, C in [(64, 128), (64, 256), (32, 368), (16, 232), (16, 488), (16, 616), (8, 816)]:
    M = N * H * H
    x = torch.randn(N, H, H, C, device="cuda").to(dt)
    g = torch.randn(N, H, H, C, device="cuda").to(dt)
    y, dx = torch.zeros_like(x), torch.zeros_like(x)
    gamma, beta = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
    mean, rstd = torch.zeros(M, device="cuda"), torch.ones(M, device="cuda")
    stats = torch.zeros(ops.STAT_STRIPES * 2 * 2 * C, dtype=torch.float64, device="cuda")
    tf = timeit(lambda: ops.layernorm2d_fwd(x, gamma, beta, 1e-6, y, N, H, H, C, code, c_pad=C, mean=mean, rstd=rstd))
    tb = timeit(lambda: ops.layernorm2d_bwd(g, x, gamma, mean, rstd, dx, stats, N, H, H, C, code, c_pad=C))
    print(f"{H:3d}x{H:<3d} C {C:4d} (M {M:6d}): fwd {tf*1e3:7.1f} us {2*M*C*2/tf/1e9:5.2f} TB/s | bwd {tb*1e3:7.1f} us {3*M*C*2/tb/1e9:5.2f} TB/s")
