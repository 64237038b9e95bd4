"""3x3 conv 1024 -> 64 @128^2 x 32 (the data gradient of the 64 -> 1024 pre-shuffle conv): PSSR_IGEMM_V3_64=0/1"""
import sys; sys.path.insert(0, str(__import__('pathlib').Path(__file__).resolve().parents[2]))
import torch
from pssr2_amd import ops, _lib as L
dt = torch.bfloat16; code = L.BF16
N, H, W = 32, 128, 128
def timeit(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for ci, co in ((1024, 64), (512, 64), (64, 64), (96, 64)):
    x = torch.randn(N, H, W, ci, device="cuda").to(dt)
    w = torch.randn(co, ci, 3, 3, device="cuda") / (ci * 9) ** 0.5
    pw = ops.pack_conv_weight(w, code)
    out = torch.zeros(N, H, W, co, device="cuda", dtype=dt)
    bias = torch.zeros(co, device="cuda")
    fl = 2.0 * N * H * W * ci * co * 9
    t = timeit(lambda: ops.conv2d(x, ci, pw, out, co, n=N, h=H, w=W, bias=bias))
    print(f"{ci}->{co}: {t*1e3:8.1f} us {fl/t/1e9:7.1f} TF/s")
