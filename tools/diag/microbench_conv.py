import sys; sys.path.insert(0, str(__import__('pathlib').Path(__file__).resolve().parents[2]))
import os
import torch
from pssr2_amd import ops, _lib as L
dt = torch.bfloat16; code = L.BF16
N = 32
# (name, H, W, Cin, Cout, taps)
layers = [("L0 64->64", 128, 128, 64, 64, 9), ("L1 128->128", 64, 64, 128, 128, 9), ("L2 256->256", 32, 32, 256, 256, 9),
          ("L3 512->512", 16, 16, 512, 512, 9), ("L4 1024->1024", 8, 8, 1024, 1024, 9), ("dec0 768->512", 16, 16, 768, 512, 9),
          ("dec3 96->64", 128, 128, 96, 64, 9), ("head 64->1024", 128, 128, 64, 1024, 9), ("L1 1x1 64->128", 64, 64, 64, 128, 1),
          ("final 64->1 @512", 512, 512, 64, 4, 9)]
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
which = sys.argv[1] if len(sys.argv) > 1 else "all"
for name, H, W, ci, co, taps in layers:
    ks = 3 if taps == 9 else 1
    x = torch.randn(N, H, W, ci, device="cuda").to(dt)
    w = torch.randn(co, ci, ks, ks, device="cuda") / (ci * taps) ** 0.5
    if os.environ.get("MB_ZERO") == "1":        # all-zero operands: the same instruction stream without data toggling (is the clock power-limited?)
        x.zero_(); w.zero_()
    pw = ops.pack_conv_weight(w, code)
    out = torch.zeros(N, H, W, max(co, 4), device="cuda", dtype=dt)
    sc, sh = torch.ones(ci, device="cuda"), torch.zeros(ci, device="cuda")
    stats = torch.zeros(ops.STAT_STRIPES * 2 * co, dtype=torch.float64, device="cuda")
    bias = torch.zeros(co, device="cuda")
    fl = 2.0 * N * H * W * ci * co * taps
    if which in ("all", "fwd"):
        t = timeit(lambda: ops.conv2d(x, ci, pw, out, co, n=N, h=H, w=W, bias=bias))
        t2 = timeit(lambda: ops.conv2d(x, ci, pw, out, co, n=N, h=H, w=W, bias=bias, pro_scale=sc, pro_shift=sh, flags=L.FLAG_STATS, stats=stats))
        t3 = timeit(lambda: ops.conv2d(x, ci, pw, out, co, n=N, h=H, w=W, bias=bias, pro_scale=sc, pro_shift=sh))
        print(f"{name:20s} fwd plain {t*1e3:8.1f} us {fl/t/1e9:7.1f} TF/s | +prologue {t3*1e3:8.1f} us | +prologue+stats {t2*1e3:8.1f} us {fl/t2/1e9:7.1f} TF/s")
    if which in ("all", "wgrad") and co >= 16:
        dy = torch.randn(N, H, W, co, device="cuda").to(dt)
        dwp = torch.zeros(co, taps, ci, device="cuda")
        t = timeit(lambda: ops.conv2d_wgrad(dy, co, x, ci, taps, dwp, n=N, h=H, w=W, dtype=code, pro_scale=sc, pro_shift=sh))
        t2 = timeit(lambda: ops.conv2d_wgrad_parts(dy, co, x, ci, taps, n=N, h=H, w=W, dtype=code, pro_scale=sc, pro_shift=sh))
        parts = ops.conv2d_wgrad_parts(dy, co, x, ci, taps, n=N, h=H, w=W, dtype=code, pro_scale=sc, pro_shift=sh)
        dwo = torch.zeros(co, ci, ks, ks, device="cuda")
        t3 = timeit(lambda: ops.unpack_conv_wgrad(parts, dwo, k_pad=ci))
        t4 = timeit(lambda: ops.conv2d_wgrad_parts(dy, co, x, ci, taps, n=N, h=H, w=W, dtype=code))      # prologue-free: the all-DMA kernel where it applies
        print(f"{name:20s} wgrad atomic {t*1e3:8.1f} us {fl/t/1e9:7.1f} TF/s | parts({parts.shape[0]:3d}) {t2*1e3:8.1f} us {fl/t2/1e9:7.1f} TF/s | unpack {t3*1e3:6.1f} us"
              f" | no prologue {t4*1e3:8.1f} us {fl/t4/1e9:7.1f} TF/s")
