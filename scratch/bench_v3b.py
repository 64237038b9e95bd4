"""v3 component timing: full / no epilogue / no multiply / neither."""
import sys; sys.path.insert(0, '/root/repo')
import torch
from pssr2_amd import ops, _lib as L
dt = torch.bfloat16; code = L.BF16
N = 32
layers = [("L1 128->128 @64", 64, 64, 128, 128, 0), ("L2 256->256 @32", 32, 32, 256, 256, 0), ("dec2 192->128 @64", 64, 64, 192, 128, 0),
          ("pre 64(+16)->1024 @128", 128, 128, 64, 1024, 16)]
def timeit(fn, n=8):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
lib = L.lib()
for name, H, W, ci, co, c1 in layers:
    x = torch.randn(N, H, W, ci, device="cuda").to(dt)
    w = torch.randn(co, ci, 3, 3, device="cuda") / (ci * 9) ** 0.5
    pw = ops.pack_conv_weight(w, code)
    out = torch.zeros(N, H, W, co, device="cuda", dtype=dt)
    sc, sh = torch.rand(ci, device="cuda") + 0.5, torch.randn(ci, device="cuda") * 0.1
    stats = torch.zeros(ops.STAT_STRIPES * 2 * co, dtype=torch.float64, device="cuda")
    bias = torch.zeros(co, device="cuda")
    kw = {}
    if c1:
        x1 = torch.randn(N, H, W, c1, device="cuda").to(dt)
        w1 = torch.randn(co, c1, 1, 1, device="cuda") / c1 ** 0.5
        kw = dict(x1=x1, cin1=c1, w1=ops.pack_conv_weight(w1, code))
    fl = 2.0 * N * H * W * co * (ci * 9 + c1)
    line = f"{name:26s}"
    for dbg in (0, 1, 2, 3):
        lib.pssr_set_option(b"IGEMM_DBG", dbg)
        t = min(timeit(lambda: ops.conv2d(x, ci, pw, out, co, n=N, h=H, w=W, bias=bias, **kw)) for _ in range(2))
        t2 = min(timeit(lambda: ops.conv2d(x, ci, pw, out, co, n=N, h=H, w=W, bias=bias, pro_scale=sc, pro_shift=sh, flags=L.FLAG_STATS, stats=stats, **kw)) for _ in range(2))
        line += f" | dbg{dbg}: {t*1e3:7.1f} / {t2*1e3:7.1f} us"
    lib.pssr_set_option(b"IGEMM_DBG", 0)
    print(line + f"   (ideal @2.5PF {fl/2.5e15*1e6:.1f} us)", flush=True)
