"""v3 (LDS-DMA loop) vs the 128-pixel loop on the 3x3 layers of a c2 step, interleaved in one process."""
import sys; sys.path.insert(0, '/root/repo')
import torch
from pssr2_amd import ops, _lib as L
dt = torch.bfloat16; code = L.BF16
N = 32
layers = [("L1 128->128 @64", 64, 64, 128, 128, 0), ("L2 256->256 @32", 32, 32, 256, 256, 0), ("L3 512->512 @16", 16, 16, 512, 512, 0),
          ("dec0 768->512 @16", 16, 16, 768, 512, 0), ("dec1 384->256 @32", 32, 32, 384, 256, 0), ("dec2 192->128 @64", 64, 64, 192, 128, 0),
          ("pre 64(+16)->1024 @128", 128, 128, 64, 1024, 16), ("L1 dgrad+1x1 128->128", 64, 64, 128, 128, 128)]
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
    res = []
    for rnd in range(2):
        for v3 in (1, 0):
            lib.pssr_set_option(b"IGEMM_V3", v3)
            t = timeit(lambda: ops.conv2d(x, ci, pw, out, co, n=N, h=H, w=W, bias=bias, **kw))
            t2 = timeit(lambda: ops.conv2d(x, ci, pw, out, co, n=N, h=H, w=W, bias=bias, pro_scale=sc, pro_shift=sh, flags=L.FLAG_STATS, stats=stats, **kw))
            res.append((v3, t, t2))
    lib.pssr_set_option(b"IGEMM_V3", 1)
    for v3 in (1, 0):
        tt = min(r[1] for r in res if r[0] == v3); tt2 = min(r[2] for r in res if r[0] == v3)
        print(f"{name:26s} v3={v3} plain {tt*1e3:8.1f} us {fl/tt/1e9:7.1f} TF/s | +prologue+stats {tt2*1e3:8.1f} us {fl/tt2/1e9:7.1f} TF/s", flush=True)
