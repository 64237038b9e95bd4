import sys; sys.path.insert(0, '/root/repo')
import torch
from pssr2_amd import ops, _lib as L
def timeit(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (n, h, cout) in [(32, 512, 1), (8, 1024, 3)]:
    cin, blk, dt, code = 64, 2, torch.bfloat16, L.BF16
    ad = torch.randn(n, h, h, cin, device="cuda").relu().to(dt)
    wt = torch.randn(cout, cin, 3, 3, device="cuda") * 0.05
    b = torch.zeros(cout, device="cuda"); out = torch.empty(n, cout, h, h, device="cuda"); dout = torch.randn(n, cout, h, h, device="cuda")
    da = torch.empty_like(ad); dw = torch.zeros_like(wt)
    gb = ad.numel() * 2 / 1e9
    t1 = timeit(lambda: ops.head_conv_fwd(ad, blk, wt, b, out, n, h, h, cin, cout, 128.0, 128.0, code))
    t2 = timeit(lambda: ops.head_conv_dgrad(dout, 128.0, wt, ad, da, blk, n, h, h, cin, cout, code))
    t3 = timeit(lambda: ops.head_conv_wgrad(dout, 128.0, ad, blk, dw, n, h, h, cin, cout, code))
    print(f"n={n} {h}^2 cout={cout}: act {gb:.2f} GB | fwd {t1:7.1f} us ({gb/t1*1e3:.2f} TB/s) dgrad {t2:7.1f} us ({2*gb/t2*1e3:.2f} TB/s) wgrad {t3:7.1f} us ({gb/t3*1e3:.2f} TB/s)")
