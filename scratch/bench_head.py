import sys; sys.path.insert(0, '/root/repo')
import torch
from pssr2_amd import ops, _lib as L
dt = torch.bfloat16; code = L.BF16
n, cin, cout, h, w, blk = 32, 64, 1, 512, 512, 2
act = torch.randn(n, h, w, cin, device="cuda").to(dt)
da = torch.empty_like(act)
g = torch.randn(n, cout, h, w, device="cuda")
wt = torch.randn(cout, cin, 3, 3, device="cuda") / 24
dw = torch.zeros_like(wt)
bs = torch.zeros(16 * cin, device="cuda")
def timeit(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
print("dgrad", timeit(lambda: ops.head_conv_dgrad(g, 128.0, wt, act, da, blk, n, h, w, cin, cout, code)))
print("wgrad", timeit(lambda: ops.head_conv_wgrad(g, 128.0, act, blk, dw, n, h, w, cin, cout, code)))
print("bwd+bsum", timeit(lambda: ops.head_conv_bwd(g, 128.0, wt, act, da, blk, dw, bs, n, h, w, cin, cout, code)))
print("bwd", timeit(lambda: ops.head_conv_bwd(g, 128.0, wt, act, da, blk, dw, None, n, h, w, cin, cout, code)))
