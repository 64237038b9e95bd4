import sys, time; sys.path.insert(0, '/root/repo')
import torch
t0 = time.time()
from pssr2_amd import ops, _lib as L
n, cin, cout, h, w, blk = 2, 64, 1, 48, 80, 2
dt = torch.bfloat16
ad = torch.randn(n, h, w, cin, device="cuda").relu().to(dt)
wt = torch.randn(cout, cin, 3, 3, device="cuda") * 0.05
b = torch.zeros(cout, device="cuda")
out = torch.empty(n, cout, h, w, device="cuda")
dout = torch.randn(n, cout, h, w, device="cuda")
torch.cuda.synchronize(); print("setup", time.time() - t0, flush=True)
for name, fn in [("fwd", lambda: ops.head_conv_fwd(ad, blk, wt, b, out, n, h, w, cin, cout, 128.0, 128.0, L.BF16)),
                 ("dgrad", lambda: ops.head_conv_dgrad(dout, 128.0, wt, ad, torch.empty_like(ad), blk, n, h, w, cin, cout, L.BF16)),
                 ("wgrad", lambda: ops.head_conv_wgrad(dout, 128.0, ad, blk, torch.zeros_like(wt), n, h, w, cin, cout, L.BF16))]:
    t1 = time.time(); fn(); torch.cuda.synchronize(); print(name, "first", time.time() - t1, flush=True)
    t1 = time.time(); fn(); torch.cuda.synchronize(); print(name, "second", time.time() - t1, flush=True)
