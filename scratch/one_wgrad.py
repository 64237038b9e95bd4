import sys; sys.path.insert(0, '/root/repo')
import torch
from pssr2_amd import ops, _lib as L
which = sys.argv[1] if len(sys.argv) > 1 else "wgrad"
N, H, W, ci, co = 32, 64, 64, 128, 128
dt = torch.bfloat16; code = L.BF16
x = torch.randn(N, H, W, ci, device="cuda").to(dt)
dy = torch.randn(N, H, W, co, device="cuda").to(dt)
sc, sh = torch.ones(ci, device="cuda"), torch.zeros(ci, device="cuda")
if which == "wgrad":
    for _ in range(3):
        ops.conv2d_wgrad_parts(dy, co, x, ci, 9, n=N, h=H, w=W, dtype=code, pro_scale=sc, pro_shift=sh)
else:
    w = torch.randn(co, ci, 3, 3, device="cuda") / (ci * 9) ** 0.5
    pw = ops.pack_conv_weight(w, code)
    out = torch.zeros(N, H, W, co, device="cuda", dtype=dt)
    stats = torch.zeros(ops.STAT_STRIPES * 2 * co, dtype=torch.float64, device="cuda")
    for _ in range(3):
        ops.conv2d(x, ci, pw, out, co, n=N, h=H, w=W, bias=None, pro_scale=sc, pro_shift=sh, flags=L.FLAG_STATS, stats=stats)
torch.cuda.synchronize()
