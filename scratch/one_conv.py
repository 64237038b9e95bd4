import sys; sys.path.insert(0, '/root/repo')
import torch
from pssr2_amd import ops, _lib as L
mode = int(sys.argv[1])
L.lib().pssr_conv2d_pipeline_mode(mode)
N, H, W, ci, co = 32, 128, 128, 64, 1024
dt = torch.bfloat16; code = L.BF16
x = torch.randn(N, H, W, ci, device="cuda").to(dt)
w = torch.randn(co, ci, 3, 3, device="cuda") / (ci * 9) ** 0.5
pw = ops.pack_conv_weight(w, code)
out = torch.zeros(N, H, W, co, device="cuda", dtype=dt)
bias = torch.zeros(co, device="cuda")
for _ in range(3):
    ops.conv2d(x, ci, pw, out, co, n=N, h=H, w=W, bias=bias)
torch.cuda.synchronize()
