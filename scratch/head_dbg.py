import sys; sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import torch, torch.nn.functional as F
from pssr2_amd import ops, _lib as L
from test_gpu_head import _blocked, _unblocked
for (n, cin, cout, h, w, blk) in [(2, 64, 1, 48, 80, 2), (1, 32, 3, 40, 24, 2)]:
    dt = torch.bfloat16; code = ops.dtype_code(dt)
    g = torch.Generator().manual_seed(cin + cout)
    act = F.relu(torch.randn(n, cin, h, w, generator=g)).to(dt).float()
    wt = (torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5)
    dout = torch.randn(n, cout, h, w, generator=g)
    ad = _blocked(act, blk, dt)
    da = torch.full_like(ad, 7.0)
    ops.head_conv_dgrad(dout.cuda(), 128.0, wt.cuda().contiguous(), ad, da, blk, n, h, w, cin, cout, code)
    da2 = torch.full_like(ad, 7.0)
    dw2 = torch.zeros(cout, cin, 3, 3, device="cuda")
    r = 1 << blk
    bsum = torch.zeros(r * r * cin, device="cuda")
    ops.head_conv_bwd(dout.cuda(), 128.0, wt.cuda().contiguous(), ad, da2, blk, dw2, bsum, n, h, w, cin, cout, code)
    a, b = _unblocked(da, blk), _unblocked(da2, blk)
    d = (a - b).abs()
    print("max diff", d.max().item(), "n diff", (d > 0).sum().item(), "of", d.numel(), "max ref", a.abs().max().item())
    idx = (d > 0).nonzero()[:10]
    for i in idx: print(i.tolist(), a[tuple(i)].item(), b[tuple(i)].item())
