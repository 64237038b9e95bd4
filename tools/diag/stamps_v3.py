"""Diagnostic build only (libpssr_mi355_stamps.so, -DPSSR_V3_STAMPS): where a v3 stage spends its cycles.
segments: 0 = stage start -> DMA / image loads issued; 1 = multiply (asm block); 2 = image commit; 3 = wait + barrier; 4 = prologue"""
import sys; sys.path.insert(0, str(__import__('pathlib').Path(__file__).resolve().parents[2]))
import torch
import pssr2_amd._lib as L
from pathlib import Path
L._LIB_PATH = Path(__file__).resolve().parents[2] / 'pssr2_amd' / 'libpssr_mi355_stamps.so'
from pssr2_amd import ops
dt = torch.bfloat16; code = L.BF16
N = 32
layers = [("L1 128->128 @64", 64, 64, 128, 128, 0, True), ("L2 256->256 @32", 32, 32, 256, 256, 0, True),
          ("pre 64(+16)->1024 @128", 128, 128, 64, 1024, 16, False)]
lib = L.lib()
for name, H, W, ci, co, c1, pro in layers:
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
    if pro:
        kw.update(pro_scale=sc, pro_shift=sh, flags=L.FLAG_STATS, stats=stats)
    nblk = (H // 16) * (W // 16) * N * (co // 128)
    buf = torch.zeros(nblk * 4 * 10, dtype=torch.int32, device="cuda")       # 8 per wave + 2 per wave (epilogue times) behind them
    lib.pssr_debug_stamp_buffer(L.ptr(buf))
    for _ in range(3):
        ops.conv2d(x, ci, pw, out, co, n=N, h=H, w=W, bias=bias, **kw)
    torch.cuda.synchronize()
    b = buf[:nblk * 32].view(nblk, 4, 8).cpu().long()
    nst = int(b[0, 0, 5])
    seg = b[:, :, :5].float()
    tot = seg[:, :, :4].sum(-1)
    print(f"{name}: {nblk} workgroups, {nst} stages; cycles per stage per wave (mean over waves): "
          f"issue {seg[:,:,0].mean()/nst:.0f}  multiply {seg[:,:,1].mean()/nst:.0f}  commit {seg[:,:,2].mean()/nst:.0f}  wait+barrier {seg[:,:,3].mean()/nst:.0f}"
          f"  | prologue {seg[:,:,4].mean():.0f}  loop total {tot.mean():.0f}  (min {tot.min():.0f}, max {tot.max():.0f})", flush=True)
    # first vs late workgroups
    for lo, hi in ((0, min(512, nblk)), (max(0, nblk - 512), nblk)):
        s2 = seg[lo:hi]
        print(f"    wg {lo}-{hi}: issue {s2[:,:,0].mean()/nst:.0f} multiply {s2[:,:,1].mean()/nst:.0f} commit {s2[:,:,2].mean()/nst:.0f} wait {s2[:,:,3].mean()/nst:.0f} prologue {s2[:,:,4].mean():.0f}")
