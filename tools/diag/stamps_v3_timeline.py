"""Diagnostic build only (libpssr_mi355_stamps.so, -DPSSR_V3_STAMPS): timeline of the v3 workgroups of one launch -- kernel entry,
main loop, epilogue instructions, store acknowledgement -- in shader clocks relative to the first workgroup's entry."""
import sys; sys.path.insert(0, str(__import__('pathlib').Path(__file__).resolve().parents[2]))
import torch
import pssr2_amd._lib as L
from pathlib import Path
L._LIB_PATH = Path(__file__).resolve().parents[2] / 'pssr2_amd' / 'libpssr_mi355_stamps.so'
from pssr2_amd import ops
dt = torch.bfloat16; code = L.BF16
N = 32
layers = [("L0 64->64 @128 (BN prologue + stats)", 128, 128, 64, 64, True, 32), ("L0 64->64 @128 plain", 128, 128, 64, 64, False, 32),
          ("L1 128->128 @64 (BN prologue + stats)", 64, 64, 128, 128, True, 16)]
lib = L.lib()
for name, H, W, ci, co, pro, tw in layers:
    x = torch.randn(N, H, W, ci, device="cuda").to(dt)
    w = torch.randn(co, ci, 3, 3, device="cuda") / (ci * 9) ** 0.5
    pw = ops.pack_conv_weight(w, code)
    out = torch.zeros(N, H, W, co, device="cuda", dtype=dt)
    sc, sh = torch.rand(ci, device="cuda") + 0.5, torch.randn(ci, device="cuda") * 0.1
    stats = torch.zeros(ops.STAT_STRIPES * 2 * co, dtype=torch.float64, device="cuda")
    bias = torch.zeros(co, device="cuda")
    kw = dict(pro_scale=sc, pro_shift=sh, flags=L.FLAG_STATS, stats=stats) if pro else {}
    nblk = (H // 16) * (W // tw) * N * max(co // 128, 1)
    buf = torch.zeros(nblk * 4 * 10, dtype=torch.int32, device="cuda")
    lib.pssr_debug_stamp_buffer(L.ptr(buf))
    for _ in range(3):
        ops.conv2d(x, ci, pw, out, co, n=N, h=H, w=W, bias=bias, **kw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); ops.conv2d(x, ci, pw, out, co, n=N, h=H, w=W, bias=bias, **kw); e1.record(); torch.cuda.synchronize()
    b = buf[:nblk * 32].view(nblk, 4, 8).cpu().long() & 0xffffffff
    t = buf[nblk * 32:].view(nblk, 4, 2).cpu().long() & 0xffffffff
    t0 = b[:, :, 6].min()
    rel = lambda v: ((v - t0) & 0xffffffff).float()
    entry, loop_end, epi_end, ack = rel(b[:, :, 6]).mean(1), rel(b[:, :, 7]).mean(1), rel(t[:, :, 0]).mean(1), rel(t[:, :, 1]).mean(1)
    nst = int(b[0, 0, 5])
    pro_c = b[:, :, 4].float().mean(1)
    print(f"{name}: {nblk} workgroups, {nst} stages, launch {e0.elapsed_time(e1) * 1e3:.1f} us; last store acknowledged at {ack.max():.0f} clk")
    order = entry.argsort()
    for lo, hi in ((0, 256), (256, 512), (512, 768), (768, 1024)):
        if lo >= nblk: break
        sel = order[lo:min(hi, nblk)]
        print(f"   workgroups {lo:4d}-{hi:4d} by entry time: entry {entry[sel].mean():7.0f} | first stage ready +{pro_c[sel].mean():6.0f} | loop end {loop_end[sel].mean():7.0f} "
              f"(loop {(loop_end[sel] - entry[sel]).mean() - pro_c[sel].mean():6.0f}) | epilogue issued {epi_end[sel].mean():7.0f} (+{(epi_end[sel] - loop_end[sel]).mean():6.0f}) "
              f"| stores acknowledged {ack[sel].mean():7.0f} (+{(ack[sel] - epi_end[sel]).mean():6.0f})", flush=True)
