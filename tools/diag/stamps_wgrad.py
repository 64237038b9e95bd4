"""Diagnostic build only (libpssr_mi355_stamps.so, make -C pssr2_amd/csrc stamps): where a weight-gradient pixel tile spends its cycles.
segments: 0 = exposed load wait, 1 = commit (prologue + LDS writes), 2 = barrier, 3 = issue next tile's loads, 4 = multiply, 5 = end barrier"""
import sys; sys.path.insert(0, str(__import__('pathlib').Path(__file__).resolve().parents[2]))
import torch
import pssr2_amd._lib as L
from pathlib import Path
L._LIB_PATH = Path(__file__).resolve().parents[2] / 'pssr2_amd' / 'libpssr_mi355_stamps.so'
from pssr2_amd import ops
dt = torch.bfloat16; code = L.BF16
N = 32
layers = [("L0 64->64 @128", 128, 128, 64, 64), ("L2 256->256 @32", 32, 32, 256, 256), ("pre 64->1024 @128", 128, 128, 64, 1024)]
lib = L.lib()
for name, H, W, ci, co in layers:
    x = torch.randn(N, H, W, ci, device="cuda").to(dt)
    dy = torch.randn(N, H, W, co, device="cuda").to(dt)
    sc, sh = torch.rand(ci, device="cuda") + 0.5, torch.randn(ci, device="cuda") * 0.1
    buf = torch.zeros(4096 * 8 * 8, dtype=torch.int32, device="cuda")
    lib.pssr_debug_wgrad_stamp_buffer(L.ptr(buf))
    for _ in range(3):
        parts = ops.conv2d_wgrad_parts(dy, co, x, ci, 9, n=N, h=H, w=W, dtype=code, pro_scale=sc, pro_shift=sh)
    torch.cuda.synchronize()
    b = buf.view(8192, 4, 8).cpu().long()   # two-group kernel: [workgroup][group][wave]; one-group kernel fills the first half as [workgroup][wave]
    live = b[:, 0, 6] > 0
    b = b[live]
    pro, epi = (b[:, :, 6] >> 8).float(), b[:, :, 7].float()
    b[:, :, 6] &= 0xff
    b = b.float()
    print(f"  prologue {pro.mean():.0f} cycles, epilogue (partial-slab stores drained) {epi.mean():.0f} cycles")
    nt = b[:, :, 6].mean()
    seg = b[:, :, :6] / b[:, :, 6:7]
    names = ["load wait", "commit", "barrier", "issue", "multiply", "end barrier"]
    print(f"{name}: {int(live.sum())} workgroups x {nt:.0f} tiles; cycles per tile: "
          + "  ".join(f"{n} {seg[:,:,i].mean():.1f}" for i, n in enumerate(names)) + f"  | total {seg.sum(-1).mean():.1f}", flush=True)
    for gsel in (0, 1):
        sg = seg[gsel::2]
        if len(sg): print(f"    group {gsel}: " + "  ".join(f"{sg[:,:,i].mean():.1f}" for i in range(6)) + f"  | total {sg.sum(-1).mean():.1f}")
