"""Diagnostic build only (make -C pssr2_amd/csrc stamps): where head_bwd_kernel (3x3 conv 64 -> 1 @512^2 backward) spends a tile.
segments: 0 stage (wait for the prefetch, activation + gradient tile to LDS), 1 barrier, 2 issue next tile, 3 dP (dgrad MFMAs + LDS out),
4 dW, 5 barrier, 6 masked write-out, 7 barrier"""
import sys, time; sys.path.insert(0, str(__import__('pathlib').Path(__file__).resolve().parents[2]))
import torch
import pssr2_amd._lib as L
from pathlib import Path
L._LIB_PATH = Path(__file__).resolve().parents[2] / 'pssr2_amd' / 'libpssr_mi355_stamps.so'
from pssr2_amd import ops
N, H, W, C, blk = 32, 512, 512, 64, 2
lib = L.lib()
act = (torch.randn(N, H, W, C, device="cuda")).to(torch.bfloat16)
dact = torch.empty_like(act)
g = torch.randn(N, 1, H, W, device="cuda")
wt = torch.randn(1, C, 3, 3, device="cuda") * 0.05
dw = torch.zeros_like(wt)
bsum = torch.zeros(16 * C, device="cuda")
buf = torch.zeros(512 * 4 * 12, dtype=torch.int32, device="cuda")
lib.pssr_debug_head_stamp_buffer(L.ptr(buf))
for _ in range(3):
    ops.head_conv_bwd(g, 1.0, wt, act, dact, blk, dw, bsum, N, H, W, C, 1, L.BF16)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); ops.head_conv_bwd(g, 1.0, wt, act, dact, blk, dw, bsum, N, H, W, C, 1, L.BF16); e1.record(); torch.cuda.synchronize()
print(f"kernel {e0.elapsed_time(e1)*1e3:.0f} us (stamped build)")
b = buf.view(512, 4, 12).cpu().float()
nt = b[:, :, 8:9]
seg = b[:, :, :8] / nt
names = ["stage", "barrier", "issue", "dP", "dW", "barrier", "write-out", "barrier"]
print(f"{nt.mean():.0f} tiles per workgroup; cycles per tile: " + "  ".join(f"{n} {seg[:,:,i].mean():.0f}" for i, n in enumerate(names)) + f" | total {seg.sum(-1).mean():.0f}")
