"""RDNet's 1x1 layers (pssr/models/_rdnet.py Block: conv1 C -> 4C, GELU, conv2 4C -> g) one at a time at batch 32: forward, input gradient and
weight gradient, microseconds / TFLOP/s / algorithmic TB/s.  python tools/diag/microbench_pw.py [fwd|dgrad|wgrad|all]"""
import sys; sys.path.insert(0, str(__import__('pathlib').Path(__file__).resolve().parents[2]))
import torch
from pssr2_amd import ops, _lib as L
dt = torch.bfloat16; code = L.BF16
N = 32
# (stage, H = W, C_in of the block, growth)
blocks = [("s0b0", 64, 128, 64), ("s0b2", 64, 256, 64), ("s1b2", 32, 368, 104), ("s2b0", 16, 232, 128), ("s3b2", 16, 560, 128), ("s5b2", 16, 616, 128),
          ("s6b2", 8, 816, 224)]
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
which = sys.argv[1] if len(sys.argv) > 1 else "all"
P16 = lambda v: (v + 15) // 16 * 16
def line(tag, t, fl, by):
    print(f"  {tag:34s} {t*1e3:8.1f} us {fl/t/1e9:7.1f} TF/s {by/t/1e9:6.2f} TB/s")
for name, H, C, g in blocks:
    M = N * H * H
    I = 4 * C
    Cp, gp = P16(C), P16(g)
    x = torch.randn(N, H, H, Cp, device="cuda").to(dt)
    z = torch.randn(N, H, H, I, device="cuda").to(dt)
    t_ = torch.randn(N, H, H, gp, device="cuda").to(dt)
    w1 = torch.randn(I, Cp, 1, 1, device="cuda") / Cp ** 0.5
    w2 = torch.randn(gp, I, 1, 1, device="cuda") / I ** 0.5
    pw1, pw2 = ops.pack_conv_weight(w1, code), ops.pack_conv_weight(w2, code)
    pw1t = ops.pack_conv_weight(w1.permute(1, 0, 2, 3).contiguous(), code)      # dgrad of conv1: 4C -> C
    pw2t = ops.pack_conv_weight(w2.permute(1, 0, 2, 3).contiguous(), code)      # dgrad of conv2: g -> 4C
    b1, b2 = torch.zeros(I, device="cuda"), torch.zeros(gp, device="cuda")
    oz, ot, ox = torch.zeros_like(z), torch.zeros_like(t_), torch.zeros_like(x)
    stats = torch.zeros(ops.STAT_STRIPES * 2 * I, dtype=torch.float64, device="cuda")
    print(f"{name}: {H}x{H} x{N}  C {C} -> {I} -> g {g}   (M = {M})")
    if which in ("all", "fwd"):
        line("fwd conv1 C->4C (bias)", timeit(lambda: ops.conv2d(x, Cp, pw1, oz, I, n=N, h=H, w=H, bias=b1)), 2.0 * M * Cp * I, 2.0 * M * (Cp + I))
        line("fwd conv2 4C->g (gelu in, bias)", timeit(lambda: ops.conv2d(z, I, pw2, ot, gp, n=N, h=H, w=H, bias=b2, gelu_in=True)), 2.0 * M * I * gp, 2.0 * M * (gp + I))
    if which in ("all", "dgrad"):
        line("dgrad conv2 g->4C (gelu' z, stats)", timeit(lambda: ops.conv2d(t_, gp, pw2t, oz, I, n=N, h=H, w=H, epilogue=L.EPI_DGRAD_GELU, flags=L.FLAG_STATS,
                                                                             aux=z, stats=stats)), 2.0 * M * I * gp, 2.0 * M * (gp + 2 * I))
        line("dgrad conv1 4C->C", timeit(lambda: ops.conv2d(z, I, pw1t, ox, Cp, n=N, h=H, w=H)), 2.0 * M * Cp * I, 2.0 * M * (Cp + I))
    if which in ("all", "wgrad"):
        t1 = timeit(lambda: ops.conv2d_wgrad_parts(z, I, x, Cp, 1, n=N, h=H, w=H, dtype=code))
        parts = ops.conv2d_wgrad_parts(z, I, x, Cp, 1, n=N, h=H, w=H, dtype=code)
        d1 = torch.zeros(I, Cp, 1, 1, device="cuda")
        u1 = timeit(lambda: ops.unpack_conv_wgrad(parts, d1, k_pad=Cp, accumulate=True))
        line(f"wgrad conv1 ({parts.shape[0]} parts) + unpack {u1*1e3:.1f} us", t1, 2.0 * M * Cp * I, 2.0 * M * (Cp + I))
        t2 = timeit(lambda: ops.conv2d_wgrad_parts(t_, gp, z, I, 1, n=N, h=H, w=H, dtype=code, gelu_in=True))
        parts = ops.conv2d_wgrad_parts(t_, gp, z, I, 1, n=N, h=H, w=H, dtype=code, gelu_in=True)
        d2 = torch.zeros(gp, I, 1, 1, device="cuda")
        u2 = timeit(lambda: ops.unpack_conv_wgrad(parts, d2, k_pad=I, accumulate=True))
        line(f"wgrad conv2 ({parts.shape[0]} parts, gelu in) + unpack {u2*1e3:.1f} us", t2, 2.0 * M * I * gp, 2.0 * M * (gp + I))
