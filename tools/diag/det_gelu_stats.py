"""Diagnostic: are the f64 statistic rows of a 1x1 data-gradient convolution (GELU-derivative epilogue, conv_flat_kernel) the same on
every launch?  Repeats each configuration and counts the launches whose rows differ from the first one."""
import sys; sys.path.insert(0, str(__import__('pathlib').Path(__file__).resolve().parents[2]))
import torch
from pssr2_amd import ops, _lib as L
dt = torch.bfloat16; code = L.BF16
torch.manual_seed(0)
REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 20
for (N, H, W, ci, co) in [(32, 128, 128, 32, 128), (32, 64, 64, 64, 256), (32, 32, 32, 64, 256)]:
    x = (torch.randn(N, H, W, ci, device="cuda") * 1e-3).to(dt)
    z = torch.randn(N, H, W, co, device="cuda").to(dt)
    w = torch.randn(ci, co, 1, 1, device="cuda") / co ** 0.5      # forward conv co -> ci; its data gradient maps ci -> co
    pw = ops.pack_conv_weight(w, code, mode=1)
    for name, epi, kw in (("gelu", L.EPI_DGRAD_GELU, dict(aux=z)), ("store", L.EPI_STORE, {})):
        first, bad, worst, where = None, 0, 0.0, set()
        for rep in range(REPS):
            out = torch.zeros(N, H, W, co, device="cuda", dtype=dt)
            stats = torch.zeros(ops.STAT_STRIPES * 2 * co, dtype=torch.float64, device="cuda")
            ops.conv2d(x, ci, pw, out, co, n=N, h=H, w=W, epilogue=epi, flags=L.FLAG_STATS, stats=stats, **kw)
            torch.cuda.synchronize()
            s = stats.view(2, 32, 2, co).clone()
            if first is None:
                first = s
            elif not torch.equal(s, first):
                bad += 1
                worst = max(worst, float((s - first).abs().max()))
                where |= set((int(i[1]), int(i[2]), int(i[3]) // 8 * 8) for i in (s != first).nonzero()[:200])
        print((N, H, W, ci, co), name, f"{bad}/{REPS - 1} launches differ from the first; worst {worst:.3e}; (stripe, sum/sq, channel group) {sorted(where)[:12]}", flush=True)
