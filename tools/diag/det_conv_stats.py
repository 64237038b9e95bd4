"""Diagnostic: statistic rows of the 3x3 convolutions (forward: BN prologue + statistics; data gradient: ReLU mask + statistics) launch after
launch at the c2 layer shapes.  Counts launches whose f64 rows differ from the first."""
import sys; sys.path.insert(0, str(__import__('pathlib').Path(__file__).resolve().parents[2]))
import torch
from pssr2_amd import ops, _lib as L
dt = torch.bfloat16; code = L.BF16
torch.manual_seed(0)
REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 10
N = 32
for (H, W, ci, co) in [(128, 128, 64, 64), (64, 64, 128, 128), (32, 32, 256, 256), (16, 16, 512, 512)]:
    x = torch.randn(N, H, W, ci, device="cuda").to(dt)
    aux = torch.randn(N, H, W, co, device="cuda").to(dt)
    w = torch.randn(co, ci, 3, 3, device="cuda") / (ci * 9) ** 0.5
    pw = ops.pack_conv_weight(w, code)
    sc, sh = torch.rand(ci, device="cuda") + 0.5, torch.randn(ci, device="cuda") * 0.1
    asc, ash, am, ai = (torch.rand(co, device="cuda") + 0.5 for _ in range(4))
    bias = torch.randn(co, device="cuda")
    for name, kw in (("fwd pro+stats", dict(bias=bias, pro_scale=sc, pro_shift=sh, flags=L.FLAG_STATS)),
                     ("dgrad mask+stats", dict(epilogue=L.EPI_DGRAD_MASK, flags=L.FLAG_STATS, aux=aux, aux_scale=asc, aux_shift=ash, aux_mean=am, aux_invstd=ai))):
        first, bad, worst, bad_out, where = None, 0, 0.0, 0, set()
        for rep in range(REPS):
            out = torch.zeros(N, H, W, co, device="cuda", dtype=dt)
            stats = torch.zeros(ops.STAT_STRIPES * 2 * co, dtype=torch.float64, device="cuda")
            ops.conv2d(x, ci, pw, out, co, n=N, h=H, w=W, stats=stats, **kw)
            torch.cuda.synchronize()
            s = stats.view(2, 32, 2, co).clone()
            ref = out.float()
            if first is None:
                first, out0 = s, out.clone()
                # reference sums from the stored output
                if name.startswith("fwd"):
                    chk = (ref.reshape(-1, co).double().sum(0), (ref.reshape(-1, co).double() ** 2).sum(0))
                    got = s.sum((0, 1))
                    print("   sums vs torch:", float((got[0].cpu() - chk[0].cpu()).abs().max()), float((got[1].cpu() - chk[1].cpu()).abs().max() / chk[1].abs().max()))
            else:
                if not torch.equal(out, out0):
                    bad_out += 1
                if not torch.equal(s, first):
                    bad += 1
                    worst = max(worst, float((s - first).abs().max()))
                    d = (s != first)
                    where |= set((int(i[1]), int(i[2])) for i in d.nonzero()[:64])
        print((H, W, ci, co), name, f"stats differ in {bad}/{REPS - 1} launches (worst {worst:.3e}; (stripe, sum/sq) {sorted(where)[:10]}), outputs differ in {bad_out}", flush=True)
