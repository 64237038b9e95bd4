import sys, collections; sys.path.insert(0, '/root/repo')
import torch
from pssr2_amd import ops
import pssr2_amd.engine as E, pssr2_amd.rd_engine as R
from pssr2_amd.models import RDResUNet, ResUNet
from pssr2_amd.util import SSIMLoss
which = sys.argv[1] if len(sys.argv) > 1 else "rd"
ev = []
def wrap(name, fn, key):
    def timed(*a, **kw):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); r = fn(*a, **kw); e1.record()
        ev.append((key(*a, **kw), e0, e1)); return r
    return timed
kconv = lambda x, cin0, w0, out, cout, **kw: ("conv", kw["n"], kw["h"], kw["w"], cin0, kw.get("cin1", 0), cout, w0.taps, kw.get("epilogue", 0), 2.0 * kw["n"] * kw["h"] * kw["w"] * cout * (w0.taps * cin0 + kw.get("cin1", 0)))
kwg = lambda dy, cout, x, cin_pad, taps, **kw: ("wgrad", kw["n"], kw["h"], kw["w"], cin_pad, 0, cout, taps, 0, 2.0 * kw["n"] * kw["h"] * kw["w"] * cout * taps * cin_pad)
c2 = wrap("conv", ops.conv2d, kconv); w2 = wrap("wg", ops.conv2d_wgrad_parts, kwg)
for mod in (ops, E.ops, R.ops): mod.conv2d = c2; mod.conv2d_wgrad_parts = w2
m = (RDResUNet(channels=1) if which == "rd" else ResUNet(channels=1)).cuda(); m.compute_dtype = torch.bfloat16
loss_fn = SSIMLoss(channels=1, mix=0.8)
x = torch.rand(32, 1, 128, 128, device="cuda") * 255
hr = torch.rand(32, 1, 512, 512, device="cuda")
for step in range(3):
    ev.clear()
    for p in m.parameters(): p.grad = None
    y = m(x); loss_fn(y / 255, hr).backward()
torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0])
for k, a, b in ev:
    agg[k][0] += 1; agg[k][1] += a.elapsed_time(b) * 1e3
tot = sum(v[1] for v in agg.values())
print("total conv+wgrad us", tot)
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"{k[0]:5s} n{k[1]} {k[2]:3d}x{k[3]:<3d} cin {k[4]:5d}+{k[5]:<4d} cout {k[6]:5d} taps {k[7]} epi {k[8]} x{v[0]:3d} {v[1]:9.1f} us  {k[9]*v[0]/v[1]/1e6:7.1f} TF/s")
