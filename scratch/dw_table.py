import sys, collections; sys.path.insert(0, '/root/repo')
import os; os.environ["PSSR_WGRAD_STREAM"] = "0"
import torch
from pssr2_amd import ops
import pssr2_amd.rd_engine as R
from pssr2_amd.models import RDResUNet
from pssr2_amd.util import SSIMLoss
ev = []
def wrap(name, fn, key):
    def timed(*a, **kw):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); r = fn(*a, **kw); e1.record()
        ev.append((key(*a, **kw), e0, e1)); return r
    return timed
kf = lambda x, wp, bias, out, n, h, w, c, dtype, **kw: ("dw", n, h, w, c, x.shape[-1], out.shape[-1], int(kw.get("accumulate", False)))
kw_ = lambda dy, x, dw, n, h, w, c, dtype, **kw: ("dwwg", n, h, w, c, dy.shape[-1], x.shape[-1], 0)
f2 = wrap("dw", ops.dwconv7, kf); w2 = wrap("wg", ops.dwconv7_wgrad, kw_)
for mod in (ops, R.ops): mod.dwconv7 = f2; mod.dwconv7_wgrad = w2
m = RDResUNet(channels=1).cuda(); m.compute_dtype = torch.bfloat16
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
print("total dw us", sum(v[1] for v in agg.values()))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    el = k[1] * k[2] * k[3] * k[4]
    t = v[1] / v[0]
    print(f"{k[0]:5s} n{k[1]} {k[2]:3d}x{k[3]:<3d} c {k[4]:5d} strides {k[5]:5d}/{k[6]:<5d} acc {k[7]} x{v[0]:3d} avg {t:8.1f} us  {el*98/t/1e6:6.1f} TF/s  {el*4/t/1e3:7.1f} GB/s(alg)")
