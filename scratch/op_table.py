import sys, collections, types; sys.path.insert(0, '/root/repo')
import os; os.environ["PSSR_WGRAD_STREAM"] = "0"
import torch
from pssr2_amd import ops
from pssr2_amd.models import RDResUNet, ResUNet
from pssr2_amd.util import SSIMLoss
which = sys.argv[1] if len(sys.argv) > 1 else "resunet"
ev = []
def wrap(name, fn):
    def timed(*a, **kw):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); r = fn(*a, **kw); e1.record()
        ints = tuple(x for x in a if isinstance(x, int) and not isinstance(x, bool))[:4] + tuple(v for k, v in sorted(kw.items()) if k in ("n", "h", "w", "c", "npix") )
        ev.append(((name, ints), e0, e1)); return r
    return timed
skip = {"dtype_code", "pad_to", "pack_conv_weight", "packed_weight_bytes"}
for nm in dir(ops):
    f = getattr(ops, nm)
    if isinstance(f, types.FunctionType) and not nm.startswith("_") and nm not in skip and f.__module__ == ops.__name__:
        setattr(ops, nm, wrap(nm, f))
m = (RDResUNet(channels=1) if which != "resunet" else ResUNet(channels=1)).cuda(); m.compute_dtype = torch.bfloat16
loss_fn = SSIMLoss(channels=1, mix=0.8)
x = torch.rand(32, 1, 128, 128, device="cuda") * 255
hr = torch.rand(32, 1, 512, 512, device="cuda")
for step in range(3):
    ev.clear()
    for p in m.parameters(): p.grad = None
    y = m(x); loss_fn(y / 255, hr).backward()
torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0]); byname = collections.defaultdict(lambda: [0, 0.0])
for k, a, b in ev:
    t = a.elapsed_time(b) * 1e3
    agg[k][0] += 1; agg[k][1] += t; byname[k[0]][0] += 1; byname[k[0]][1] += t
print("total us", sum(v[1] for v in byname.values()))
for k, v in sorted(byname.items(), key=lambda kv: -kv[1][1]): print(f"{k:28s} x{v[0]:4d} {v[1]:9.1f} us  avg {v[1]/v[0]:7.1f}")
print()
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:50]:
    if k[0] in ("conv2d", "conv2d_wgrad_parts"): continue
    print(f"{k[0]:24s} x{v[0]:3d} avg {v[1]/v[0]:8.1f} us  {k[1]}")
