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
names = ["layernorm2d_bwd", "layernorm2d_fwd", "ese_bwd", "ese_gate", "channel_sum_nhwc", "image_channel_dot", "scale_nc"]
def mk(name):
    def key(*a, **kw):
        ints = tuple(x for x in a if isinstance(x, int) and not isinstance(x, bool))[:5]
        shapes = tuple(tuple(x.shape) for x in a[:2] if isinstance(x, torch.Tensor))
        return (name, ints, shapes, tuple(sorted((k, v) for k, v in kw.items() if isinstance(v, (int, bool)))))
    return key
for nm in names:
    f = wrap(nm, getattr(ops, nm), mk(nm))
    for mod in (ops, R.ops): setattr(mod, nm, f)
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
byname = collections.defaultdict(float)
for k, v in agg.items(): byname[k[0]] += v[1]
print({k: round(v) for k, v in byname.items()})
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
    print(f"{k[0]:18s} x{v[0]:3d} avg {v[1]/v[0]:8.1f} us  {k[1]} {k[2]} {k[3]}")
