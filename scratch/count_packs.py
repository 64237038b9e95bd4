import sys, collections, traceback; sys.path.insert(0, '/root/repo')
import torch
from pssr2_amd import ops
from pssr2_amd.models import ResUNet
from pssr2_amd.optim import FusedAdamW
calls = collections.Counter()
orig = ops.pack_conv_weight
def counted(w, dtype, **kw):
    st = traceback.extract_stack(limit=4)
    calls[(tuple(w.shape), kw.get("mode"), st[0].name, st[0].lineno, st[1].name, st[1].lineno)] += 1
    return orig(w, dtype, **kw)
ops.pack_conv_weight = counted
m = ResUNet(channels=1).cuda(); m.compute_dtype = torch.bfloat16
opt = FusedAdamW(m.parameters(), lr=1e-4)
x = torch.rand(4, 1, 64, 64, device="cuda") * 255
for step in range(3):
    calls.clear()
    opt.zero_grad()
    y = m(x); (y.mean()).backward(); opt.step()
    print("step", step, sum(calls.values()))
for k, v in calls.items(): print(v, k)
