import sys; sys.path.insert(0, '/root/repo')
import torch
from pssr2_amd import ops
import pssr2_amd.engine as E
from pssr2_amd.models import ResUNet
from pssr2_amd.util import SSIMLoss
orig = ops.copy_f32_batch
log = []
def spy(pairs):
    log.append([d.numel() for d, s in pairs]); return orig(pairs)
ops.copy_f32_batch = spy; E.ops.copy_f32_batch = spy
m = ResUNet(channels=1).cuda(); m.compute_dtype = torch.bfloat16
x = torch.rand(32, 1, 128, 128, device="cuda") * 255; hr = torch.rand(32, 1, 512, 512, device="cuda")
l = SSIMLoss(channels=1, mix=0.8)(m(x) / 255, hr); l.backward()
torch.cuda.synchronize()
for b in log: print(len(b), "items, numel", b)
