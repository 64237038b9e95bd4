import sys; sys.path.insert(0, str(__import__('pathlib').Path(__file__).resolve().parents[2]))
import torch
from pssr2_amd.util import SSIMLoss
x = torch.rand(32, 1, 512, 512, device="cuda", requires_grad=True); y = torch.rand(32, 1, 512, 512, device="cuda")
lf = SSIMLoss(mix=0.8)
from torch.profiler import profile, ProfilerActivity
for _ in range(3):
    l = lf(x, y); l.backward()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    for _ in range(5):
        l = lf(x, y); l.backward()
    torch.cuda.synchronize()
for e in sorted(prof.key_averages(), key=lambda e: -e.device_time_total)[:8]:
    print(f"{e.device_time_total/5:9.1f} us/iter  {e.count//5:3d} calls  {e.key[:80]}")
