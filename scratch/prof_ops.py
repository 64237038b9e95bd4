import sys; sys.path.insert(0, '/root/repo')
import torch
from pssr2_amd.models import ResUNet
from pssr2_amd.optim import FusedAdamW
from pssr2_amd.util import SSIMLoss
torch.manual_seed(0)
model = ResUNet().cuda().train(); model.compute_dtype = torch.bfloat16
opt = FusedAdamW(model.parameters(), lr=1e-3); loss_fn = SSIMLoss(mix=0.8)
lr = torch.rand(8, 1, 128, 128, device="cuda") * 255; hr = torch.rand(8, 1, 512, 512, device="cuda") * 255
def step():
    y = model(lr); loss = loss_fn(y / 255, hr / 255); loss.backward(); opt.step(); opt.zero_grad()
for _ in range(3): step()
y = model(lr); loss = loss_fn(y / 255, hr / 255); loss.backward()
flat = model._engine._flat_grad
ps = list(model.parameters())
print("grads aliasing flat:", sum(1 for p in ps if p.grad is not None and p.grad._base is flat), "of", len(ps), "none:", sum(1 for p in ps if p.grad is None))
opt.step(); opt.zero_grad()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=False) as prof:
    step(); torch.cuda.synchronize()
ka = prof.key_averages()
rows = sorted(ka, key=lambda e: -e.count)
for e in rows[:40]:
    print(f"{e.count:6d} {e.key[:90]}")
