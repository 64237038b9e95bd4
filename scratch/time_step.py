import sys, time; sys.path.insert(0, '/root/repo')
import torch
from pssr2_amd.models import ResUNet
bs = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dt = torch.bfloat16 if (len(sys.argv) < 3 or sys.argv[2] == "bf16") else torch.float32
torch.manual_seed(0)
model = ResUNet().cuda().train(); model.compute_dtype = dt
x = torch.rand(bs, 1, 128, 128, device="cuda") * 255
hr = torch.rand(bs, 1, 512, 512, device="cuda") * 255
def step():
    y = model(x)
    loss = torch.nn.functional.mse_loss(y / 255, hr / 255)
    loss.backward()
    for p in model.parameters(): p.grad = None
for _ in range(2): step()
torch.cuda.synchronize(); t = time.time()
n = 5
for _ in range(n): step()
torch.cuda.synchronize(); dtm = (time.time() - t) / n
print(f"batch {bs} {dt}: {dtm*1e3:.1f} ms/step -> {bs/dtm:.1f} tiles/s (fwd+bwd, no optimizer)")
model.eval()
with torch.no_grad():
    for _ in range(2): model(x)
    torch.cuda.synchronize(); t = time.time()
    for _ in range(n): model(x)
    torch.cuda.synchronize(); dtm = (time.time() - t) / n
print(f"infer: {dtm*1e3:.1f} ms/batch -> {bs/dtm:.1f} tiles/s; mem {torch.cuda.max_memory_allocated()/2**30:.1f} GiB")
