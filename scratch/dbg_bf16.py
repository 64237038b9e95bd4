import sys; sys.path.insert(0, '/root/repo')
import numpy as np, torch
from pssr2_amd.models import ResUNet
from oracle import model_ref as M
g = np.load('/root/repo/tests/golden/model.npz')
name = 'tiny'
n, cin, hw, scale, depth, nlev, cout = (int(v) for v in g[f"{name}_cfg"])
sd = {k.split("/", 1)[1]: torch.tensor(g[k]) for k in g.files if k.startswith(f"{name}_sd/")}
x = torch.tensor(g[f"{name}_x"]); target = torch.tensor(g[f"{name}_target"])
def oracle(dt, autocast=False, dev="cpu"):
    params = {k: v.clone().to(dev).to(dt).requires_grad_(True) if v.dtype.is_floating_point and "running" not in k else (v.to(dev).to(dt) if v.dtype.is_floating_point else v.to(dev)) for k, v in sd.items()}
    with torch.autocast(dev, dtype=torch.bfloat16, enabled=autocast):
        yo, _ = M.resunet_forward(x.to(dev).to(dt), params, nlev, depth, scale, train=True)
    torch.nn.functional.mse_loss(yo.float() / 255, target.to(dev) / 255).backward()
    return {k: v.grad.cpu() for k, v in params.items() if getattr(v, "grad", None) is not None}
g64 = oracle(torch.float64)
gac = oracle(torch.float32, autocast=True, dev="cuda")
model = ResUNet(channels=[cin, cout], hidden=[int(v) for v in g[f"{name}_hidden"]], scale=scale, depth=depth)
model.load_state_dict(sd); model.cuda().train(); model.compute_dtype = torch.bfloat16
y = model(x.cuda())
torch.nn.functional.mse_loss(y / 255, target.cuda() / 255).backward()
def cos(a, b):
    a, b = a.flatten().double(), b.flatten().double()
    return float(torch.dot(a, b) / (a.norm() * b.norm() + 1e-30))
for pname, p in model.named_parameters():
    if pname.endswith("weight") and p.dim() == 4:
        print(f"{pname:32s} cos(hip_bf16, f64)={cos(p.grad.cpu(), g64[pname]):.4f}  cos(torch_autocast_bf16, f64)={cos(gac[pname], g64[pname]):.4f}  |g| ratio={p.grad.norm().item()/g64[pname].norm().item():.3f}")
