import sys; sys.path.insert(0, '/root/repo')
import numpy as np, torch
from pssr2_amd.models import ResUNet
from oracle import model_ref as M
g = np.load('/root/repo/tests/golden/model.npz')
name = sys.argv[1] if len(sys.argv) > 1 else 'tiny'
n, cin, hw, scale, depth, nlev, cout = (int(v) for v in g[f"{name}_cfg"])
model = ResUNet(channels=[cin, cout], hidden=[int(v) for v in g[f"{name}_hidden"]], scale=scale, depth=depth)
sd = {k.split("/", 1)[1]: torch.tensor(g[k]) for k in g.files if k.startswith(f"{name}_sd/")}
model.load_state_dict(sd); model.cuda().train()
x = torch.tensor(g[f"{name}_x"]); target = torch.tensor(g[f"{name}_target"])
def oracle(dt):
    params = {k: v.clone().to(dt).requires_grad_(True) if v.dtype.is_floating_point and "running" not in k else (v.to(dt) if v.dtype.is_floating_point else v) for k, v in sd.items()}
    yo, _ = M.resunet_forward(x.to(dt), params, nlev, depth, scale, train=True)
    torch.nn.functional.mse_loss(yo / 255, target.to(dt) / 255).backward()
    return {k: v.grad for k, v in params.items() if getattr(v, "grad", None) is not None}
g64, g32 = oracle(torch.float64), oracle(torch.float32)
y = model(x.cuda())
torch.nn.functional.mse_loss(y / 255, target.cuda() / 255).backward()
rows = []
for pname, p in model.named_parameters():
    t = g64[pname]; sc = t.abs().max().item() + 1e-30
    e_gpu = (p.grad.cpu().double() - t).abs().max().item() / sc
    e_cpu = (g32[pname].double() - t).abs().max().item() / sc
    e_fix = (torch.tensor(g[f"{name}_grad/{pname}"]).double() - t).abs().max().item() / sc
    rows.append((pname, e_gpu, e_cpu, e_fix, sc))
for r in rows:
    if r[4] > 1e-8: print(f"{r[0]:36s} gpu={r[1]:.1e} cpu_f32={r[2]:.1e} reference_f32={r[3]:.1e}")
