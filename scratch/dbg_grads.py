import sys; sys.path.insert(0, '/root/repo')
import numpy as np, torch
from pssr2_amd.models import ResUNet
g = np.load('/root/repo/tests/golden/model.npz')
name = 'tiny'
n, cin, hw, scale, depth, nlev, cout = (int(v) for v in g[f"{name}_cfg"])
model = ResUNet(channels=[cin, cout], hidden=[int(v) for v in g[f"{name}_hidden"]], scale=scale, depth=depth)
sd = {k.split("/", 1)[1]: torch.tensor(g[k]) for k in g.files if k.startswith(f"{name}_sd/")}
model.load_state_dict(sd); model.cuda().train()
x = torch.tensor(g[f"{name}_x"]).cuda()
y = model(x)
target = torch.tensor(g[f"{name}_target"]).cuda()
loss = torch.nn.functional.mse_loss(y / 255, target / 255)
loss.backward()
for pname, p in model.named_parameters():
    ref = g[f"{name}_grad/{pname}"]
    got = p.grad.cpu().numpy()
    sc = np.abs(ref).max()
    print(f"{pname:40s} err={np.abs(got-ref).max()/(sc+1e-12):.2e} scale={sc:.2e}")
