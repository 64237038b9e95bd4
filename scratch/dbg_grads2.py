import sys; sys.path.insert(0, '/root/repo')
import numpy as np, torch
from pssr2_amd.models import ResUNet
from oracle import model_ref as M
g = np.load('/root/repo/tests/golden/model.npz')
name = 'tiny'
n, cin, hw, scale, depth, nlev, cout = (int(v) for v in g[f"{name}_cfg"])
model = ResUNet(channels=[cin, cout], hidden=[int(v) for v in g[f"{name}_hidden"]], scale=scale, depth=depth)
sd = {k.split("/", 1)[1]: torch.tensor(g[k]) for k in g.files if k.startswith(f"{name}_sd/")}
model.load_state_dict(sd); model.cuda().train()
x = torch.tensor(g[f"{name}_x"])
target = torch.tensor(g[f"{name}_target"])
# oracle with recorded intermediates
params = {k: v.clone().requires_grad_(True) if v.dtype.is_floating_point and "running" not in k else v for k, v in sd.items()}
rec = {}
yo, _ = M.resunet_forward(x, params, nlev, depth, scale, train=True, record=rec)
torch.nn.functional.mse_loss(yo / 255, target / 255).backward()
y = model(x.cuda())
loss = torch.nn.functional.mse_loss(y / 255, target.cuda() / 255)
eng = model._engine
p = eng.saved[0]
loss.backward()
bw = p.bwd
def cmp(nm, got, ref):
    got = got.float().cpu(); sc = ref.abs().max()
    d = (got - ref).abs()
    print(f"{nm:30s} maxerr/scale={d.max()/sc:.2e} nbad={(d > 1e-4*sc).sum().item()}/{d.numel()}")
    return d
nhwc = lambda t: t.permute(0, 3, 1, 2)
# forward checks
cmp("dec.l0.out", nhwc(p.dec[0].out[..., :16]), rec["decoder.1.out"].detach())
cmp("dec.l0.y3", nhwc(p.dec[0].y[3][..., :16]), rec["decoder.1.y3"].detach())
# backward checks
d = cmp("d dec.l0.out (dfeat)", nhwc(bw.dout[0][..., :16]), rec["decoder.1.out"].grad)
print("  bad locations (n,c,y,x):", torch.nonzero(d > 1e-4 * rec["decoder.1.out"].grad.abs().max())[:12].tolist())
cmp("d pre_out", nhwc(bw.dpre.view(n, hw, hw, 16, 16).permute(0,1,2,4,3).reshape(n, hw, hw, 256)), rec["reconstruction.pre_out"].grad)
cmp("d dec.l0.in (dcat0)", nhwc(bw.dcat[0][..., :24]), rec["decoder.1.in"].grad)
cmp("d dec.l1.out", nhwc(bw.dout[1][..., :32]), rec["decoder.0.out"].grad)
