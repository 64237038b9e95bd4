import sys; sys.path.insert(0, '/root/repo')
import numpy as np, torch
from pssr2_amd.models import ResUNet
from pssr2_amd import engine as E, ops
from oracle import model_ref as M
g = np.load('/root/repo/tests/golden/model.npz')
name = 'tiny'
n, cin, hw, scale, depth, nlev, cout = (int(v) for v in g[f"{name}_cfg"])
model = ResUNet(channels=[cin, cout], hidden=[int(v) for v in g[f"{name}_hidden"]], scale=scale, depth=depth)
sd = {k.split("/", 1)[1]: torch.tensor(g[k]) for k in g.files if k.startswith(f"{name}_sd/")}
model.load_state_dict(sd); model.cuda().train()
x = torch.tensor(g[f"{name}_x"])
target = torch.tensor(g[f"{name}_target"])
params = {k: v.clone().requires_grad_(True) if v.dtype.is_floating_point and "running" not in k else v for k, v in sd.items()}
rec = {}
yo, _ = M.resunet_forward(x, params, nlev, depth, scale, train=True, record=rec)
torch.nn.functional.mse_loss(yo / 255, target / 255).backward()
seq = []
o_apply, o_relu, o_conv = ops.bn_bwd_apply, ops.relu_bwd_stats, ops.conv2d
def w_apply(g_, y, a, b, c, dy, npix, cc, dtype, **kw):
    o_apply(g_, y, a, b, c, dy, npix, cc, dtype, **kw); torch.cuda.synchronize(); seq.append(("dy", dy.clone(), g_.clone()))
def w_relu(dout, out, y, mean, invstd, dz, stats, npix, c, dtype, **kw):
    o_relu(dout, out, y, mean, invstd, dz, stats, npix, c, dtype, **kw); torch.cuda.synchronize(); seq.append(("dz", dz.clone(), stats.clone()))
ops.bn_bwd_apply, ops.relu_bwd_stats = w_apply, w_relu
y = model(x.cuda())
loss = torch.nn.functional.mse_loss(y / 255, target.cuda() / 255)
loss.backward()
def cmp(nm, got, ref):
    got = got.float().cpu(); sc = ref.abs().max()
    d = (got - ref).abs()
    print(f"{nm:30s} maxerr/scale={d.max()/sc:.2e} nbad={(d > 1e-4*sc).sum().item()}/{d.numel()}")
    return d
nhwc = lambda t: t.permute(0, 3, 1, 2)
# first block processed = decoder.1 (level 0): seq[0]=dz, seq[1]=dy3, seq[2]=dy2, seq[3]=dy1, seq[4]=dy0
print([s[0] for s in seq[:6]])
out = rec["decoder.1.out"]; dz_ref = out.grad * (out > 0)
cmp("dz", nhwc(seq[0][1][..., :16]), dz_ref)
st = seq[0][2].cpu()
print("sum dz gpu/ref:", st[:4].tolist(), dz_ref.double().sum((0, 2, 3))[:4].tolist())
for i, k in enumerate([3, 2, 1, 0]):
    d = cmp(f"dy{k}", nhwc(seq[1 + i][1][..., :16]), rec[f"decoder.1.y{k}"].grad)
    if i:
        # g (masked dgrad) reference: grad wrt post-relu activation masked
        pass
    bad = torch.nonzero(d > 1e-4 * rec[f"decoder.1.y{k}"].grad.abs().max())
    if len(bad):
        print("   first bad (n,c,y,x):", bad[:8].tolist(), " chan hist:", torch.bincount(bad[:, 1], minlength=16).tolist())
