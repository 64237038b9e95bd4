import sys; sys.path.insert(0, '/root/repo')
import numpy as np, torch
from pssr2_amd.models import ResUNet
from pssr2_amd import engine as E
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
snaps = {}
orig = E.Engine._block_backward
def wrapped(self, p, bw, grads, blk, module, src, cin, first, out_buf, out_coff, dout, dsrc, dsrc_c):
    key = [k for k, m in model.named_modules() if m is module][0]
    snaps[key + ".dout"] = dout.clone()
    orig(self, p, bw, grads, blk, module, src, cin, first, out_buf, out_coff, dout, dsrc, dsrc_c)
    torch.cuda.synchronize()
    snaps[key + ".dsrc"] = dsrc.clone()
    snaps[key + ".dz"] = bw.dz[blk.level].clone()
    snaps[key + ".dy0"] = bw.dy[blk.level].clone()
E.Engine._block_backward = wrapped
y = model(x.cuda())
loss = torch.nn.functional.mse_loss(y / 255, target.cuda() / 255)
loss.backward()
def cmp(nm, got, ref):
    got = got.float().cpu(); sc = ref.abs().max()
    d = (got - ref).abs()
    print(f"{nm:30s} maxerr/scale={d.max()/sc:.2e} nbad={(d > 1e-4*sc).sum().item()}/{d.numel()}")
    return d
nhwc = lambda t: t.permute(0, 3, 1, 2)
for j, c in ((1, 16), (0, 32)):
    k = f"decoder.{j}"
    d = cmp(k + ".dout", nhwc(snaps[k + ".dout"][..., :c]), rec[k + ".out"].grad)
    cmp(k + ".dy0", nhwc(snaps[k + ".dy0"][..., :c]), rec[k + ".y0"].grad)
    cin = rec[k + ".in"].shape[1]
    d = cmp(k + ".dsrc", nhwc(snaps[k + ".dsrc"][..., :cin]), rec[k + ".in"].grad)
    bad = torch.nonzero(d > 1e-4 * rec[k + ".in"].grad.abs().max())
    print("   bad channel histogram:", torch.bincount(bad[:, 1], minlength=cin).tolist())
for i, c in ((2, 64), (1, 32), (0, 16)):
    k = f"encoder.{i}"
    cmp(k + ".dout", nhwc(snaps[k + ".dout"][..., :c]), rec[k + ".out"].grad)
    cmp(k + ".dy0", nhwc(snaps[k + ".dy0"][..., :c]), rec[k + ".y0"].grad)
