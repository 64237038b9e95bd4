import sys, numpy as np, torch
sys.path.insert(0, ".")
from tests.test_gpu_rdmodel import RD_KW, _cfg
from oracle import rdnet_ref as R
from pssr2_amd.models import RDResUNet
name = sys.argv[1] if len(sys.argv) > 1 else "rd_a"
g = np.load("tests/golden/rdmodel.npz")
model = RDResUNet(**RD_KW[name])
sd0 = {k.split("/", 1)[1]: torch.tensor(g[k]) for k in g.files if k.startswith(f"{name}_sd/")}
model.load_state_dict(sd0); model.cuda().train()
x = torch.tensor(g[f"{name}_x"]).cuda(); target = torch.tensor(g[f"{name}_target"])
y = model(x)
torch.nn.functional.mse_loss(y / 255, target.cuda() / 255).backward()
cfg = _cfg(RD_KW[name])
p64 = {k: (v.double().requires_grad_(True) if "running" not in k else v.double()) if v.dtype.is_floating_point else v for k, v in sd0.items()}
rec = {}
y64, _ = R.rdresunet_forward(x.cpu().double(), p64, cfg, train=True, record=rec)
torch.nn.functional.mse_loss(y64 / 255, target.double() / 255).backward()
print("fwd err", (y.detach().cpu().double() - y64).abs().max().item())
for pname, prm in model.named_parameters():
    truth = p64[pname].grad; got = prm.grad.cpu().double()
    fix = torch.tensor(g[f"{name}_grad/{pname}"]).double()
    scale = truth.abs().max().item()
    e = (got - truth).abs().max().item() / max(scale, 1e-30)
    print(f"{pname:70s} {e:.2e} fixnoise {(fix-truth).abs().max().item()/max(scale,1e-30):.1e} scale {scale:.2e}")
print("---- intermediate gradients")
eng = model._engine
p = list(eng.plans.values())[0]; bw = p.bwd
def cmp(name, got_nhwc, ref_nchw, coff=0):
    c = ref_nchw.shape[1]
    got = got_nhwc[..., coff:coff + c].float().cpu().permute(0, 3, 1, 2).double()
    s = ref_nchw.abs().max().item()
    print(f"{name:30s} relerr {(got - ref_nchw).abs().max().item() / s:.2e}  scale {s:.2e}")
for k in range(len(p.dec)):
    cmp(f"d decoder.{k}.out", bw.dout[k], rec[f"decoder.{k}.out"].grad)
    cmp(f"d decoder.{k}.in", bw.dcat[k], rec[f"decoder.{k}.in"].grad)
    cmp(f"  decoder.{k}.out", p.dec[k].out, rec[f"decoder.{k}.out"].detach())
    for i in range(4):
        cmp(f"  decoder.{k}.y{i}", p.dec[k].y[i], rec[f"decoder.{k}.y{i}"].detach())
for k in range(len(p.dec)):
    cmp(f"d decoder.{k}.y0 (bw.dy)", bw.dy[k], rec[f"decoder.{k}.y0"].grad)
    sc = p.shuf_c[k]
    if sc:
        cmp(f"d decoder.{k}.in[:shuf]", bw.dcat[k], rec[f"decoder.{k}.in"].grad[:, :sc])
print("---- isolate decoder.3 dgrad-mask step")
import torch.nn.functional as F
from pssr2_amd import ops, _lib as L
k = 3; blk = p.dec[k]; mod = model.decoder[k]
y3g = rec["decoder.3.y3"].grad.float()          # dy of the last conv (truth)
y2 = rec["decoder.3.y2"].detach().float()
w9 = sd0["decoder.3.conv.9.weight"]
bn = blk.bn[2]
bw_ = sd0["decoder.3.conv.7.weight"]; bb_ = sd0["decoder.3.conv.7.bias"]
mu = y2.mean((0, 2, 3)); var = y2.var((0, 2, 3), unbiased=False); is_ = 1 / torch.sqrt(var + 1e-5)
print("scale err", (bn.scale.cpu() - bw_ * is_).abs().max().item(), "shift err", (bn.shift.cpu() - (bb_ - mu * bw_ * is_)).abs().max().item(),
      "mean err", (bn.mean.cpu() - mu).abs().max().item(), "invstd err", (bn.invstd.cpu() - is_).abs().max().item())
bnout = (y2 - mu.view(1, -1, 1, 1)) * (bw_ * is_).view(1, -1, 1, 1) + bb_.view(1, -1, 1, 1)
gref = F.conv_transpose2d(y3g, w9, padding=1) * (bnout > 0)
n_, c_, h_, w_ = y3g.shape
dyd = y3g.permute(0, 2, 3, 1).contiguous().cuda()
gd = torch.zeros(n_, h_, w_, c_, device="cuda")
st = torch.zeros(ops.STAT_STRIPES, 2 * c_, dtype=torch.float64, device="cuda")
pwd = ops.pack_conv_weight(w9.cuda().contiguous(), L.F32, mode=1)
ops.conv2d(dyd, c_, pwd, gd, c_, n=n_, h=h_, w=w_, epilogue=L.EPI_DGRAD_MASK, flags=L.FLAG_STATS, aux=blk.y[2], aux_scale=bn.scale, aux_shift=bn.shift,
           aux_mean=bn.mean, aux_invstd=bn.invstd, stats=st)
got = gd.cpu().permute(0, 3, 1, 2)
print("g relerr", ((got - gref).abs().max() / gref.abs().max()).item(), "nonzero mismatch", ((got != 0) != (gref != 0)).sum().item())
print("sum g: hip", st.sum(0)[:4].cpu().numpy(), "ref", gref.double().sum((0, 2, 3))[:4].numpy())
print("conv.7.bias grad: truth", p64["decoder.3.conv.7.bias"].grad[:4].numpy(), "hip", dict(model.named_parameters())["decoder.3.conv.7.bias"].grad[:4].cpu().numpy())
