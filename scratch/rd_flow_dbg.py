import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np, torch
import test_gpu_atrous as T
from _atrous_cfgs import ATROUS_CFGS
from oracle import model_ref as M, rdnet_ref as R
g = np.load('/root/repo/tests/golden/atrous.npz')
name = "rd_atrous_psp"
model, sd, x, target = T._build(name, g)
model.compute_dtype = torch.float32
model.train()
out = model(x.cuda())
torch.nn.functional.mse_loss(out / 255, target.cuda() / 255).backward()
eng = model._engine; p = list(eng.plans.values())[-1]; bw = p.bwd
family, kw, hw, n = ATROUS_CFGS[name]
p64 = {k: (v.double().requires_grad_(True) if "running" not in k else v.double()) if v.dtype.is_floating_point else v for k, v in sd.items()}
cfg = R.RDConfig(**{k: v for k, v in kw.items() if k not in ("dilations", "pool_sizes", "encoder_pool")})
rec = {}
y, _ = R.rdresunet_forward(x.double(), p64, cfg, train=True, record=rec, dilations=kw["dilations"], pool_sizes=kw["pool_sizes"], encoder_pool=True)
torch.nn.functional.mse_loss(y / 255, target.double() / 255).backward()
def cmp(nm, got, ref):
    got = got.double().permute(0, 3, 1, 2).cpu()[:, :ref.shape[1]]
    print(f"{nm:34s} max|ref| {float(ref.abs().max()):9.2e} err {float((got - ref).abs().max()):9.2e}  rel-l2 {float((got-ref).norm()/ref.norm()):8.1e}")
print(sorted(rec.keys()))
cmp("fwd rpool out", p.rpool_out, rec["reconstruction_pool.out"].detach())
cmp("fwd epool out", p.epool_out, rec["encoder_pool.out"].detach())
cmp("fwd decoder.1.out", p.dec[1].out, rec["decoder.1.out"].detach())
cmp("fwd decoder.0.out", p.dec[0].out, rec["decoder.0.out"].detach())
cmp("bwd d(rpool out)", bw.drpool, rec["reconstruction_pool.out"].grad)
cmp("bwd d(decoder.1.out)", bw.dout[1], rec["decoder.1.out"].grad)
cmp("bwd d(decoder.1.in)", bw.dcat[1], rec["decoder.1.in"].grad)
cmp("bwd d(decoder.0.out)", bw.dout[0], rec["decoder.0.out"].grad)
cmp("bwd d(epool out)", bw.depool, rec["encoder_pool.out"].grad)
import torch.nn.functional as F
ref_dfeat = F.pixel_shuffle(rec["decoder.1.out"].grad, 2)
cmp("bwd dfeat vs oracle", bw.dfeat, ref_dfeat)
h0 = eng.h0
dfeat_nchw = bw.dfeat[..., :h0].double().permute(0, 3, 1, 2).cpu()
unsh = F.pixel_unshuffle(dfeat_nchw, 2)
got = bw.dout[1].double().permute(0, 3, 1, 2).cpu()[:, :unsh.shape[1]]
print("unshuffle(bw.dfeat) vs bw.dout[1]: err", float((got - unsh).abs().max()), "scale", float(unsh.abs().max()))
# chunk-wise error of dfeat
d = (bw.dfeat[..., :h0].double().permute(0, 3, 1, 2).cpu() - ref_dfeat).abs()
print("dfeat err per channel", [f"{float(d[:, c].max()):.1e}" for c in range(h0)])
# ---- argmax agreement of the 2x2 max pooling between the engine's and the oracle's pooled input (chunk 1 = channels 8..15 of feat)
fe = p.feat[..., :h0].double().permute(0, 3, 1, 2).cpu()[:, 8:16]
fo = F.pixel_shuffle(rec["decoder.1.out"].detach(), 2)[:, 8:16]
print("feat chunk1 fwd err", float((fe - fo).abs().max()), "scale", float(fo.abs().max()))
_, ie = F.max_pool2d(fe, 2, return_indices=True)
_, io = F.max_pool2d(fo, 2, return_indices=True)
print("argmax differs in", int((ie != io).sum()), "of", ie.numel(), "windows")
win = F.unfold(fo.reshape(-1, 1, *fo.shape[-2:]), 2, stride=2)          # [N*C, 4, L]
top2 = win.topk(2, dim=1).values
gap = (top2[:, 0] - top2[:, 1])
print("windows with top-2 gap == 0:", int((gap == 0).sum()), " gap < 1e-6:", int((gap < 1e-6).sum()), " of which max > 0:", int(((gap < 1e-6) & (top2[:, 0] > 0)).sum()))
# ---- sensitivity: the same torch f64 PSP graph on the engine's feat and on the oracle's feat, same output gradient
mod = model.reconstruction_pool; st = p.rpool
sdp = {k: v.detach().double().cpu() for k, v in mod.state_dict().items()}
dout64 = rec["reconstruction_pool.out"].grad
def psp_grad(xin):
    xin = xin.clone().requires_grad_(True)
    chunks = torch.chunk(xin, 2, dim=1)
    us = []
    for i, (ch, k) in enumerate(zip(chunks, st.sizes)):
        ch = F.interpolate(F.max_pool2d(ch, kernel_size=k), size=xin.shape[-2:], mode="bilinear")
        us.append(F.conv2d(ch, sdp[f"convs.{i}.0.weight"], sdp[f"convs.{i}.0.bias"]))
    a = torch.cat([F.relu(F.batch_norm(u, None, None, sdp[f"convs.{i}.1.weight"], sdp[f"convs.{i}.1.bias"], True, 0.1, 1e-5)) for i, u in enumerate(us)], 1)
    v = F.conv2d(a, sdp["conv_out.weight"], sdp["conv_out.bias"])
    y = F.relu(F.batch_norm(v, None, None, sdp["norm_out.weight"], sdp["norm_out.bias"], True, 0.1, 1e-5))
    y.backward(dout64)
    return xin.grad, [float(u.var(dim=(0, 2, 3), unbiased=False).min()) for u in us]
fe_all = p.feat[..., :h0].double().permute(0, 3, 1, 2).cpu()
fo_all = F.pixel_shuffle(rec["decoder.1.out"].detach(), 2)
ge, var_e = psp_grad(fe_all)
go, var_o = psp_grad(fo_all)
print("min channel variance of the chunk conv outputs:", var_o)
print("torch PSP grad on engine feat vs on oracle feat: rel-l2", float((ge - go).norm() / go.norm()), " | oracle-feat graph vs full oracle:", float((go - ref_dfeat).norm() / ref_dfeat.norm()))
