import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np, torch
import test_gpu_atrous as T
g = np.load('/root/repo/tests/golden/atrous.npz')
model, sd, x, target = T._build("rd_atrous_psp", g)
model.compute_dtype = torch.float32
model.train()
out = model(x.cuda())
eng = model._engine
p = list(eng.plans.values())[-1]
for nm, st in (("rpool", p.rpool), ("epool", p.epool)):
    u = st.u[..., :st.C].float()
    mu = u.mean(dim=(0, 1, 2)); var = u.var(dim=(0, 1, 2), unbiased=False)
    print(nm, "C", st.C, "small", st.small)
    print("  mean   err", (st.all.mean[:st.C] - mu).abs().cpu().numpy().round(6))
    print("  invstd err", (st.all.invstd[:st.C] - 1 / torch.sqrt(var + 1e-5)).abs().cpu().numpy().round(6))
    for i, bs in enumerate(st.bn):
        sl = slice(i * st.small, (i + 1) * st.small)
        print("  chunk", i, "bs.mean err", float((bs.mean - mu[sl]).abs().max()), "bs.invstd err", float((bs.invstd - 1 / torch.sqrt(var[sl] + 1e-5)).abs().max()))

# ---- backward intermediates of the reconstruction pool vs torch f64 on the engine's own input / output gradient
import torch.nn.functional as F
torch.nn.functional.mse_loss(out / 255, target.cuda() / 255).backward()
bw = p.bwd
st = p.rpool; b = st.bwd; mod = model.reconstruction_pool
C, small = st.C, st.small
xin = p.feat[..., :C].double().permute(0, 3, 1, 2).cpu().requires_grad_(True)
dout = bw.drpool[..., :C].double().permute(0, 3, 1, 2).cpu()
sdp = {k: v.detach().double().cpu() for k, v in mod.state_dict().items()}
chunks = torch.chunk(xin, 2, dim=1)
us = []
for i, (ch, k) in enumerate(zip(chunks, st.sizes)):
    ch = F.interpolate(F.max_pool2d(ch, kernel_size=k), size=xin.shape[-2:], mode="bilinear")
    u = F.conv2d(ch, sdp[f"convs.{i}.0.weight"], sdp[f"convs.{i}.0.bias"]); u.retain_grad(); us.append(u)
a = torch.cat([F.relu(F.batch_norm(u, None, None, sdp[f"convs.{i}.1.weight"], sdp[f"convs.{i}.1.bias"], True, 0.1, 1e-5)) for i, u in enumerate(us)], 1)
v = F.conv2d(a, sdp["conv_out.weight"], sdp["conv_out.bias"]); v.retain_grad()
y = F.relu(F.batch_norm(v, None, None, sdp["norm_out.weight"], sdp["norm_out.bias"], True, 0.1, 1e-5))
print("rpool fwd err", float((y.detach() - p.rpool_out[..., :C].double().permute(0, 3, 1, 2).cpu()).abs().max()))
y.backward(dout)
def cmp(name, got, ref):
    got = got.double().permute(0, 3, 1, 2).cpu()
    print(f"  {name:10s} max|ref| {float(ref.abs().max()):9.2e}  err {float((got - ref).abs().max()):9.2e}   per-channel err {[(float((got[:, c] - ref[:, c]).abs().max())) for c in range(ref.shape[1])]}")
cmp("dv", b.dv[..., :C], v.grad)
cmp("dconv", b.dconv[..., :C], torch.cat([u.grad for u in us], 1))
cmp("dsrc", bw.dfeat[..., :C], xin.grad)
