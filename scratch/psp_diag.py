"""PSP block through the engine helpers vs torch f64 autograd, piece by piece."""
import sys; sys.path.insert(0,'/root/repo')
import torch, torch.nn.functional as F
from pssr2_amd import atrous as A, ops, _lib as L
from pssr2_amd.models import PSP_Pooling, ResUNet
torch.manual_seed(0)
C, sizes, n, h, w = 16, [1, 2], 2, 32, 32
host = ResUNet(hidden=[16, 32], depth=0, pool_sizes=sizes).cuda()      # only used as an engine host for weight caches / grad slots
eng = host._engine
mod = host.reconstruction_pool
with torch.no_grad():
    for m in mod.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.weight.uniform_(0.5, 1.5); m.bias.uniform_(-0.2, 0.2)
eng._grad_layout(torch.device("cuda")); eng._flat_grad.zero_(); eng._side_begin(torch.device("cuda")); eng._side_on = False
dt, code = torch.float32, L.F32
x = torch.randn(n, C, h, w)
gy = torch.randn(n, C, h, w)
st = A.make_psp_state(mod, n, h, w, dt, "cuda")
src = torch.zeros(n, h, w, 16, device="cuda"); src[..., :C] = x.permute(0, 2, 3, 1).cuda()
dst = torch.zeros(n, h, w, 16, device="cuda")
mod.train()
A.psp_forward(eng, st, mod, src, 0, n, code, dst, 0, True)
sd = {"p." + k: v.detach().double().cpu() for k, v in mod.state_dict().items()}
from oracle import model_ref as M
xr = x.double().requires_grad_(True)
params = {k: (v.clone().requires_grad_(True) if v.dtype.is_floating_point and "running" not in k else v) for k, v in sd.items()}
y = M.psp_forward(xr, params, "p", sizes, True, {})
print("fwd err", float((dst[..., :C].permute(0, 3, 1, 2).cpu().double() - y.detach()).abs().max()))
(y * gy.double()).sum().backward()
grads = {}
dout = torch.zeros(n, h, w, 16, device="cuda"); dout[..., :C] = gy.permute(0, 2, 3, 1).cuda()
dsrc = torch.zeros(n, h, w, 16, device="cuda")
A.psp_backward(eng, st, mod, grads, src, 0, n, code, dst, 0, dout, 0, dsrc, 0)
torch.cuda.synchronize()
print("dx err", float((dsrc[..., :C].permute(0, 3, 1, 2).cpu().double() - xr.grad).abs().max()), "scale", float(xr.grad.abs().max()))
for k, prm in mod.named_parameters():
    ref = params["p." + k].grad
    got = eng._gviews[eng._gindex[id(prm)]].detach().double().cpu()
    print(f"{k:24s} max|g| {float(ref.abs().max()):9.2e} err {float((got - ref).abs().max()):9.2e}")
