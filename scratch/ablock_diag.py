"""ResBlockA through the engine helpers vs torch f64 autograd."""
import sys; sys.path.insert(0,'/root/repo')
import torch, torch.nn.functional as F
from pssr2_amd import atrous as A, ops, _lib as L
from pssr2_amd.models import ResBlockA, ResUNet
from oracle import model_ref as M
torch.manual_seed(0)
for cin, c, dils, depth, n, h, w in ((40, 32, [1, 3], 1, 2, 16, 16), (16, 16, [2], 2, 1, 12, 20), (24, 64, [1, 3, 5], 0, 2, 16, 16)):
    host = ResUNet(hidden=[16, 32], depth=0, pool_sizes=[1, 2]).cuda()
    eng = host._engine
    mod = ResBlockA(cin, c, dils, depth).cuda()
    host.extra = mod                      # registers the parameters with the host model (gradient slots)
    with torch.no_grad():
        for m in mod.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.weight.uniform_(0.5, 1.5); m.bias.uniform_(-0.2, 0.2)
    eng._grad_layout(torch.device("cuda")); eng._flat_grad.zero_(); eng._side_begin(torch.device("cuda")); eng._side_on = False
    dt, code = torch.float32, L.F32
    x = torch.randn(n, cin, h, w); gy = torch.randn(n, c, h, w)
    st = A.make_ablock_state(mod, n, h, w, cin, dt, "cuda")
    cp = ops.pad_to(cin, 16)
    src = torch.zeros(n, h, w, cp, device="cuda"); src[..., :cin] = x.permute(0, 2, 3, 1).cuda()
    dst = torch.zeros(n, h, w, ops.pad_to(c, 16), device="cuda")
    mod.train()
    A.ablock_forward(eng, st, mod, src, 0, n, code, dst, 0, True)
    sd = {"b." + k: v.detach().double().cpu() for k, v in mod.state_dict().items()}
    xr = x.double().requires_grad_(True)
    params = {k: (v.clone().requires_grad_(True) if v.dtype.is_floating_point and "running" not in k else v) for k, v in sd.items()}
    y = M.resblock_a_forward(xr, params, "b", dils, depth, True, {})
    print(f"cin {cin} c {c} dils {dils} depth {depth}: fwd err", float((dst[..., :c].permute(0, 3, 1, 2).cpu().double() - y.detach()).abs().max()))
    (y * gy.double()).sum().backward()
    grads = {}
    dout = torch.zeros(n, h, w, ops.pad_to(c, 16), device="cuda"); dout[..., :c] = gy.permute(0, 2, 3, 1).cuda()
    dsrc = torch.zeros(n, h, w, cp, device="cuda")
    A.ablock_backward(eng, st, mod, grads, src, 0, n, code, dst, 0, dout, 0, dsrc, True)
    torch.cuda.synchronize()
    print("   dx err", float((dsrc[..., :cin].permute(0, 3, 1, 2).cpu().double() - xr.grad).abs().max()), "scale", float(xr.grad.abs().max()))
    for k, prm in mod.named_parameters():
        ref = params["b." + k].grad
        got = eng._gviews[eng._gindex[id(prm)]].detach().double().cpu()
        sc = float(ref.abs().max())
        print(f"   {k:28s} max|g| {sc:9.2e} err {float((got - ref).abs().max()):9.2e}")
