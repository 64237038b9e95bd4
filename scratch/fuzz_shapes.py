import sys; sys.path.insert(0, '/root/repo')
import torch
from pssr2_amd.models import RDResUNet, ResUNet
from pssr2_amd.util import SSIMLoss
torch.manual_seed(0)
for name, mk in (("resunet", lambda: ResUNet(channels=1)), ("rdresunet", lambda: RDResUNet(channels=1))):
    for dt in (torch.bfloat16, torch.float32):
        m = mk().cuda(); m.compute_dtype = dt
        loss_fn = SSIMLoss(channels=1, mix=0.8)
        for n, s in ((3, 48), (5, 80), (1, 112), (2, 160)):
            if s % 16: continue
            x = torch.rand(n, 1, s, s, device="cuda") * 255
            hr = torch.rand(n, 1, 4 * s, 4 * s, device="cuda")
            m.train()
            for p in m.parameters(): p.grad = None
            y = m(x); l = loss_fn(y / 255, hr); l.backward()
            g = sum(float(p.grad.abs().sum()) for p in m.parameters() if p.grad is not None)
            m.eval()
            with torch.no_grad(): ye = m(x)
            ok = torch.isfinite(y).all().item() and torch.isfinite(ye).all().item() and g == g and g < 1e30
            print(f"{name:10s} {str(dt)[6:]:9s} n={n} {s}x{s}: loss {float(l):.5f} |grad| {g:.4e} finite={ok}")
            assert ok
print("fuzz ok")
