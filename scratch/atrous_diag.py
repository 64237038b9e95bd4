import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np, torch
import test_gpu_atrous as T
g = np.load('/root/repo/tests/golden/atrous.npz')
for name in sys.argv[1:]:
    model, sd, x, target = T._build(name, g)
    model.compute_dtype = torch.float32
    model.train()
    out = model(x.cuda())
    torch.nn.functional.mse_loss(out / 255, target.cuda() / 255).backward()
    _, g64 = T._oracle64(name, sd, x, target)
    for k, prm in model.named_parameters():
        ref64 = g64[k]; got = prm.grad.detach().double().cpu(); fix = torch.tensor(g[f"{name}_grad/{k}"]).double()
        sc = float(ref64.abs().max()) + 1e-30
        d = (got - ref64).abs()
        print(f"{name} {k:45s} max|g| {sc:9.2e} hip max {float(d.max())/sc:8.1e} l2 {float(d.norm()/ (ref64.norm()+1e-30)):8.1e} | fixture max {float((fix-ref64).abs().max())/sc:8.1e} l2 {float((fix-ref64).norm()/(ref64.norm()+1e-30)):8.1e}  n>{1e-3:.0e}: {int((d/sc>1e-3).sum())}/{d.numel()}")
