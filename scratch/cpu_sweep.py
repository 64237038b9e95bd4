import time, torch, sys, os
sys.path.insert(0,'/root/repo')
from oracle import loss_ref, model_ref
print("cpus", os.cpu_count(), "threads", torch.get_num_threads(), flush=True)
sd = model_ref.make_state_dict(seed=1, randomize_bn=False)
params = {k: (v.clone().requires_grad_(True) if v.dtype.is_floating_point and "running" not in k else v) for k, v in sd.items()}
opt = torch.optim.AdamW([p for p in params.values() if p.requires_grad], lr=1e-3)
def step(lr, hr):
    y,_ = model_ref.resunet_forward(lr, params, 5,3,4, train=True)
    loss = loss_ref.ssim_loss(y/255, hr/255, mix=0.8)
    loss.backward(); opt.step(); opt.zero_grad()
for res in (64, 128):
    lr = torch.rand(4,1,res,res)*255; hr = torch.rand(4,1,4*res,4*res)*255
    for th in (8, 16, 32, 64, 128):
        if th > os.cpu_count(): continue
        torch.set_num_threads(th)
        step(lr, hr)
        ts=[]
        for i in range(2):
            t0=time.perf_counter(); step(lr, hr); ts.append(time.perf_counter()-t0)
        print(f"lr_res {res} threads {th}: {min(ts):.2f} s/step -> {4/min(ts):.2f} tiles/s", flush=True)
