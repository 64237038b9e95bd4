import time, torch, sys, os
sys.path.insert(0, str(__import__('pathlib').Path(__file__).resolve().parents[2]))
from oracle import loss_ref, model_ref
sd = model_ref.make_state_dict(seed=1, randomize_bn=False)
params = {k: (v.clone().requires_grad_(True) if v.dtype.is_floating_point and "running" not in k else v) for k, v in sd.items()}
opt = torch.optim.AdamW([p for p in params.values() if p.requires_grad], lr=1e-3)
def step(lr, hr):
    y,_ = model_ref.resunet_forward(lr, params, 5,3,4, train=True)
    loss = loss_ref.ssim_loss(y/255, hr/255, mix=0.8)
    loss.backward(); opt.step(); opt.zero_grad()
res = 128
for b, ths in ((8, (16, 32)), (16, (32, 64)), (32, (32, 64, 128))):
    lr = torch.rand(b,1,res,res)*255; hr = torch.rand(b,1,4*res,4*res)*255
    for th in ths:
        torch.set_num_threads(th)
        t0=time.perf_counter(); step(lr, hr); tw=time.perf_counter()-t0
        t0=time.perf_counter(); step(lr, hr); t1=time.perf_counter()-t0
        print(f"batch {b} threads {th}: warm {tw:.2f} s, {t1:.2f} s/step -> {b/t1:.2f} tiles/s", flush=True)
with torch.no_grad():
    for b, th in ((4,16),(16,32),(32,64),(32,128)):
        torch.set_num_threads(th)
        lr = torch.rand(b,1,res,res)*255
        model_ref.resunet_forward(lr, params, 5,3,4, train=False)
        t0=time.perf_counter(); model_ref.resunet_forward(lr, params, 5,3,4, train=False); t1=time.perf_counter()-t0
        print(f"infer batch {b} threads {th}: {t1:.2f} s -> {b/t1:.2f} tiles/s", flush=True)
