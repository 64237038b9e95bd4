import sys; sys.path.insert(0, '/root/repo')
import numpy as np, torch, ctypes as C
from pssr2_amd import ops, _lib as L
shape = (1, 512, 512)
shp = (3, 100, 37)
rng = np.random.default_rng(sum(shp))
base = rng.normal(120, 40, size=shp)
which = int(sys.argv[1]) if len(sys.argv) > 1 else 0
hr = np.clip(base + rng.normal(0, 5, size=shp), 0, 255).astype(np.uint8)[which:which+1]
hat = np.clip(0.7 * base + 30 + rng.normal(0, 9, size=shp), 0, 255).astype(np.uint8)[which:which+1]
dbg = torch.zeros(16, dtype=torch.float64, device="cuda")
L.lib().pssr__normalize_preds_debug(C.c_void_p(dbg.data_ptr()))
a, b = ops.normalize_preds_u8(torch.tensor(hr).cuda(), torch.tensor(hat).cuda())
torch.cuda.synchronize()
d = dbg.cpu().numpy()
x = hr[0].astype(np.float32); h = hat[0].astype(np.float32)
base_max = np.percentile(x, 99.9); base_mean = np.mean(x); x_min = np.percentile(x, 0.1)
hn = (x - np.float32(x_min)) / (np.float32(base_max) - np.float32(x_min) + np.float32(1e-20))
mean_hn = np.mean(hn); hn2 = hn - mean_hn; mean_hat = np.mean(h); h2 = h - mean_hat
var_hat = np.var(h2.flatten()); cov = np.cov(h2.flatten(), hn2.flatten())[0,1]; amp = cov / var_hat
mn = hn2.min(); aa = (hn2 - mn) * base_max; bb = (amp * h2 - mn) * base_max
names = ["base_max","base_mean","x_min","mean_hn","mean_hat","var_hat","amp","a_mean","b_mean","mn","m2","cov"]
vals = [base_max, base_mean, x_min, mean_hn, mean_hat, var_hat, amp, aa.mean(), bb.mean(), mn, np.mean(h2), cov]
for n_, v, dv in zip(names, vals, d):
    print(f"{n_:10s} ref {float(v)!r:28s} dev {float(dv)!r:28s} {'OK' if float(v)==float(dv) else 'DIFF'}")

from oracle import metrics_ref as M
wa, wb = M.normalize_preds(hr, hat)
print("a mismatches", int((a.cpu().numpy() != wa).sum()), "b mismatches", int((b.cpu().numpy() != wb).sum()))
