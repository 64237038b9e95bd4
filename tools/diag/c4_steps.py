"""Per-step wall times of the c4 configuration (ResUNet 3-ch 256^2 -> 1024^2, fp16 + loss scaling, batch 8) through train_paired:
python tools/diag/c4_steps.py <tiles>"""
import sys, time; sys.path.insert(0, str(__import__('pathlib').Path(__file__).resolve().parents[2]))
import numpy as np, torch
from pssr2_amd.crappifiers import AdditiveGaussian
from pssr2_amd.data import DeviceTileDataset
from pssr2_amd.models import ResUNet
from pssr2_amd.optim import FusedAdamW
from pssr2_amd.train import train_paired
from pssr2_amd.util import SSIMLoss
n = int(sys.argv[1]) if len(sys.argv) > 1 else 72
rng = np.random.default_rng(0)
tiles = rng.integers(0, 256, (n, 3, 1024, 1024), dtype=np.uint8)
torch.manual_seed(0)
model = ResUNet(channels=3).cuda(); model.compute_dtype = torch.float16
ds = DeviceTileDataset(torch.from_numpy(tiles).cuda(), hr_res=1024, lr_scale=4, crappifier=AdditiveGaussian(13, 0, 0), val_split=0.1, rotation=True, device="cuda", seed=9)
opt = FusedAdamW(model.parameters(), lr=1e-4)
times = []
def cb():
    torch.cuda.synchronize(); times.append(time.perf_counter())
train_paired(model, ds, 8, SSIMLoss(channels=3, mix=0.8), opt, 2, device="cuda", callbacks=[cb], log_frequency=1000)
d = np.diff(np.array(times)) * 1e3
print(n, "tiles:", " ".join(f"{v:.1f}" for v in d))
