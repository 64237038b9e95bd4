"""Where does the host-fed training step (train_paired over a DataLoader) lose its 0.7 ms against the device-resident one?  The same stepper
fed (a) one pinned batch over and over -- no loader at all --, (b) fresh pinned batches made by a thread, (c) the real DataLoader."""
import sys, time; sys.path.insert(0, str(__import__('pathlib').Path(__file__).resolve().parents[2]))
import numpy as np
import torch
from pssr2_amd import fastpath
from pssr2_amd.models import ResUNet
from pssr2_amd.optim import FusedAdamW
from pssr2_amd.util import SSIMLoss

dev = "cuda"
torch.manual_seed(0)
model = ResUNet().to(dev)
model.compute_dtype = torch.bfloat16
model.train()
opt = FusedAdamW(model.parameters(), lr=1e-3)
B = 32
st = fastpath.TrainStepper(model, None, B, SSIMLoss(mix=0.8), opt, False, 255, None, 0, dev, host=True)
hr = (torch.rand(B, 1, 512, 512) * 255).pin_memory()
lr = (torch.rand(B, 1, 128, 128) * 255).pin_memory()


def run(next_batch, n=60, warm=10, label=""):
    for i in range(n + warm):
        if i == warm:
            torch.cuda.synchronize(); t0 = time.perf_counter()
        st.step(next_batch())
    torch.cuda.synchronize()
    print(f"{label:60s} {(time.perf_counter() - t0) / n * 1e3:8.3f} ms / step", flush=True)


run(lambda: (hr, lr), label="(a) the same pinned batch every step")
pool = [((torch.rand(B, 1, 512, 512) * 255).pin_memory(), (torch.rand(B, 1, 128, 128) * 255).pin_memory()) for _ in range(4)]
k = [0]
def rot():
    k[0] += 1
    return pool[k[0] % 4]
run(rot, label="(b) four pinned batches in rotation")
def fresh():
    return (hr.clone().pin_memory(), lr.clone().pin_memory())
run(fresh, label="(c) a new pinned allocation + 35 MB host copy per step on the launching thread")
def unpinned():
    return (hr.clone(), lr.clone())
run(unpinned, label="(d) unpinned host batches")
