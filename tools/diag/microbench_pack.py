"""weight re-pack (pssr_pack_conv_weight): 1024 x 1024 x 3 x 3 f32 -> packed bf16, forward (mode 0) and data-gradient (mode 1) layouts"""
import sys; sys.path.insert(0, str(__import__('pathlib').Path(__file__).resolve().parents[2]))
import torch
from pssr2_amd import ops, _lib as L
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for co, ci in ((1024, 1024), (512, 768), (64, 64)):
    w = torch.randn(co, ci, 3, 3, device="cuda")
    for mode in (0, 1):
        pw = ops.pack_conv_weight(w, L.BF16, mode=mode)
        t = timeit(lambda: ops.pack_conv_weight(w, L.BF16, mode=mode, out=pw))
        print(f"{co}x{ci} mode {mode}: {t:7.1f} us  {w.numel() * 6 / t / 1e6:.2f} TB/s")
