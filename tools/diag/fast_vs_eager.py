import sys, os; sys.path.insert(0, str(__import__('pathlib').Path(__file__).resolve().parents[2])); sys.path.insert(0, str(__import__('pathlib').Path(__file__).resolve().parents[2] / 'tests'))
import io, contextlib
import numpy as np, torch
import test_gpu_fastpath as T
from pssr2_amd.optim import FusedAdamW
import pssr2_amd.optim as O
def run(graph, eps):
    orig = FusedAdamW.__init__
    def init(self, params, lr=1e-3, betas=(0.9, 0.999), eps_=eps, weight_decay=1e-2):
        orig(self, params, lr=lr, betas=betas, eps=eps_, weight_decay=weight_decay)
    FusedAdamW.__init__ = init
    try:
        with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
            return T._run_train(graph, True, scheduler=True)
    finally:
        FusedAdamW.__init__ = orig
for eps in (1e-8, 1e-3):
    a = run(True, eps); b = run(False, eps); c = run(False, eps)
    print("eps", eps)
    print(" graph :", np.array(a[0]))
    print(" eager :", np.array(b[0]))
    print(" eager2:", np.array(c[0]))
    print(" |graph-eager| rel", np.abs(np.array(a[0]) / np.array(b[0]) - 1).max(), " |eager-eager2| rel", np.abs(np.array(c[0]) / np.array(b[0]) - 1).max())
    wa = max(float((a[2][k] - b[2][k]).abs().max()) for k in a[2] if 'num_batches' not in k)
    wc = max(float((c[2][k] - b[2][k]).abs().max()) for k in a[2] if 'num_batches' not in k)
    print(" max weight diff graph-eager", wa, " eager-eager2", wc)
