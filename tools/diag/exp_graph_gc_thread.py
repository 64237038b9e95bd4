"""Does destroying a thread_local-captured torch.cuda.CUDAGraph (and its pool) from ANOTHER thread abort?  (GPU test run r3c aborted
inside a garbage collection on the autograd thread after a failed test had left captured graphs behind.)"""
import gc, sys, threading
import torch
mode = sys.argv[1] if len(sys.argv) > 1 else "thread_local"
x = torch.zeros(1 << 20, device="cuda")
def make():
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode=mode):
        y = x * 2 + 1
        z = torch.empty(1 << 22, device="cuda").fill_(3.0)
    g.replay()
    torch.cuda.synchronize()
    cyc = {"g": g, "y": y, "z": z}
    cyc["self"] = cyc                # cyclic garbage: only the collector frees it
make()
def other():
    w = torch.ones(8, device="cuda")
    for _ in range(3):
        gc.collect()
        w = w + 1
    torch.cuda.synchronize()
    print("other thread done", float(w[0]))
t = threading.Thread(target=other); t.start(); t.join()
# the same from an autograd worker thread
class F(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a): return a * 2
    @staticmethod
    def backward(ctx, g):
        gc.collect()
        return g * 2
make()
a = torch.ones(4, device="cuda", requires_grad=True)
F.apply(a).sum().backward()
torch.cuda.synchronize()
print("ok", mode)
