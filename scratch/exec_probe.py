import subprocess, sys, torch
print("before init: child rc", subprocess.run([sys.executable, "-c", "print('child ok (parent not initialised)')"]).returncode, flush=True)
torch.zeros(1).cuda(); torch.cuda.synchronize()
print("cuda initialised in parent", flush=True)
r = subprocess.run([sys.executable, "-c", "import torch; print('child ok, cuda', torch.zeros(1).cuda().item())"], capture_output=True, text=True)
print("after init: child rc", r.returncode, r.stdout.strip(), r.stderr.strip()[-300:], flush=True)
import torch.multiprocessing as mp
def w(q): q.put(7)
ctx = mp.get_context("spawn"); q = ctx.Queue(); p = ctx.Process(target=w, args=(q,)); p.start(); p.join(60)
print("spawn child exit", p.exitcode, q.get(timeout=5) if p.exitcode == 0 else None, flush=True)
