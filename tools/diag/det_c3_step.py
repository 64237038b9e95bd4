"""The c3 training step run several times from the same state: which parameter gradients differ between runs, and by how much
(tests/test_gpu_fullsize.py::test_config_c3_training_step_full_size asks for bit equality)."""
import sys; sys.path.insert(0, str(__import__('pathlib').Path(__file__).resolve().parents[2]))
import torch
from pssr2_amd.models import RDResUNet
from pssr2_amd.util import SSIMLoss
torch.manual_seed(0)
model = RDResUNet(channels=1).cuda().train(); model.compute_dtype = torch.bfloat16
g = torch.Generator().manual_seed(5)
x = (torch.rand(32, 1, 128, 128, generator=g) * 255).cuda()
hr = (torch.rand(32, 1, 512, 512, generator=g) * 255).cuda()
loss_fn = SSIMLoss(mix=0.8)
names = [n for n, _ in model.named_parameters()]
def run():
    for p in model.parameters(): p.grad = None
    for b in model.modules():
        if isinstance(b, torch.nn.BatchNorm2d): b.reset_running_stats()
    loss = loss_fn(model(x) / 255, hr / 255); loss.backward()
    torch.cuda.synchronize()
    return float(loss), [p.grad.detach().clone() for p in model.parameters()]
ref = run()
for k in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
    cur = run()
    bad = [(n, float((a - b).abs().max()), float(a.abs().max())) for n, a, b in zip(names, ref[1], cur[1]) if not torch.equal(a, b)]
    print(k, "loss equal", ref[0] == cur[0], "differing:", bad[:6])
# which pass is right: the same step with everything on one stream (PSSR_WGRAD_STREAM=0 semantics)
model._engine.side_wgrad = False
one = run()
for tag, other in (("first pass", ref), ("later pass", cur)):
    bad = [(n, float((a - b).abs().max()), float(a.abs().max())) for n, a, b in zip(names, one[1], other[1]) if not torch.equal(a, b)]
    print("one stream vs", tag, ":", bad[:8])
