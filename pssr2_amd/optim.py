"""FusedAdamW: ``torch.optim.AdamW`` semantics with the update done by one HIP kernel per flat
buffer (csrc/optim.hip).  Parameters and gradients are flattened into one f32 buffer each so that
the step is a single launch and the data-parallel all-reduce sees a few large messages."""
from __future__ import annotations

import torch

from . import ops


class FusedAdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        if lr < 0 or eps < 0 or not 0 <= betas[0] < 1 or not 0 <= betas[1] < 1 or weight_decay < 0:
            raise ValueError("invalid AdamW hyper-parameter")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._flat = {}
        self.device_state = False      # True: step count / lr are read from device memory (hipGraph capture)

    def _flatten(self, gi, group):
        """Move the group's parameters into one flat buffer (views keep shapes and identity)."""
        params = [p for p in group["params"]]
        dev = params[0].device
        if not params[0].is_cuda:
            raise RuntimeError("FusedAdamW runs on an MI355X (HIP) device only")
        offs, tot = [], 0
        for p in params:
            offs.append(tot)
            tot += (p.numel() + 3) // 4 * 4
        flat = torch.zeros(tot, dtype=torch.float32, device=dev)
        for p, o in zip(params, offs):
            flat[o:o + p.numel()].copy_(p.data.reshape(-1))
            p.data = flat[o:o + p.numel()].view(p.shape)
        st = dict(flat=flat, grad=torch.zeros_like(flat), m=torch.zeros_like(flat), v=torch.zeros_like(flat), offs=offs, step=0,
                  ptrs=[p.data_ptr() for p in params])
        self._flat[gi] = st
        return st

    @torch.no_grad()
    def step(self, closure=None, grad_scale=1.0, amp=None):
        """``amp`` (device_state mode): ``(state, growth, backoff, interval)`` of a ``LossScaler`` kept on the device -- the step is skipped
        and the scale adapted there (pssr_adamw_step_amp), nothing comes back to the host."""
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for gi, group in enumerate(self.param_groups):
            st = self._flat.get(gi)
            params = group["params"]
            if st is None or any(p.data_ptr() != q for p, q in zip(params, st["ptrs"])):
                old = st
                st = self._flatten(gi, group)
                if old is not None and old["flat"].numel() == st["flat"].numel():
                    st["m"], st["v"], st["step"] = old["m"], old["v"], old["step"]
            g = st["grad"]
            base = getattr(params[0].grad, "_base", None) if params[0].grad is not None else None
            if (base is not None and base.dtype == torch.float32 and base.numel() == g.numel() and
                    all(p.grad is not None and p.grad._base is base and p.grad.data_ptr() == base.data_ptr() + 4 * o
                        for p, o in zip(params, st["offs"]))):
                g = base                  # the engine's flat gradient buffer has exactly our layout: no copy
            else:
                for p, o in zip(params, st["offs"]):
                    if p.grad is None:
                        g[o:o + p.numel()].zero_()
                    elif p.grad.data_ptr() != g.data_ptr() + 4 * o:
                        g[o:o + p.numel()].copy_(p.grad.reshape(-1))
            st["step"] += 1
            b1, b2 = group["betas"]
            if self.device_state:
                # step counter and learning rate live on the device: safe to capture in a hipGraph
                if "dev" not in st:
                    import struct
                    lr_bits = struct.unpack("<I", struct.pack("<f", float(group["lr"])))[0]
                    st["dev"] = torch.tensor([st["step"] - 1, lr_bits], dtype=torch.int64, device=st["flat"].device)
                    st["dev_lr"] = group["lr"]
                elif st["dev_lr"] != group["lr"] and not torch.cuda.is_current_stream_capturing():
                    import struct
                    st["dev"][1] = struct.unpack("<I", struct.pack("<f", float(group["lr"])))[0]
                    st["dev_lr"] = group["lr"]
                if amp is not None:
                    ops.adamw_step_amp(st["flat"], g, st["m"], st["v"], st["dev"], b1, b2, group["eps"], group["weight_decay"], amp[0], amp[1], amp[2],
                                       amp[3], grad_scale)
                else:
                    ops.adamw_step_dev(st["flat"], g, st["m"], st["v"], st["dev"], b1, b2, group["eps"], group["weight_decay"], grad_scale)
            else:
                ops.adamw_step(st["flat"], g, st["m"], st["v"], group["lr"], b1, b2, group["eps"], group["weight_decay"], st["step"],
                               grad_scale)
            for p in params:      # the kernel wrote the parameters behind torch's back: bump their version counters
                torch.autograd.graph.increment_version(p)
        return loss

    def sync_device_lr(self):
        """Write the groups' current learning rates to the device-side state (``device_state`` mode): a replayed hipGraph reads
        the rate from HBM, so a scheduler's change has to be pushed there (pssr2_amd/fastpath.py)."""
        import struct
        for gi, group in enumerate(self.param_groups):
            st = self._flat.get(gi)
            if st is not None and "dev" in st and st.get("dev_lr") != group["lr"]:
                st["dev"][1] = struct.unpack("<I", struct.pack("<f", float(group["lr"])))[0]
                st["dev_lr"] = group["lr"]

    def flat_grad(self, gi=0):
        """Flat gradient buffer of a group (the engine writes into it directly when it can)."""
        st = self._flat.get(gi)
        return None if st is None else st["grad"]


class LossScaler:
    """Dynamic loss scaling for ``compute_dtype = torch.float16`` (BASELINE config 4).  The reference trains in fp32 only
    (no AMP anywhere in pssr/), so this has no upstream counterpart: activations and their gradients are stored in fp16 on
    the MI355X path, hence the loss is multiplied by ``scale`` before ``backward`` and the (f32) parameter gradients are
    divided by it inside the optimizer step; a step whose gradients are not finite is skipped and the scale halved, and
    the scale doubles after ``growth_interval`` good steps (the torch.amp.GradScaler policy)."""

    def __init__(self, init_scale=2.0 ** 12, growth_factor=2.0, backoff_factor=0.5, growth_interval=200):
        self.scale_value, self.growth, self.backoff, self.interval = float(init_scale), growth_factor, backoff_factor, growth_interval
        self.good_steps, self.skipped = 0, 0
        self.amp = None         # device mode (to_device): int32[4] = scale (f32 bits), good steps, skipped steps, non-finite flag

    def scale(self, loss):
        return loss * self.scale_value

    # ---- device mode: the whole policy runs in stream order (pssr_amp_check / pssr_adamw_step_amp), so that an fp16 step -- loss scaling,
    # finiteness check, skipped-or-taken FusedAdamW step, scale update -- replays as one hipGraph without a host round trip per step
    def to_device(self, device):
        if self.amp is None or self.amp.device != torch.device(device):
            self.amp = torch.zeros(4, dtype=torch.int32, device=device)
            self.push()
        return self

    @property
    def scale_dev(self):
        """The loss scale as a device float32[1] (a view of the state: what the captured step multiplies the loss by)."""
        return self.amp.view(torch.float32)[0:1]

    def push(self):
        """Host attributes -> device state."""
        self.amp.copy_(torch.tensor([0, self.good_steps, self.skipped, 0], dtype=torch.int32))
        self.amp.view(torch.float32)[0:1].fill_(self.scale_value)

    def pull(self):
        """Device state -> host attributes (synchronises; the drivers call it when an epoch ends)."""
        if self.amp is not None:
            host = self.amp.cpu()
            self.scale_value = float(host.view(torch.float32)[0])
            self.good_steps, self.skipped = int(host[1]), int(host[2])
        return self

    def step_dev(self, optim, flat_grad, grad_scale=1.0):
        """One optimizer step under the device-side policy; ``flat_grad``: the f32 buffer holding every gradient the step reads."""
        ops.amp_check(flat_grad, self.amp)
        optim.step(grad_scale=grad_scale, amp=(self.amp, self.growth, self.backoff, self.interval))

    def step(self, optim, params):
        if self.amp is not None:            # host-side step of a scaler that also lives on the device: keep the two in step
            self.pull()
            try:
                return self._step_host(optim, params)
            finally:
                self.push()
        return self._step_host(optim, params)

    def _step_host(self, optim, params):
        grads = [p.grad for p in params if p.grad is not None]
        if not grads:
            return False
        base = getattr(grads[0], "_base", None)
        if base is not None and all(getattr(g, "_base", None) is base for g in grads):
            finite = bool(torch.isfinite(base).all())           # the engine's flat gradient buffer: one pass
        else:
            finite = all(bool(torch.isfinite(g).all()) for g in grads)
        if finite:
            if isinstance(optim, FusedAdamW):
                optim.step(grad_scale=1.0 / self.scale_value)
            else:
                for g in ([base] if base is not None and all(getattr(g, "_base", None) is base for g in grads) else grads):
                    g.mul_(1.0 / self.scale_value)
                optim.step()
            self.good_steps += 1
            if self.good_steps % self.interval == 0:
                self.scale_value *= self.growth
        else:
            self.skipped += 1
            self.good_steps = 0
            self.scale_value *= self.backoff
        return finite
