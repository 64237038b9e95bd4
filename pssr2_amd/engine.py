"""Execution engine of the MI355X ResUNet: the whole forward and backward pass as an explicit
sequence of libpssr_mi355.so launches over pre-allocated NHWC buffers.

The reference runs ``ResUNet.forward`` (pssr/models/resunet.py:65-96) op by op through torch
autograd.  Here one ``torch.autograd.Function`` covers the network: ``Engine.forward`` issues the
fused kernels (conv + BatchNorm statistics epilogue, BatchNorm+ReLU prologue, residual tail, pool,
pixel-shuffle into the concat buffer, two-source head, blocked-layout final conv) and keeps only the
raw conv outputs; ``Engine.backward`` walks the same structure in reverse.  torch supplies device
memory, the stream and the autograd hook — no torch operator computes on the hot path.
"""
from __future__ import annotations


import torch

from . import _lib as L
from . import ops

BN_EPS, BN_MOMENTUM = 1e-5, 0.1     # torch.nn.BatchNorm2d defaults used by the reference blocks


def _blocked_order(r: int):
    """How Reconstruction's F.pixel_shuffle(x, r) (pssr/models/_blocks.py:17) is carried out: (blk, explicit).  Power-of-two factors: the
    high-resolution tensor IS `pre`'s output read in sub-pixel-blocked order (blk = log2 r: the kernels index it with shifts, nothing is
    moved).  Any other factor (3, 5, 6 ...): blk = 0 and an explicit shuffle between `pre`'s low-resolution output and a plain
    high-resolution tensor (pssr_pixel_shuffle; its inverse on the way back)."""
    if r < 1:
        raise ValueError(f"upscaling factor must be a positive integer, got scale={r}")
    l = r.bit_length() - 1
    return (l, False) if (1 << l) == r else (0, True)


_COPY_BATCH = __import__("os").environ.get("PSSR_COPY_BATCH", "1") != "0"
# PSSR_MATERIALISE=1 (off by default): write relu(bn(y)) out once per layer under the forward pass so that EVERY 3x3 weight gradient takes
# the all-DMA kernel.  Measured (c2 step): the kernels get 20 % faster stand-alone (57 -> 46 us per layer) and the step does not move
# (11.51 / 11.49 vs 11.37 / 11.53 ms): the backward phase is two balanced queues, and the forward convolutions slow down by what the
# 27 extra launches cost
_ABLATE_XCOL = __import__("os").environ.get("PSSR_ABLATE_XCOL", "0") == "1"      # timing ablation (wrong gradients): skip the two passes over d(pre) that serve the input channel
_XCOL_SIDE = __import__("os").environ.get("PSSR_XCOL_SIDE", "0") == "1"
_NO_MATERIALISE = __import__("os").environ.get("PSSR_MATERIALISE", "0") != "1"
# PSSR_ABLATE=<comma list> (timing ablations for tools/diag/ab_env.sh: WRONG gradients on purpose, never set in a product run): leave single
# launches of the step out to price what folding them into a neighbour could buy at most -- xdgrad / xwgrad (the two passes over d(pre) that
# serve the input channel), poolbwd (maxpool2_bwd), unshuf (inverse pixel shuffle), shuf (forward pixel shuffle), pool (forward max pool),
# apply (bn_bwd_apply), relustats (relu_bwd_stats), unpack (partial-slab reduction of the weight gradients), gzero (flat gradient memset)
_FUSE_DOUT = __import__("os").environ.get("PSSR_FUSE_DOUT", "1") != "0"
_OVERWRITE_GRADS = __import__("os").environ.get("PSSR_OVERWRITE_GRADS", "1") != "0"
_EVAL_SHUF = __import__("os").environ.get("PSSR_EVAL_SHUF", "1") != "0"         # eval mode: F.pixel_shuffle(x, 2) done by the producing conv's stores (FLAG_SHUF2)
_EVAL_AFFINE = __import__("os").environ.get("PSSR_EVAL_AFFINE", "1") != "0"     # eval mode: BatchNorm + ReLU in the producing conv's epilogue (FLAG_AFFINE)
_HEAD_FUSE = __import__("os").environ.get("PSSR_HEAD_FUSE", "1") != "0"      # eval mode: Reconstruction.conv inside pre's epilogue (EPI_HEADQ)
_ABL = frozenset(x for x in __import__("os").environ.get("PSSR_ABLATE", "").split(",") if x)

class _Arena:
    """Bump allocator for the many small per-channel vectors (one memset zeroes all statistics)."""

    def __init__(self):
        self.req = []

    def take(self, n):
        self.req.append(n)
        return len(self.req) - 1

    def build(self, dtype, device):
        offs, tot = [], 0
        for n in self.req:
            offs.append(tot)
            tot += (n + 3) // 4 * 4
        self.buf = torch.zeros(max(tot, 4), dtype=dtype, device=device)
        self.views = [self.buf[o:o + n] for o, n in zip(offs, self.req)]
        return self


class _BNState:
    def __init__(self, c, f32: _Arena, f64: _Arena):
        self.c = c
        self.i32 = [f32.take(c) for _ in range(7)]      # scale shift mean invstd coefA coefB coefC
        self.i64 = [f64.take(2 * c * ops.STAT_STRIPES) for _ in range(2)]  # forward stats, backward stats (striped)

    def bind(self, f32: _Arena, f64: _Arena):
        self.scale, self.shift, self.mean, self.invstd, self.ca, self.cb, self.cc = (f32.views[i] for i in self.i32)
        self.stats, self.bstats = (f64.views[i] for i in self.i64)
        self.eval_key = None            # (data_ptr, version) of the tensors the cached eval-mode scale / shift were folded from


class _Conv:
    """Packed-weight cache of one nn.Conv2d (re-packed only when the parameter version changes)."""

    def __init__(self, module, specs, wepoch=None):
        self.m = module
        self.specs = specs          # name -> dict(mode, ci_begin, ci_count, n_perm, w3x3_from_1x1)
        self.packed = {}
        self.version = {}
        self.wepoch = wepoch if wepoch is not None else [0]     # Engine._wepoch: bumped when weights changed behind torch's back

    def ver(self):
        return (self.wepoch[0], self.m.weight._version)

    def weight_for(self, spec):
        w = self.m.weight
        if spec.get("center"):      # 1x1 weight consumed through the 3x3 im2col'ed input: centre tap only
            w3 = torch.zeros(w.shape[0], w.shape[1], 3, 3, device=w.device, dtype=w.dtype)
            w3[:, :, 1, 1] = w.detach()[:, :, 0, 0]
            return w3
        return w.detach()

    def item(self, name, dtype):
        """pssr_pack_item of an already packed (name, dtype) entry, for the batched re-pack."""
        spec, pw, w = self.specs[name], self.packed[(name, dtype)], self.m.weight
        center = bool(spec.get("center"))
        it = L.PackItem()
        it.w, it.packed = w.data_ptr(), pw.data.data_ptr()
        perm = spec.get("n_perm")
        it.n_perm = perm.data_ptr() if perm is not None else None
        it.cout, it.cin, it.ks = w.shape[0], w.shape[1], 3 if center else w.shape[2]
        it.ci_begin = spec.get("ci_begin", 0)
        it.ci_count = spec.get("ci_count") if spec.get("ci_count") is not None else w.shape[1] - it.ci_begin
        it.mode, it.k_pad, it.n_pad, it.dtype, it.center = spec["mode"], pw.k_pad, pw.n_pad, dtype, int(center)
        return it

    def get(self, name, dtype):
        spec = self.specs[name]
        key = (name, dtype)
        ver = self.ver()
        if self.version.get(key) != ver or key not in self.packed:
            self.packed[key] = ops.pack_conv_weight(self.weight_for(spec).contiguous(), dtype, mode=spec["mode"],
                                                    ci_begin=spec.get("ci_begin", 0), ci_count=spec.get("ci_count"),
                                                    n_perm=spec.get("n_perm"), out=self.packed.get(key))
            self.version[key] = ver
        return self.packed[key]


class Engine:
    _warned_bf16_infer = False       # the infer_dtype = bfloat16 notice is printed once per process (storage_dtype)

    def __init__(self, model):
        self.model = model
        self.plans = {}
        self.saved = None
        self._convs = {}
        self.reducer = None          # pssr2_amd.distributed.GradReducer when data-parallel
        self._wepoch = [0]           # part of every packed-weight / folded-BatchNorm cache key (mark_weights_changed)
        # sync_bn (model.sync_bn = True, data-parallel runs only): BatchNorm statistics over the GLOBAL batch -- the striped
        # [sum, sum of squares] (forward) and [sum g, sum g*xhat] (backward) buffers of every BatchNorm are SUM-all-reduced before
        # they are finalised, which is the reference's single-process semantics on N GPUs (off by default like DDP)
        self._flat_grad = None
        # weight-gradient launches (wgrad + partial-slab reduction) on a second HIP stream: nothing on the backward's
        # dependent chain waits for them, so they fill the chip while the chain runs its tiny BatchNorm-coefficient
        # kernels and kernel tails (PSSR_WGRAD_STREAM=0 keeps everything on the launch stream)
        import os
        self.side_wgrad = os.environ.get("PSSR_WGRAD_STREAM", "1") != "0"
        self._side = None
        self._side_on = False
        self._pending = {}

    # ------------------------------------------------------------------ hipGraph replay support (pssr2_amd/fastpath.py)
    def mark_weights_changed(self):
        """Parameters or BatchNorm buffers were written without torch noticing (a replayed hipGraph that contains the optimizer
        step): every packed-weight copy and every folded eval-mode BatchNorm affine is stale from now on."""
        self._wepoch[0] += 1

    def publish_grads(self):
        """Point ``.grad`` of every parameter that has none at its slot of the flat gradient buffer (what the end of an eager
        backward does): a replayed backward graph writes the buffer but runs no Python."""
        if self._flat_grad is None:
            return
        for prm, view in zip(self.model.parameters(), self._gviews):
            if prm.requires_grad and prm.grad is None:
                prm.grad = view

    # ------------------------------------------------------------------ flat gradient buffer
    def _grad_layout(self, device):
        """All gradients live in one flat f32 buffer in parameter order (4-element aligned slots)."""
        params = list(self.model.parameters())
        key = (str(device), tuple(p.numel() for p in params))
        if self._flat_grad is not None and self._flat_key == key:
            return
        offs, tot = [], 0
        for p in params:
            offs.append(tot)
            tot += (p.numel() + 3) // 4 * 4
        self._flat_grad = torch.zeros(tot, dtype=torch.float32, device=device)
        self._flat_key = key
        self._goffs = offs
        self._gsizes = [p.numel() for p in params]
        self._gindex = {id(p): i for i, p in enumerate(params)}
        self._gviews = [self._flat_grad[o:o + p.numel()].view(p.shape) for p, o in zip(params, offs)]

    def attach_reducer(self, bucket_bytes=64 << 20, group=None):
        from .distributed import GradReducer
        dev = next(self.model.parameters()).device
        self._grad_layout(dev)
        self.reducer = GradReducer(self._flat_grad, self._goffs, self._gsizes, bucket_bytes, group)
        return self.reducer

    def _begin_backward(self, device):
        """Zero the flat gradient buffer.  A .grad still aliasing it (no zero_grad since the last backward, or
        zero_grad(set_to_none=False)) is detached into its own storage first so that accumulation stays correct."""
        self._grad_layout(device)
        for prm, view in zip(self.model.parameters(), self._gviews):
            if prm.grad is not None and prm.grad.data_ptr() == view.data_ptr():
                prm.grad = prm.grad.clone()
        # the ~240 MB memset of the flat buffer (32 us at the head of every backward pass) is only needed where a slot is ACCUMULATED into:
        # a plain ResUNet overwrites every weight-gradient slot (one unpack per weight; Reconstruction.pre's two sources write disjoint
        # channel ranges), BatchNorm / bias gradients are stored or copied, and the slots nothing writes -- biases of convolutions in front of
        # a batch-statistics BatchNorm, whose gradient is exactly zero -- keep the zeros of the allocation (_overwrite_grads)
        if not getattr(self, "_overwrite_grads", False) and "gzero" not in _ABL:
            self._flat_grad.zero_()
        if self.reducer is not None:
            self.reducer.begin()
        self._side_begin(device)
        self._flush_fwd()
        if getattr(self, "_fwd_side", False):
            # the forward pass left bn_relu_apply launches on the second stream (finished long ago): whoever reads them comes after
            torch.cuda.current_stream().wait_stream(self._side)
            self._fwd_side = False

    def _finish_backward(self, grads):
        """Publish the gradients.  The engine owns the .grad of its parameters: a parameter without a gradient gets the
        view of the flat buffer itself (no AccumulateGrad copy, and FusedAdamW / the all-reduce see one flat tensor); an
        existing gradient is accumulated into, as autograd would."""
        self._flush_folds()
        self._flush_moves()
        if self.reducer is not None:
            self.reducer.finish()
        self._side_join()
        self.__dict__.get("_keepalive", []).clear()         # temporaries the second stream was reading (RDEngine._bias_grad_job)
        if getattr(self.model, "autograd_grads", False):
            # model.autograd_grads = True: hand the gradients to autograd like any torch.autograd.Function does (copies out of the
            # flat buffer, which the next backward pass overwrites) and leave .grad to it -- what torch.autograd.grad(loss, params),
            # post-accumulate-grad hooks and third-party training loops built on them need.  Costs one copy of every gradient per
            # step and loses the flat-buffer fast path of FusedAdamW; off by default.
            return {id(prm): view.clone() for prm, view in zip(self.model.parameters(), self._gviews) if prm.requires_grad}
        with torch.no_grad():
            for prm, view in zip(self.model.parameters(), self._gviews):
                if not prm.requires_grad:
                    continue
                if prm.grad is None:
                    prm.grad = view
                else:
                    prm.grad.add_(view)
        return {}

    def grad_split_offset(self):
        """First element of the flat gradient buffer that is final when ``backward(split_cb=...)`` calls back: the deepest
        encoder block's first parameter (parameters are laid out in module order: norm, encoder.0.., decoder.., reconstruction)."""
        first = next(self.model.encoder[self.L - 1].parameters())
        return self._goffs[self._gindex[id(first)]]

    def _gbuf(self, param):
        """Zeroed gradient slot of a parameter (a view of the flat buffer)."""
        return self._gviews[self._gindex[id(param)]]

    # ------------------------------------------------------------------ second stream for the weight gradients
    def _side_begin(self, device):
        self._side_on = bool(self.side_wgrad) and self.reducer is None and device.type == "cuda"
        self._pending = {}
        self._deferred = None
        # queues of a pass that did not reach _finish_backward (an exception mid-backward): their entries must not ride along with this
        # pass's launches (two writers of one gradient slot in one batched launch; stale temporaries pinned)
        for q in ("_folds", "_moves", "_keepalive"):
            self.__dict__.get(q, []).clear()
        if self._side_on and self._side is None:
            self._side = torch.cuda.Stream(device)

    def _on_side(self, reads, fn):
        """Run fn() on the side stream after everything issued so far on the launch stream; `reads` are the launch-stream
        buffers it reads (their next writer waits for it, see _before_write).

        The dependency (an event on the launch stream) is taken NOW, the side-stream launch itself is deferred until the launch
        stream has issued its own next kernels (_flush_side): in the captured hipGraph the node with two successors then has its
        launch-stream successor created first.  The graph executor keeps the first successor on the node's own hardware queue and
        moves the other one; with the weight gradient created first the dependent chain hopped queues at every fork, and every hop
        is a cross-queue signal -- 60 idle gaps of ~15 us per c2 step in the kernel trace, each right behind a bn_bwd_apply."""
        self._flush_side()
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        self._deferred = (ev, reads, fn)

    def _flush_side(self):
        d = getattr(self, "_deferred", None)
        if d is None:
            return
        self._deferred = None
        ev, reads, fn = d
        with torch.cuda.stream(self._side):
            self._side.wait_event(ev)
            fn()
            done = torch.cuda.Event()
            done.record(self._side)
        for b in reads:
            self._pending[b.data_ptr()] = done

    def _before_write(self, *bufs):
        """The launch stream is about to overwrite these buffers: wait for side-stream readers still using them."""
        if not self._side_on:
            return
        self._flush_side()
        main = torch.cuda.current_stream()
        for b in bufs:
            ev = self._pending.pop(b.data_ptr(), None)
            if ev is not None:
                main.wait_event(ev)

    def reset_backward_state(self):
        """Forget a backward pass that did not run to its end (an aborted graph capture): no saved forward, no deferred side-stream
        launch, no events of buffers the side stream was still reading."""
        self.saved = None
        self._deferred = None
        self._pending = {}
        self._side_on = False
        self.__dict__.get("_folds", []).clear()
        self.__dict__.get("_moves", []).clear()
        self.__dict__.get("_keepalive", []).clear()

    def _side_join(self):
        if self._side_on:
            self._flush_side()
            ev = torch.cuda.Event()
            ev.record(self._side)
            torch.cuda.current_stream().wait_event(ev)
            self._pending = {}

    def _fold(self, s64, dst):
        """Queue ``ops.f64_to_f32(s64, dst)`` for the next _ready / the end of the pass (one launch per 16 folds).  Only for sums that
        nothing on the device reads before then (parameter gradients) and that live in storage nothing reuses meanwhile."""
        self.__dict__.setdefault("_folds", []).append((s64, dst, False))

    def _flush_folds(self):
        q = self.__dict__.get("_folds")
        if q:
            ops.f64_to_f32_batch(q)
            q.clear()

    def _ready(self, grads, params):
        """Copy small side results into their slots and tell the reducer these parameters are final."""
        self._flush_folds()
        if self._side_on:
            self._flush_side()
        idx, moves = [], []
        for prm in params:
            i = self._gindex[id(prm)]
            g = grads.get(id(prm))
            if g is not None and g.data_ptr() != self._gviews[i].data_ptr():
                if _COPY_BATCH and g.dtype == torch.float32 and g.is_contiguous() and g.numel() == self._gviews[i].numel():
                    moves.append((self._gviews[i], g))                 # one launch for all of them (memcpy nodes cost ~10 us each)
                else:
                    self._gviews[i].copy_(g.view(prm.shape))
            grads[id(prm)] = self._gviews[i]
            idx.append(i)
        if moves:
            if self.reducer is None:
                # nothing reads these slots before the pass ends (or the split callback): one launch there instead of one per block
                # on the dependent chain
                self.__dict__.setdefault("_moves", []).extend(moves)
            else:
                ops.copy_f32_batch(moves)
        if self.reducer is not None:
            self.reducer.mark_ready(idx)

    def _flush_moves(self):
        q = self.__dict__.get("_moves")
        if q:
            ops.copy_f32_batch(q)
            q.clear()

    # ------------------------------------------------------------------ static structure
    def _structure(self, device):
        m = self.model
        if getattr(self, "_built_for", None) == device:
            return
        self.cin, self.cout = m.channels
        self.hidden = list(m.hidden)
        self.L = len(self.hidden)
        self.atrous = m.norm is None          # pssr/models/resunet.py:50: the atrous variant has no input BatchNorm
        self.r = m.reconstruction.scale
        self.blk, self.explicit_shuffle = _blocked_order(self.r)
        self.xc = ops.pad_to(9 * self.cin, 16)
        h0 = self.h0 = self.hidden[0]
        r2 = self.r * self.r
        # sub-pixel-major channel order of Reconstruction.pre: n' = sub*h0 + c  <-  n = c*r2 + sub (explicit shuffle: torch's own order)
        idx = torch.arange(r2 * h0)
        self.pre_perm = (idx if self.explicit_shuffle else (idx % h0) * r2 + idx // h0).to(torch.int32).to(device)
        self.pre_perm_long = self.pre_perm.long()
        self._convs = {}
        self._built_for = device

    def storage_dtype(self, train):
        """Storage type of a pass.  Training: ``model.compute_dtype``.  Inference (eval-mode forward): ``model.infer_dtype`` when set,
        else float16 for a bfloat16 model, else ``compute_dtype``.  bf16 keeps 8 significant bits of every stored activation and
        weight; over the ~50 layers of a ResUNet that adds up to 0.5-2e-3 dB of PSNR against the f32 path -- AT the 1e-3 dB parity
        criterion, not inside it.  fp16 storage (11 bits, the same MFMA rate, the same bytes) measures
        1-3e-4 dB; its narrower range is no issue for a forward pass through BatchNorm'ed activations (gradients are what need bf16's
        range or loss scaling).  A network whose activations overflow fp16 (max 65504) shows as inf / nan outputs: set
        ``model.infer_dtype = torch.bfloat16`` for it (tests/test_gpu_parity_trained.py)."""
        m = self.model
        if train:
            return m.compute_dtype
        inf = getattr(m, "infer_dtype", None)
        if inf is not None:
            if inf == torch.bfloat16 and not Engine._warned_bf16_infer:
                Engine._warned_bf16_infer = True
                import warnings
                warnings.warn("pssr2_amd: model.infer_dtype = torch.bfloat16 stores inference activations with 8 significant bits; measured "
                              "against the float32 path that is 0.5-2e-3 dB of PSNR per tile, i.e. AT the 1e-3 dB parity criterion rather than "
                              "inside it (tests/test_gpu_parity_trained.py).  The default (float16 storage, 1-3e-4 dB) stays inside; keep "
                              "bfloat16 only for networks whose activations overflow float16.", stacklevel=3)
            return inf
        return torch.float16 if m.compute_dtype == torch.bfloat16 else m.compute_dtype

    def _repack_all(self, code=None):
        """Re-pack every cached conv weight (of storage type ``code``) whose parameter changed -- an optimizer step changes all of them
        -- with ONE launch instead of one per (conv, form).  Entries that were never packed yet stay on the lazy path of _Conv.get;
        copies in another storage type (the fp16 copies of a bf16-trained model's validation passes) wait for a pass that uses them."""
        convs = [c for c in self._convs.values() if c.packed]
        stale = [(c, key) for c in convs for key in c.packed if (code is None or key[1] == code) and c.version.get(key) != c.ver()]
        if len(stale) < 8:
            return
        sig = tuple((id(c), key, c.m.weight.data_ptr(), c.packed[key].data.data_ptr()) for c, key in stale)
        cache = self.__dict__.setdefault("_pack_tables", {})
        if cache.get(code, (None, None))[0] != sig:
            import ctypes as C
            arr = (L.PackItem * len(stale))(*[c.item(*key) for c, key in stale])
            host = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8)
            if code in cache:
                # a captured graph (a training or validation step of fastpath.py) may hold the previous table BY ADDRESS: a pass that packs
                # new forms after that capture (the eval-mode forms of a validation pass, say) changes the set, and a freed table read by
                # the old graph's re-pack launch is a kernel writing through stale pointers.  Tables are a few KB: never handed back
                self.__dict__.setdefault("_pack_tables_kept", []).append(cache[code][1])
            cache[code] = (sig, host.to(stale[0][0].m.weight.device))
        L.check(L.lib().pssr_pack_conv_weight_batch(L.ptr(cache[code][1]), len(stale), L.stream_ptr()), "pssr_pack_conv_weight_batch")
        for c, key in stale:
            c.version[key] = c.ver()

    def _conv(self, module, **specs):
        c = self._convs.get(id(module))
        if c is None:
            c = self._convs[id(module)] = _Conv(module, specs, self._wepoch)
        return c

    def _pw_any(self, module, name, code, **spec):
        """Packed weight of any conv module under a packing spec (mode ...), cached per parameter version."""
        c = self._convs.get(id(module))
        if c is None:
            c = self._convs[id(module)] = _Conv(module, {}, self._wepoch)
        key = f"{name}:{spec.get('mode', 0)}"
        if key not in c.specs:
            c.specs[key] = dict(spec)
        return c.get(key, code)

    def _check_supported(self, dtype_code, h, w, train):
        kch = 8 if dtype_code == L.F32 else 16
        for i, hc in enumerate(self.hidden):
            if hc % kch:
                raise ValueError(f"hidden[{i}]={hc}: the MI355X path needs channel counts that are multiples of {kch} "
                                 f"for compute dtype {ops.TORCH_DTYPE[dtype_code]}")
            if i and hc % 16:
                raise ValueError(f"hidden[{i}]={hc} must be a multiple of 16 (pixel-shuffle slice alignment)")
        if h % (1 << (self.L - 1)) or w % (1 << (self.L - 1)):
            raise ValueError(f"input size {h}x{w} must be divisible by 2^{self.L - 1} (max-pool / pixel-shuffle symmetry)")
        if train and min(h, w) >> (self.L - 1) < 3:
            raise ValueError("training needs at least 3x3 pixels at the deepest level on the MI355X path")

    # ------------------------------------------------------------------ per-shape plan
    def _plan(self, n, h, w, dt, device):
        key = (n, h, w, dt, str(device))
        p = self.plans.get(key)
        if p is not None:
            return p
        code = ops.dtype_code(dt)
        Lv, hid = self.L, self.hidden
        p = type("Plan", (), {})()
        p.n, p.h, p.w, p.dt, p.code = n, h, w, dt, code
        p.dims = [(h >> i, w >> i) for i in range(Lv)]
        f32, f64 = _Arena(), _Arena()
        m = self.model

        def buf(hh, ww, c):
            return torch.zeros(n, hh, ww, ops.pad_to(c, 16), dtype=dt, device=device)

        p.bn_in = _BNState(self.cin, f32, f64)
        p.xcol = buf(h, w, self.xc)
        p.enc, p.dec = [], []
        nl = max(m.depth, 0) + 1
        # concat buffers: cat[l] = [shuffle(level l+1 output) | encoder l output]
        p.cat = [buf(*p.dims[l], hid[l + 1] // 4 + hid[l]) for l in range(Lv - 1)]
        p.pooled = [buf(*p.dims[l + 1], hid[l]) for l in range(Lv - 1)]
        from . import atrous as A
        from .models import ResBlockA
        for i in range(Lv):
            b = type("B", (), {})()
            b.level, b.c = i, hid[i]
            b.a = None
            if isinstance(m.encoder[i], ResBlockA):
                b.a = A.make_ablock_state(m.encoder[i], n, *p.dims[i], self.cin if i == 0 else hid[i - 1], dt, device)
                b.y, b.bn = [], []
            else:
                b.y = [buf(*p.dims[i], hid[i]) for _ in range(nl)]
                b.act = None
                b.bn = [_BNState(hid[i], f32, f64) for _ in range(nl)]
            b.out = None if i < Lv - 1 else buf(*p.dims[i], hid[i])       # encoder outputs live in cat[i]
            p.enc.append(b)
        for l in range(Lv - 1):
            b = type("B", (), {})()
            b.level, b.c = l, hid[l]
            b.a = None
            mod = m.decoder[Lv - 2 - l]
            if isinstance(mod, ResBlockA):
                b.a = A.make_ablock_state(mod, n, *p.dims[l], hid[l + 1] // 4 + hid[l], dt, device)
                b.y, b.bn = [], []
            else:
                b.y = [buf(*p.dims[l], hid[l]) for _ in range(nl)]
                b.act = None
                b.bn = [_BNState(hid[l], f32, f64) for _ in range(nl)]
            b.out = buf(*p.dims[l], hid[l])
            p.dec.append(b)                                              # p.dec[l] is the block at level l
        p.xin = buf(h, w, self.cin) if self.atrous else None              # plain "x / 128 - 1" for an atrous first block
        p.epool = p.rpool = None
        if getattr(m, "encoder_pool", None) is not None:
            p.epool = A.make_psp_state(m.encoder_pool, n, *p.dims[Lv - 1], dt, device)
            p.epool_out = buf(*p.dims[Lv - 1], hid[Lv - 1])
        if getattr(m, "reconstruction_pool", None) is not None:
            p.rpool = A.make_psp_state(m.reconstruction_pool, n, *p.dims[0], dt, device)
            p.rpool_out = buf(*p.dims[0], hid[0])
        r2 = self.r * self.r
        p.pre = torch.zeros(n, h, w, r2 * hid[0], dtype=dt, device=device)
        p.f32 = f32.build(torch.float32, device)
        p.f64 = f64.build(torch.float64, device)
        p.bn_in.bind(f32, f64)
        for b in p.enc + p.dec:
            for s in b.bn:
                s.bind(f32, f64)
        p.ones_pre = torch.ones(r2 * hid[0], dtype=torch.float32, device=device)
        p.zeros_pre = torch.zeros(r2 * hid[0], dtype=torch.float32, device=device)
        p.bwd = None
        self.plans[key] = p
        return p

    def _bwd_buffers(self, p, device):
        if p.bwd is not None:
            return p.bwd
        n, dt, hid, Lv = p.n, p.dt, self.hidden, self.L
        b = type("Bwd", (), {})()

        def buf(hh, ww, c):
            return torch.zeros(n, hh, ww, ops.pad_to(c, 16), dtype=dt, device=device)

        b.dz = [buf(*p.dims[l], hid[l]) for l in range(Lv)]
        b.dy = [buf(*p.dims[l], hid[l]) for l in range(Lv)]
        b.dy2 = [buf(*p.dims[l], hid[l]) for l in range(Lv)] if self.side_wgrad else b.dy
        b.dy3 = [buf(*p.dims[l], hid[l]) for l in range(Lv)] if self.side_wgrad else b.dy
        b.g = [buf(*p.dims[l], hid[l]) for l in range(Lv)]
        b.dout = [buf(*p.dims[l], hid[l]) for l in range(Lv)]            # gradient of a block output at level l
        b.dcat = [buf(*p.dims[l], hid[l + 1] // 4 + hid[l]) for l in range(Lv - 1)]
        b.dpooled = [buf(*p.dims[l + 1], hid[l]) for l in range(Lv - 1)]
        b.dxcol_a = buf(p.h, p.w, self.xc)
        b.dxcol_b = buf(p.h, p.w, self.xc)
        r = self.r
        b.g_hr = torch.zeros(n, p.h * r, p.w * r, 16, dtype=dt, device=device)
        b.dpre = torch.zeros(n, p.h, p.w, r * r * hid[0], dtype=dt, device=device)
        b.sum64 = torch.zeros(ops.STAT_STRIPES * max(16, r * r * hid[0]), dtype=torch.float64, device=device)
        b.drpool = buf(*p.dims[0], hid[0]) if getattr(p, "rpool", None) is not None else None
        b.depool = buf(*p.dims[Lv - 1], hid[Lv - 1]) if getattr(p, "epool", None) is not None else None
        p.bwd = b
        return b

    def _count_batches(self):
        """num_batches_tracked += 1 on every BatchNorm2d with ONE kernel: the counters are re-seated as views of one flat
        int64 tensor (values kept; state_dict / load_state_dict see ordinary buffers)."""
        bns = [mod for mod in self.model.modules() if isinstance(mod, torch.nn.BatchNorm2d) and mod.num_batches_tracked is not None]
        if not bns:
            return
        key = tuple(b.num_batches_tracked.data_ptr() for b in bns)
        if getattr(self, "_nbt_key", None) != key:
            flat = torch.stack([b.num_batches_tracked.detach().reshape(()) for b in bns]).contiguous()
            for i, b in enumerate(bns):
                b.num_batches_tracked.data = flat[i]
            self._nbt_flat = flat
            self._nbt_key = tuple(b.num_batches_tracked.data_ptr() for b in bns)
        self._nbt_flat.add_(1)

    # ------------------------------------------------------------------ BatchNorm statistics (optionally over all ranks)
    def _sync_world(self):
        if not getattr(self.model, "sync_bn", False):
            return 1
        from . import distributed as D
        return D.rank_world()[1]

    def bn_stats_count(self, stats, count):
        """Forward statistics of a BatchNorm about to be finalised: SUM over ranks under sync_bn.  Returns the sample count."""
        world = self._sync_world()
        if world > 1:
            torch.distributed.all_reduce(stats)
        return count * world

    def bn_coefs(self, bstats, count, gamma, mean, invstd, ca, cb, cc, dgamma, dbeta):
        """pssr_bn_bwd_coefs; under sync_bn the parameter gradients come from this rank's sums (the gradient all-reduce averages
        them like every other gradient) and the input-gradient coefficients from the sums over all ranks."""
        world = self._sync_world()
        if world > 1:
            ops.bn_bwd_coefs(bstats, count, gamma, mean, invstd, ca, cb, cc, dgamma, dbeta)
            torch.distributed.all_reduce(bstats)
            ops.bn_bwd_coefs(bstats, count * world, gamma, mean, invstd, ca, cb, cc, None, None)
        else:
            ops.bn_bwd_coefs(bstats, count, gamma, mean, invstd, ca, cb, cc, dgamma, dbeta)

    # ------------------------------------------------------------------ forward
    def _bn_forward(self, p, st, bn_module, count, train):
        if train:
            st.eval_key = None          # scale / shift now hold batch statistics, and the running statistics move
            count = self.bn_stats_count(st.stats, count)
            ops.bn_finalize(st.stats, count, bn_module.weight, bn_module.bias, BN_EPS, BN_MOMENTUM,
                            bn_module.running_mean, bn_module.running_var, st.scale, st.shift, st.mean, st.invstd)
        else:
            # eval-mode scale / shift are constants of the parameters: folded once, redone when a tensor is replaced or written
            # through torch (load_state_dict, an optimizer step) or after a training forward (37 five-microsecond launches
            # were 4.6 % of an inference pass)
            ts = (bn_module.weight, bn_module.bias, bn_module.running_mean, bn_module.running_var)
            key = (self._wepoch[0],) + tuple((t.data_ptr(), t._version) for t in ts)
            if getattr(st, "eval_key", None) != key:
                ops.bn_eval_affine(*ts, BN_EPS, st.scale, st.shift)
                st.eval_key = key

    def _materialise_ok(self, p, c):
        """The weight gradients of this plan read materialised activations (16-bit storage, power-of-two channel counts, the all-DMA
        kernel switched on: tunable WGRAD_DMA) instead of applying BatchNorm+ReLU in their loaders."""
        if p.code == L.F32 or c < 8 or (c & (c - 1)) or _NO_MATERIALISE:
            return False
        if getattr(self, "_wgrad_dma", None) is None:
            self._wgrad_dma = L.lib().pssr_get_option(b"WGRAD_DMA") > 0
        return self._wgrad_dma

    def _materialise(self, p, blk, k):
        """act[k] = relu(bn_k(y[k])) on the SECOND stream, under the forward pass (which leaves that stream idle): the weight gradient of
        conv k + 1 then takes both operands by LDS-DMA (conv_wgrad16d_kernel) -- 13 us of HBM-bound work per layer beside MFMA-bound
        convolutions buys a weight-gradient kernel without staging registers, prologue arithmetic or LDS commit."""
        if getattr(blk, "act", None) is None:
            blk.act = [torch.zeros_like(blk.y[0]) for _ in range(len(blk.y) - 1)]
        st = blk.bn[k]
        hh, ww = p.dims[blk.level]
        if self._side is None:
            self._side = torch.cuda.Stream(blk.y[0].device)
        # the dependency is taken now, the launch waits until the launch stream has issued its own next kernel (_flush_fwd): in a captured
        # graph the FIRST-created successor of a node stays on the node's hardware queue, and it has to be the next convolution, not this
        # side kernel (Engine._on_side: the same rule for the weight gradients)
        self._flush_fwd()
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        n_el, c, code = p.n * hh * ww, blk.c, p.code
        self._fwd_deferred = (ev, lambda: ops.bn_relu_apply(blk.y[k], st.scale, st.shift, blk.act[k], n_el, c, code))

    def _flush_fwd(self):
        d = getattr(self, "_fwd_deferred", None)
        if d is None:
            return
        self._fwd_deferred = None
        ev, fn = d
        with torch.cuda.stream(self._side):
            self._side.wait_event(ev)
            fn()
        self._fwd_side = True

    def _folded_last(self, module, st, conv, code, perm=None):
        """(packed weight, bias) of a block's last convolution with its eval-mode BatchNorm folded in: W' = W * scale[co],
        b' = conv.bias * scale + shift + respass.bias.  Cached per (parameter versions, BatchNorm affine).  ``perm`` (int64 [cout]): output
        channels in that order (FLAG_SHUF2: sub-pixel-major)."""
        cache = self.__dict__.setdefault("_fold_cache", {})
        rp = module.respass
        key = (self._wepoch[0], conv.weight._version, conv.bias._version, rp.bias._version, st.eval_key)
        ck = (id(conv), code, perm is not None)
        ent = cache.get(ck)
        if ent is None or ent[0] != key:
            with torch.no_grad():
                wf = conv.weight.detach() * st.scale.view(-1, 1, 1, 1)
                bf = torch.addcmul(st.shift, conv.bias.detach(), st.scale).add_(rp.bias.detach())
                if perm is not None:
                    wf, bf = wf[perm], bf[perm]
                wf = wf.contiguous()
            pw = ops.pack_conv_weight(wf, code, mode=0, out=ent[1] if ent is not None else None)
            if ent is not None:                 # in place: a captured eval graph (fastpath.EvalStepper) holds these two buffers by address
                bf = ent[2].copy_(bf)
            ent = cache[ck] = (key, pw, bf)
        return ent[1], ent[2]

    def _shuf_perm(self, c, device):
        """Sub-pixel-major order of a block output that goes through F.pixel_shuffle(x, 2): row s * c / 4 + q <- torch channel 4 q + s."""
        cache = self.__dict__.setdefault("_shuf_perms", {})
        if (c, device) not in cache:
            idx = torch.arange(c, device=device)
            perm = (idx % (c // 4)) * 4 + idx // (c // 4)
            cache[(c, device)] = (perm, perm.to(torch.int32))
        return cache[(c, device)]

    def _block_forward(self, p, blk, module, src, cin, first, dst, dst_coff, train, shuf=None):
        """``shuf``: the [n, 2h, 2w, .] concat buffer whose first c / 4 channels are F.pixel_shuffle(block output, 2).  Returns True when the
        block wrote them there itself (eval mode: FLAG_SHUF2 on its last convolution; ``dst`` is then left unwritten), else None."""
        if getattr(blk, "a", None) is not None:          # ResBlockA (pssr2_amd/atrous.py); a first block reads the plain input
            from . import atrous as A
            A.ablock_forward(self, blk.a, module, p.xin if first else src, 0, p.n, p.code, dst, dst_coff, train)
            return
        n = p.n
        hh, ww = p.dims[blk.level]
        count = float(n * hh * ww)
        nl = len(blk.y)
        if not train and p.code != L.F32 and _EVAL_AFFINE and blk.c % 8 == 0:
            # eval mode, 16-bit storage: the BatchNorm + ReLU behind convolution k is applied by convolution k itself on its f32
            # accumulators (FLAG_AFFINE | FLAG_RELU: the affine is a constant of the parameters), so y[k] holds the ACTIVATED map and
            # convolution k + 1 stages it without a prologue (the BatchNorm + ReLU prologue costs the 3x3 loops ~7 % of their time)
            for k in range(nl):
                conv, bn = module.conv[3 * k], module.conv[3 * k + 1]
                self._bn_forward(p, blk.bn[k], bn, count, False)
                if k == 0:
                    spec = (dict(fwd=dict(mode=2), dgrad=dict(mode=3)) if first else dict(fwd=dict(mode=0), dgrad=dict(mode=1)))
                    inp, icn = src, cin
                else:
                    spec = dict(fwd=dict(mode=0), dgrad=dict(mode=1))
                    inp, icn = blk.y[k - 1], blk.c
                if k < nl - 1:
                    pw = self._conv(conv, **spec).get("fwd", p.code)
                    ops.conv2d(inp, icn, pw, blk.y[k], blk.c, n=n, h=hh, w=ww, bias=conv.bias, flags=L.FLAG_RELU | L.FLAG_AFFINE,
                               aux_scale=blk.bn[k].scale, aux_shift=blk.bn[k].shift)
                    continue
                # the last convolution takes the residual 1x1 (module.respass on the block input) as its second source and ends in
                # the block's ReLU: out = relu(bn(conv(a)) + respass(src)) with the BatchNorm's scale folded into THIS convolution's
                # weight rows (exact in eval mode; no division by a scale that may be zero) -- the raw map y[-1] is never stored and
                # the separate tail launch (a pass over src, y[-1] and the output) is gone
                rp = module.respass
                if first:
                    pwr = self._conv(rp, fwd=dict(mode=2, center=True), dgrad=dict(mode=3, center=True)).get("fwd", p.code)
                else:
                    pwr = self._conv(rp, fwd=dict(mode=0), dgrad=dict(mode=1)).get("fwd", p.code)
                if spec["fwd"]["mode"] == 0:
                    if shuf is not None and _EVAL_SHUF and blk.c % 32 == 0 and not first:
                        # the consumer of this block is F.pixel_shuffle(., 2) into a concat buffer: the stores go there directly
                        perm_l, perm_i = self._shuf_perm(blk.c, inp.device)
                        pwf, bf = self._folded_last(module, blk.bn[k], conv, p.code, perm_l)
                        pwr = self._pw_any(rp, "fwd_shuf", p.code, mode=0, n_perm=perm_i)
                        ops.conv2d(inp, icn, pwf, shuf, blk.c, n=n, h=hh, w=ww, out_coff=0, bias=bf, x1=src, cin1=cin, w1=pwr,
                                   flags=L.FLAG_RELU | L.FLAG_SHUF2)
                        return True
                    pwf, bf = self._folded_last(module, blk.bn[k], conv, p.code)
                    ops.conv2d(inp, icn, pwf, dst, blk.c, n=n, h=hh, w=ww, out_coff=dst_coff, bias=bf, x1=src, cin1=cin, w1=pwr,
                               flags=L.FLAG_RELU)
                    return
                # (a block of ONE convolution on the network input: both sources are flat-K 1x1 forms -- keep the separate tail)
                pw = self._conv(conv, **spec).get("fwd", p.code)
                ops.conv2d(inp, icn, pw, blk.y[k], blk.c, n=n, h=hh, w=ww, bias=conv.bias)
            nl = 0          # (the loop below is the training / f32 form)
        for k in range(nl):
            conv = module.conv[3 * k]
            bn = module.conv[3 * k + 1]
            if k == 0:
                if first:
                    pw = self._conv(conv, fwd=dict(mode=2), dgrad=dict(mode=3)).get("fwd", p.code)
                else:
                    pw = self._conv(conv, fwd=dict(mode=0), dgrad=dict(mode=1)).get("fwd", p.code)
                ops.conv2d(src, cin, pw, blk.y[0], blk.c, n=n, h=hh, w=ww, bias=conv.bias,
                           flags=L.FLAG_STATS if train else 0, stats=blk.bn[0].stats if train else None)
            else:
                pw = self._conv(conv, fwd=dict(mode=0), dgrad=dict(mode=1)).get("fwd", p.code)
                prev = blk.bn[k - 1]
                ops.conv2d(blk.y[k - 1], blk.c, pw, blk.y[k], blk.c, n=n, h=hh, w=ww, bias=conv.bias,
                           pro_scale=prev.scale, pro_shift=prev.shift,
                           flags=L.FLAG_STATS if train else 0, stats=blk.bn[k].stats if train else None)
                self._flush_fwd()
            self._bn_forward(p, blk.bn[k], bn, count, train)
            if train and k < nl - 1 and getattr(self, "_will_backward", False) and self._materialise_ok(p, blk.c):
                self._materialise(p, blk, k)
        rp = module.respass
        if first:
            pw = self._conv(rp, fwd=dict(mode=2, center=True), dgrad=dict(mode=3, center=True)).get("fwd", p.code)
        else:
            pw = self._conv(rp, fwd=dict(mode=0), dgrad=dict(mode=1)).get("fwd", p.code)
        last = blk.bn[-1]
        ops.conv2d(src, cin, pw, dst, blk.c, n=n, h=hh, w=ww, out_coff=dst_coff, bias=rp.bias, epilogue=L.EPI_TAIL,
                   aux=blk.y[-1], aux_scale=last.scale, aux_shift=last.shift)

    def forward(self, x, train):
        m = self.model
        if not x.is_cuda:
            raise RuntimeError("pssr2_amd.ResUNet runs on an MI355X (HIP) device only; there is no CPU fallback")
        x = x.contiguous().float()
        self._structure(x.device)
        n, c, h, w = x.shape
        if c != self.cin:
            raise ValueError(f"expected {self.cin} input channels, got {c}")
        dt = self.storage_dtype(train)
        code = ops.dtype_code(dt)
        self._check_supported(code, h, w, train)
        p = self._plan(n, h, w, dt, x.device)
        self._repack_all(code)
        Lv, hid = self.L, self.hidden
        if train:
            p.f64.buf.zero_()
            self._count_batches()
        if self.atrous:
            p.bn_in.scale.fill_(1.0), p.bn_in.shift.zero_()          # no input BatchNorm: xcol is the im2col of x / 128 - 1 itself
            ops.input_plain(x, p.xin, code)
        else:
            if train:
                ops.channel_stats_nchw(x, p.bn_in.stats, 1 / 128, -1.0)
            self._bn_forward(p, p.bn_in, m.norm, float(n * h * w), train)
        ops.input_im2col(x, p.xcol, p.bn_in.scale, p.bn_in.shift, code)
        # encoder
        for i in range(Lv):
            blk = p.enc[i]
            src, cin = (p.xcol, self.xc) if i == 0 else (p.pooled[i - 1], ops.pad_to(hid[i - 1], 16))
            if i < Lv - 1:
                dst, off = p.cat[i], hid[i + 1] // 4
            else:
                dst, off = blk.out, 0
            # (the deepest block feeds the first pixel shuffle of the decoder -- unless PSP pooling sits in between)
            shuffled = self._block_forward(p, blk, m.encoder[i], src, cin, i == 0, dst, off, train,
                                           shuf=p.cat[Lv - 2] if (i == Lv - 1 and Lv > 1 and p.epool is None and not train) else None)
            if i < Lv - 1 and "pool" not in _ABL:
                ops.maxpool2(dst, p.pooled[i], n, *p.dims[i], hid[i], code, in_coff=off)
        deep = p.enc[Lv - 1].out
        if p.epool is not None:         # pssr/models/resunet.py:78-79
            from . import atrous as A
            A.psp_forward(self, p.epool, m.encoder_pool, deep, 0, n, code, p.epool_out, 0, train)
            deep = p.epool_out
        # decoder
        for l in range(Lv - 2, -1, -1):
            prev = deep if l == Lv - 2 else p.dec[l + 1].out
            if "shuf" not in _ABL and not shuffled:       # (shuffled: the block before stored into cat[l] itself)
                ops.pixel_shuffle(prev, p.cat[l], n, *p.dims[l + 1], hid[l + 1] // 4, 2, code)
            blk = p.dec[l]
            shuffled = self._block_forward(p, blk, m.decoder[Lv - 2 - l], p.cat[l], p.cat[l].shape[-1], False, blk.out, 0, train,
                                           shuf=p.cat[l - 1] if (l > 0 and not train) else None)
        feat = p.dec[0].out if Lv > 1 else deep
        if p.rpool is not None:         # pssr/models/resunet.py:87-88
            from . import atrous as A
            A.psp_forward(self, p.rpool, m.reconstruction_pool, feat, 0, n, code, p.rpool_out, 0, train)
            feat = p.rpool_out
        out = self._head_forward(p, feat, x, train)
        self._flush_fwd()
        self.saved = (p, x) if train else None
        return out

    # ------------------------------------------------------------------ reconstruction head (shared with RDEngine)
    def _head_forward(self, p, feat, x, train=True):
        """relu(conv3x3([feat | x0])) in sub-pixel-major channel order (== pixel-shuffled, blocked layout) -> conv3x3 ->
        x*128+128 (pssr/models/_blocks.py:15-18, pssr/models/resunet.py:90-95).  With 64 hidden channels, one output channel and 16-bit
        storage `pre`'s epilogue multiplies its activation with the final convolution's taps in registers (nine products per high-resolution
        pixel, stored in per-(tap, sub-pixel) planes at low resolution) and a second kernel sums the nine shifted products of every output
        pixel: Reconstruction.conv never reads the
        64-channel high-resolution tensor (1.07 GB at batch 32), and in eval mode -- nothing is kept for a backward pass -- that tensor
        is not written either."""
        rec = self.model.reconstruction
        n, h, w, code, h0, r = p.n, p.h, p.w, p.code, self.h0, self.r
        cpre = self._conv(rec.pre,
                          fwd0=dict(mode=0, ci_begin=0, ci_count=h0, n_perm=self.pre_perm),
                          fwd1=dict(mode=2, ci_begin=h0, ci_count=self.cin, n_perm=self.pre_perm),
                          dgrad0=dict(mode=1, ci_begin=0, ci_count=h0, n_perm=self.pre_perm),
                          dgrad1=dict(mode=3, ci_begin=h0, ci_count=self.cin, n_perm=self.pre_perm))
        p.pre_bias = rec.pre.bias.detach()[self.pre_perm_long].contiguous()
        out = torch.empty(n, self.cout, h * r, w * r, dtype=torch.float32, device=x.device)
        if _HEAD_FUSE and ops.head_q_supported(code, h0, self.cout, r, h, w):
            if getattr(p, "head_q", None) is None:
                p.head_q = torch.empty(9, r * r, n, h, w, dtype=torch.float32, device=x.device)     # [tap][sub-pixel] planes of tap products
            if train:       # the activation is kept for the backward pass (FLAG_HEADQ: stored AND multiplied with the head's taps)
                ops.conv2d(feat, h0, cpre.get("fwd0", code), p.pre, r * r * h0, n=n, h=h, w=w, bias=p.pre_bias, x1=p.xcol, cin1=self.xc,
                           w1=cpre.get("fwd1", code), flags=L.FLAG_RELU | L.FLAG_HEADQ, head_w=rec.conv.weight, head_q=p.head_q)
            else:
                ops.conv2d(feat, h0, cpre.get("fwd0", code), p.head_q, r * r * h0, n=n, h=h, w=w, bias=p.pre_bias, x1=p.xcol, cin1=self.xc,
                           w1=cpre.get("fwd1", code), epilogue=L.EPI_HEADQ, head_w=rec.conv.weight, head_q=p.head_q)
            ops.head_q_gather(p.head_q, rec.conv.bias, out, n, h, w, 128.0, 128.0)
            return out
        ops.conv2d(feat, h0, cpre.get("fwd0", code), p.pre, r * r * h0, n=n, h=h, w=w, bias=p.pre_bias,
                   x1=p.xcol, cin1=self.xc, w1=cpre.get("fwd1", code), flags=L.FLAG_RELU)
        pre_hr = self._pre_hr(p)
        if self.explicit_shuffle:
            ops.pixel_shuffle(p.pre, pre_hr, n, h, w, h0, r, code)
        if ops.head_conv_supported(code, h0, self.cout):
            ops.head_conv_fwd(pre_hr, self.blk, rec.conv.weight, rec.conv.bias, out, n, h * r, w * r, h0, self.cout, 128.0, 128.0, code)
        else:
            cfin = self._conv(rec.conv, fwd=dict(mode=0), dgrad=dict(mode=1))
            ops.conv2d(pre_hr, h0, cfin.get("fwd", code), out, self.cout, n=n, h=h * r, w=w * r, bias=rec.conv.bias,
                       epilogue=L.EPI_FINAL, in0_blk=self.blk, out_scale=128.0, out_shift=128.0)
        return out

    def _rows_head(self, code):
        """head_conv_bwd_rows (one pass: data + weight gradient + pre's bias sums per (sub-pixel, channel)) takes this configuration."""
        return ops.head_conv_supported(code, self.h0, self.cout) and not self.explicit_shuffle and self.blk <= 2 and self.h0 in (32, 64, 128)

    def _pre_hr(self, p):
        """Reconstruction.pre's activation as the high-resolution tensor the final convolution reads: a view in blocked order, or (factors
        that are not powers of two) its own buffer filled by an explicit pixel shuffle."""
        n, h, w, r, h0 = p.n, p.h, p.w, self.r, self.h0
        if not self.explicit_shuffle:
            return p.pre.view(n, h * r, w * r, h0)
        if getattr(p, "pre_hr", None) is None:
            p.pre_hr = torch.zeros(n, h * r, w * r, h0, dtype=p.pre.dtype, device=p.pre.device)
        return p.pre_hr

    def _head_backward(self, p, bw, grads, dout, feat, dfeat):
        """Backward of the head: parameter gradients of Reconstruction, d(feat) into `dfeat`, d(xcol) into bw.dxcol_b."""
        rec = self.model.reconstruction
        n, h, w, code, h0, r = p.n, p.h, p.w, p.code, self.h0, self.r
        dev = dout.device
        H, W = h * r, w * r
        dout = dout.contiguous().float()
        pre_hr = self._pre_hr(p)
        if self.explicit_shuffle:
            if getattr(bw, "dpre_hr", None) is None:
                bw.dpre_hr = torch.zeros_like(pre_hr)
            dpre_hr = bw.dpre_hr
        else:
            dpre_hr = bw.dpre.view(n, H, W, h0)
        if ops.head_conv_supported(code, h0, self.cout):
            # final conv straight from the f32 NCHW gradient ("x*128+128" folded into g_scale)
            gw = grads[id(rec.conv.weight)] = self._gbuf(rec.conv.weight)
            gb = torch.empty(2 * self.cout, dtype=torch.float32, device=dev)
            grads[id(rec.conv.bias)] = gb[:self.cout]
            # dgrad + wgrad + the bias sums of Reconstruction.pre in one pass over the HR activation
            gpb = torch.empty(r * r * h0, dtype=torch.float32, device=dev) if self._rows_head(code) else None
            if gpb is not None:
                # order-independent sums (ops.head_conv_bwd_rows): dW and the bias sums land in zeroed f64 statistic buffers first.  The
                # three statistic buffers of this stretch (output-gradient sums, dW, pre's bias sums) are slices of ONE allocation: one
                # memset and one batched fold instead of three each (these launches sit alone between the loss and the head's backward)
                S = ops.STAT_STRIPES
                sizes = (S * 2 * self.cout, S * gw.numel(), S * gpb.numel())
                if getattr(bw, "h64", None) is None or bw.h64.numel() != sum(sizes):
                    bw.h64 = torch.zeros(sum(sizes), dtype=torch.float64, device=dev)
                else:
                    bw.h64.zero_()
                s_out, dw64, s_pre = torch.split(bw.h64, sizes)
                ops.channel_stats_nchw(dout, s_out, 128.0, 0.0)
                ops.head_conv_bwd_rows(dout, 128.0, rec.conv.weight, pre_hr, dpre_hr, self.blk, dw64, s_pre, n, H, W, h0, self.cout, code)
                ops.f64_to_f32_batch([(s_out, gb, False), (dw64, gw.view(-1), False), (s_pre, gpb, False)])
            else:
                bw.sum64.zero_()
                ops.channel_stats_nchw(dout, bw.sum64, 128.0, 0.0)
                ops.f64_to_f32(bw.sum64[:ops.STAT_STRIPES * 2 * self.cout], gb)
                ops.head_conv_wgrad(dout, 128.0, pre_hr, self.blk, gw, n, H, W, h0, self.cout, code)
                ops.head_conv_dgrad(dout, 128.0, rec.conv.weight, pre_hr, dpre_hr, self.blk, n, H, W, h0, self.cout, code)
        else:
            gpb = None
            ops.nchw_to_nhwc(dout, bw.g_hr, 128.0, code)
            bw.sum64.zero_()
            ops.channel_sum_nhwc(bw.g_hr, n * H * W, 16, bw.sum64, code)
            gb = torch.empty(16, dtype=torch.float32, device=dev)
            ops.f64_to_f32(bw.sum64, gb)
            grads[id(rec.conv.bias)] = gb[:self.cout]
            self._wgrad(p, grads, rec.conv, bw.g_hr, 16, pre_hr, h0, 9, in_blk=self.blk, hh=H, ww=W)
            cfin = self._conv(rec.conv, fwd=dict(mode=0), dgrad=dict(mode=1))
            ops.conv2d(bw.g_hr, 16, cfin.get("dgrad", code), dpre_hr, h0, n=n, h=H, w=W, epilogue=L.EPI_DGRAD_MASK,
                       aux=pre_hr, aux_scale=p.ones_pre, aux_shift=p.zeros_pre, out_blk=self.blk, aux_blk=self.blk)
        # ---- Reconstruction.pre (two sources)
        cpre_n = r * r * h0
        if self.explicit_shuffle:
            ops.pixel_shuffle(bw.dpre, dpre_hr, n, h, w, h0, r, code, inverse=True)
        if gpb is None:
            bw.sum64.zero_()
            ops.channel_sum_nhwc(bw.dpre, n * h * w, cpre_n, bw.sum64, code)
            gpb = torch.empty(cpre_n, dtype=torch.float32, device=dev)
            ops.f64_to_f32(bw.sum64, gpb)
        gb_pre = torch.empty_like(gpb)
        gb_pre[self.pre_perm_long] = gpb
        grads[id(rec.pre.bias)] = gb_pre
        self._wgrad(p, grads, rec.pre, bw.dpre, cpre_n, feat, h0, 9, mode=0, ci_begin=0, ci_count=h0, n_perm=self.pre_perm, hh=h, ww=w)
        if not _ABLATE_XCOL and "xwgrad" not in _ABL:
            self._wgrad(p, grads, rec.pre, bw.dpre, cpre_n, p.xcol, self.xc, 1, mode=2, ci_begin=h0, ci_count=self.cin,
                        n_perm=self.pre_perm, hh=h, ww=w)
        self._ready(grads, list(rec.parameters()))
        cpre = self._convs[id(rec.pre)]
        ops.conv2d(bw.dpre, cpre_n, cpre.get("dgrad0", code), dfeat, h0, n=n, h=h, w=w)
        if not _ABLATE_XCOL and "xdgrad" not in _ABL and not self.atrous:
            # the 16-channel data gradient of the input source only feeds the input BatchNorm's parameter gradients at the very end of the
            # pass: on the second stream it is off the dependent chain (the two queues of the backward phase end within 0.1 ms of each
            # other, so this pays only together with something that lightens the weight-gradient queue: PSSR_XCOL_SIDE)
            pw1 = cpre.get("dgrad1", code)

            def dgrad_x():
                ops.conv2d(bw.dpre, cpre_n, pw1, bw.dxcol_b, self.xc, n=n, h=h, w=w)
            if self._side_on and _XCOL_SIDE:
                self._on_side([bw.dpre, bw.dxcol_b], dgrad_x)
            else:
                dgrad_x()

    # ------------------------------------------------------------------ backward
    def _wgrad(self, p, grads, conv_module, dy, cout, src, cin_pad, taps, *, mode=0, ci_begin=0, ci_count=None,
               n_perm=None, pro=None, dy_blk=0, in_blk=0, hh, ww, center=False, dy_view_c=None):
        code = p.code
        esz = 4 if code == L.F32 else 2
        co_eff = cout if (cout * esz) % 16 == 0 else ops.pad_to(cout, 16)
        w = conv_module.weight
        gname = id(w)

        def parts():
            return ops.conv2d_wgrad_parts(dy, co_eff, src, cin_pad, taps, n=p.n, h=hh, w=ww, dtype=code, dy_blk=dy_blk, in_blk=in_blk,
                                          pro_scale=pro.scale if pro else None, pro_shift=pro.shift if pro else None)
        if center:
            g3 = torch.zeros(w.shape[0], w.shape[1], 3, 3, dtype=torch.float32, device=w.device)
            ops.unpack_conv_wgrad(parts(), g3, mode=mode, ci_begin=ci_begin, ci_count=ci_count, n_perm=n_perm, k_pad=cin_pad)
            grads[gname] = g3[:, :, 1:2, 1:2].contiguous()
            return
        if gname not in grads:
            grads[gname] = self._gbuf(w)          # zeroed at the start of backward
        slot = grads[gname]

        def run():
            # the slot was zeroed with the whole flat buffer at the start of backward: accumulate (no separate zero pass)
            pr = parts()
            if "unpack" not in _ABL:
                ops.unpack_conv_wgrad(pr, slot, mode=mode, ci_begin=ci_begin, ci_count=ci_count, n_perm=n_perm, k_pad=cin_pad,
                                      accumulate=not getattr(self, "_overwrite_grads", False))
        if self._side_on:
            self._on_side([dy], run)
        else:
            run()

    def _fused_dout(self, p, blk, kind):
        """May relu_bwd_stats form the gradient of this block's output in its loader ('pool': skip + max-pool backward, 'unshuffle':
        inverse pixel shuffle) instead of reading a tensor another launch wrote?  Plain ResBlocks in 16-bit storage with power-of-two
        widths (PSSR_FUSE_DOUT=0: the separate launches, for A/B runs and the bit-identity test)."""
        if getattr(blk, "a", None) is not None or not _FUSE_DOUT:
            return False
        return ops.relu_bwd_stats_fused_ok(p.code, blk.c, *p.dims[blk.level], unshuffle=kind == "unshuffle")

    def _block_backward(self, p, bw, grads, blk, module, src, cin, first, out_buf, out_coff, dout, dsrc, dsrc_c, dout_from=None):
        """dout: gradient of the block output (buffer at this level).  Writes the gradient of `src` into dsrc.
        dout_from (16-bit storage): ("pool", dpool, dskip, dskip_coff) or ("unshuffle", dhi) -- `dout` was NOT materialised, relu_bwd_stats
        forms it from these in its loader (_fused_dout)."""
        if getattr(blk, "a", None) is not None:
            from . import atrous as A
            A.ablock_backward(self, blk.a, module, grads, p.xin if first else src, 0, p.n, p.code, out_buf, out_coff, dout, 0, dsrc, not first)
            return
        n, code = p.n, p.code
        lvl = blk.level
        hh, ww = p.dims[lvl]
        npix = n * hh * ww
        count = float(npix)
        nl = len(blk.y)
        dz, g = bw.dz[lvl], bw.g[lvl]
        # a ring of three: side-stream wgrads may still read the two older ones (with two buffers the launch stream waited for the
        # weight gradient of the layer before at every layer: ~9 us of cross-queue signalling each time in the kernel trace)
        ring = [bw.dy[lvl], bw.dy2[lvl], bw.dy3[lvl]] if self._side_on else [bw.dy[lvl]]
        ri = 0
        dy = ring[0]
        last = blk.bn[-1]
        bn_last = module.conv[3 * (nl - 1) + 1]
        self._before_write(dz)
        if "relustats" in _ABL:
            pass
        elif dout_from is None:
            ops.relu_bwd_stats(dout, out_buf, blk.y[-1], last.mean, last.invstd, dz, last.bstats, npix, blk.c, code, out_coff=out_coff)
        elif dout_from[0] == "pool":
            ops.relu_bwd_stats_pool(dout_from[1], dout_from[2], dout_from[3], out_buf, out_coff, blk.y[-1], last.mean, last.invstd, dz, last.bstats,
                                    n, hh, ww, blk.c, code)
        else:
            ops.relu_bwd_stats_unshuffle(dout_from[1], out_buf, out_coff, blk.y[-1], last.mean, last.invstd, dz, last.bstats, n, hh, ww, blk.c, code)
        dgam, dbet = self._gbuf(bn_last.weight), self._gbuf(bn_last.bias)
        self.bn_coefs(last.bstats, count, bn_last.weight, last.mean, last.invstd, last.ca, last.cb, last.cc, dgam, dbet)
        grads[id(bn_last.weight)], grads[id(bn_last.bias)] = dgam, dbet
        grads[id(module.respass.bias)] = dbet               # d(respass bias) = sum dz = dbeta of the last BN (copied by _ready)
        self._before_write(dy)
        if "apply" not in _ABL:
            ops.bn_bwd_apply(dz, blk.y[-1], last.ca, last.cb, last.cc, dy, npix, blk.c, code)
        for k in range(nl - 1, 0, -1):
            conv = module.conv[3 * k]
            prev, bn_prev = blk.bn[k - 1], module.conv[3 * (k - 1) + 1]
            # conv.bias sits in front of a batch-statistics BN: its gradient is exactly zero (slot stays zeroed)
            if getattr(blk, "act", None) is not None:
                self._wgrad(p, grads, conv, dy, blk.c, blk.act[k - 1], blk.c, 9, hh=hh, ww=ww)      # materialised in the forward pass
            else:
                self._wgrad(p, grads, conv, dy, blk.c, blk.y[k - 1], blk.c, 9, pro=prev, hh=hh, ww=ww)
            pwd = self._conv(conv, fwd=dict(mode=0), dgrad=dict(mode=1)).get("dgrad", code)
            ops.conv2d(dy, blk.c, pwd, g, blk.c, n=n, h=hh, w=ww, epilogue=L.EPI_DGRAD_MASK, flags=L.FLAG_STATS,
                       aux=blk.y[k - 1], aux_scale=prev.scale, aux_shift=prev.shift, aux_mean=prev.mean, aux_invstd=prev.invstd,
                       stats=prev.bstats)
            dgam, dbet = self._gbuf(bn_prev.weight), self._gbuf(bn_prev.bias)
            self.bn_coefs(prev.bstats, count, bn_prev.weight, prev.mean, prev.invstd, prev.ca, prev.cb, prev.cc, dgam, dbet)
            grads[id(bn_prev.weight)], grads[id(bn_prev.bias)] = dgam, dbet
            ri = (ri + 1) % len(ring)
            dy_nxt = ring[ri]
            self._before_write(dy_nxt)
            if "apply" not in _ABL:
                ops.bn_bwd_apply(g, blk.y[k - 1], prev.ca, prev.cb, prev.cc, dy_nxt, npix, blk.c, code)
            dy = dy_nxt
        conv0, rp = module.conv[0], module.respass
        if first:
            self._wgrad(p, grads, conv0, dy, blk.c, src, cin, 1, mode=2, hh=hh, ww=ww)
            self._wgrad(p, grads, rp, dz, blk.c, src, cin, 1, mode=2, hh=hh, ww=ww, center=True)
            c0 = self._conv(conv0, fwd=dict(mode=2), dgrad=dict(mode=3)).get("dgrad", code)
            c1 = self._conv(rp, fwd=dict(mode=2, center=True), dgrad=dict(mode=3, center=True)).get("dgrad", code)
        else:
            self._wgrad(p, grads, conv0, dy, blk.c, src, cin, 9, hh=hh, ww=ww)
            self._wgrad(p, grads, rp, dz, blk.c, src, cin, 1, hh=hh, ww=ww)
            c0 = self._conv(conv0, fwd=dict(mode=0), dgrad=dict(mode=1)).get("dgrad", code)
            c1 = self._conv(rp, fwd=dict(mode=0), dgrad=dict(mode=1)).get("dgrad", code)
        ops.conv2d(dy, blk.c, c0, dsrc, dsrc_c, n=n, h=hh, w=ww, x1=dz, cin1=blk.c, w1=c1)
        self._ready(grads, list(module.parameters()))

    def backward(self, dout, split_cb=None):
        """``split_cb`` (optional) is called once, when the gradients of Reconstruction, the decoder and the deepest encoder
        block are final in the flat buffer (everything from ``grad_split_offset()`` on: ~85 % of a default ResUNet's bytes) and
        the side stream is joined: a data-parallel driver launches their all-reduce there, under the rest of the backward."""
        if self.saved is None:
            raise RuntimeError("backward called without a training-mode forward (or called twice)")
        p, x = self.saved
        self.saved = None
        m = self.model
        dev = x.device
        bw = self._bwd_buffers(p, dev)
        n, h, w, code = p.n, p.h, p.w, p.code
        Lv, hid, r = self.L, self.hidden, self.r
        h0 = hid[0]
        grads = {}
        plain = all(getattr(b, "a", None) is None for b in p.enc + p.dec) and p.epool is None and p.rpool is None
        rows_head = self._rows_head(code)
        self._overwrite_grads = plain and rows_head and not self.atrous and _OVERWRITE_GRADS
        self._begin_backward(dev)
        from . import atrous as A
        deep = p.epool_out if p.epool is not None else p.enc[Lv - 1].out
        feat0 = p.dec[0].out if Lv > 1 else deep
        feat = p.rpool_out if p.rpool is not None else feat0
        if p.rpool is not None:
            self._head_backward(p, bw, grads, dout, feat, bw.drpool)
            A.psp_backward(self, p.rpool, m.reconstruction_pool, grads, feat0, 0, n, code, p.rpool_out, 0, bw.drpool, 0, bw.dout[0], 0)
        else:
            self._head_backward(p, bw, grads, dout, feat, bw.dout[0])
        # ---- decoder, bottom-up in the data-flow sense (level 0 first)
        unshuf = None               # set when the next block up reads its output gradient straight out of dcat (no inverse-shuffle launch)
        for l in range(0, Lv - 1):
            blk = p.dec[l]
            self._block_backward(p, bw, grads, blk, m.decoder[Lv - 2 - l], p.cat[l], p.cat[l].shape[-1], False,
                                 blk.out, 0, bw.dout[l], bw.dcat[l], hid[l + 1] // 4 + hid[l], dout_from=unshuf)
            # split dcat: [0, h_{l+1}/4) -> un-shuffle to the producer at level l+1
            nxt = p.dec[l + 1] if l + 1 < Lv - 1 else p.enc[Lv - 1]
            if (l + 1 < Lv - 1 or p.epool is None) and self._fused_dout(p, nxt, "unshuffle"):
                unshuf = ("unshuffle", bw.dcat[l])
            else:
                unshuf = None
                if "unshuf" not in _ABL:
                    ops.pixel_shuffle(bw.dout[l + 1], bw.dcat[l], n, *p.dims[l + 1], hid[l + 1] // 4, 2, code, inverse=True)
        if p.epool is not None:         # bw.dout[Lv-1] is the gradient of the pooled map: back through the PSP block
            A.psp_backward(self, p.epool, m.encoder_pool, grads, p.enc[Lv - 1].out, 0, n, code, p.epool_out, 0, bw.dout[Lv - 1], 0, bw.depool, 0)
            bw.dout[Lv - 1], bw.depool = bw.depool, bw.dout[Lv - 1]
        # ---- encoder, deepest first
        for i in range(Lv - 1, -1, -1):
            blk = p.enc[i]
            dfrom = None
            if i < Lv - 1:
                off = hid[i + 1] // 4
                # block output feeds the pool (dpooled) and the skip (dcat slice)
                if self._fused_dout(p, blk, "pool"):
                    dfrom = ("pool", bw.dpooled[i], bw.dcat[i], off)
                elif "poolbwd" not in _ABL:
                    ops.maxpool2_bwd(p.cat[i], bw.dpooled[i], bw.dcat[i], bw.dout[i], n, *p.dims[i], hid[i], code,
                                     act_coff=off, dskip_coff=off)
                out_buf, out_off = p.cat[i], off
            else:
                out_buf, out_off = blk.out, 0
                dfrom = unshuf if Lv > 1 else None
            if i == 0:
                src, cin, dsrc, dsrc_c = p.xcol, self.xc, bw.dxcol_a, self.xc
            else:
                src, cin, dsrc, dsrc_c = p.pooled[i - 1], ops.pad_to(hid[i - 1], 16), bw.dpooled[i - 1], hid[i - 1]
            self._block_backward(p, bw, grads, blk, m.encoder[i], src, cin, i == 0, out_buf, out_off, bw.dout[i], dsrc, dsrc_c, dout_from=dfrom)
            if split_cb is not None and i == Lv - 1 and Lv > 1:
                self._flush_folds()
                self._flush_moves()
                self._side_join()
                split_cb()
        if self.atrous:                 # no input BatchNorm, and the network input needs no gradient
            return self._finish_backward(grads)
        # ---- input BatchNorm parameters
        st = p.bn_in
        st.bstats.zero_()
        self._before_write(bw.dxcol_b)          # (its producer may have run on the second stream)
        ops.input_norm_bwd(bw.dxcol_a, bw.dxcol_b, x, st.mean, st.invstd, st.bstats, code)
        dgam, dbet = self._gbuf(m.norm.weight), self._gbuf(m.norm.bias)
        self.bn_coefs(st.bstats, float(n * h * w), m.norm.weight, st.mean, st.invstd, st.ca, st.cb, st.cc, dgam, dbet)
        grads[id(m.norm.weight)], grads[id(m.norm.bias)] = dgam, dbet
        self._ready(grads, list(m.norm.parameters()))
        return self._finish_backward(grads)
