"""Execution engine of the MI355X RDResUNet: RDNet encoder + ResUNet decoder + head as an explicit sequence of
libpssr_mi355.so launches over pre-allocated NHWC buffers (forward and backward).

The reference runs ``RDResUNet.forward`` (pssr/models/rdresunet.py:104-130) and ``RDNet.forward``
(pssr/models/_rdnet.py:95-104) op by op through torch autograd.  Here:

  * every dense stage owns ONE buffer [N, H, W, C_out]; a block writes its new feature at its channel offset, so the
    ``torch.cat`` of DenseStage.forward / DenseBlock.forward (_rdnet.py:132-138,169-170) never happens.  A stage whose output
    is a decoder skip lives directly inside the decoder's concat buffer ([pixel-shuffled previous decoder output | skip]);
  * a block is  dw7x7 -> LayerNorm2d -> 1x1 conv (MFMA implicit GEMM) -> [GELU fused into the next conv's loader] ->
    1x1 conv -> [ESE gate] * layer-scale (one pointwise kernel writing into the stage buffer);
  * the stride-2 transition conv is a 1x1 conv over the space-to-depth layout its LayerNorm writes;
  * the decoder blocks and the reconstruction head are the ResUNet kernels (engine.Engine);
  * backward walks the same structure in reverse; the gradient of a stage buffer has the same layout, and the depthwise
    input-gradient kernel accumulates into it (a feature is read by every later block of its stage).
"""
from __future__ import annotations

import torch

from . import _lib as L
from . import ops
from .engine import Engine, _Arena, _BNState, _Conv, _blocked_order

LN_EPS = 1e-6


def _obj(**kw):
    o = type("O", (), {})()
    o.__dict__.update(kw)
    return o



class _ZeroArena:
    """Zero-initialised scratch for one backward pass from ONE fill: the statistic / bias-sum buffers of every block are
    consecutive slices of one tensor (the per-use ``zero_()`` launches were ~100 five-microsecond kernels per step).  The first
    pass measures the need and serves its requests with individual fills."""

    def __init__(self, dtype):
        self.dtype, self.buf, self.need, self.pos, self.asked = dtype, None, 0, 0, 0

    def begin(self, device):
        if self.need and (self.buf is None or self.buf.numel() < self.need):
            self.buf = torch.zeros(self.need, dtype=self.dtype, device=device)
        elif self.buf is not None:
            self.buf.zero_()
        self.pos = self.asked = 0

    def owns(self, t):
        """t is a slice of the arena (unique for the rest of the pass), not a reused fallback buffer."""
        b = self.buf
        return b is not None and b.data_ptr() <= t.data_ptr() < b.data_ptr() + b.numel() * b.element_size()

    def take(self, n, fallback):
        """n zeroed elements: a slice of the arena, or (first pass / grown need) ``fallback`` zeroed in place."""
        n_al = (n + 3) // 4 * 4
        self.asked += n_al
        self.need = max(self.need, self.asked)
        if self.buf is not None and self.pos + n_al <= self.buf.numel():
            out = self.buf[self.pos:self.pos + n]
            self.pos += n_al
            return out
        fb = fallback[:n]
        fb.zero_()
        return fb


class RDEngine(Engine):
    _z64 = None
    _z32 = None

    # ------------------------------------------------------------------ static structure
    def _structure(self, device):
        m = self.model
        if getattr(self, "_built_for", None) == device:
            return
        enc = m.encoder
        self.cin, self.cout = m.channels
        self.hidden = list(m.hidden)
        self.atrous = m.norm is None          # pssr/models/rdresunet.py:80
        self.r = m.reconstruction.scale
        self.blk, self.explicit_shuffle = _blocked_order(self.r)
        self.ps = enc.patch_size
        self.xc = ops.pad_to(9 * self.cin, 16)
        self.pc = ops.pad_to(self.cin * self.ps * self.ps, 16)
        self.h0 = self.hidden[-1] // m.ratios[-1] ** 2
        h0, r2 = self.h0, self.r * self.r
        idx = torch.arange(r2 * h0)
        self.pre_perm = (idx if self.explicit_shuffle else (idx % h0) * r2 + idx // h0).to(torch.int32).to(device)
        self.pre_perm_long = self.pre_perm.long()
        # stages: spatial level (number of down-samplings after the stem), skip index into the decoder (or None)
        ns = enc.num_stages
        lvl, levels = 0, []
        for i in range(ns):
            if i and enc.ds_blocks[i]:
                lvl += 1
            levels.append(lvl)
        self.stage_level = levels
        skip_stages = [i for i in range(ns) if i + 1 == ns or enc.ds_blocks[i + 1]]
        self.skip_of_stage = {s: len(skip_stages) - 1 - j for j, s in enumerate(skip_stages)}     # decoder index k fed by stage s
        self.n_levels = lvl + 1
        self._convs = {}
        self._built_for = device

    def _check_supported_rd(self, code, h, w, train):
        enc, m = self.model.encoder, self.model
        kch = 8 if code == L.F32 else 16
        g_align = 4 if code == L.F32 else 8
        name = str(ops.TORCH_DTYPE[code])
        if enc.ds_blocks[0]:
            raise ValueError("ds_blocks[0] must be False (RDNet's first stage has no transition, pssr/models/_rdnet.py:54)")
        if h % (self.ps << (self.n_levels - 1)) or w % (self.ps << (self.n_levels - 1)):
            raise ValueError(f"input size {h}x{w} must be divisible by patch_size * 2^{self.n_levels - 1}")
        if enc.n_init_features % 8:
            raise ValueError(f"rdnet_init={enc.n_init_features} must be a multiple of 8 on the MI355X path")
        for gr in enc.growth_rates:
            if gr % g_align:
                raise ValueError(f"growth rate {gr}: the MI355X path needs multiples of {g_align} for compute dtype {name}")
        for k, hc in enumerate(self.hidden):
            if hc % kch:
                raise ValueError(f"hidden[{k}]={hc}: the MI355X path needs channel counts that are multiples of {kch} for compute dtype {name}")
            if (hc // m.ratios[k + 1] ** 2) % (4 if code == L.F32 else 8):
                raise ValueError(f"hidden[{k}]={hc} / {m.ratios[k + 1]}^2 must keep the skip slice 16-byte aligned")
        if self.h0 % kch:
            raise ValueError(f"head width hidden[-1]/patch_size^2 = {self.h0} must be a multiple of {kch} for compute dtype {name}")
        if train and min(h, w) // (self.ps << (self.n_levels - 1)) < 3:
            raise ValueError("training needs at least 3x3 pixels at the deepest level on the MI355X path")

    # ------------------------------------------------------------------ per-shape plan
    def _plan(self, n, h, w, dt, device):
        key = (n, h, w, dt, str(device))
        p = self.plans.get(key)
        if p is not None:
            return p
        m, enc = self.model, self.model.encoder
        code = ops.dtype_code(dt)
        hid, nd = self.hidden, len(self.hidden)
        p = _obj(n=n, h=h, w=w, dt=dt, code=code, bwd=None)
        f32, f64 = _Arena(), _Arena()
        p.small = _Arena()          # f32 [N, C] side tensors of the ESE gates (means are accumulated with atomics: zeroed per forward)

        def buf(hh, ww, c):
            return torch.zeros(n, hh, ww, ops.pad_to(c, 16), dtype=dt, device=device)

        p.enc_dims = [(h // self.ps >> l, w // self.ps >> l) for l in range(self.n_levels)]
        # decoder block k runs at the resolution of skip k: deepest first
        p.dims = [p.enc_dims[self.n_levels - 1 - k] for k in range(nd)]
        p.bn_in = _BNState(self.cin, f32, f64)
        p.xcol = buf(h, w, self.xc)
        p.xpatch = buf(*p.enc_dims[0], self.pc)
        # decoder concat buffers: cat[k] = [pixel_shuffle(decoder k-1 output) | skip k]; cat[0] is the last stage itself
        p.shuf_c = [0] + [hid[k - 1] // m.ratios[k] ** 2 for k in range(1, nd)]
        p.cat = [buf(*p.dims[k], p.shuf_c[k] + m.skips[k]) for k in range(nd)]
        nl = max(m.depth, 0) + 1
        p.dec = []
        from . import atrous as A
        from .models import ResBlockA
        for k in range(nd):
            b = _obj(level=k, c=hid[k], a=None)
            if isinstance(m.decoder[k], ResBlockA):
                b.a = A.make_ablock_state(m.decoder[k], n, *p.dims[k], p.shuf_c[k] + m.skips[k], dt, device)
                b.y, b.bn = [], []
            else:
                b.y = [buf(*p.dims[k], hid[k]) for _ in range(nl)]
                b.act = None
                b.bn = [_BNState(hid[k], f32, f64) for _ in range(nl)]
            b.out = buf(*p.dims[k], hid[k])
            p.dec.append(b)
        p.feat = buf(h, w, self.h0)
        p.xin = None
        p.epool = p.rpool = None
        if getattr(m, "encoder_pool", None) is not None:         # acts on the deepest skip (rdresunet.py:112-113)
            p.epool = A.make_psp_state(m.encoder_pool, n, *p.dims[0], dt, device)
            p.epool_out = buf(*p.dims[0], m.skips[0])
        if getattr(m, "reconstruction_pool", None) is not None:
            p.rpool = A.make_psp_state(m.reconstruction_pool, n, h, w, dt, device)
            p.rpool_out = buf(h, w, self.h0)
        r2 = self.r * self.r
        p.pre = torch.zeros(n, h, w, r2 * self.h0, dtype=dt, device=device)
        # encoder stages
        p.stem_y = buf(*p.enc_dims[0], enc.n_init_features)
        p.stem_stat = [torch.empty(n * p.enc_dims[0][0] * p.enc_dims[0][1], dtype=torch.float32, device=device) for _ in range(2)]
        p.stages = []
        for i in range(enc.num_stages):
            hh, ww = p.enc_dims[self.stage_level[i]]
            npix = n * hh * ww
            st = _obj(idx=i, h=hh, w=ww, npix=npix, c_in=enc.stage_in[i], c_out=enc.stage_out[i], g=enc.growth_rates[i], ese=enc.ese_blocks[i])
            k = self.skip_of_stage.get(i)
            if k is not None:
                st.F, st.coff, st.skip = p.cat[k], p.shuf_c[k], k
            else:
                st.F, st.coff, st.skip = buf(hh, ww, st.c_out), 0, None
            if i:
                ph, pw = p.enc_dims[self.stage_level[i - 1]]
                st.ds = bool(enc.ds_blocks[i])
                st.tc = enc.trans_in[i]
                st.tcp = ops.pad_to(st.tc, 16)
                st.tr_ln = torch.zeros(n, hh, ww, (4 if st.ds else 1) * st.tcp, dtype=dt, device=device)
                st.tr_stat = [torch.empty(n * ph * pw, dtype=torch.float32, device=device) for _ in range(2)]
            st.blocks = []
            c = st.c_in
            for b in range(enc.n_blocks[i]):
                mod = self._stage_module(i)[b]
                inter = mod.layers.layers[2].weight.shape[0]
                bk = _obj(c_in=c, inter=inter, off=c, mod=mod)
                bk.dw = buf(hh, ww, c)
                bk.ln = buf(hh, ww, c)
                bk.z = buf(hh, ww, inter)
                bk.t = buf(hh, ww, st.g)
                bk.stat = [torch.empty(npix, dtype=torch.float32, device=device) for _ in range(2)]
                bk.wp = torch.empty(49, c, dtype=torch.float32, device=device)
                bk.wpf = torch.empty(49, c, dtype=torch.float32, device=device)
                if st.ese:
                    bk.s_mean, bk.u, bk.gate = (p.small.take(n * st.g) for _ in range(3))
                st.blocks.append(bk)
                c += st.g
            p.stages.append(st)
        p.f32 = f32.build(torch.float32, device)
        p.f64 = f64.build(torch.float64, device)
        p.small.build(torch.float32, device)
        p.bn_in.bind(f32, f64)
        for b in p.dec:
            for s in b.bn:
                s.bind(f32, f64)
        for st in p.stages:
            for bk in st.blocks:
                if st.ese:
                    bk.s_mean, bk.u, bk.gate = (p.small.views[i].view(n, st.g) for i in (bk.s_mean, bk.u, bk.gate))
        p.ones_pre = torch.ones(r2 * self.h0, dtype=torch.float32, device=device)
        p.zeros_pre = torch.zeros(r2 * self.h0, dtype=torch.float32, device=device)
        self.plans[key] = p
        return p

    def _stage_module(self, i):
        """The DenseStage (nn.Sequential of DenseBlocks) of stage i."""
        return self.model.encoder.dense_stages[i][-1]

    def _bwd_buffers(self, p, device):
        if p.bwd is not None:
            return p.bwd
        n, dt, hid, nd = p.n, p.dt, self.hidden, len(self.hidden)
        m, enc = self.model, self.model.encoder
        b = _obj()

        def buf(hh, ww, c):
            return torch.zeros(n, hh, ww, ops.pad_to(c, 16), dtype=dt, device=device)

        # decoder working buffers, indexed by decoder block k (Engine._block_backward indexes them by blk.level)
        b.dz = [buf(*p.dims[k], hid[k]) for k in range(nd)]
        b.dy = [buf(*p.dims[k], hid[k]) for k in range(nd)]
        b.dy2 = [buf(*p.dims[k], hid[k]) for k in range(nd)] if self.side_wgrad else b.dy
        b.dy3 = [buf(*p.dims[k], hid[k]) for k in range(nd)] if self.side_wgrad else b.dy
        b.g = [buf(*p.dims[k], hid[k]) for k in range(nd)]
        b.dout = [buf(*p.dims[k], hid[k]) for k in range(nd)]
        b.dcat = [buf(*p.dims[k], p.shuf_c[k] + m.skips[k]) for k in range(nd)]
        b.dfeat = buf(p.h, p.w, self.h0)
        b.drpool = buf(p.h, p.w, self.h0) if p.rpool is not None else None
        b.depool = buf(*p.dims[0], m.skips[0]) if p.epool is not None else None
        b.dxcol_b = buf(p.h, p.w, self.xc)
        b.dxpatch = buf(*p.enc_dims[0], self.pc)
        r = self.r
        b.g_hr = torch.zeros(n, p.h * r, p.w * r, 16, dtype=dt, device=device)
        b.dpre = torch.zeros(n, p.h, p.w, r * r * self.h0, dtype=dt, device=device)
        b.sum64 = torch.zeros(ops.STAT_STRIPES * max(16, r * r * self.h0), dtype=torch.float64, device=device)
        # encoder: gradient buffers with the layout of the stage buffers; per-stage scratch shared by its blocks
        b.G, b.scr = [], []
        small = _Arena()
        b.stat64 = _Arena()
        maxc = 0
        for st in p.stages:
            if st.skip is not None:
                b.G.append((b.dcat[st.skip], st.coff))
            else:
                b.G.append((buf(st.h, st.w, st.c_out), 0))
            cmax = st.blocks[-1].c_in
            imax = st.blocks[-1].inter
            sc = _obj(ddw=buf(st.h, st.w, cmax), dln=buf(st.h, st.w, cmax), dz=buf(st.h, st.w, imax), dt=buf(st.h, st.w, st.g))
            if st.idx:
                sc.dtr = torch.zeros_like(st.tr_ln)
            sc.A, sc.du, sc.add = (small.take(n * st.g) for _ in range(3))
            b.scr.append(sc)
            maxc = max(maxc, st.c_out, imax)
        b.dstem = buf(*p.enc_dims[0], enc.n_init_features)
        b.stat_ln = torch.zeros(ops.STAT_STRIPES * 2 * max(maxc, enc.n_init_features), dtype=torch.float64, device=device)
        small.build(torch.float32, device)
        b.small = small
        for st, sc in zip(p.stages, b.scr):
            sc.A, sc.du, sc.add = (small.views[i].view(n, st.g) for i in (sc.A, sc.du, sc.add))
        p.bwd = b
        return b

    # ------------------------------------------------------------------ small helpers
    def _pw(self, module, name, code, **spec):
        """Packed weight of a conv module under a named packing spec (cached per parameter version)."""
        c = self._convs.get(id(module))
        if c is None:
            c = self._convs[id(module)] = _Conv(module, {}, self._wepoch)
        if name not in c.specs:
            c.specs[name] = spec
        return c.get(name, code)

    def _ln_grads(self, bw, grads, ln_module, c, s64):
        iw, ib = self._gindex[id(ln_module.weight)], self._gindex[id(ln_module.bias)]
        if self._goffs[ib] == self._goffs[iw] + c:
            # weight and bias slots are adjacent in the flat gradient buffer: convert straight into them (no side copy)
            gb = self._flat_grad[self._goffs[iw]:self._goffs[iw] + 2 * c]
        else:
            gb = torch.empty(2 * c, dtype=torch.float32, device=bw.stat_ln.device)
        self._fold64(s64, gb)
        grads[id(ln_module.weight)], grads[id(ln_module.bias)] = gb[:c], gb[c:]

    def _fold64(self, s64, dst):
        """Striped f64 sums -> f32 parameter gradient: queued for one batched launch (Engine._fold) when the sums sit in the arena."""
        if self._z64 is not None and self._z64.owns(s64):
            self._fold(s64, dst)
        else:
            ops.f64_to_f32(s64, dst)

    def _bias_grad(self, bw, grads, bias, t, npix, c, code, coff=0):
        """d bias = per-channel sum of the output gradient (f64 striped accumulation)."""
        s64 = self._z64.take(ops.STAT_STRIPES * c, bw.stat_ln)
        ops.channel_sum_nhwc(t, npix, c, s64, code, coff=coff)
        g = self._gbuf(bias)               # straight into the parameter's slot of the flat gradient buffer
        self._fold64(s64, g)
        grads[id(bias)] = g

    def _bias_grad_job(self, bw, grads, bias, t, npix, c, code, coff=0):
        """The same sum as _bias_grad as a closure for the second stream: it rides in front of the weight gradient that reads the same
        tensor (49 launches of ~11 us that nothing on the dependent chain waits for)."""
        s64 = self._z64.take(ops.STAT_STRIPES * c, bw.stat_ln)
        if not self._z64.owns(s64):         # first pass: the arena's fallback buffer is reused by the next request, this sum is read later
            s64 = torch.zeros(ops.STAT_STRIPES * c, dtype=torch.float64, device=bw.stat_ln.device)
            # allocated on the launch stream, used on the second one: kept alive until the streams have joined (a block freed when the
            # closure dies goes back to the launch stream's pool and can be handed out again while the side kernels still read it)
            self.__dict__.setdefault("_keepalive", []).append(s64)
        g = self._gbuf(bias)
        grads[id(bias)] = g

        def run():
            ops.channel_sum_nhwc(t, npix, c, s64, code, coff=coff)
            ops.f64_to_f32(s64, g)
        return run

    def _wgrad1x1(self, p, grads, conv_module, dy, cout, dy_coff, src, cin_pad, hh, ww, *, mode=0, gelu_in=False, first=None):
        code = p.code
        esz = 4 if code == L.F32 else 2
        rows = cout if (cout * esz) % 16 == 0 else ops.pad_to(cout, 16)
        w = conv_module.weight
        slot = grads[id(w)] = self._gbuf(w)

        def run():
            if first is not None:
                first()
            dwp = ops.conv2d_wgrad_parts(dy, rows, src, cin_pad, 1, n=p.n, h=hh, w=ww, dtype=code, dy_coff=dy_coff, gelu_in=gelu_in)
            ops.unpack_conv_wgrad(dwp, slot, mode=mode, k_pad=cin_pad, accumulate=True)     # slot zeroed at the start of backward
        if self._side_on:
            self._on_side([dy], run)       # second stream: see Engine._on_side / _before_write
        else:
            run()

    # ------------------------------------------------------------------ forward
    def forward(self, x, train):
        m, enc = self.model, self.model.encoder
        if not x.is_cuda:
            raise RuntimeError("pssr2_amd.RDResUNet runs on an MI355X (HIP) device only; there is no CPU fallback")
        x = x.contiguous().float()
        self._structure(x.device)
        n, c, h, w = x.shape
        if c != self.cin:
            raise ValueError(f"expected {self.cin} input channels, got {c}")
        dt = self.storage_dtype(train)
        code = ops.dtype_code(dt)
        self._check_supported_rd(code, h, w, train)
        p = self._plan(n, h, w, dt, x.device)
        self._repack_all(code)
        hid, nd = self.hidden, len(self.hidden)
        if train:
            p.f64.buf.zero_()
            self._count_batches()
        if self.atrous:
            p.bn_in.scale.fill_(1.0), p.bn_in.shift.zero_()          # no input BatchNorm
        else:
            if train:
                ops.channel_stats_nchw(x, p.bn_in.stats, 1 / 128, -1.0)
            self._bn_forward(p, p.bn_in, m.norm, float(n * h * w), train)
        ops.input_im2col(x, p.xcol, p.bn_in.scale, p.bn_in.shift, code)
        ops.input_patchify(x, p.xpatch, p.bn_in.scale, p.bn_in.shift, self.ps, code)
        p.small.buf.zero_()
        # ---- stem: conv(k = s = patch) as a 1x1 conv over patches, then LayerNorm2d into stage 0's buffer
        st0 = p.stages[0]
        stem_conv, stem_ln = enc.stem.stem[0], enc.stem.stem[1]
        c0 = enc.n_init_features
        ops.conv2d(p.xpatch, self.pc, self._pw(stem_conv, "fwd", code, mode=2), p.stem_y, c0, n=n, h=st0.h, w=st0.w, bias=stem_conv.bias)
        ops.layernorm2d_fwd(p.stem_y, stem_ln.weight, stem_ln.bias, LN_EPS, st0.F, n, st0.h, st0.w, c0, code, out_coff=st0.coff, c_pad=c0,
                            mean=p.stem_stat[0] if train else None, rstd=p.stem_stat[1] if train else None)
        # ---- dense stages (every depthwise weight, and its rotated copy for the backward pass, packed by one launch)
        dw_items = []
        for st in p.stages:
            for bk in st.blocks:
                dww = bk.mod.layers.layers[0].weight
                dw_items.append((dww, bk.wp, False))
                if train:
                    dw_items.append((dww, bk.wpf, True))
        ops.dwconv7_pack_batch(dw_items)
        for i, st in enumerate(p.stages):
            if i:
                prev = p.stages[i - 1]
                seq = enc.dense_stages[i]
                ln, conv = seq[0], seq[1]
                ops.layernorm2d_fwd(prev.F, ln.weight, ln.bias, LN_EPS, st.tr_ln, n, prev.h, prev.w, st.tc, code, in_coff=prev.coff, s2d=st.ds,
                                    c_pad=st.tcp, mean=st.tr_stat[0] if train else None, rstd=st.tr_stat[1] if train else None)
                pw = self._pw(conv, "fwd", code, mode=4 if st.ds else 0)
                ops.conv2d(st.tr_ln, st.tr_ln.shape[-1], pw, st.F, st.c_in, n=n, h=st.h, w=st.w, out_coff=st.coff, bias=conv.bias)
            for bk in st.blocks:
                lay = bk.mod.layers.layers
                dwc, ln, c1, c2 = lay[0], lay[1], lay[2], lay[4]
                ops.dwconv7(st.F, bk.wp, dwc.bias, bk.dw, n, st.h, st.w, bk.c_in, code, in_coff=st.coff)
                ops.layernorm2d_fwd(bk.dw, ln.weight, ln.bias, LN_EPS, bk.ln, n, st.h, st.w, bk.c_in, code,
                                    mean=bk.stat[0] if train else None, rstd=bk.stat[1] if train else None)
                ops.conv2d(bk.ln, bk.ln.shape[-1], self._pw(c1, "fwd", code, mode=0), bk.z, bk.inter, n=n, h=st.h, w=st.w, bias=c1.bias)
                ops.conv2d(bk.z, bk.inter, self._pw(c2, "fwd", code, mode=0), bk.t, st.g, n=n, h=st.h, w=st.w, bias=c2.bias, gelu_in=True)
                gate = None
                if st.ese:
                    fc = lay[5].fc
                    ops.image_channel_dot(bk.t, None, n, st.h * st.w, st.g, 1.0 / (st.h * st.w), bk.s_mean, code)
                    ops.ese_gate(bk.s_mean, fc.weight, fc.bias, bk.u, bk.gate)
                    gate = bk.gate
                ops.scale_nc(bk.t, gate, bk.mod.gamma, None, st.F, n, st.h * st.w, st.g, code, out_coff=st.coff + bk.off)
        # ---- decoder
        from . import atrous as A
        for k in range(nd):
            blk = p.dec[k]
            src = p.cat[k]
            if k == 0 and p.epool is not None:
                A.psp_forward(self, p.epool, m.encoder_pool, p.cat[0], 0, n, code, p.epool_out, 0, train)
                src = p.epool_out
            r = m.ratios[k + 1]
            nxt = p.cat[k + 1] if k + 1 < nd else p.feat
            # (eval mode, ratio 2: the block's last convolution stores its output shuffled into `nxt` itself -- Engine._block_forward)
            if self._block_forward(p, blk, m.decoder[k], src, src.shape[-1], False, blk.out, 0, train, shuf=nxt if (r == 2 and not train) else None):
                continue
            ops.pixel_shuffle(blk.out, nxt, n, *p.dims[k], hid[k] // (r * r) if k + 1 < nd else self.h0, r, code)
        feat = p.feat
        if p.rpool is not None:
            A.psp_forward(self, p.rpool, m.reconstruction_pool, p.feat, 0, n, code, p.rpool_out, 0, train)
            feat = p.rpool_out
        out = self._head_forward(p, feat, x, train)
        self._flush_fwd()
        self.saved = (p, x) if train else None
        return out

    # ------------------------------------------------------------------ backward
    def _dense_block_backward(self, p, bw, grads, st, sc, bk, G, gcoff):
        n, code = p.n, p.code
        hw = st.h * st.w
        lay = bk.mod.layers.layers
        dwc, ln, c1, c2 = lay[0], lay[1], lay[2], lay[4]
        g = st.g
        # ---- layer scale (+ ESE gate): dt, dgamma (, d fc)
        A = self._z32.take(n * g, sc.A.view(-1)).view(n, g)
        ops.image_channel_dot(G, bk.t, n, hw, g, 1.0, A, code, a_coff=gcoff + bk.off)
        dgam = self._gbuf(bk.mod.gamma)
        if st.ese:
            fc = lay[5].fc
            dbfc = self._gbuf(fc.bias)
            dwfc = self._gbuf(fc.weight)
            ops.ese_bwd(A, bk.gate, bk.u, bk.mod.gamma, bk.s_mean, fc.weight, hw, sc.du, dgam, dbfc, dwfc, sc.add)
            grads[id(fc.weight)], grads[id(fc.bias)] = dwfc, dbfc
            self._before_write(sc.dt)
            ops.scale_nc(G, bk.gate, bk.mod.gamma, sc.add, sc.dt, n, hw, g, code, t_coff=gcoff + bk.off)
        else:
            ops.ese_bwd(A, None, None, bk.mod.gamma, None, None, hw, None, dgam, None, None, None)
            self._before_write(sc.dt)
            ops.scale_nc(G, None, bk.mod.gamma, None, sc.dt, n, hw, g, code, t_coff=gcoff + bk.off)
        grads[id(bk.mod.gamma)] = dgam
        # ---- second 1x1 conv (input = gelu(z))
        self._wgrad1x1(p, grads, c2, sc.dt, g, 0, bk.z, bk.inter, st.h, st.w, gelu_in=True,
                       first=self._bias_grad_job(bw, grads, c2.bias, sc.dt, st.npix, g, code))
        s64 = self._z64.take(ops.STAT_STRIPES * 2 * bk.inter, bw.stat_ln)
        self._before_write(sc.dz)
        ops.conv2d(sc.dt, sc.dt.shape[-1], self._pw(c2, "dgrad", code, mode=1), sc.dz, bk.inter, n=n, h=st.h, w=st.w,
                   epilogue=L.EPI_DGRAD_GELU, flags=L.FLAG_STATS, aux=bk.z, stats=s64)
        sums = torch.empty(2 * bk.inter, dtype=torch.float32, device=G.device)
        self._fold64(s64, sums)
        grads[id(c1.bias)] = sums[:bk.inter]
        # ---- first 1x1 conv (input = LayerNorm output)
        cpad = bk.ln.shape[-1]
        self._wgrad1x1(p, grads, c1, sc.dz, bk.inter, 0, bk.ln, cpad, st.h, st.w)
        ops.conv2d(sc.dz, bk.inter, self._pw(c1, "dgrad", code, mode=1), sc.dln, bk.c_in, n=n, h=st.h, w=st.w)
        # ---- LayerNorm2d
        s64 = self._z64.take(ops.STAT_STRIPES * 2 * bk.c_in, bw.stat_ln)
        self._before_write(sc.ddw)
        ops.layernorm2d_bwd(sc.dln, bk.dw, ln.weight, bk.stat[0], bk.stat[1], sc.ddw, s64, n, st.h, st.w, bk.c_in, code, c_pad=bk.c_in)
        self._ln_grads(bw, grads, ln, bk.c_in, s64)
        # ---- depthwise 7x7
        dwb_run = self._bias_grad_job(bw, grads, dwc.bias, sc.ddw, st.npix, bk.c_in, code)
        dww = self._gbuf(dwc.weight)
        ddw, c_in = sc.ddw, bk.c_in

        def dw_run():
            dwb_run()
            ops.dwconv7_wgrad(ddw, st.F, dww.view(c_in, 49), n, st.h, st.w, c_in, code, x_coff=st.coff)
        if self._side_on:
            self._on_side([ddw], dw_run)
        else:
            dw_run()
        grads[id(dwc.weight)] = dww
        ops.dwconv7(sc.ddw, bk.wpf, None, G, n, st.h, st.w, bk.c_in, code, out_coff=gcoff, accumulate=True)
        self._ready(grads, list(bk.mod.parameters()))

    def grad_split_offset(self):
        """First element of the flat gradient buffer that is final at the split callback of ``backward``: the decoder's first
        parameter (module order: norm, encoder, decoder, reconstruction; the backward runs reconstruction, decoder, encoder)."""
        first = next(self.model.decoder.parameters())
        return self._goffs[self._gindex[id(first)]]

    def backward(self, dout, split_cb=None):
        if self._z64 is None:
            self._z64, self._z32 = _ZeroArena(torch.float64), _ZeroArena(torch.float32)
        if self.saved is None:
            raise RuntimeError("backward called without a training-mode forward (or called twice)")
        p, x = self.saved
        self.saved = None
        m, enc = self.model, self.model.encoder
        dev = x.device
        bw = self._bwd_buffers(p, dev)
        n, h, w, code = p.n, p.h, p.w, p.code
        hid, nd, r, h0 = self.hidden, len(self.hidden), self.r, self.h0
        grads = {}
        self._begin_backward(dev)
        for arena in (self._z64, self._z32):
            arena.begin(dev)
        from . import atrous as A
        if p.rpool is not None:
            self._head_backward(p, bw, grads, dout, p.rpool_out, bw.drpool)
            A.psp_backward(self, p.rpool, m.reconstruction_pool, grads, p.feat, 0, n, code, p.rpool_out, 0, bw.drpool, 0, bw.dfeat, 0)
        else:
            self._head_backward(p, bw, grads, dout, p.feat, bw.dfeat)
        # ---- decoder, last block first
        for k in range(nd - 1, -1, -1):
            blk = p.dec[k]
            rr = m.ratios[k + 1]
            dhi = bw.dcat[k + 1] if k + 1 < nd else bw.dfeat
            dfrom = None
            if rr == 2 and self._fused_dout(p, blk, "unshuffle"):
                dfrom = ("unshuffle", dhi)          # relu_bwd_stats reads the block's output gradient out of the finer level's buffer
            else:
                ops.pixel_shuffle(bw.dout[k], dhi, n, *p.dims[k], hid[k] // (rr * rr) if k + 1 < nd else h0, rr, code, inverse=True)
            if k == 0 and p.epool is not None:
                self._block_backward(p, bw, grads, blk, m.decoder[k], p.epool_out, p.epool_out.shape[-1], False, blk.out, 0, bw.dout[k], bw.depool,
                                     m.skips[0], dout_from=dfrom)
                A.psp_backward(self, p.epool, m.encoder_pool, grads, p.cat[0], 0, n, code, p.epool_out, 0, bw.depool, 0, bw.dcat[0], 0)
            else:
                self._block_backward(p, bw, grads, blk, m.decoder[k], p.cat[k], p.cat[k].shape[-1], False, blk.out, 0, bw.dout[k], bw.dcat[k],
                                     p.shuf_c[k] + m.skips[k], dout_from=dfrom)
        if split_cb is not None:       # reconstruction + decoder gradients (the tail of the flat buffer) are final
            self._flush_folds()
            self._flush_moves()
            self._side_join()
            split_cb()
        # ---- encoder, last stage first
        bw.small.buf.zero_()
        for i in range(len(p.stages) - 1, -1, -1):
            st, sc = p.stages[i], bw.scr[i]
            G, gcoff = bw.G[i]
            for bk in reversed(st.blocks):
                self._dense_block_backward(p, bw, grads, st, sc, bk, G, gcoff)
            if i:
                prev = p.stages[i - 1]
                seq = enc.dense_stages[i]
                ln, conv = seq[0], seq[1]
                kc = st.tr_ln.shape[-1]
                self._wgrad1x1(p, grads, conv, G, st.c_in, gcoff, st.tr_ln, kc, st.h, st.w, mode=4 if st.ds else 0,
                               first=self._bias_grad_job(bw, grads, conv.bias, G, st.npix, st.c_in, code, coff=gcoff))
                # the dgrad reads the first c_in channels of G at its offset (K padded to 16: the packed weight rows beyond
                # c_in are zero and the gradient buffer holds finite values there)
                ops.conv2d(G, ops.pad_to(st.c_in, 16), self._pw(conv, "dgrad", code, mode=5 if st.ds else 1), sc.dtr, kc,
                           n=n, h=st.h, w=st.w, in0_coff=gcoff)
                Gp, gpo = bw.G[i - 1]
                s64 = self._z64.take(ops.STAT_STRIPES * 2 * st.tc, bw.stat_ln)
                ops.layernorm2d_bwd(sc.dtr, prev.F, ln.weight, st.tr_stat[0], st.tr_stat[1], Gp, s64, n, prev.h, prev.w, st.tc, code,
                                    x_coff=prev.coff, dx_coff=gpo, s2d=st.ds, c_pad=st.tcp, accumulate=prev.skip is not None)
                self._ln_grads(bw, grads, ln, st.tc, s64)
                self._ready(grads, [ln.weight, ln.bias, conv.weight, conv.bias])
        # ---- stem
        st0 = p.stages[0]
        G0, g0o = bw.G[0]
        stem_conv, stem_ln = enc.stem.stem[0], enc.stem.stem[1]
        c0 = enc.n_init_features
        s64 = self._z64.take(ops.STAT_STRIPES * 2 * c0, bw.stat_ln)
        ops.layernorm2d_bwd(G0, p.stem_y, stem_ln.weight, p.stem_stat[0], p.stem_stat[1], bw.dstem, s64, n, st0.h, st0.w, c0, code,
                            g_coff=g0o, c_pad=c0)
        self._ln_grads(bw, grads, stem_ln, c0, s64)
        self._bias_grad(bw, grads, stem_conv.bias, bw.dstem, st0.npix, c0, code)
        self._wgrad1x1(p, grads, stem_conv, bw.dstem, c0, 0, p.xpatch, self.pc, st0.h, st0.w, mode=2)
        ops.conv2d(bw.dstem, ops.pad_to(c0, 16), self._pw(stem_conv, "dgrad", code, mode=3), bw.dxpatch, self.pc, n=n, h=st0.h, w=st0.w)
        self._ready(grads, list(enc.stem.parameters()))
        if self.atrous:                 # no input BatchNorm
            return self._finish_backward(grads)
        # ---- input BatchNorm parameters (gradient sources: head im2col + stem patches)
        stn = p.bn_in
        stn.bstats.zero_()
        self._before_write(bw.dxcol_b)          # (its producer may have run on the second stream: Engine._head_backward)
        ops.input_norm_bwd2(None, bw.dxcol_b, bw.dxpatch, self.ps, x, stn.mean, stn.invstd, stn.bstats, code)
        dgam, dbet = self._gbuf(m.norm.weight), self._gbuf(m.norm.bias)
        self.bn_coefs(stn.bstats, float(n * h * w), m.norm.weight, stn.mean, stn.invstd, stn.ca, stn.cb, stn.cc, dgam, dbet)
        grads[id(m.norm.weight)], grads[id(m.norm.bias)] = dgam, dbet
        self._ready(grads, list(m.norm.parameters()))
        return self._finish_backward(grads)
