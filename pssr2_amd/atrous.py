"""Atrous blocks and PSP pooling for the MI355X engines (pssr/models/_blocks.py:43-92).

``ResBlockA``: every dilated 3x3 convolution runs as ``im2col_dil`` (the pre-activation BatchNorm + ReLU fused into the gather)
+ the 1x1 implicit-GEMM kernels over K = 9 * C (weights packed in mode 4; mode 5 for the input gradient) + ``col2im_dil`` (ReLU
mask and BatchNorm-backward statistics fused) -- see csrc/atrous.hip for why.  ``PSP_Pooling``: per channel chunk k x k max
pooling, bilinear resize back, conv1x1 + BatchNorm (+ ReLU as the prologue of the consumer), then conv1x1 + BatchNorm + ReLU.

The functions take the engine (for its packed-weight cache, gradient slots and BatchNorm helpers), a per-shape state object built
by ``make_*_state`` at plan time, and NHWC buffers; both ``Engine`` (ResUNet) and ``RDEngine`` (RDResUNet decoder) call them.
"""
from __future__ import annotations

import torch

from . import _lib as L
from . import ops

BN_EPS, BN_MOMENTUM = 1e-5, 0.1


def _ns(**kw):
    return type("S", (), kw)()


def _bn_alloc(c, dev):
    """scale, shift, mean, invstd, coefA, coefB, coefC (f32 [c]) + forward / backward statistics (f64, striped)."""
    st = _ns(c=c)
    st.scale, st.shift, st.mean, st.invstd, st.ca, st.cb, st.cc = (torch.zeros(c, dtype=torch.float32, device=dev) for _ in range(7))
    st.stats = torch.zeros(ops.STAT_STRIPES * 2 * c, dtype=torch.float64, device=dev)
    st.bstats = torch.zeros(ops.STAT_STRIPES * 2 * c, dtype=torch.float64, device=dev)
    st.eval_key = None
    return st


def _bn_fwd(eng, st, bn, stats, count, train, synced=False):
    if train:
        st.eval_key = None
        if not synced:
            count = eng.bn_stats_count(stats, count)          # sync_bn: statistics over all ranks
        ops.bn_finalize(stats, count, bn.weight, bn.bias, BN_EPS, BN_MOMENTUM, bn.running_mean, bn.running_var, st.scale[:st.c], st.shift[:st.c],
                        st.mean[:st.c], st.invstd[:st.c])
    else:
        ts = (bn.weight, bn.bias, bn.running_mean, bn.running_var)
        key = (eng._wepoch[0],) + tuple((t.data_ptr(), t._version) for t in ts)
        if st.eval_key != key:
            ops.bn_eval_affine(*ts, BN_EPS, st.scale[:st.c], st.shift[:st.c])
            st.eval_key = key


def _bias_grad(eng, grads, bias, t, npix, c, code, coff=0):
    """d(bias) = sum over pixels of the NHWC slice t[..., coff:coff+c]."""
    s64 = torch.zeros(ops.STAT_STRIPES * c, dtype=torch.float64, device=t.device)
    ops.channel_sum_nhwc(t, npix, c, s64, code, coff=coff)
    g = eng._gbuf(bias)
    ops.f64_to_f32(s64, g)
    grads[id(bias)] = g


def _wgrad1(eng, grads, conv, dy, cout, dy_coff, src, cin_pad, n, h, w, code, *, mode=0, pro=None, src_coff=0):
    """1x1-kernel weight gradient of `conv` (mode 0: an ordinary 1x1 conv; mode 4: a 3x3 weight over the im2col layout)."""
    esz = 4 if code == L.F32 else 2
    if (cout * esz) % 16:               # rows beyond cout are zero-padded channels of dy (the unpack drops them)
        assert dy_coff == 0
        cout = ops.pad_to(cout, 16)
    parts = ops.conv2d_wgrad_parts(dy, cout, src, cin_pad, 1, n=n, h=h, w=w, dtype=code, dy_coff=dy_coff, in_coff=src_coff,
                                   pro_scale=pro.scale if pro is not None else None, pro_shift=pro.shift if pro is not None else None)
    slot = eng._gbuf(conv.weight)
    ops.unpack_conv_wgrad(parts, slot, mode=mode, k_pad=cin_pad, accumulate=True)
    grads[id(conv.weight)] = slot


# ======================================================================================================== ResBlockA
def make_ablock_state(module, n, hh, ww, cin_real, dt, dev):
    """Buffers of one ResBlockA at one resolution."""
    c = module.respass.weight.shape[0]
    nl = max(module.depth, 0) + 1
    cp_in, cp = ops.pad_to(cin_real, 16), ops.pad_to(c, 16)

    def buf(ch):
        return torch.zeros(n, hh, ww, ops.pad_to(ch, 16), dtype=dt, device=dev)
    st = _ns(kind="A", c=c, cin=cin_real, nl=nl, dils=list(module.dilation_values), hh=hh, ww=ww)
    st.br = []
    for _ in st.dils:
        st.br.append(_ns(y=[buf(c) for _ in range(nl)], bn=[_bn_alloc(cin_real if k == 0 else c, dev) for k in range(nl)]))
    st.r = buf(c)
    st.stats_in = torch.zeros(ops.STAT_STRIPES * 2 * cin_real, dtype=torch.float64, device=dev)
    st.col = torch.zeros(n, hh, ww, 9 * max(cp_in, cp), dtype=dt, device=dev)       # im2col scratch (forward, and recomputed for wgrad)
    st.dcol = None                                                                  # backward scratch, made on first use
    st.bwd = None
    return st


def ablock_forward(eng, st, module, src, src_coff, n, code, dst, dst_coff, train):
    """relu(sum_d branch_d(src) + respass(src)) into dst[..., dst_coff:dst_coff + c]; src[..., src_coff:src_coff + cin] is the block input."""
    hh, ww, c, cin = st.hh, st.ww, st.c, st.cin
    if ww < module.min_size:
        raise ValueError(f"Tensor size {(n, cin, hh, ww)} is smaller than than dilation kernel size {module.min_size}.")
    npix = n * hh * ww
    count = float(npix)
    count_in = count
    if train:
        st.stats_in.zero_()
        ops.channel_stats_nhwc(src, cin, npix, st.stats_in, code, coff=src_coff)
        count_in = eng.bn_stats_count(st.stats_in, count)     # shared by the first BatchNorm of every branch: reduced over ranks once
    for br, dil, seq in zip(st.br, st.dils, module.dilations):
        for k in range(st.nl):
            bn, conv = seq[3 * k], seq[3 * k + 2]
            bs = br.bn[k]
            if train and k + 1 < st.nl:
                br.bn[k + 1].stats.zero_()
            if k == 0:
                _bn_fwd(eng, bs, bn, st.stats_in, count_in, train, synced=True)
            else:
                _bn_fwd(eng, bs, bn, bs.stats, count, train)
            inp, ioff, ci = (src, src_coff, cin) if k == 0 else (br.y[k - 1], 0, c)
            cp = ops.pad_to(ci, 16)
            col = st.col.view(-1)[:npix * 9 * cp].view(n, hh, ww, 9 * cp)
            ops.im2col_dil(inp, ci, col, cp, n, hh, ww, dil, code, in_coff=ioff, scale=bs.scale, shift=bs.shift)
            pw = eng._pw_any(conv, "fwd", code, mode=4)
            more = train and k + 1 < st.nl
            ops.conv2d(col, 9 * cp, pw, br.y[k], c, n=n, h=hh, w=ww, bias=conv.bias, flags=L.FLAG_STATS if more else 0,
                       stats=br.bn[k + 1].stats if more else None)
    rp = module.respass
    cpi = ops.pad_to(cin, 16)
    if src_coff or src.shape[-1] < cpi:
        raise RuntimeError("ResBlockA input must start its buffer and be padded to 16 channels")
    ops.conv2d(src, cpi, eng._pw_any(rp, "fwd", code, mode=0), st.r, c, n=n, h=hh, w=ww, bias=rp.bias)
    ops.sum_relu([(br.y[-1], 0) for br in st.br] + [(st.r, 0)], dst, npix, c, code, relu=True, out_coff=dst_coff)


def ablock_backward(eng, st, module, grads, src, src_coff, n, code, out_buf, out_coff, dout, dout_coff, dsrc, need_dsrc):
    """Gradient of ablock_forward: parameter gradients into the engine's slots; d(src) written to dsrc[..., :cin] when need_dsrc."""
    hh, ww, c, cin = st.hh, st.ww, st.c, st.cin
    npix = n * hh * ww
    count = float(npix)
    dev, dt = src.device, src.dtype
    if st.bwd is None:
        def buf(ch):
            return torch.zeros(n, hh, ww, ops.pad_to(ch, 16), dtype=dt, device=dev)
        cpm = max(ops.pad_to(cin, 16), ops.pad_to(c, 16))
        st.bwd = _ns(dz=buf(c), dy=[buf(c), buf(c)], g=buf(max(c, cin)), t=buf(cin), acc=buf(cin),
                     dcol=torch.zeros(n, hh, ww, 9 * cpm, dtype=dt, device=dev))
    b = st.bwd
    ops.relu_mask(dout, out_buf, b.dz, npix, c, code, do_coff=dout_coff, o_coff=out_coff)
    rp = module.respass
    cpi = ops.pad_to(cin, 16)
    _bias_grad(eng, grads, rp.bias, b.dz, npix, c, code)
    _wgrad1(eng, grads, rp, b.dz, c, 0, src, cpi, n, hh, ww, code, mode=0)
    parts = []          # contributions to d(src)
    if need_dsrc:
        ops.conv2d(b.dz, ops.pad_to(c, 16), eng._pw_any(rp, "dgrad", code, mode=1), b.acc, cin, n=n, h=hh, w=ww)
    for br, dil, seq in zip(st.br, st.dils, module.dilations):
        dy = b.dz
        for k in range(st.nl - 1, -1, -1):
            bn, conv = seq[3 * k], seq[3 * k + 2]
            bs = br.bn[k]
            inp, ioff, ci = (src, src_coff, cin) if k == 0 else (br.y[k - 1], 0, c)
            cp = ops.pad_to(ci, 16)
            if k == st.nl - 1:
                grads[id(conv.bias)] = grads[id(rp.bias)]       # both biases add to the block sum: d = sum dz (copied by _ready)
            # (a conv bias in front of a batch-statistics BatchNorm has an exactly zero gradient: its slot stays zeroed)
            col = st.col.view(-1)[:npix * 9 * cp].view(n, hh, ww, 9 * cp)
            ops.im2col_dil(inp, ci, col, cp, n, hh, ww, dil, code, in_coff=ioff, scale=bs.scale, shift=bs.shift)
            _wgrad1(eng, grads, conv, dy, c, 0, col, 9 * cp, n, hh, ww, code, mode=4)
            if k == 0 and not need_dsrc and not bn.weight.requires_grad:
                continue
            dcol = b.dcol.view(-1)[:npix * 9 * cp].view(n, hh, ww, 9 * cp)
            ops.conv2d(dy, ops.pad_to(c, 16), eng._pw_any(conv, "dgrad", code, mode=5), dcol, 9 * cp, n=n, h=hh, w=ww)
            bs.bstats.zero_()
            ops.col2im_dil(dcol, cp, b.g, ci, n, hh, ww, dil, code, y=inp, y_coff=ioff, scale=bs.scale, shift=bs.shift, mean=bs.mean,
                           invstd=bs.invstd, stats=bs.bstats)
            dgam, dbet = eng._gbuf(bn.weight), eng._gbuf(bn.bias)
            eng.bn_coefs(bs.bstats, count, bn.weight, bs.mean[:ci], bs.invstd[:ci], bs.ca[:ci], bs.cb[:ci], bs.cc[:ci], dgam, dbet)
            grads[id(bn.weight)], grads[id(bn.bias)] = dgam, dbet
            if k > 0:
                nxt = b.dy[k & 1]
                ops.bn_bwd_apply(b.g, inp, bs.ca, bs.cb, bs.cc, nxt, npix, ci, code)
                dy = nxt
            elif need_dsrc:
                ops.bn_bwd_apply(b.g, inp, bs.ca, bs.cb, bs.cc, b.t, npix, ci, code, y_coff=ioff)
                ops.sum_relu([(b.acc, 0), (b.t, 0)], b.acc, npix, ci, code, relu=False)
    if need_dsrc:
        ops.sum_relu([(b.acc, 0)], dsrc, npix, cin, code, relu=False)
    eng._ready(grads, list(module.parameters()))


# ======================================================================================================== PSP_Pooling
def make_psp_state(module, n, hh, ww, dt, dev):
    C = module.channels
    sizes = list(module.sizes)
    ns = len(sizes)
    small = C // ns
    if small * ns != C or small % 4:
        raise ValueError(f"PSP_Pooling: {C} channels in {ns} chunks: the MI355X path needs chunks of a multiple of 4 channels")
    for k in sizes:
        if k > min(hh, ww):
            raise ValueError(f"PSP_Pooling: pooling ratio {k} exceeds the {hh}x{ww} feature map")
    sp, Cp = ops.pad_to(small, 16), ops.pad_to(C, 16)

    def buf(h_, w_, ch):
        return torch.zeros(n, h_, w_, ch, dtype=dt, device=dev)
    st = _ns(kind="P", C=C, small=small, sp=sp, Cp=Cp, sizes=sizes, hh=hh, ww=ww)
    st.pool = [buf(hh // k, ww // k, sp) for k in sizes]
    st.up = [buf(hh, ww, sp) if k > 1 else None for k in sizes]
    st.u, st.v = buf(hh, ww, Cp), buf(hh, ww, Cp)
    st.bn = [_bn_alloc(small, dev) for _ in sizes]
    st.all = _bn_alloc(Cp, dev)          # the chunk BatchNorms side by side (prologue / mask vectors of conv_out over C channels)
    st.all.c = C
    st.bn_out = _bn_alloc(C, dev)
    st.bwd = None
    return st


def psp_forward(eng, st, module, src, src_coff, n, code, dst, dst_coff, train):
    hh, ww, C, small, sp = st.hh, st.ww, st.C, st.small, st.sp
    npix = n * hh * ww
    count = float(npix)
    for i, k in enumerate(st.sizes):
        conv, bn = module.convs[i][0], module.convs[i][1]
        ops.maxpool_k(src, st.pool[i], n, hh, ww, small, k, code, in_coff=src_coff + i * small)
        if k > 1:
            ops.bilinear_up(st.pool[i], st.up[i], n, hh // k, ww // k, hh, ww, small, code)
        inp = st.up[i] if k > 1 else st.pool[i]
        bs = st.bn[i]
        if train:
            bs.stats.zero_()
        ops.conv2d(inp, sp, eng._pw_any(conv, "fwd", code, mode=0), st.u, small, n=n, h=hh, w=ww, out_coff=i * small, bias=conv.bias,
                   flags=L.FLAG_STATS if train else 0, stats=bs.stats if train else None)
        _bn_fwd(eng, bs, bn, bs.stats, count, train)
        sl = slice(i * small, (i + 1) * small)
        for name in ("scale", "shift", "mean", "invstd"):
            getattr(st.all, name)[sl].copy_(getattr(bs, name))
    co, bo = module.conv_out, module.norm_out
    if train:
        st.bn_out.stats.zero_()
    ops.conv2d(st.u, st.Cp, eng._pw_any(co, "fwd", code, mode=0), st.v, C, n=n, h=hh, w=ww, bias=co.bias, pro_scale=st.all.scale,
               pro_shift=st.all.shift, flags=L.FLAG_STATS if train else 0, stats=st.bn_out.stats if train else None)
    _bn_fwd(eng, st.bn_out, bo, st.bn_out.stats, count, train)
    ops.affine_relu(st.v, st.bn_out.scale, st.bn_out.shift, dst, npix, C, code, out_coff=dst_coff)


def psp_backward(eng, st, module, grads, src, src_coff, n, code, out_buf, out_coff, dout, dout_coff, dsrc, dsrc_coff):
    """dsrc[..., dsrc_coff : dsrc_coff + C] = d(input)."""
    hh, ww, C, small, sp, Cp = st.hh, st.ww, st.C, st.small, st.sp, st.Cp
    npix = n * hh * ww
    count = float(npix)
    dev, dt = src.device, src.dtype
    if st.bwd is None:
        def buf(h_, w_, ch):
            return torch.zeros(n, h_, w_, ch, dtype=dt, device=dev)
        st.bwd = _ns(dz=buf(hh, ww, Cp), dv=buf(hh, ww, Cp), g=buf(hh, ww, Cp), dconv=buf(hh, ww, Cp), dchunk=buf(hh, ww, sp), dup=buf(hh, ww, sp),
                     dpool=[buf(hh // k, ww // k, sp) for k in st.sizes], gamma_all=torch.zeros(Cp, dtype=torch.float32, device=dev),
                     dgam_all=torch.zeros(Cp, dtype=torch.float32, device=dev), dbet_all=torch.zeros(Cp, dtype=torch.float32, device=dev))
    b = st.bwd
    co, bo = module.conv_out, module.norm_out
    so = st.bn_out
    # ---- out = relu(BN_out(v))
    if dout_coff:
        raise RuntimeError("psp_backward expects the output gradient at channel offset 0")
    so.bstats.zero_()
    ops.relu_bwd_stats(dout, out_buf, st.v, so.mean, so.invstd, b.dz, so.bstats, npix, C, code, out_coff=out_coff)
    dgam, dbet = eng._gbuf(bo.weight), eng._gbuf(bo.bias)
    eng.bn_coefs(so.bstats, count, bo.weight, so.mean, so.invstd, so.ca, so.cb, so.cc, dgam, dbet)
    grads[id(bo.weight)], grads[id(bo.bias)] = dgam, dbet
    ops.bn_bwd_apply(b.dz, st.v, so.ca, so.cb, so.cc, b.dv, npix, C, code)
    # ---- conv_out over relu(BN_i(u_i)) (its bias precedes a batch-statistics BatchNorm: exactly zero gradient)
    _wgrad1(eng, grads, co, b.dv, C, 0, st.u, Cp, n, hh, ww, code, mode=0, pro=st.all)
    sa = st.all
    sa.bstats.zero_()
    ops.conv2d(b.dv, Cp, eng._pw_any(co, "dgrad", code, mode=1), b.g, C, n=n, h=hh, w=ww, epilogue=L.EPI_DGRAD_MASK, flags=L.FLAG_STATS,
               aux=st.u, aux_scale=sa.scale, aux_shift=sa.shift, aux_mean=sa.mean, aux_invstd=sa.invstd, stats=sa.bstats)
    for i in range(len(st.sizes)):
        b.gamma_all[i * small:(i + 1) * small].copy_(module.convs[i][1].weight.detach())
    eng.bn_coefs(sa.bstats, count, b.gamma_all[:C], sa.mean[:C], sa.invstd[:C], sa.ca[:C], sa.cb[:C], sa.cc[:C], b.dgam_all[:C], b.dbet_all[:C])
    ops.bn_bwd_apply(b.g, st.u, sa.ca, sa.cb, sa.cc, b.dconv, npix, C, code)
    for i, k in enumerate(st.sizes):
        conv, bn = module.convs[i][0], module.convs[i][1]
        sl = slice(i * small, (i + 1) * small)
        dgam, dbet = eng._gbuf(bn.weight), eng._gbuf(bn.bias)
        dgam.copy_(b.dgam_all[sl]), dbet.copy_(b.dbet_all[sl])
        grads[id(bn.weight)], grads[id(bn.bias)] = dgam, dbet
        ops.sum_relu([(b.dconv, i * small)], b.dchunk, npix, small, code, relu=False)        # the chunk's gradient at channel 0 of a padded buffer
        inp = st.up[i] if k > 1 else st.pool[i]
        _wgrad1(eng, grads, conv, b.dchunk, small, 0, inp, sp, n, hh, ww, code, mode=0)
        ops.conv2d(b.dchunk, sp, eng._pw_any(conv, "dgrad", code, mode=1), b.dup, small, n=n, h=hh, w=ww)
        if k > 1:
            ops.bilinear_up_bwd(b.dup, b.dpool[i], n, hh // k, ww // k, hh, ww, small, code)
            dp = b.dpool[i]
        else:
            dp = b.dup
        ops.maxpool_k_bwd(src, dp, dsrc, n, hh, ww, small, k, code, act_coff=src_coff + i * small, dx_coff=dsrc_coff + i * small)
    eng._ready(grads, list(module.parameters()))
