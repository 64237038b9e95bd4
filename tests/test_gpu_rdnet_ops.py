"""RDNet kernels (csrc/rdnet.hip + GELU / space-to-depth paths of the conv kernels), through the C ABI,
vs torch-CPU fp32 references of the same ops (F.conv2d groups=C, F.layer_norm, F.gelu, the 5-line ESE)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DTS = [torch.float32, torch.bfloat16]


def _nhwc(x, cs, dt, coff=0, fill=0.0):
    n, c, h, w = x.shape
    out = torch.full((n, h, w, cs), fill, dtype=dt, device="cuda")
    out[..., coff:coff + c] = x.permute(0, 2, 3, 1).to("cuda").to(dt)
    return out


def _nchw(t, coff, c):
    return t[..., coff:coff + c].float().cpu().permute(0, 3, 1, 2)


def _close(got, ref, dt, k=1, scale=None):
    scale = float(ref.abs().max()) if scale is None else scale
    if dt == torch.float32:
        np.testing.assert_allclose(got.numpy(), ref.numpy(), rtol=2e-5 * max(1, k) ** 0.5, atol=2e-6 * max(1, k) ** 0.5 * scale + 1e-7)
    else:
        np.testing.assert_allclose(got.numpy(), ref.numpy(), rtol=3e-2, atol=2e-2 * scale + 1e-6)


def test_patchify():
    from pssr2_amd import ops, _lib as L
    x = torch.rand(3, 3, 8, 12) * 255
    sc, sh = torch.tensor([1.1, 0.9, 1.3]), torch.tensor([0.1, -0.2, 0.05])
    xp = torch.full((3, 4, 6, 16), 9.0, device="cuda")
    ops.input_patchify(x.cuda(), xp, sc.cuda(), sh.cuda(), 2, L.F32)
    xn = (x / 128 - 1) * sc.view(1, 3, 1, 1) + sh.view(1, 3, 1, 1)
    ref = F.unfold(xn, 2, stride=2).view(3, 12, 4, 6).permute(0, 2, 3, 1)      # channel = ci*4 + dy*2 + dx
    np.testing.assert_allclose(xp[..., :12].cpu().numpy(), ref.numpy(), rtol=1e-6, atol=1e-6)
    assert (xp[..., 12:] == 0).all()


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("shape", [(2, 24, 9, 13), (1, 264, 16, 16), (3, 8, 4, 4), (2, 40, 6, 24), (1, 1040, 8, 8), (4, 136, 32, 32)])
def test_dwconv7(shape, dt):
    from pssr2_amd import ops
    n, c, h, w = shape
    code = ops.dtype_code(dt)
    g = torch.Generator().manual_seed(c)
    x = torch.randn(n, c, h, w, generator=g).to(dt).float().requires_grad_(True)
    wt = (torch.randn(c, 1, 7, 7, generator=g) / 7).requires_grad_(True)
    b = torch.randn(c, generator=g)
    ref = F.conv2d(x, wt, b, padding=3, groups=c)
    dy = torch.randn(ref.shape, generator=g).to(dt).float()
    ref.backward(dy)
    xd = _nhwc(x.detach(), c + 24, dt, coff=8, fill=3.0)
    wp = ops.dwconv7_pack(wt.detach().cuda().contiguous(), torch.empty(49, c, device="cuda"))
    out = torch.full((n, h, w, c + 8), -1.0, dtype=dt, device="cuda")
    ops.dwconv7(xd, wp, b.cuda(), out, n, h, w, c, code, in_coff=8, out_coff=4)
    _close(_nchw(out, 4, c), ref.detach(), dt, 49)
    assert (out[..., :4] == -1).all() and (out[..., 4 + c:] == -1).all()
    # input gradient: flipped weights, accumulated into an existing gradient
    wf = ops.dwconv7_pack(wt.detach().cuda().contiguous(), torch.empty(49, c, device="cuda"), flip=True)
    dyd = _nhwc(dy, c, dt)
    base = torch.randn(n, c, h, w, generator=g).to(dt).float()
    dx = _nhwc(base, c + 16, dt, coff=16)
    ops.dwconv7(dyd, wf, None, dx, n, h, w, c, code, out_coff=16, accumulate=True)
    _close(_nchw(dx, 16, c), x.grad + base, dt, 49)
    # weight gradient
    dw = torch.zeros(c, 49, device="cuda")
    ops.dwconv7_wgrad(dyd, xd, dw, n, h, w, c, code, x_coff=8)
    _close(dw.cpu().view(c, 1, 7, 7), wt.grad, dt, n * h * w)


def test_dwconv7_wgrad_slabs_accumulate_and_are_reproducible():
    """w % 8 == 0: per-workgroup slabs + fixed-order sum (no atomics) -- dw += ..., identical bits on every run; the workspace size is checked."""
    from pssr2_amd import ops, _lib as L
    import ctypes as C
    n, c, h, w = 8, 200, 16, 24
    g = torch.Generator().manual_seed(3)
    x = torch.randn(n, c, h, w, generator=g)
    dy = torch.randn(n, c, h, w, generator=g)
    xd, dyd = _nhwc(x, c + 8, torch.float32, coff=4), _nhwc(dy, c, torch.float32)
    code = ops.dtype_code(torch.float32)
    runs = []
    for _ in range(2):
        dw = torch.full((c, 49), 0.5, device="cuda")
        ops.dwconv7_wgrad(dyd, xd, dw, n, h, w, c, code, x_coff=4)
        runs.append(dw.cpu())
    assert torch.equal(runs[0], runs[1])
    wt = torch.zeros(c, 1, 7, 7, requires_grad=True)
    F.conv2d(x, wt, padding=3, groups=c).backward(dy)
    ref = wt.grad.view(c, 49)
    torch.testing.assert_close(runs[0] - 0.5, ref, rtol=2e-4, atol=2e-3)
    lib = L.lib()
    lib.pssr_dwconv7_wgrad_workspace_bytes.restype = C.c_int64
    need = lib.pssr_dwconv7_wgrad_workspace_bytes(n, h, w, c)
    assert need > 0 and lib.pssr_dwconv7_wgrad_workspace_bytes(n, h, 9, c) == 0
    ws = torch.empty(need, dtype=torch.uint8, device="cuda")
    rc = lib.pssr_dwconv7_wgrad_ws(L.ptr(dyd), c, 0, L.ptr(xd), c + 8, 4, L.ptr(dw), n, h, w, c, code, L.ptr(ws), C.c_int64(need - 1), L.stream_ptr())
    assert rc != 0                                           # too small a workspace is refused, not overrun


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("shape,s2d", [((2, 24, 6, 10), False), ((1, 1040, 4, 4), False), ((2, 40, 8, 4), True), ((1, 472, 4, 6), True)])
def test_layernorm2d(shape, s2d, dt):
    from pssr2_amd import ops
    n, c, h, w = shape
    code = ops.dtype_code(dt)
    g = torch.Generator().manual_seed(c + h)
    x = (torch.randn(n, c, h, w, generator=g) * 2 + 0.5).to(dt).float().requires_grad_(True)
    gam = (0.5 + torch.rand(c, generator=g)).requires_grad_(True)
    bet = (torch.rand(c, generator=g) - 0.5).requires_grad_(True)
    ref = F.layer_norm(x.permute(0, 2, 3, 1), (c,), gam, bet, 1e-6).permute(0, 3, 1, 2)
    dy = torch.randn(ref.shape, generator=g).to(dt).float()
    ref.backward(dy)
    cpad = ops.pad_to(c, 16)
    xd = _nhwc(x.detach(), c + 8, dt, coff=8, fill=5.0)
    npix = n * h * w
    mean, rstd = torch.empty(npix, device="cuda"), torch.empty(npix, device="cuda")
    if s2d:
        out = torch.full((n, h // 2, w // 2, 4 * cpad), -1.0, dtype=dt, device="cuda")
    else:
        out = torch.full((n, h, w, cpad), -1.0, dtype=dt, device="cuda")
    ops.layernorm2d_fwd(xd, gam.detach().cuda(), bet.detach().cuda(), 1e-6, out, n, h, w, c, code, in_coff=8, s2d=s2d, mean=mean, rstd=rstd)

    def unfold(t):      # s2d layout -> NCHW
        if not s2d:
            return t[..., :c].float().cpu().permute(0, 3, 1, 2)
        v = t.float().cpu().view(n, h // 2, w // 2, 2, 2, cpad)[..., :c]      # n, oy, ox, dy, dx, c
        return v.permute(0, 5, 1, 3, 2, 4).reshape(n, c, h, w)

    def fold(v):        # NCHW -> s2d / plain NHWC device tensor with zero pad
        if not s2d:
            return _nhwc(v, cpad, dt)
        t = torch.zeros(n, h // 2, w // 2, 2, 2, cpad)
        t[..., :c] = v.view(n, c, h // 2, 2, w // 2, 2).permute(0, 2, 4, 3, 5, 1)
        return t.view(n, h // 2, w // 2, 4 * cpad).to(dt).cuda()

    _close(unfold(out), ref.detach(), dt, 1)
    pads = out.view(-1, cpad)[:, c:]
    assert (pads == 0).all()
    mu = x.detach().mean(1).reshape(-1)
    np.testing.assert_allclose(mean.cpu().numpy(), mu.numpy(), rtol=1e-5, atol=1e-5)
    # backward
    from pssr2_amd.ops import STAT_STRIPES
    stats = torch.zeros(STAT_STRIPES * 2 * c, dtype=torch.float64, device="cuda")
    base = torch.randn(n, c, h, w, generator=g).to(dt).float()
    dx = _nhwc(base, c + 4, dt, coff=4)
    ops.layernorm2d_bwd(fold(dy), xd, gam.detach().cuda(), mean, rstd, dx, stats, n, h, w, c, code, x_coff=8, dx_coff=4, s2d=s2d, accumulate=True)
    _close(_nchw(dx, 4, c), x.grad + base, dt, 1)
    gb = torch.empty(2 * c, device="cuda")
    ops.f64_to_f32(stats, gb)
    _close(gb[:c].cpu(), gam.grad, dt, npix)
    _close(gb[c:].cpu(), bet.grad, dt, npix)


@pytest.mark.parametrize("dt", DTS)
def test_transition_s2d_conv(dt):
    """LayerNorm2d (s2d layout) -> 2x2 stride-2 conv as a 1x1 conv with mode-4 weights; dgrad (mode 5); wgrad (unpack mode 4)."""
    from pssr2_amd import ops
    n, c, h, w, co = 2, 40, 8, 12, 24
    code = ops.dtype_code(dt)
    g = torch.Generator().manual_seed(4)
    a = torch.randn(n, c, h, w, generator=g).to(dt).float().requires_grad_(True)     # stands for the LN output
    wt = (torch.randn(co, c, 2, 2, generator=g) / (4 * c) ** 0.5).requires_grad_(True)
    b = torch.randn(co, generator=g)
    wq = wt.detach().to(dt).float().requires_grad_(True)
    ref = F.conv2d(a, wq, b, stride=2)
    dy = torch.randn(ref.shape, generator=g).to(dt).float()
    ref.backward(dy)
    cpad = ops.pad_to(c, 16)
    t = torch.zeros(n, h // 2, w // 2, 2, 2, cpad)
    t[..., :c] = a.detach().view(n, c, h // 2, 2, w // 2, 2).permute(0, 2, 4, 3, 5, 1)
    ad = t.view(n, h // 2, w // 2, 4 * cpad).to(dt).cuda()
    pw = ops.pack_conv_weight(wt.detach().cuda().contiguous(), code, mode=4)
    assert pw.k_pad == 4 * cpad
    out = torch.zeros(n, h // 2, w // 2, ops.pad_to(co, 16), dtype=dt, device="cuda")
    ops.conv2d(ad, 4 * cpad, pw, out, co, n=n, h=h // 2, w=w // 2, bias=b.cuda())
    _close(_nchw(out, 0, co), ref.detach(), dt, 4 * c)
    pd = ops.pack_conv_weight(wt.detach().cuda().contiguous(), code, mode=5)
    dyd = _nhwc(dy, ops.pad_to(co, 16), dt)
    da = torch.zeros_like(ad)
    ops.conv2d(dyd, ops.pad_to(co, 16), pd, da, 4 * cpad, n=n, h=h // 2, w=w // 2)
    got = da.float().cpu().view(n, h // 2, w // 2, 2, 2, cpad)[..., :c].permute(0, 5, 1, 3, 2, 4).reshape(n, c, h, w)
    _close(got, a.grad, dt, co)
    dwp = torch.zeros(ops.pad_to(co, 16) if dt == torch.bfloat16 else co, 1, 4 * cpad, device="cuda")
    ops.conv2d_wgrad(dyd, dwp.shape[0], ad, 4 * cpad, 1, dwp, n=n, h=h // 2, w=w // 2, dtype=code)
    dw = torch.zeros(co, c, 2, 2, device="cuda")
    ops.unpack_conv_wgrad(dwp, dw, mode=4, k_pad=4 * cpad)
    _close(dw.cpu(), wq.grad, dt, n * h * w // 4)


@pytest.mark.parametrize("dt", DTS)
def test_gelu_fused_1x1(dt):
    """z -> gelu -> 1x1 conv (PRO_GELU), its dgrad with the GELU derivative (EPI_DGRAD_GELU + bias-gradient sums), and
    the weight gradient with the GELU prologue."""
    from pssr2_amd import ops, _lib as L
    n, ci, co, h, w = 2, 96, 24, 6, 10
    code = ops.dtype_code(dt)
    g = torch.Generator().manual_seed(8)
    z = (torch.randn(n, ci, h, w, generator=g) * 1.5).to(dt).float().requires_grad_(True)
    wt = (torch.randn(co, ci, 1, 1, generator=g) / ci ** 0.5)
    wq = wt.to(dt).float().requires_grad_(True)
    b = torch.randn(co, generator=g)
    aq = F.gelu(z).to(dt).float() if dt == torch.bfloat16 else F.gelu(z)
    ref = F.conv2d(F.gelu(z), wq, b)
    dy = torch.randn(ref.shape, generator=g).to(dt).float()
    ref.backward(dy)
    zd = _nhwc(z.detach(), ci, dt)
    pw = ops.pack_conv_weight(wt.cuda().contiguous(), code, mode=0)
    cop = ops.pad_to(co, 16)
    out = torch.zeros(n, h, w, cop, dtype=dt, device="cuda")
    ops.conv2d(zd, ci, pw, out, co, n=n, h=h, w=w, bias=b.cuda(), gelu_in=True)
    _close(_nchw(out, 0, co), ref.detach(), dt, ci)
    # dgrad * gelu'(z), statistics = per-channel sum of the result (bias gradient of the previous 1x1 conv)
    pd = ops.pack_conv_weight(wt.cuda().contiguous(), code, mode=1)
    dyd = _nhwc(dy, cop, dt)
    dz = torch.zeros(n, h, w, ci, dtype=dt, device="cuda")
    stats = torch.zeros(ops.STAT_STRIPES * 2 * ci, dtype=torch.float64, device="cuda")
    ops.conv2d(dyd, cop, pd, dz, ci, n=n, h=h, w=w, epilogue=L.EPI_DGRAD_GELU, flags=L.FLAG_STATS, aux=zd, stats=stats)
    _close(_nchw(dz, 0, ci), z.grad, dt, co)
    sums = torch.empty(2 * ci, device="cuda")
    ops.f64_to_f32(stats, sums)
    _close(sums[:ci].cpu(), dz.float().sum((0, 1, 2)).cpu(), torch.float32, n * h * w)
    # wgrad with the GELU prologue
    rows = cop if dt == torch.bfloat16 else co
    dwp = torch.zeros(rows, 1, ci, device="cuda")
    ops.conv2d_wgrad(dyd, rows, zd, ci, 1, dwp, n=n, h=h, w=w, dtype=code, gelu_in=True)
    dw = torch.zeros(co, ci, 1, 1, device="cuda")
    ops.unpack_conv_wgrad(dwp, dw, mode=0, k_pad=ci)
    _close(dw.cpu(), wq.grad, dt, n * h * w)
    del aq


@pytest.mark.parametrize("dt", DTS)
def test_gelu_propagates_nonfinite(dt):
    """A NaN / inf in z must reach the consumers of gelu(z) and gelu'(z) (the polynomial GELU of the 16-bit builds is built from min / max,
    which return their non-NaN operand): the output pixel of the fused 1x1 conv and the input gradient at that element are non-finite,
    every other pixel stays finite."""
    from pssr2_amd import ops, _lib as L
    n, ci, co, h, w = 1, 32, 16, 4, 8
    code = ops.dtype_code(dt)
    g = torch.Generator().manual_seed(3)
    z = torch.randn(n, ci, h, w, generator=g)
    z[0, 5, 1, 2] = float("nan")
    z[0, 7, 2, 3] = float("inf")
    wt = torch.randn(co, ci, 1, 1, generator=g) / ci ** 0.5
    zd = _nhwc(z, ci, dt)
    out = torch.zeros(n, h, w, co, dtype=dt, device="cuda")
    ops.conv2d(zd, ci, ops.pack_conv_weight(wt.cuda().contiguous(), code, mode=0), out, co, n=n, h=h, w=w, gelu_in=True)
    fin = torch.isfinite(out.float()).all(-1)[0].cpu()
    bad = torch.zeros(h, w, dtype=torch.bool)
    bad[1, 2] = bad[2, 3] = True
    assert torch.equal(~fin, bad), fin
    dy = _nhwc(torch.randn(n, co, h, w, generator=g), co, dt)
    dz = torch.zeros(n, h, w, ci, dtype=dt, device="cuda")
    ops.conv2d(dy, co, ops.pack_conv_weight(wt.cuda().contiguous(), code, mode=1), dz, ci, n=n, h=h, w=w, epilogue=L.EPI_DGRAD_GELU, aux=zd)
    dzf = torch.isfinite(dz.float())[0].cpu()
    assert not dzf[1, 2, 5] and not dzf[2, 3, 7]
    dzf[1, 2, 5] = dzf[2, 3, 7] = True
    assert dzf.all()


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("ese", [True, False])
@pytest.mark.parametrize("c", [24, 328])
def test_ese_layerscale(ese, dt, c):
    from pssr2_amd import ops
    n, h, w = 3, 5, 7
    code = ops.dtype_code(dt)
    g = torch.Generator().manual_seed(2)
    t = torch.randn(n, c, h, w, generator=g).to(dt).float().requires_grad_(True)
    wfc = (torch.randn(c, c, 1, 1, generator=g) * 2).requires_grad_(True)
    bfc = (torch.randn(c, generator=g) * 2).requires_grad_(True)
    gam = (0.5 + torch.rand(c, generator=g)).requires_grad_(True)
    if ese:
        s = t.mean((2, 3), keepdim=True)
        u = F.conv2d(s, wfc, bfc)
        ref = t * (F.relu6(u + 3) / 6) * gam.view(1, -1, 1, 1)
    else:
        ref = t * gam.view(1, -1, 1, 1)
    dout = torch.randn(ref.shape, generator=g).to(dt).float()
    ref.backward(dout)
    hw = h * w
    td = _nhwc(t.detach(), ops.pad_to(c, 16), dt)
    dev = dict(device="cuda", dtype=torch.float32)
    sm, uu, gate = torch.zeros(n, c, **dev), torch.empty(n, c, **dev), torch.empty(n, c, **dev)
    gd = gam.detach().cuda()
    out = torch.zeros(n, h, w, c + 8, dtype=dt, device="cuda")
    if ese:
        ops.image_channel_dot(td, None, n, hw, c, 1.0 / hw, sm, code)
        ops.ese_gate(sm, wfc.detach().cuda().view(c, c).contiguous(), bfc.detach().cuda(), uu, gate)
    ops.scale_nc(td, gate if ese else None, gd, None, out, n, hw, c, code, out_coff=8)
    _close(_nchw(out, 8, c), ref.detach(), dt, 1)
    # backward
    dd = _nhwc(dout, c + 8, dt, coff=8)
    A = torch.zeros(n, c, **dev)
    ops.image_channel_dot(dd, td, n, hw, c, 1.0, A, code, a_coff=8)
    dgam = torch.empty(c, **dev)
    dt_buf = torch.zeros(n, h, w, ops.pad_to(c, 16), dtype=dt, device="cuda")
    if ese:
        du, dbfc, dwfc, add = torch.empty(n, c, **dev), torch.empty(c, **dev), torch.empty(c, c, **dev), torch.empty(n, c, **dev)
        ops.ese_bwd(A, gate, uu, gd, sm, wfc.detach().cuda().view(c, c).contiguous(), hw, du, dgam, dbfc, dwfc, add)
        ops.scale_nc(dd, gate, gd, add, dt_buf, n, hw, c, code, t_coff=8)
        _close(dbfc.cpu(), bfc.grad, dt, hw)
        _close(dwfc.cpu().view(c, c, 1, 1), wfc.grad, dt, hw)
    else:
        ops.ese_bwd(A, None, None, gd, None, None, hw, None, dgam, None, None, None)
        ops.scale_nc(dd, None, gd, None, dt_buf, n, hw, c, code, t_coff=8)
    _close(dgam.cpu(), gam.grad, dt, n * hw)
    _close(_nchw(dt_buf, 0, c), t.grad, dt, 1)


def test_dwconv7_pack_batch_equals_single_packs():
    """pssr_dwconv7_pack_batch (every block's forward and rotated copy by one launch) writes what pssr_dwconv7_pack writes per weight;
    more items than one launch holds are split."""
    from pssr2_amd import ops
    g = torch.Generator().manual_seed(11)
    ws = [torch.randn(c, 1, 7, 7, generator=g).cuda() for c in (8, 64, 104, 816)] * 13          # 52 items > 48 per launch
    items, want = [], []
    for i, w in enumerate(ws):
        flip = bool(i & 1)
        items.append((w, torch.empty(49, w.shape[0], device="cuda"), flip))
        want.append(ops.dwconv7_pack(w, torch.empty(49, w.shape[0], device="cuda"), flip=flip))
    ops.dwconv7_pack_batch(items)
    torch.cuda.synchronize()
    for (w, got, flip), ref in zip(items, want):
        assert torch.equal(got, ref), (w.shape, flip)
