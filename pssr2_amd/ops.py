"""Thin Python wrappers over the C ABI (one function per entry point of include/pssr_mi355.h).

Tensors are only carriers of device memory here: every wrapper passes raw pointers, sizes and the
current HIP stream to libpssr_mi355.so.  Nothing in this module computes with torch.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib as L

TORCH_DTYPE = {L.F32: torch.float32, L.BF16: torch.bfloat16, L.F16: torch.float16}


def dtype_code(dt: torch.dtype) -> int:
    if dt == torch.float32:
        return L.F32
    if dt == torch.bfloat16:
        return L.BF16
    if dt == torch.float16:
        return L.F16
    raise ValueError(f"unsupported compute dtype {dt}; use torch.float32, torch.bfloat16 or torch.float16")


def pad_to(v: int, m: int) -> int:
    return (v + m - 1) // m * m


class PackedWeight:
    """A conv weight in the kernel's K-chunked layout plus the GEMM dims it was packed with."""
    __slots__ = ("data", "taps", "k_pad", "n_pad", "dtype")

    def __init__(self, data, taps, k_pad, n_pad, dtype):
        self.data, self.taps, self.k_pad, self.n_pad, self.dtype = data, taps, k_pad, n_pad, dtype


def pack_conv_weight(w: torch.Tensor, dtype: int, mode: int = 0, ci_begin: int = 0, ci_count: int | None = None,
                     n_perm: torch.Tensor | None = None, out: PackedWeight | None = None) -> PackedWeight:
    """OIHW f32 -> packed (mode 0 forward, 1 dgrad, 2 flat-K im2col, 3 flat-K dgrad)."""
    assert w.is_cuda and w.dtype == torch.float32 and w.is_contiguous() and w.dim() == 4
    cout, cin, ks, _ = w.shape
    ci_count = cin - ci_begin if ci_count is None else ci_count
    s2dk = ks * ks * pad_to(cin, 16)
    gk = ci_count if mode == 0 else cout if mode in (1, 3, 5) else s2dk if mode == 4 else ci_count * ks * ks
    gn = ci_count if mode == 1 else ci_count * ks * ks if mode == 3 else s2dk if mode == 5 else cout
    taps = 1 if mode >= 2 else ks * ks
    k_pad, n_pad = pad_to(gk, 16), pad_to(gn, 128)
    if out is None:
        nbytes = L.lib().pssr_packed_weight_bytes(taps, k_pad, n_pad, dtype)
        out = PackedWeight(torch.empty(nbytes, dtype=torch.uint8, device=w.device), taps, k_pad, n_pad, dtype)
    L.check(L.lib().pssr_pack_conv_weight(L.ptr(w), L.ptr(out.data), cout, cin, ks, ci_begin, ci_count, mode,
                                          L.ptr(n_perm), k_pad, n_pad, dtype, L.stream_ptr()), "pssr_pack_conv_weight")
    return out


# [True] while a forward pass issues its launches (Engine.forward): nothing runs beside them, which conv2d passes on as PSSR_FLAG_SOLO
SOLO = [False]


def conv2d(x, cin0, w0: PackedWeight, out, cout, *, n, h, w, in0_coff=0, out_coff=0, bias=None,
           x1=None, cin1=0, w1: PackedWeight | None = None, in1_coff=0,
           pro_scale=None, pro_shift=None, epilogue=L.EPI_STORE, flags=0,
           aux=None, aux_coff=0, aux_scale=None, aux_shift=None, aux_mean=None, aux_invstd=None, stats=None,
           in0_blk=0, out_blk=0, aux_blk=0, out_scale=1.0, out_shift=0.0, gelu_in=False, head_w=None, head_q=None):
    """x/out/aux: NHWC tensors [n, h, w, cstride] in the compute dtype (channel slices via *_coff).  EPI_HEADQ: ``out`` is unused (pass
    ``head_q``), the tap products go to ``head_q`` [9, 16, n, h, w] f32."""
    d = L.ConvDesc()
    d.dtype = w0.dtype
    d.n, d.h, d.w = n, h, w
    d.in0, d.in0_cstride, d.in0_coff, d.cin0, d.taps0, d.w0 = L.ptr(x), x.shape[-1], in0_coff, cin0, w0.taps, L.ptr(w0.data)
    if w1 is not None:
        d.in1, d.in1_cstride, d.in1_coff, d.cin1, d.taps1, d.w1 = L.ptr(x1), x1.shape[-1], in1_coff, cin1, w1.taps, L.ptr(w1.data)
        assert w1.n_pad == w0.n_pad and w1.dtype == w0.dtype
    d.prologue = L.PRO_GELU if gelu_in else L.PRO_BN_RELU if pro_scale is not None else L.PRO_NONE
    d.pro_scale, d.pro_shift = L.ptr(pro_scale), L.ptr(pro_shift)
    d.out, d.out_cstride, d.out_coff, d.cout, d.n_pad = L.ptr(out), out.shape[-1], out_coff, cout, w0.n_pad
    d.bias = L.ptr(bias)
    d.epilogue, d.flags = epilogue, flags | (L.FLAG_SOLO if SOLO[0] else 0)
    if aux is not None:
        d.aux, d.aux_cstride, d.aux_coff = L.ptr(aux), aux.shape[-1], aux_coff
    d.aux_scale, d.aux_shift, d.aux_mean, d.aux_invstd = L.ptr(aux_scale), L.ptr(aux_shift), L.ptr(aux_mean), L.ptr(aux_invstd)
    d.stats = L.ptr(stats)
    d.in0_blk, d.out_blk, d.aux_blk, d.out_scale, d.out_shift = in0_blk, out_blk, aux_blk, out_scale, out_shift
    if epilogue == L.EPI_FINAL:      # f32 NCHW output [n, cout, h, w]
        d.out_cstride, d.out_coff = cout, 0
    if epilogue == L.EPI_HEADQ:
        d.out_cstride, d.out_coff = cout, 0
    if epilogue == L.EPI_HEADQ or flags & L.FLAG_HEADQ:
        d.head_w, d.head_q = L.ptr(head_w), L.ptr(head_q)
    ws_bytes = L.lib().pssr_conv2d_workspace_bytes(C.byref(d))      # > 0: the library wants to split K (under-filled grid)
    if ws_bytes > 0:
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=out.device)
        d.workspace, d.workspace_bytes = L.ptr(ws), ws_bytes
    L.check(L.lib().pssr_conv2d(C.byref(d), L.stream_ptr()), "pssr_conv2d")
    return out


def _wgrad_desc(dy, cout, x, cin_pad, taps, *, n, h, w, dtype, dy_coff=0, in_coff=0, dy_blk=0, in_blk=0, pro_scale=None, pro_shift=None,
                gelu_in=False):
    d = L.WgradDesc()
    d.dtype, d.n, d.h, d.w = dtype, n, h, w
    d.dy, d.dy_cstride, d.dy_coff, d.dy_blk, d.cout = L.ptr(dy), dy.shape[-1], dy_coff, dy_blk, cout
    d.in_, d.in_cstride, d.in_coff, d.in_blk, d.cin_pad = L.ptr(x), x.shape[-1], in_coff, in_blk, cin_pad
    d.taps = taps
    d.prologue = L.PRO_GELU if gelu_in else L.PRO_BN_RELU if pro_scale is not None else L.PRO_NONE
    d.pro_scale, d.pro_shift = L.ptr(pro_scale), L.ptr(pro_shift)
    return d


def conv2d_wgrad(dy, cout, x, cin_pad, taps, dw, **kw):
    """Atomic mode: dw (f32 [cout, taps, cin_pad], zeroed by the caller) += dy^T (*) prologue(x)."""
    d = _wgrad_desc(dy, cout, x, cin_pad, taps, **kw)
    d.dw, d.dw_parts = L.ptr(dw), 0
    L.check(L.lib().pssr_conv2d_wgrad(C.byref(d), L.stream_ptr()), "pssr_conv2d_wgrad")
    return dw


def conv2d_wgrad_parts(dy, cout, x, cin_pad, taps, **kw):
    """Partial-slab mode: returns dw f32 [parts, cout, taps, cin_pad] (uninitialised workspace filled by the kernel); the
    parts are summed by ``unpack_conv_wgrad``."""
    d = _wgrad_desc(dy, cout, x, cin_pad, taps, **kw)
    parts = L.lib().pssr_conv2d_wgrad_parts(C.byref(d))
    if parts <= 0:
        L.check(parts if parts < 0 else -1, "pssr_conv2d_wgrad_parts")
    dw = torch.empty(parts, cout, taps, cin_pad, dtype=torch.float32, device=dy.device)
    d.dw, d.dw_parts = L.ptr(dw), parts
    L.check(L.lib().pssr_conv2d_wgrad(C.byref(d), L.stream_ptr()), "pssr_conv2d_wgrad")
    return dw


def unpack_conv_wgrad(dw_packed, dw_oihw, *, mode=0, ci_begin=0, ci_count=None, n_perm=None, k_pad, accumulate=False):
    """dw_packed: f32 [rows, taps, k_pad] or [parts, rows, taps, k_pad] (the parts are summed)."""
    cout, cin, ks, _ = dw_oihw.shape
    ci_count = cin - ci_begin if ci_count is None else ci_count
    parts, rows = (dw_packed.shape[0], dw_packed.shape[1]) if dw_packed.dim() == 4 else (1, dw_packed.shape[0])
    L.check(L.lib().pssr_unpack_conv_wgrad_parts(L.ptr(dw_packed), parts, rows, L.ptr(dw_oihw), cout, cin, ks, ci_begin, ci_count, mode,
                                                 L.ptr(n_perm), k_pad, int(accumulate), L.stream_ptr()), "pssr_unpack_conv_wgrad_parts")
    return dw_oihw


# ----------------------------------------------------------------------------------------------
# per-channel / pointwise kernels (csrc/elementwise.hip)
def _ref(t, coff=0):
    return L.ptr(t), t.shape[-1], coff


def channel_stats_nchw(x, stats, pre_scale=1.0, pre_shift=0.0):
    n, c, h, w = x.shape
    L.check(L.lib().pssr_channel_stats_nchw(L.ptr(x), n, c, C.c_int64(h * w), C.c_float(pre_scale), C.c_float(pre_shift),
                                            L.ptr(stats), L.stream_ptr()), "pssr_channel_stats_nchw")


def bn_finalize(stats, count, gamma, beta, eps, momentum, running_mean, running_var, scale, shift, mean, invstd):
    c = scale.numel()
    L.check(L.lib().pssr_bn_finalize(L.ptr(stats), C.c_double(count), L.ptr(gamma), L.ptr(beta), C.c_float(eps), C.c_float(momentum),
                                     L.ptr(running_mean), L.ptr(running_var), L.ptr(scale), L.ptr(shift), L.ptr(mean), L.ptr(invstd),
                                     c, L.stream_ptr()), "pssr_bn_finalize")


def bn_eval_affine(gamma, beta, running_mean, running_var, eps, scale, shift):
    L.check(L.lib().pssr_bn_eval_affine(L.ptr(gamma), L.ptr(beta), L.ptr(running_mean), L.ptr(running_var), C.c_float(eps),
                                        L.ptr(scale), L.ptr(shift), scale.numel(), L.stream_ptr()), "pssr_bn_eval_affine")


def bn_bwd_coefs(stats, count, gamma, mean, invstd, coef_a, coef_b, coef_c, dgamma, dbeta):
    L.check(L.lib().pssr_bn_bwd_coefs(L.ptr(stats), C.c_double(count), L.ptr(gamma), L.ptr(mean), L.ptr(invstd), L.ptr(coef_a),
                                      L.ptr(coef_b), L.ptr(coef_c), L.ptr(dgamma), L.ptr(dbeta), coef_a.numel(), L.stream_ptr()),
            "pssr_bn_bwd_coefs")


def bn_bwd_apply(g, y, coef_a, coef_b, coef_c, dy, npix, c, dtype, g_coff=0, y_coff=0, dy_coff=0):
    L.check(L.lib().pssr_bn_bwd_apply(*_ref(g, g_coff), *_ref(y, y_coff), L.ptr(coef_a), L.ptr(coef_b), L.ptr(coef_c),
                                      *_ref(dy, dy_coff), C.c_int64(npix), c, dtype, L.stream_ptr()), "pssr_bn_bwd_apply")


def bn_relu_apply(y, scale, shift, a, npix, c, dtype, y_coff=0, a_coff=0):
    """a = relu(scale * y + shift) in the storage type: pssr_bn_relu_apply (what a BatchNorm+ReLU prologue would stage, written out)."""
    L.check(L.lib().pssr_bn_relu_apply(L.ptr(y), y.shape[-1], y_coff, L.ptr(scale), L.ptr(shift), L.ptr(a), a.shape[-1], a_coff, C.c_int64(npix), c, dtype,
                                       L.stream_ptr()), "pssr_bn_relu_apply")


def input_im2col(x, xcol, scale, shift, dtype, pre_scale=1 / 128, pre_shift=-1.0):
    n, c, h, w = x.shape
    L.check(L.lib().pssr_input_im2col(L.ptr(x), L.ptr(xcol), n, c, h, w, xcol.shape[-1], C.c_float(pre_scale), C.c_float(pre_shift),
                                      L.ptr(scale), L.ptr(shift), dtype, L.stream_ptr()), "pssr_input_im2col")


def input_norm_bwd(dxcol_a, dxcol_b, x, mean, invstd, stats, dtype, pre_scale=1 / 128, pre_shift=-1.0):
    n, c, h, w = x.shape
    L.check(L.lib().pssr_input_norm_bwd(L.ptr(dxcol_a), L.ptr(dxcol_b), dxcol_a.shape[-1], L.ptr(x), C.c_float(pre_scale),
                                        C.c_float(pre_shift), L.ptr(mean), L.ptr(invstd), n, c, h, w, L.ptr(stats), dtype,
                                        L.stream_ptr()), "pssr_input_norm_bwd")


def maxpool2(x, out, n, h, w, c, dtype, in_coff=0, out_coff=0):
    L.check(L.lib().pssr_maxpool2(*_ref(x, in_coff), *_ref(out, out_coff), n, h, w, c, dtype, L.stream_ptr()), "pssr_maxpool2")


def maxpool2_bwd(act, dpool, dskip, dout, n, h, w, c, dtype, act_coff=0, dskip_coff=0):
    ds = _ref(dskip, dskip_coff) if dskip is not None else (None, 0, 0)
    L.check(L.lib().pssr_maxpool2_bwd(*_ref(act, act_coff), *_ref(dpool), *ds, *_ref(dout), n, h, w, c, dtype, L.stream_ptr()),
            "pssr_maxpool2_bwd")


def pixel_shuffle(lo, hi, n, h, w, c_hi, r, dtype, lo_coff=0, hi_coff=0, inverse=False):
    L.check(L.lib().pssr_pixel_shuffle(*_ref(lo, lo_coff), *_ref(hi, hi_coff), n, h, w, c_hi, r, int(inverse), dtype,
                                       L.stream_ptr()), "pssr_pixel_shuffle")


def relu_bwd_stats(dout, out, y, mean, invstd, dz, stats, npix, c, dtype, out_coff=0):
    L.check(L.lib().pssr_relu_bwd_stats(*_ref(dout), *_ref(out, out_coff), *_ref(y), L.ptr(mean), L.ptr(invstd), *_ref(dz),
                                        L.ptr(stats), C.c_int64(npix), c, dtype, L.stream_ptr()), "pssr_relu_bwd_stats")


def relu_bwd_stats_fused_ok(dtype, c, h=2, w=2, unshuffle=False):
    """The loader-fused forms below take 16-bit storage and power-of-two channel counts (every BatchNorm of the default models)."""
    return dtype != L.F32 and (c & (c - 1)) == 0 and c >= (32 if unshuffle else 8) and (unshuffle or (h % 2 == 0 and w % 2 == 0))


def relu_bwd_stats_pool(dpool, dskip, dskip_coff, out, out_coff, y, mean, invstd, dz, stats, n, h, w, c, dtype):
    """relu_bwd_stats whose d(out) = dskip + max-pool-backward(dpool) is formed in the loader (no maxpool2_bwd launch, no d(out) tensor)."""
    L.check(L.lib().pssr_relu_bwd_stats_pool(*_ref(dpool), *_ref(dskip, dskip_coff), *_ref(out, out_coff), *_ref(y), L.ptr(mean), L.ptr(invstd),
                                             *_ref(dz), L.ptr(stats), n, h, w, c, dtype, L.stream_ptr()), "pssr_relu_bwd_stats_pool")


def relu_bwd_stats_unshuffle(dhi, out, out_coff, y, mean, invstd, dz, stats, n, h, w, c, dtype):
    """relu_bwd_stats whose d(out) is the inverse pixel shuffle (r = 2) of channels [0, c / 4) of ``dhi`` (twice the resolution)."""
    L.check(L.lib().pssr_relu_bwd_stats_unshuffle(*_ref(dhi), *_ref(out, out_coff), *_ref(y), L.ptr(mean), L.ptr(invstd), *_ref(dz), L.ptr(stats),
                                                  n, h, w, c, dtype, L.stream_ptr()), "pssr_relu_bwd_stats_unshuffle")


def channel_sum_nhwc(x, npix, c, out, dtype, coff=0, cstride=None):
    cs = x.shape[-1] if cstride is None else cstride
    L.check(L.lib().pssr_channel_sum_nhwc(L.ptr(x), cs, coff, C.c_int64(npix), c, L.ptr(out), dtype, L.stream_ptr()),
            "pssr_channel_sum_nhwc")


def nchw_to_nhwc(x, out, scale, dtype):
    n, c, h, w = x.shape
    L.check(L.lib().pssr_nchw_to_nhwc(L.ptr(x), L.ptr(out), n, c, C.c_int64(h * w), out.shape[-1], C.c_float(scale), dtype,
                                      L.stream_ptr()), "pssr_nchw_to_nhwc")


def clip_u8(x, out):
    L.check(L.lib().pssr_clip_u8(L.ptr(x), L.ptr(out), C.c_int64(x.numel()), L.stream_ptr()), "pssr_clip_u8")


STAT_STRIPES = 64     # PSSR_STAT_ROWS: rows of every statistic buffer (32 stripes x {multiple-of-2^-20 part, remainder})


def f64_to_f32(src, dst, accumulate=False, stripes=STAT_STRIPES):
    L.check(L.lib().pssr_f64_to_f32(L.ptr(src), L.ptr(dst), dst.numel(), int(accumulate), stripes, L.stream_ptr()), "pssr_f64_to_f32")


def f64_to_f32_batch(items, stripes=STAT_STRIPES):
    """[(src f64 [stripes * n], dst f32 [n], accumulate), ...] folded by one launch per 16 items (pssr_f64_to_f32_batch)."""
    for k in range(0, len(items), L.COPY_BATCH_MAX):
        chunk = items[k:k + L.COPY_BATCH_MAX]
        fb = L.FoldBatch()
        for i, (src, dst, acc) in enumerate(chunk):
            if src.dtype != torch.float64 or dst.dtype != torch.float32 or src.numel() < stripes * dst.numel() or not (src.is_contiguous() and dst.is_contiguous()):
                raise ValueError("f64_to_f32_batch needs contiguous float64 [stripes * n] sources and float32 [n] destinations")
            fb.dst[i], fb.src[i], fb.n[i], fb.accumulate[i] = dst.data_ptr(), src.data_ptr(), dst.numel(), int(acc)
        L.check(L.lib().pssr_f64_to_f32_batch(C.byref(fb), len(chunk), stripes, L.stream_ptr()), "pssr_f64_to_f32_batch")


# ----------------------------------------------------------------------------------------------
# loss + optimizer (csrc/loss.hip, csrc/optim.hip)
def _win(win):
    arr = (C.c_float * len(win))(*[float(v) for v in win])
    return arr, len(win)


def ssim_level_fwd(x, y, planes, h, w, win, c1, c2, sums, l1_sum=None):
    arr, k = _win(win)
    L.check(L.lib().pssr_ssim_level_fwd(L.ptr(x), L.ptr(y), planes, h, w, arr, k, C.c_float(c1), C.c_float(c2), L.ptr(sums),
                                        L.ptr(l1_sum), L.stream_ptr()), "pssr_ssim_level_fwd")


def ssim_level_fwd_adj(x, y, planes, h, w, win, c1, c2, use_ssim, sums, l1_sum, stripes, stripe_stride, adj, in_div=1.0):
    k = len(win); arr = (C.c_float * k)(*win)
    L.check(L.lib().pssr_ssim_level_fwd_adj(L.ptr(x), L.ptr(y), C.c_float(in_div), planes, h, w, arr, k, C.c_float(c1), C.c_float(c2), int(use_ssim), L.ptr(sums),
                                            L.ptr(l1_sum), stripes, C.c_int64(stripe_stride), L.ptr(adj), L.stream_ptr()), "pssr_ssim_level_fwd_adj")


def msssim_weights_striped(sums, stripes, stripe_stride, folded, levels, planes, nvalid, level_weights, ms, mix, l1_sum, l1_numel, grad_out,
                           loss_out, wts, l1_coef):
    L.check(L.lib().pssr_msssim_weights_striped(L.ptr(sums), stripes, C.c_int64(stripe_stride), L.ptr(folded), levels, planes, L.ptr(nvalid),
                                                L.ptr(level_weights), int(ms), C.c_float(mix), L.ptr(l1_sum), C.c_double(l1_numel),
                                                L.ptr(grad_out), L.ptr(loss_out), L.ptr(wts), L.ptr(l1_coef), L.stream_ptr()),
            "pssr_msssim_weights_striped")


def ssim_level_bwd_adj(x, y, adj, planes, h, w, win, wts, dcoarse, hc, wc, l1_coef, dx, in_div=1.0):
    k = len(win); arr = (C.c_float * k)(*win)
    L.check(L.lib().pssr_ssim_level_bwd_adj(L.ptr(x), L.ptr(y), C.c_float(in_div), L.ptr(adj), planes, h, w, arr, k, L.ptr(wts), L.ptr(dcoarse), hc, wc,
                                            L.ptr(l1_coef), L.ptr(dx), L.stream_ptr()), "pssr_ssim_level_bwd_adj")


def avgpool2_planes(x, out, planes, h, w, in_div=1.0):
    L.check(L.lib().pssr_avgpool2_planes_div(L.ptr(x), C.c_float(in_div), L.ptr(out), planes, h, w, L.stream_ptr()), "pssr_avgpool2_planes_div")


def avgpool2_pair(x, y, xo, yo, planes, h, w, in_div=1.0):
    L.check(L.lib().pssr_avgpool2_pair_div(L.ptr(x), L.ptr(y), C.c_float(in_div), L.ptr(xo), L.ptr(yo), planes, h, w, L.stream_ptr()),
            "pssr_avgpool2_pair_div")


def msssim_weights(sums, levels, planes, nvalid, level_weights, ms, mix, l1_sum, l1_numel, grad_out, loss_out, wts, l1_coef):
    L.check(L.lib().pssr_msssim_weights(L.ptr(sums), levels, planes, L.ptr(nvalid), L.ptr(level_weights), int(ms), C.c_float(mix),
                                        L.ptr(l1_sum), C.c_double(l1_numel), L.ptr(grad_out), L.ptr(loss_out), L.ptr(wts),
                                        L.ptr(l1_coef), L.stream_ptr()), "pssr_msssim_weights")


def ssim_level_bwd(x, y, planes, h, w, win, c1, c2, wts, use_ssim, dcoarse, hc, wc, l1_coef, dx):
    arr, k = _win(win)
    L.check(L.lib().pssr_ssim_level_bwd(L.ptr(x), L.ptr(y), planes, h, w, arr, k, C.c_float(c1), C.c_float(c2), L.ptr(wts),
                                        int(use_ssim), L.ptr(dcoarse), hc, wc, L.ptr(l1_coef), L.ptr(dx), L.stream_ptr()),
            "pssr_ssim_level_bwd")


def adamw_step(p, g, m, v, lr, beta1, beta2, eps, weight_decay, step, grad_scale=1.0):
    L.check(L.lib().pssr_adamw_step(L.ptr(p), L.ptr(g), L.ptr(m), L.ptr(v), C.c_int64(p.numel()), C.c_float(lr), C.c_float(beta1),
                                    C.c_float(beta2), C.c_float(eps), C.c_float(weight_decay), C.c_int64(step), C.c_float(grad_scale),
                                    L.stream_ptr()), "pssr_adamw_step")


# ----------------------------------------------------------------------------------------------
# pair generation / crappifiers (csrc/crappify.hip)
ROUND_CLIP, CLIP = 2, 1


def bilinear_down_u8(hr, h, w):
    """uint8 [..., H, W] -> uint8 [..., h, w] (Pillow BILINEAR, bit-exact)."""
    assert hr.dtype == torch.uint8 and hr.is_cuda and hr.is_contiguous()
    H, W = hr.shape[-2:]
    planes = hr.numel() // (H * W)
    tmp = torch.empty(planes * H * w, dtype=torch.uint8, device=hr.device)
    lr = torch.empty(*hr.shape[:-2], h, w, dtype=torch.uint8, device=hr.device)
    L.check(L.lib().pssr_bilinear_down_u8(L.ptr(hr), L.ptr(tmp), L.ptr(lr), planes, H, W, h, w, L.stream_ptr()), "pssr_bilinear_down_u8")
    return lr


def u8_to_f32(x):
    out = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    L.check(L.lib().pssr_u8_to_f32(L.ptr(x), L.ptr(out), C.c_int64(x.numel()), L.stream_ptr()), "pssr_u8_to_f32")
    return out


def crappify_gaussian(x, intensity, gain, spread, seed, tile_offset, flags, noise=None, out=None, tile_counter=None):
    out = torch.empty_like(x) if out is None else out
    tiles = x.shape[0]
    L.check(L.lib().pssr_crappify_gaussian(L.ptr(x), L.ptr(out), tiles, C.c_int64(x.numel() // tiles), C.c_float(intensity), C.c_float(gain),
                                           C.c_float(spread), C.c_uint64(seed), C.c_uint64(tile_offset), L.ptr(noise), flags,
                                           L.ptr(tile_counter), L.stream_ptr()), "pssr_crappify_gaussian")
    return out


def crappify_poisson(x, intensity, gain, spread, seed, tile_offset, flags, out=None, tile_counter=None):
    out = torch.empty_like(x) if out is None else out
    tiles = x.shape[0]
    L.check(L.lib().pssr_crappify_poisson(L.ptr(x), L.ptr(out), tiles, C.c_int64(x.numel() // tiles), C.c_float(intensity), C.c_float(gain),
                                          C.c_float(spread), C.c_uint64(seed), C.c_uint64(tile_offset), flags, L.ptr(tile_counter),
                                          L.stream_ptr()), "pssr_crappify_poisson")
    return out


def crappify_poisson_samples(x, samples, intensity, gain, flags, out=None):
    """Poisson mix with injected samples (f64, same shape as x): pssr_crappify_poisson_samples."""
    out = torch.empty_like(x) if out is None else out
    if samples.dtype != torch.float64 or samples.numel() != x.numel():
        raise ValueError("samples: float64, one per element of x")
    L.check(L.lib().pssr_crappify_poisson_samples(L.ptr(x), L.ptr(samples.contiguous()), L.ptr(out), C.c_int64(x.numel()), C.c_double(intensity),
                                                  C.c_double(gain), flags, L.stream_ptr()), "pssr_crappify_poisson_samples")
    return out


def gaussian_blur(x, sigma, gain, flags):
    h, w = x.shape[-2:]
    planes = x.numel() // (h * w)
    tmp, out = torch.empty_like(x), torch.empty_like(x)
    L.check(L.lib().pssr_gaussian_blur(L.ptr(x), L.ptr(tmp), L.ptr(out), planes, h, w, C.c_float(sigma), C.c_float(gain), flags,
                                       L.stream_ptr()), "pssr_gaussian_blur")
    return out


def counter_add(counter, inc):
    L.check(L.lib().pssr_counter_add(L.ptr(counter), C.c_uint64(inc), L.stream_ptr()), "pssr_counter_add")


def adamw_step_dev(p, g, m, v, state, beta1, beta2, eps, weight_decay, grad_scale=1.0):
    L.check(L.lib().pssr_adamw_step_dev(L.ptr(p), L.ptr(g), L.ptr(m), L.ptr(v), C.c_int64(p.numel()), L.ptr(state), C.c_float(beta1),
                                        C.c_float(beta2), C.c_float(eps), C.c_float(weight_decay), C.c_float(grad_scale),
                                        L.stream_ptr()), "pssr_adamw_step_dev")


def amp_check(g, amp):
    """amp[3] |= any(g is inf / NaN); amp: device int32[4] (see include/pssr_mi355.h)."""
    L.check(L.lib().pssr_amp_check(L.ptr(g), C.c_int64(g.numel()), L.ptr(amp), L.stream_ptr()), "pssr_amp_check")


def adamw_step_amp(p, g, m, v, state, beta1, beta2, eps, weight_decay, amp, growth, backoff, interval, grad_scale=1.0):
    L.check(L.lib().pssr_adamw_step_amp(L.ptr(p), L.ptr(g), L.ptr(m), L.ptr(v), C.c_int64(p.numel()), L.ptr(state), C.c_float(beta1),
                                        C.c_float(beta2), C.c_float(eps), C.c_float(weight_decay), C.c_float(grad_scale), L.ptr(amp),
                                        C.c_float(growth), C.c_float(backoff), int(interval), L.stream_ptr()), "pssr_adamw_step_amp")


# ----------------------------------------------------------------------------------------------
# RDNet encoder kernels (csrc/rdnet.hip)
def input_patchify(x, xpatch, scale, shift, patch, dtype, pre_scale=1 / 128, pre_shift=-1.0):
    n, c, h, w = x.shape
    L.check(L.lib().pssr_input_patchify(L.ptr(x), L.ptr(xpatch), n, c, h, w, patch, xpatch.shape[-1], C.c_float(pre_scale),
                                        C.c_float(pre_shift), L.ptr(scale), L.ptr(shift), dtype, L.stream_ptr()), "pssr_input_patchify")


def input_norm_bwd2(dxcol_a, dxcol_b, dpatch, patch, x, mean, invstd, stats, dtype, pre_scale=1 / 128, pre_shift=-1.0):
    n, c, h, w = x.shape
    col = dxcol_a if dxcol_a is not None else dxcol_b
    L.check(L.lib().pssr_input_norm_bwd2(L.ptr(dxcol_a), L.ptr(dxcol_b), col.shape[-1] if col is not None else 0, L.ptr(dpatch),
                                         dpatch.shape[-1] if dpatch is not None else 0, patch, L.ptr(x), C.c_float(pre_scale),
                                         C.c_float(pre_shift), L.ptr(mean), L.ptr(invstd), n, c, h, w, L.ptr(stats), dtype,
                                         L.stream_ptr()), "pssr_input_norm_bwd2")


def dwconv7_pack_batch(items):
    """[(weight [C,1,7,7] f32, packed [49, C] f32, flip), ...] by one launch per 48 items (pssr_dwconv7_pack_batch)."""
    for k in range(0, len(items), L.DWPACK_BATCH_MAX):
        chunk = items[k:k + L.DWPACK_BATCH_MAX]
        b = L.DwPackBatch()
        for i, (w, out, flip) in enumerate(chunk):
            c = w.shape[0]
            if w.dtype != torch.float32 or out.dtype != torch.float32 or not (w.is_contiguous() and out.is_contiguous()) or w.numel() != 49 * c \
                    or out.numel() < 49 * c:
                raise ValueError("dwconv7_pack_batch needs contiguous float32 [C,1,7,7] weights and [49, C] destinations")
            b.w[i], b.packed[i], b.c[i], b.flip[i] = w.data_ptr(), out.data_ptr(), c, int(flip)
        L.check(L.lib().pssr_dwconv7_pack_batch(C.byref(b), len(chunk), L.stream_ptr()), "pssr_dwconv7_pack_batch")


def dwconv7_pack(w, out, flip=False):
    """torch depthwise weight [C,1,7,7] f32 -> [49][C] f32 (flip: rotated 180 degrees for the input gradient)."""
    c = w.shape[0]
    L.check(L.lib().pssr_dwconv7_pack(L.ptr(w), L.ptr(out), c, int(flip), L.stream_ptr()), "pssr_dwconv7_pack")
    return out


def dwconv7(x, wp, bias, out, n, h, w, c, dtype, in_coff=0, out_coff=0, accumulate=False):
    L.check(L.lib().pssr_dwconv7(*_ref(x, in_coff), L.ptr(wp), L.ptr(bias), *_ref(out, out_coff), n, h, w, c, int(accumulate), dtype,
                                 L.stream_ptr()), "pssr_dwconv7")


def dwconv7_wgrad(dy, x, dw, n, h, w, c, dtype, dy_coff=0, x_coff=0):
    if w % 8 == 0:      # per-workgroup slabs summed in a fixed order (no atomics)
        lib = L.lib()
        lib.pssr_dwconv7_wgrad_workspace_bytes.restype = C.c_int64
        nbytes = lib.pssr_dwconv7_wgrad_workspace_bytes(n, h, w, c)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dw.device)
        L.check(lib.pssr_dwconv7_wgrad_ws(*_ref(dy, dy_coff), *_ref(x, x_coff), L.ptr(dw), n, h, w, c, dtype, L.ptr(ws), C.c_int64(nbytes),
                                          L.stream_ptr()), "pssr_dwconv7_wgrad_ws")
        return
    L.check(L.lib().pssr_dwconv7_wgrad(*_ref(dy, dy_coff), *_ref(x, x_coff), L.ptr(dw), n, h, w, c, dtype, L.stream_ptr()), "pssr_dwconv7_wgrad")


def layernorm2d_fwd(x, gamma, beta, eps, out, n, h, w, c, dtype, in_coff=0, out_coff=0, s2d=False, c_pad=None, mean=None, rstd=None):
    c_pad = pad_to(c, 16) if c_pad is None else c_pad
    L.check(L.lib().pssr_layernorm2d_fwd(*_ref(x, in_coff), L.ptr(gamma), L.ptr(beta), C.c_float(eps), *_ref(out, out_coff), int(s2d), c_pad,
                                         n, h, w, c, L.ptr(mean), L.ptr(rstd), dtype, L.stream_ptr()), "pssr_layernorm2d_fwd")


def layernorm2d_bwd(g, x, gamma, mean, rstd, dx, stats, n, h, w, c, dtype, g_coff=0, x_coff=0, dx_coff=0, s2d=False, c_pad=None,
                    accumulate=False):
    c_pad = pad_to(c, 16) if c_pad is None else c_pad
    L.check(L.lib().pssr_layernorm2d_bwd(*_ref(g, g_coff), int(s2d), c_pad, *_ref(x, x_coff), L.ptr(gamma), L.ptr(mean), L.ptr(rstd),
                                         *_ref(dx, dx_coff), int(accumulate), n, h, w, c, L.ptr(stats), dtype, L.stream_ptr()),
            "pssr_layernorm2d_bwd")


def image_channel_dot(a, b, n, hw, c, scale, out, dtype, a_coff=0, b_coff=0):
    """out[img][c] += scale * sum over the image's pixels of a * b, the same bits on every run (one workgroup per image and 32 channels,
    fixed summation order)."""
    bref = _ref(b, b_coff) if b is not None else (None, 0, 0)
    L.check(L.lib().pssr_image_channel_dot(*_ref(a, a_coff), *bref, n, hw, c, C.c_float(scale), L.ptr(out), dtype, L.stream_ptr()),
            "pssr_image_channel_dot")


def ese_gate(s_mean, w_fc, b_fc, u, gate):
    n, c = s_mean.shape
    L.check(L.lib().pssr_ese_gate(L.ptr(s_mean), L.ptr(w_fc), L.ptr(b_fc), n, c, L.ptr(u), L.ptr(gate), L.stream_ptr()), "pssr_ese_gate")


def scale_nc(t, gate, gamma, add, out, n, hw, c, dtype, t_coff=0, out_coff=0):
    L.check(L.lib().pssr_scale_nc(*_ref(t, t_coff), L.ptr(gate), L.ptr(gamma), L.ptr(add), *_ref(out, out_coff), n, hw, c, dtype,
                                  L.stream_ptr()), "pssr_scale_nc")


def ese_bwd(A, gate, u, gamma, s_mean, w_fc, hw, du, dgamma, db_fc, dw_fc, add):
    n, c = A.shape
    L.check(L.lib().pssr_ese_bwd(L.ptr(A), L.ptr(gate), L.ptr(u), L.ptr(gamma), L.ptr(s_mean), L.ptr(w_fc), n, c, hw, L.ptr(du),
                                 L.ptr(dgamma), L.ptr(db_fc), L.ptr(dw_fc), L.ptr(add), L.stream_ptr()), "pssr_ese_bwd")


# ----------------------------------------------------------------------------------------------
# Reconstruction.conv for C_out <= 3 (csrc/head_conv.hip), bf16 storage
def head_conv_supported(dtype, cin, cout):
    return dtype in (L.BF16, L.F16) and 1 <= cout <= 3 and cin % 32 == 0 and 32 <= cin <= 128


def head_conv_fwd(act, blk, weight, bias, out, n, h, w, cin, cout, out_scale, out_shift, dtype, act_coff=0):
    L.check(L.lib().pssr_head_conv_fwd(L.ptr(act), act.shape[-1], act_coff, blk, L.ptr(weight), L.ptr(bias), L.ptr(out), n, h, w, cin, cout,
                                       C.c_float(out_scale), C.c_float(out_shift), dtype, L.stream_ptr()), "pssr_head_conv_fwd")


def head_conv_dgrad(g, g_scale, weight, act, dact, blk, n, h, w, cin, cout, dtype):
    L.check(L.lib().pssr_head_conv_dgrad(L.ptr(g), C.c_float(g_scale), L.ptr(weight), L.ptr(act), act.shape[-1], 0, L.ptr(dact), dact.shape[-1], 0,
                                         blk, n, h, w, cin, cout, dtype, L.stream_ptr()), "pssr_head_conv_dgrad")


def head_conv_wgrad(g, g_scale, act, blk, dw, n, h, w, cin, cout, dtype):
    L.check(L.lib().pssr_head_conv_wgrad(L.ptr(g), C.c_float(g_scale), L.ptr(act), act.shape[-1], 0, blk, L.ptr(dw), n, h, w, cin, cout, dtype,
                                         L.stream_ptr()), "pssr_head_conv_wgrad")


# ----------------------------------------------------------------------------------------------
# whole-sheet prediction (csrc/tiles.hip)
def sliding_tiles_u8(sheet, size, stride, tile0, ntile):
    """uint8 sheet [C,H,W] on the device -> float32 tiles [ntile, C, size, size] (row-major sliding windows)."""
    c, h, w = sheet.shape
    out = torch.empty(ntile, c, size, size, dtype=torch.float32, device=sheet.device)
    L.check(L.lib().pssr_sliding_tiles_u8(L.ptr(sheet), L.ptr(out), c, h, w, size, stride, tile0, ntile, L.stream_ptr()), "pssr_sliding_tiles_u8")
    return out


def patch_tiles_u8(tiles, n_rows, n_cols, overlap, margin):
    """uint8 tiles [n_rows*n_cols, C, S, S] -> uint8 sheet [C, n_rows*step+overlap, n_cols*step+overlap] (overlap-averaged)."""
    t, c, size, _ = tiles.shape
    assert t == n_rows * n_cols and tiles.dtype == torch.uint8 and tiles.is_contiguous()
    step = size - overlap
    out = torch.empty(c, n_rows * step + overlap, n_cols * step + overlap, dtype=torch.uint8, device=tiles.device)
    L.check(L.lib().pssr_patch_tiles_u8(L.ptr(tiles), L.ptr(out), c, n_rows, n_cols, size, overlap, margin, L.stream_ptr()), "pssr_patch_tiles_u8")
    return out


def head_conv_bwd(g, g_scale, weight, act, dact, blk, dw, bias_sum, n, h, w, cin, cout, dtype):
    """dgrad + wgrad (+ the bias sums of the pixel-shuffle conv in front) in one pass over the activation."""
    L.check(L.lib().pssr_head_conv_bwd(L.ptr(g), C.c_float(g_scale), L.ptr(weight), L.ptr(act), act.shape[-1], 0, L.ptr(dact), dact.shape[-1], 0,
                                       blk, L.ptr(dw), L.ptr(bias_sum), n, h, w, cin, cout, dtype, L.stream_ptr()), "pssr_head_conv_bwd")


def head_q_supported(dtype, h0, cout, r, h, w):
    """The inference form of Reconstruction (EPI_HEADQ + head_q_gather): 16-bit storage, 64 hidden channels, one output channel, 4x."""
    return dtype != L.F32 and h0 == 64 and cout == 1 and r == 4 and h >= 16 and w >= 16 and L.lib().pssr_get_option(b"IGEMM_V3") > 0


def head_q_gather(q, bias, out, n, h, w, out_scale, out_shift):
    L.check(L.lib().pssr_head_q_gather(L.ptr(q), L.ptr(bias), L.ptr(out), n, h, w, C.c_float(out_scale), C.c_float(out_shift), L.stream_ptr()),
            "pssr_head_q_gather")


def head_conv_bwd_rows(g, g_scale, weight, act, dact, blk, dw_rows, bias_rows, n, h, w, cin, cout, dtype):
    """head_conv_bwd with order-independent sums: dw_rows / bias_rows are zeroed f64 [STAT_STRIPES (rows)][...] buffers, folded by f64_to_f32."""
    L.check(L.lib().pssr_head_conv_bwd_rows(L.ptr(g), C.c_float(g_scale), L.ptr(weight), L.ptr(act), act.shape[-1], 0, L.ptr(dact), dact.shape[-1], 0,
                                            blk, L.ptr(dw_rows), L.ptr(bias_rows), n, h, w, cin, cout, dtype, L.stream_ptr()), "pssr_head_conv_bwd_rows")


def crappify_saltpepper(x, amount, gain, spread, seed, tile_offset, flags, out=None, tile_counter=None):
    out = torch.empty_like(x) if out is None else out
    tiles = x.shape[0]
    L.check(L.lib().pssr_crappify_saltpepper(L.ptr(x), L.ptr(out), tiles, C.c_int64(x.numel() // tiles), C.c_float(amount), C.c_float(gain),
                                             C.c_float(spread), C.c_uint64(seed), C.c_uint64(tile_offset), flags, L.ptr(tile_counter),
                                             L.stream_ptr()), "pssr_crappify_saltpepper")
    return out


def gaussian_blur_tiles(x, sigma, spread, gain, seed, tile_offset, flags, tile_counter=None):
    """x: [tiles, frames, h, w] f32; every tile is blurred with its own sigma = max(N(sigma, spread), 0)."""
    tiles, h, w = x.shape[0], x.shape[-2], x.shape[-1]
    tmp, out = torch.empty_like(x), torch.empty_like(x)
    L.check(L.lib().pssr_gaussian_blur_tiles(L.ptr(x), L.ptr(tmp), L.ptr(out), tiles, x.numel() // (tiles * h * w), h, w, C.c_float(sigma),
                                             C.c_float(spread), C.c_float(gain), C.c_uint64(seed), C.c_uint64(tile_offset), flags,
                                             L.ptr(tile_counter), L.stream_ptr()), "pssr_gaussian_blur_tiles")
    return out


def gen_pair_geometry_u8(stacks, rotations, res):
    """_gen_pair's crop / reflect pad / rot90 / flip for a batch of uint8 stacks [c, h, w] resident on the device.
    rotations: per stack ``False`` or ``[rot, axis]`` as the reference draws them (axis 1, 2 or (1, 2))."""
    c = stacks[0].shape[0]
    items = (L.GatherItem * len(stacks))()
    for i, (st, rot) in enumerate(zip(stacks, rotations)):
        if st.dtype != torch.uint8 or not st.is_cuda or st.dim() != 3 or st.shape[0] != c or not st.is_contiguous():
            raise ValueError("gen_pair_geometry_u8 needs contiguous uint8 [c, h, w] stacks on the device")
        axis = -1
        if rot:
            axis = 3 if isinstance(rot[1], (tuple, list)) else int(rot[1])
        items[i].src, items[i].sh, items[i].sw = st.data_ptr(), st.shape[1], st.shape[2]
        items[i].rot, items[i].flip_axis = int(bool(rot and rot[0])), axis
    table = torch.frombuffer(bytearray(bytes(items)), dtype=torch.uint8).to(stacks[0].device)
    out = torch.empty(len(stacks), c, res, res, dtype=torch.uint8, device=stacks[0].device)
    L.check(L.lib().pssr_gen_pair_geometry_u8(L.ptr(table), len(stacks), L.ptr(out), c, res, L.stream_ptr()), "pssr_gen_pair_geometry_u8")
    return out


def normalize_preds_u8(hr, hr_hat, pmin=0.1, pmax=99.9):
    """uint8 device tensors [..., H, W] of equal shape -> (hr_norm, hr_hat_norm) uint8, as pssr.util.normalize_preds (bit-exact)."""
    if hr.dtype != torch.uint8 or hr_hat.dtype != torch.uint8 or hr.shape != hr_hat.shape or not hr.is_cuda:
        raise ValueError("normalize_preds_u8 needs two uint8 device tensors of the same shape")
    hr, hr_hat = hr.contiguous(), hr_hat.contiguous()
    px = hr.shape[-1] * hr.shape[-2]
    n = hr.numel() // px
    lib = L.lib()
    lib.pssr_normalize_preds_workspace_bytes.restype = C.c_int64
    ws = torch.empty(n * lib.pssr_normalize_preds_workspace_bytes(C.c_int64(px)), dtype=torch.uint8, device=hr.device)
    a, b = torch.empty_like(hr), torch.empty_like(hr_hat)
    L.check(lib.pssr_normalize_preds_u8(L.ptr(hr), L.ptr(hr_hat), L.ptr(a), L.ptr(b), n, C.c_int64(px), C.c_float(pmin), C.c_float(pmax), L.ptr(ws),
                                        L.stream_ptr()), "pssr_normalize_preds_u8")
    return a, b


def normalize_preds_resized_u8(hr, hr_hat, pmin=0.1, pmax=99.9):
    """uint8 device tensors [n, H, W] / [n, h, w] of different sizes (pssr/util.py:176-179): pssr_normalize_preds_resized_u8."""
    if hr.dtype != torch.uint8 or hr_hat.dtype != torch.uint8 or hr.dim() != 3 or hr_hat.dim() != 3 or hr.shape[0] != hr_hat.shape[0] or not hr.is_cuda:
        raise ValueError("normalize_preds_resized_u8 needs two uint8 device tensors [n, H, W] and [n, h, w]")
    hr, hr_hat = hr.contiguous(), hr_hat.contiguous()
    n, (H, W), (h, w) = hr.shape[0], hr.shape[1:], hr_hat.shape[1:]
    lib = L.lib()
    lib.pssr_normalize_preds_resized_workspace_bytes.restype = C.c_int64
    ws = torch.empty(lib.pssr_normalize_preds_resized_workspace_bytes(n, H, W, h, w), dtype=torch.uint8, device=hr.device)
    a, b = torch.empty_like(hr), torch.empty_like(hr_hat)
    L.check(lib.pssr_normalize_preds_resized_u8(L.ptr(hr), H, W, L.ptr(hr_hat), h, w, L.ptr(a), L.ptr(b), n, C.c_float(pmin), C.c_float(pmax), L.ptr(ws),
                                                L.stream_ptr()), "pssr_normalize_preds_resized_u8")
    return a, b


def image_metrics_u8(hr, hr_hat):
    """uint8 device tensors [..., H, W] of equal shape -> float64 device tensor [n, 2]: per image the exact sum of squared differences
    and the mean SSIM as skimage.metrics.structural_similarity(data_range=255) defines it (pssr/predict.py:199-203)."""
    if hr.dtype != torch.uint8 or hr_hat.dtype != torch.uint8 or hr.shape != hr_hat.shape or not hr.is_cuda or hr.dim() < 2:
        raise ValueError("image_metrics_u8 needs two uint8 device tensors of the same shape [..., H, W]")
    hr, hr_hat = hr.contiguous(), hr_hat.contiguous()
    h, w = hr.shape[-2:]
    if min(h, w) < 7:
        raise ValueError("win_size exceeds image extent: the 7x7 SSIM window needs images of at least 7x7 pixels")
    n = hr.numel() // (h * w)
    lib = L.lib()
    lib.pssr_image_metrics_workspace_bytes.restype = C.c_int64
    ws = torch.empty(lib.pssr_image_metrics_workspace_bytes(n, h, w), dtype=torch.uint8, device=hr.device)
    out = torch.empty(n, 2, dtype=torch.float64, device=hr.device)
    L.check(lib.pssr_image_metrics_u8(L.ptr(hr), L.ptr(hr_hat), L.ptr(out), n, h, w, L.ptr(ws), L.stream_ptr()), "pssr_image_metrics_u8")
    return out


def copy_f32_batch(pairs):
    """[(dst, src), ...] contiguous float32 device tensors of equal numel: all copied by one kernel launch per 16 pairs."""
    for k in range(0, len(pairs), L.COPY_BATCH_MAX):
        chunk = pairs[k:k + L.COPY_BATCH_MAX]
        items = L.CopyBatch()
        for i, (dst, src) in enumerate(chunk):
            if dst.dtype != torch.float32 or src.dtype != torch.float32 or dst.numel() != src.numel() or not (dst.is_contiguous() and src.is_contiguous()):
                raise ValueError("copy_f32_batch needs contiguous float32 tensors of equal size")
            items.dst[i], items.src[i], items.n[i] = dst.data_ptr(), src.data_ptr(), dst.numel()
        L.check(L.lib().pssr_copy_f32_batch(C.byref(items), len(chunk), L.stream_ptr()), "pssr_copy_f32_batch")


# ------------------------------------------------------------------ atrous / PSP variants (csrc/atrous.hip)
def input_plain(x, out, dtype, pre_scale=1 / 128, pre_shift=-1.0):
    n, c, h, w = x.shape
    L.check(L.lib().pssr_input_plain(L.ptr(x), L.ptr(out), n, c, h, w, out.shape[-1], C.c_float(pre_scale), C.c_float(pre_shift), dtype,
                                     L.stream_ptr()), "pssr_input_plain")


def im2col_dil(x, c, col, cp, n, h, w, dil, dtype, in_coff=0, scale=None, shift=None):
    L.check(L.lib().pssr_im2col_dil(L.ptr(x), x.shape[-1], in_coff, c, L.ptr(scale), L.ptr(shift), L.ptr(col), cp, n, h, w, dil, dtype,
                                    L.stream_ptr()), "pssr_im2col_dil")


def col2im_dil(dcol, cp, out, c, n, h, w, dil, dtype, out_coff=0, y=None, y_coff=0, scale=None, shift=None, mean=None, invstd=None, stats=None):
    L.check(L.lib().pssr_col2im_dil(L.ptr(dcol), cp, L.ptr(out), out.shape[-1], out_coff, c, n, h, w, dil, L.ptr(y),
                                    y.shape[-1] if y is not None else 0, y_coff, L.ptr(scale), L.ptr(shift), L.ptr(mean), L.ptr(invstd),
                                    L.ptr(stats), dtype, L.stream_ptr()), "pssr_col2im_dil")


def channel_stats_nhwc(x, c, npix, stats, dtype, coff=0):
    L.check(L.lib().pssr_channel_stats_nhwc(L.ptr(x), x.shape[-1], coff, c, C.c_int64(npix), L.ptr(stats), dtype, L.stream_ptr()),
            "pssr_channel_stats_nhwc")


def sum_relu(ins, out, npix, c, dtype, relu=True, out_coff=0):
    """ins: list of (tensor, channel offset)."""
    k = len(ins)
    ptrs = (C.c_void_p * k)(*[t.data_ptr() for t, _ in ins])
    css = (C.c_int * k)(*[t.shape[-1] for t, _ in ins])
    cos = (C.c_int * k)(*[o for _, o in ins])
    L.check(L.lib().pssr_sum_relu(ptrs, css, cos, k, L.ptr(out), out.shape[-1], out_coff, C.c_int64(npix), c, int(relu), dtype, L.stream_ptr()),
            "pssr_sum_relu")


def relu_mask(dout, out, dz, npix, c, dtype, do_coff=0, o_coff=0, dz_coff=0):
    L.check(L.lib().pssr_relu_mask(L.ptr(dout), dout.shape[-1], do_coff, L.ptr(out), out.shape[-1], o_coff, L.ptr(dz), dz.shape[-1], dz_coff,
                                   C.c_int64(npix), c, dtype, L.stream_ptr()), "pssr_relu_mask")


def affine_relu(x, scale, shift, out, npix, c, dtype, coff=0, out_coff=0):
    L.check(L.lib().pssr_affine_relu(L.ptr(x), x.shape[-1], coff, L.ptr(scale), L.ptr(shift), L.ptr(out), out.shape[-1], out_coff, C.c_int64(npix), c,
                                     dtype, L.stream_ptr()), "pssr_affine_relu")


def maxpool_k(x, out, n, h, w, c, k, dtype, in_coff=0, out_coff=0):
    L.check(L.lib().pssr_maxpool_k(L.ptr(x), x.shape[-1], in_coff, L.ptr(out), out.shape[-1], out_coff, n, h, w, c, k, dtype, L.stream_ptr()),
            "pssr_maxpool_k")


def maxpool_k_bwd(act, dpool, dx, n, h, w, c, k, dtype, act_coff=0, dp_coff=0, dx_coff=0):
    L.check(L.lib().pssr_maxpool_k_bwd(L.ptr(act), act.shape[-1], act_coff, L.ptr(dpool), dpool.shape[-1], dp_coff, L.ptr(dx), dx.shape[-1], dx_coff,
                                       n, h, w, c, k, dtype, L.stream_ptr()), "pssr_maxpool_k_bwd")


def bilinear_up(x, out, n, hs, ws, h, w, c, dtype, in_coff=0, out_coff=0):
    L.check(L.lib().pssr_bilinear_up(L.ptr(x), x.shape[-1], in_coff, L.ptr(out), out.shape[-1], out_coff, n, hs, ws, h, w, c, dtype, L.stream_ptr()),
            "pssr_bilinear_up")


def bilinear_up_bwd(dout, din, n, hs, ws, h, w, c, dtype, do_coff=0, di_coff=0):
    L.check(L.lib().pssr_bilinear_up_bwd(L.ptr(dout), dout.shape[-1], do_coff, L.ptr(din), din.shape[-1], di_coff, n, hs, ws, h, w, c, dtype,
                                         L.stream_ptr()), "pssr_bilinear_up_bwd")
