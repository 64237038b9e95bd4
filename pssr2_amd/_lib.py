"""ctypes binding of libpssr_mi355.so (the C ABI declared in include/pssr_mi355.h).

The product path has no CPU fallback: if the shared library is missing or a symbol is absent this
module raises at first use, and every op raises ``RuntimeError`` on a non-zero status.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

_LIB_PATH = Path(__file__).resolve().parent / "libpssr_mi355.so"
_lib = None

F32, BF16, F16 = 0, 1, 2
PRO_NONE, PRO_BN_RELU, PRO_GELU = 0, 1, 2
EPI_STORE, EPI_TAIL, EPI_DGRAD_MASK, EPI_FINAL, EPI_DGRAD_GELU, EPI_HEADQ = 0, 1, 2, 3, 4, 5
FLAG_RELU, FLAG_STATS, FLAG_AFFINE, FLAG_HEADQ, FLAG_SHUF2, FLAG_SOLO = 1, 2, 4, 8, 16, 32

c_void_p, c_int, c_i64, c_float = C.c_void_p, C.c_int, C.c_int64, C.c_float


class ConvDesc(C.Structure):
    """struct pssr_conv_desc (include/pssr_mi355.h)."""
    _fields_ = [
        ("dtype", C.c_int32), ("n", C.c_int32), ("h", C.c_int32), ("w", C.c_int32),
        ("in0", c_void_p), ("in0_cstride", C.c_int32), ("in0_coff", C.c_int32), ("cin0", C.c_int32), ("taps0", C.c_int32),
        ("w0", c_void_p),
        ("in1", c_void_p), ("in1_cstride", C.c_int32), ("in1_coff", C.c_int32), ("cin1", C.c_int32), ("taps1", C.c_int32),
        ("w1", c_void_p),
        ("prologue", C.c_int32), ("pro_scale", c_void_p), ("pro_shift", c_void_p),
        ("out", c_void_p), ("out_cstride", C.c_int32), ("out_coff", C.c_int32), ("cout", C.c_int32), ("n_pad", C.c_int32),
        ("bias", c_void_p),
        ("epilogue", C.c_int32), ("flags", C.c_int32),
        ("aux", c_void_p), ("aux_cstride", C.c_int32), ("aux_coff", C.c_int32),
        ("aux_scale", c_void_p), ("aux_shift", c_void_p), ("aux_mean", c_void_p), ("aux_invstd", c_void_p),
        ("stats", c_void_p),
        ("in0_blk", C.c_int32), ("out_blk", C.c_int32), ("aux_blk", C.c_int32),
        ("out_scale", C.c_float), ("out_shift", C.c_float),
        ("workspace", c_void_p), ("workspace_bytes", C.c_int64),
        ("head_w", c_void_p), ("head_q", c_void_p),
    ]


class WgradDesc(C.Structure):
    """struct pssr_wgrad_desc (include/pssr_mi355.h)."""
    _fields_ = [
        ("dtype", C.c_int32), ("n", C.c_int32), ("h", C.c_int32), ("w", C.c_int32),
        ("dy", c_void_p), ("dy_cstride", C.c_int32), ("dy_coff", C.c_int32), ("dy_blk", C.c_int32), ("cout", C.c_int32),
        ("in_", c_void_p), ("in_cstride", C.c_int32), ("in_coff", C.c_int32), ("in_blk", C.c_int32), ("cin_pad", C.c_int32),
        ("taps", C.c_int32),
        ("prologue", C.c_int32), ("pro_scale", c_void_p), ("pro_shift", c_void_p),
        ("dw", c_void_p), ("dw_parts", C.c_int32),
    ]


class GatherItem(C.Structure):
    """struct pssr_gather_item (include/pssr_mi355.h)"""
    _fields_ = [("src", C.c_void_p), ("sh", C.c_int), ("sw", C.c_int), ("rot", C.c_int), ("flip_axis", C.c_int)]


COPY_BATCH_MAX = 16


class FoldBatch(C.Structure):
    """struct pssr_fold_batch"""
    _fields_ = [("dst", C.c_void_p * 16), ("src", C.c_void_p * 16), ("n", C.c_int32 * 16), ("accumulate", C.c_int32 * 16)]


DWPACK_BATCH_MAX = 48


class DwPackBatch(C.Structure):
    """struct pssr_dwpack_batch"""
    _fields_ = [("w", C.c_void_p * DWPACK_BATCH_MAX), ("packed", C.c_void_p * DWPACK_BATCH_MAX), ("c", C.c_int32 * DWPACK_BATCH_MAX),
                ("flip", C.c_int32 * DWPACK_BATCH_MAX)]


class CopyBatch(C.Structure):
    """struct pssr_copy_batch (include/pssr_mi355.h)."""
    _fields_ = [("dst", C.c_void_p * COPY_BATCH_MAX), ("src", C.c_void_p * COPY_BATCH_MAX), ("n", C.c_int64 * COPY_BATCH_MAX)]


class PackItem(C.Structure):
    """struct pssr_pack_item (include/pssr_mi355.h)."""
    _fields_ = [("w", c_void_p), ("packed", c_void_p), ("n_perm", c_void_p),
                ("cout", C.c_int32), ("cin", C.c_int32), ("ks", C.c_int32), ("ci_begin", C.c_int32), ("ci_count", C.c_int32),
                ("mode", C.c_int32), ("k_pad", C.c_int32), ("n_pad", C.c_int32), ("dtype", C.c_int32), ("center", C.c_int32)]


def lib():
    global _lib
    if _lib is None:
        if not _LIB_PATH.exists():
            raise RuntimeError(
                f"{_LIB_PATH} not found: build the HIP extension first (python -c 'import __graft_entry__ as g; g.build()' "
                "or make -C pssr2_amd/csrc). pssr2_amd has no CPU fallback.")
        import torch  # noqa: F401  (load torch's HIP runtime first: one libamdhip64 per process)
        _lib = C.CDLL(str(_LIB_PATH))
        _lib.pssr_last_error.restype = C.c_char_p
        _lib.pssr_packed_weight_bytes.restype = C.c_int64
        _lib.pssr_conv2d_workspace_bytes.restype = C.c_int64
    return _lib


def check(status: int, what: str):
    if status != 0:
        raise RuntimeError(f"{what} failed ({status}): {lib().pssr_last_error().decode()}")


def ptr(t):
    """Device/host address of a torch tensor (or None)."""
    return None if t is None else c_void_p(t.data_ptr())


def stream_ptr():
    import torch
    return c_void_p(torch.cuda.current_stream().cuda_stream)
