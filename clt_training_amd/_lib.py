"""ctypes binding of libcltmi355.so (the C ABI declared in include/clt_mi355.h).

There is NO CPU fallback: if the shared library is missing or fails to load,
importing an op raises.  Build it with ``python -m clt_training_amd.build`` (or
``__graft_entry__.build()``)."""
from __future__ import annotations

import ctypes as C
from pathlib import Path

HERE = Path(__file__).resolve().parent
import os

LIB_PATH = Path(os.environ.get("CLT_LIB", HERE / "libcltmi355.so"))  # (CLT_LIB: A/B builds while tuning)

CLT_F32, CLT_BF16 = 0, 1
CLT_DECODE_ACCUMULATE, CLT_DECODE_ROUND_ACTS = 1, 2

_vp, _i64, _i32, _f32, _sz = C.c_void_p, C.c_int64, C.c_int, C.c_float, C.c_size_t


class AdamTensor(C.Structure):
    """clt_adam_tensor (include/clt_mi355.h)."""

    _fields_ = [("p", _vp), ("g", _vp), ("m", _vp), ("v", _vp), ("numel", _i64), ("lr", _f32),
                ("p_dtype", C.c_int32), ("g_dtype", C.c_int32), ("pad_", C.c_int32)]


class DecodeSource(C.Structure):
    """clt_decode_source (include/clt_mi355.h)."""

    _fields_ = [("idx", _vp), ("v", _vp), ("post_enc", _vp), ("post_scale", _vp), ("W", _vp), ("acts_out", _vp),
                ("grad_acts", _vp), ("e_post", _vp), ("e_scale", _vp), ("lat_begin", _i64), ("weight", _f32),
                ("accumulate_grad_acts", C.c_int32)]


CLT_DECODE_MAX_SOURCES = 16


class RowsJob(C.Structure):
    """clt_rows_job (include/clt_mi355.h)."""

    _fields_ = [("perm", _vp), ("offsets", _vp), ("acts", _vp), ("v", _vp), ("post_enc", _vp), ("post_scale", _vp),
                ("W", _vp), ("exp_avg", _vp), ("exp_avg_sq", _vp), ("grad_acts", _vp), ("g_post", _vp), ("g_scale", _vp)]


CLT_ROWS_MAX_JOBS = 16


class SptJob(C.Structure):
    """clt_spt_job (include/clt_mi355.h)."""

    _fields_ = [("perm", _vp), ("offsets", _vp), ("vals", _vp), ("vals_gate", _vp), ("dense", _vp), ("p", _vp), ("m", _vp),
                ("v", _vp), ("rowsum", _vp)]

# name -> (restype, argtypes); kept in the order of include/clt_mi355.h
SIGNATURES = {
    "clt_version": (_i32, []),
    "clt_last_error": (C.c_char_p, []),
    "clt_encoder_gemm_relu": (_i32, [_vp, _vp, _i32, _vp, _i32, _vp, _i32, _i64, _i64, _i32, _vp]),
    "clt_topk_rows": (_i32, [_vp, _i32, _i64, _vp, _vp, _i64, _i64, _i32, _vp, _i32, _vp, _vp]),
    "clt_candidates_pack": (_i32, [_vp, _i32, _vp, _i64, _i64, _vp, _vp]),
    "clt_candidates_unpack": (_i32, [_vp, _i64, _i64, _i64, _vp, _i32, _vp, _vp]),
    "clt_groupmax_rows": (_i32, [_vp, _i32, _i64, _i64, _i64, _i32, _vp, _i32, _vp, _vp]),
    "clt_encode_groupmax_fwd": (_i32, [_vp, _vp, _i32, _vp, _i32, _i64, _i64, _i32, _i32, _vp, _i32, _vp, _vp, _sz, _vp]),
    "clt_encode_topk_workspace_bytes": (_sz, [_i64, _i64, _i32]),
    "clt_encode_topk_fwd": (_i32, [_vp, _vp, _i32, _vp, _i32, _i64, _i64, _i32, _i32, _vp, _i32, _vp, _vp, _sz, _vp]),
    "clt_encode_topk_fwd_layers": (_i32, [_vp, _i64, _i64, _vp, _i32, _vp, _i32, _i32, _i64, _i64, _i32, _i32, _vp, _i32,
                                          _vp, _vp, _sz, _vp]),
    "clt_decode_fwd": (_i32, [_vp, _vp, _i32, _vp, _vp, _vp, _i32, _i64, _i64, _i32, _vp, _vp, _i32, _i32, _vp, _i64, _i32,
                              _vp]),
    "clt_decode_bwd_acts": (_i32, [_vp, _vp, _i32, _vp, _vp, _vp, _i32, _i64, _i64, _i32, _vp, _i32, _vp, _i32, _vp, _vp,
                                   _i64, _i32, _vp]),
    "clt_decode_fwd_multi": (_i32, [_vp, _i32, _i32, _i32, _i64, _i32, _vp, _vp, _i32, _i32, _i64, _i32, _vp]),
    "clt_decode_bwd_acts_multi": (_i32, [_vp, _i32, _i32, _i32, _i64, _i32, _vp, _i32, _i64, _i32, _vp]),
    "clt_segments_workspace_bytes": (_sz, [_i64, _i64]),
    "clt_segments_build": (_i32, [_vp, _i64, _i64, _i64, _vp, _vp, _vp, _sz, _vp]),
    "clt_spt_dense": (_i32, [_vp, _vp, _i64, _vp, _vp, _i32, _vp, _i32, _i32, _i32, _vp, _i32, _vp, _i64, _vp, _sz, _vp]),
    "clt_rows_workspace_bytes": (_sz, [_i64, _i32]),
    "clt_spt_dense_adam": (_i32, [_vp, _vp, _i64, _vp, _vp, _i32, _vp, _i32, _i32, _i32, _vp, _vp, _vp, _i32, _vp,
                                  _f32, _f32, _f32, _f32, _i32, _i64, _vp, _sz, _vp]),
    "clt_decode_bwd_rows": (_i32, [_vp, _vp, _i64, _vp, _vp, _i32, _vp, _vp, _vp, _i32, _i32, _i32, _vp, _i32, _vp, _vp,
                                   _vp, _vp, _i64, _vp, _sz, _vp]),
    "clt_decode_bwd_rows_adam": (_i32, [_vp, _vp, _i64, _vp, _vp, _i32, _vp, _vp, _vp, _i32, _i32, _i32, _vp, _vp, _vp,
                                        _i32, _vp, _vp, _vp, _f32, _f32, _f32, _f32, _i32, _i32, _i64, _vp, _sz, _vp]),
    "clt_rows_multi_workspace_bytes": (_sz, [_i64, _i32, _i32]),
    "clt_spt_dense_adam_multi": (_i32, [_vp, _i32, _i64, _i32, _i32, _i32, _i32, _i32, _f32, _f32, _f32, _f32, _i32, _i64,
                                        _vp, _sz, _vp]),
    "clt_decode_bwd_rows_adam_multi": (_i32, [_vp, _i32, _i64, _i32, _vp, _i32, _i32, _i32, _i32, _f32, _f32, _f32, _f32,
                                              _i32, _i32, _i64, _vp, _sz, _vp]),
    "clt_rows_unit_norm": (_i32, [_vp, _i32, _i64, _i32, _f32, _vp]),
    "clt_rows_project_out": (_i32, [_vp, _vp, _i32, _i64, _i32, _vp]),
    "clt_segment_sum": (_i32, [_vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp]),
    "clt_fvu_workspace_bytes": (_sz, [_i64, _i32]),
    "clt_fvu_fwd_bwd": (_i32, [_vp, _i32, _vp, _i32, _vp, _i64, _i32, _f32, _vp, _vp, _i32, _vp, _sz, _vp]),
    "clt_fvu_fwd_bwd_static": (_i32, [_vp, _i32, _vp, _i32, _vp, _i64, _i32, _f32, _vp, _vp, _i32, _vp]),
    "clt_fvu_colsum": (_i32, [_vp, _i32, _vp, _i64, _i32, _vp, _vp, _sz, _vp]),
    "clt_fvu_partial": (_i32, [_vp, _i32, _vp, _i32, _vp, _i64, _i32, _vp, _vp, _vp, _sz, _vp]),
    "clt_fvu_grad": (_i32, [_vp, _i32, _vp, _i32, _vp, _i64, _i32, _vp, _f32, _vp, _vp, _i32, _vp]),
    "clt_adam_step": (_i32, [_vp, _i32, _vp, _vp, _i64, _i32, _f32, _f32, _f32, _f32, _i32, _vp]),
    "clt_signsgd_step": (_i32, [_vp, _i32, _vp, _vp, _i64, _i32, _f32, _vp]),
}

_lib = None


def load():
    """Load the HIP library; raises (never falls back) when it is unavailable."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise RuntimeError(
            f"{LIB_PATH} not found: the MI355X HIP extension is required (no CPU fallback). "
            "Build it with `python -m clt_training_amd.build`.")
    lib = C.CDLL(str(LIB_PATH))
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if a declared symbol is missing
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


CLT_ERR_UNSUPPORTED = -2  # (include/clt_mi355.h)


def check(rc: int, what: str):
    if rc != 0:
        msg = load().clt_last_error().decode(errors="replace")
        raise RuntimeError(f"{what} failed (status {rc}): {msg}")
