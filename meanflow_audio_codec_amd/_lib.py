"""ctypes binding of the C-ABI HIP library (``include/mfc.h``).

PyTorch is used only for device memory and streams.  There is NO fallback:
if ``libmfc.so`` is missing or a call fails, an exception is raised.
"""
from __future__ import annotations

import ctypes
import pathlib
import re
from ctypes import c_char_p, c_float, c_int, c_int64, c_uint64, c_void_p

import torch  # noqa: F401  (loads the ROCm runtime libmfc.so links against)

_HERE = pathlib.Path(__file__).resolve().parent
import os  # noqa: E402
# MFC_LIB: an alternative build of the same C ABI (A/B timing of two kernel versions in one GPU session)
LIB_PATH = pathlib.Path(os.environ["MFC_LIB"]) if os.environ.get("MFC_LIB") else _HERE / "csrc" / "libmfc.so"
HEADER = _HERE.parent / "include" / "mfc.h"

MFC_F32, MFC_BF16 = 0, 1
GEMM_TRANS_A, GEMM_TRANS_B, GEMM_ACCUM, GEMM_GELU, GEMM_LN16, GEMM_LN16T = 1, 2, 4, 8, 16, 32

_ERR = {-22: "MFC_EINVAL (bad shape/argument)", -38: "MFC_ENOSYS (unsupported)",
        -14: "MFC_EFAULT (null pointer)", -5: "MFC_EHIP (HIP launch error)"}


class MfcError(RuntimeError):
    pass


_lib = None

# name -> (restype, argtypes); kept in sync with include/mfc.h by
# tests/test_capi_symbols.py
_P = c_void_p
ABI_VERSION = 3          # MFC_ABI_VERSION of include/mfc.h


class AdamwItem(ctypes.Structure):
    """``mfc_adamw_item`` of include/mfc.h"""
    _fields_ = [("p", c_void_p), ("p_bf16", c_void_p), ("g", c_void_p), ("m", c_void_p), ("v", c_void_p),
                ("n", c_int64), ("grad_dtype", ctypes.c_int32), ("reserved", ctypes.c_int32)]


SIGNATURES = {
    "mfc_abi_version": (c_int, []),
    "mfc_build_info": (c_char_p, []),
    "mfc_mdct_num_frames": (c_int64, [c_int64, c_int, c_int]),
    "mfc_mdct_out_len": (c_int64, [c_int64, c_int, c_int]),
    "mfc_mdct_fwd": (c_int, [_P, c_int64, c_int64, c_int64, c_int, c_int, _P, _P]),
    "mfc_mdct_inv": (c_int, [_P, c_int64, c_int64, c_int, c_int, _P, c_int64, _P]),
    "mfc_resample_out_len": (c_int64, [c_int64, c_int, c_int]),
    "mfc_resample_poly": (c_int, [_P, c_int64, c_int64, c_int64, c_int, c_int, _P, c_int, _P, c_int64, _P]),
    "mfc_gemm": (c_int, [c_int, c_int, c_int64, c_int64, c_int64, _P, c_int64, _P, c_int64, _P, c_int64,
                         _P, c_int64, c_int64, c_float, _P, c_int64, c_float, c_int, _P, _P, _P]),
    "mfc_ln16_fwd": (c_int, [c_int, c_int64, _P, _P, _P, _P]),
    "mfc_ln16_jvp": (c_int, [c_int, c_int64, _P, _P, _P, _P, _P]),
    "mfc_cnx_max_blocks": (c_int64, [c_int64]),
    "mfc_gemm_ws_elems": (c_int64, [c_int, c_int64, c_int64, c_int64, c_int]),
    "mfc_gemm_adamw": (c_int, [c_int, c_int64, c_int64, c_int64, _P, c_int64, _P, c_int64, c_float, _P, _P, _P, _P,
                               c_float, c_float, c_float, c_float, c_float, c_int64, _P, c_float, _P]),
    "mfc_cnx_stats": (c_int, [c_int, c_int64, c_int, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "mfc_grn_finalize": (c_int, [c_int64, _P, _P, _P, _P, _P, _P]),
    "mfc_cnx_apply": (c_int, [c_int, c_int64, c_int, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "mfc_cnx_bwd_stats": (c_int, [c_int, c_int64, c_int, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "mfc_cnx_ws_elems": (c_int64, [c_int64, c_int]),
    "mfc_grn_bwd_finalize": (c_int, [c_int64, _P, _P, _P, _P, _P]),
    "mfc_cnx_bwd_main": (c_int, [c_int, c_int64, c_int, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "mfc_cnx_stats_save": (c_int, [c_int, c_int64, c_int, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "mfc_cnx_apply_n1": (c_int, [c_int, c_int64, c_int, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "mfc_cnx_bwd_stats_n1": (c_int, [c_int, c_int64, c_int, _P, _P, _P, _P, _P, _P, _P]),
    "mfc_cnx_bwd_main_n1": (c_int, [c_int, c_int64, c_int, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "mfc_cnx_bwd_conv": (c_int, [c_int, c_int64, c_int, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "mfc_adamw_multi": (c_int, [c_int, _P, c_float, c_float, c_float, c_float, c_float, c_float, c_int64, _P]),
    "mfc_adaln_fwd": (c_int, [c_int, c_int64, c_int64, c_int64, _P, c_int64, _P, _P, c_int64, c_int64, _P, c_int64, _P]),
    "mfc_adaln_bwd": (c_int, [c_int, c_int64, c_int64, _P, c_int64, _P, c_int64, c_int64, _P, c_int64, _P, _P, _P,
                              c_int64, _P]),
    "mfc_gate_fwd": (c_int, [c_int, c_int64, c_int64, c_int64, _P, c_int64, _P, c_int64, _P, c_int64, c_float, _P,
                             c_int64, _P]),
    "mfc_gate_bwd": (c_int, [c_int, c_int64, c_int64, _P, c_int64, _P, c_int64, _P, c_int64, c_float, _P, _P, c_int64,
                             _P]),
    "mfc_copy2d": (c_int, [c_int, c_int64, c_int64, _P, c_int64, _P, c_int64, c_float, c_int, _P]),
    "mfc_transpose": (c_int, [c_int, c_int64, c_int, c_int, _P, _P, c_float, _P, _P]),
    "mfc_chanmlp_fwd": (c_int, [c_int, c_int64, c_int64, c_int64, _P, _P, _P, _P, _P, _P, _P, _P]),
    "mfc_chanmlp_ws_elems": (c_int64, [c_int64, c_int64]),
    "mfc_chanmlp_bwd": (c_int, [c_int, c_int64, c_int64, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "mfc_time_embed": (c_int, [c_int64, c_int, _P, _P, _P, _P, _P, _P, _P, _P]),
    "mfc_sample_tr": (c_int, [c_uint64, c_uint64, c_int64, c_int64, c_int64, c_int64, c_float, c_float, c_int,
                              _P, _P, _P]),
    "mfc_flow_prepare": (c_int, [c_int, c_int64, c_int64, _P, _P, _P, c_float, c_float, c_uint64, c_uint64,
                                 c_int64, c_int64, _P, _P, _P, _P]),
    "mfc_randn": (c_int, [c_uint64, c_uint64, c_int64, c_int64, c_int64, _P, _P]),
    "mfc_randn_dev": (c_int, [c_int, c_uint64, c_uint64, _P, c_int64, c_int64, c_int64, _P, _P]),
    "mfc_gelu_fwd": (c_int, [c_int, c_int64, c_int64, c_int64, _P, _P, _P]),
    "mfc_gelu_bwd": (c_int, [c_int, c_int64, _P, _P, _P, _P]),
    "mfc_flow_loss": (c_int, [c_int, c_int, c_int, c_int64, c_int64, c_int64, _P, _P, c_int64, _P, _P, _P,
                              c_float, c_float, _P, _P, _P, _P, _P, _P]),
    "mfc_colsum": (c_int, [c_int, c_int64, c_int64, _P, c_int64, c_float, _P, c_int, _P]),
    "mfc_colsum_ws_elems": (c_int64, [c_int, c_int64, c_int64]),
    "mfc_colsum_tall": (c_int, [c_int, c_int64, c_int64, _P, c_int64, c_float, _P, c_int, _P, _P]),
    "mfc_axpby": (c_int, [c_int, c_int64, c_float, _P, c_float, _P, _P, _P]),
    "mfc_cast": (c_int, [c_int, c_int, c_int64, _P, _P, _P]),
    "mfc_adamw": (c_int, [c_int, c_int64, _P, _P, _P, c_float, _P, _P, c_float, c_float, c_float, c_float,
                          c_float, c_int64, _P]),
}


class CnxParams(ctypes.Structure):
    """mfc_cnx_params / mfc_cnx_grads (9 device pointers)."""
    _fields_ = [(n, c_void_p) for n in ("conv_w", "conv_b", "exp_w", "exp_b", "grn_gamma", "grn_beta",
                                         "con_w", "con_b", "ls")]


def header_symbols() -> list[str]:
    """Every function declared in include/mfc.h."""
    txt = HEADER.read_text()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(mfc_[a-z0-9_]+)\s*\(", txt)))


# ---- optional per-call HIP-event timing (used by bench.py for the roofline line) ----
_timing = None  # list of (name, int-args, non-null-pointer mask, start_event, end_event) while enabled


_timing_only = None  # optional predicate (name, ints, nn) -> bool: which calls get events


def enable_timing(only=None):
    """Bracket launching C-ABI calls with HIP events; ``only(name, int_args, non_null_pointer_mask)`` restricts that to
    the calls it accepts (every event pair costs a few microseconds of GPU time, so timing 500 launches per step slows
    the step it measures by a few percent; timing one kernel does not)."""
    global _timing, _timing_only
    _timing = []
    _timing_only = only


def disable_timing():
    global _timing, _timing_only
    rec, _timing, _timing_only = _timing, None, None
    return rec or []


class _Proxy:
    """Forwards to the CDLL; with timing enabled brackets every launching call with HIP events
    recorded on the stream the kernels are launched on (torch's current stream)."""

    def __init__(self, cdll):
        self._c = cdll

    def __getattr__(self, name):
        fn = getattr(self._c, name)
        if _timing is None or name not in SIGNATURES or SIGNATURES[name][0] is not c_int:
            return fn
        argtypes = SIGNATURES[name][1]

        def timed(*args):
            ints = tuple(int(a) for a, ty in zip(args, argtypes) if ty in (c_int, c_int64) and a is not None)
            nn = tuple(bool(a) for a, ty in zip(args, argtypes) if ty is c_void_p)
            if _timing_only is not None and not _timing_only(name, ints, nn):
                return fn(*args)
            s = torch.cuda.Event(enable_timing=True)
            e = torch.cuda.Event(enable_timing=True)
            s.record()
            rc = fn(*args)
            e.record()
            if _timing is not None:
                _timing.append((name, ints, nn, s, e))
            return rc
        return timed


_proxy = None


def lib():
    global _lib, _proxy
    if _proxy is not None:
        return _proxy
    if _lib is None:
        if not LIB_PATH.exists():
            raise MfcError(
                f"{LIB_PATH} not found: build it with `python -m meanflow_audio_codec_amd._build` "
                "(there is no CPU fallback for the HIP path)")
        l = ctypes.CDLL(str(LIB_PATH))
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)
            fn.restype = res
            fn.argtypes = args
        if l.mfc_abi_version() != ABI_VERSION:     # a stale build would take the new arguments as garbage pointers
            raise MfcError(f"{LIB_PATH} has ABI version {l.mfc_abi_version()}, this package binds version {ABI_VERSION}: "
                           "rebuild with `python -m meanflow_audio_codec_amd._build --force`")
        _lib = l
    _proxy = _Proxy(_lib)
    return _proxy


def check(rc: int, what: str) -> None:
    if rc != 0:
        raise MfcError(f"{what} failed: {_ERR.get(rc, rc)}")


def stream_ptr() -> int:
    return torch.cuda.current_stream().cuda_stream


def dtype_code(dt: torch.dtype) -> int:
    if dt == torch.float32:
        return MFC_F32
    if dt == torch.bfloat16:
        return MFC_BF16
    raise MfcError(f"unsupported dtype {dt}")


def require_cuda(*ts) -> None:
    for t in ts:
        if t is not None and not t.is_cuda:
            raise MfcError("HIP path needs device tensors (no CPU fallback); got a CPU tensor")


def ptr(t) -> int | None:
    return None if t is None else t.data_ptr()
