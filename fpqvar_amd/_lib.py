"""ctypes binding of libfpq_hip.so (the C ABI declared in include/fpq.h).

PyTorch is plumbing here: device memory (``tensor.data_ptr()``), the current HIP
stream and the device guard.  There is NO fallback: if the shared library is
missing or does not load, every op raises.
"""
from __future__ import annotations

import ctypes
import os
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libfpq_hip.so")

F16, F32, F64 = 0, 1, 2
TENSOR_WORKSPACE_BYTES = 8192   # FPQ_TENSOR_WORKSPACE_BYTES
TABLE_IDS = {"e2m1": 0, "e1m2": 1, "e3m0": 2, "e2m3": 3, "e3m2": 4,
             "e1m2_neg": 5, "e2m1_pos": 6, "int_neg": 7, "e2m3_pos": 8, "e2m1_neg": 9}
_DTYPES = {torch.float16: F16, torch.float32: F32, torch.float64: F64}

_lib: Optional[ctypes.CDLL] = None

_c = ctypes
class Segment(_c.Structure):
    """fpq_segment_t (include/fpq.h)."""
    _fields_ = [("x", _c.c_void_p), ("out", _c.c_void_p), ("rows", _c.c_int64)]


class GemmEpilogue(_c.Structure):
    """fpq_gemm_epilogue_t (include/fpq.h)."""
    _fields_ = [("gate", _c.c_void_p), ("residual", _c.c_void_p), ("rows_per_gate", _c.c_int64)]


class GemmSplit(_c.Structure):
    """fpq_gemm_split_t (include/fpq.h)."""
    _fields_ = [("part_cols", _c.c_int64), ("n_parts", _c.c_int32), ("out", _c.c_void_p * 3), ("row_stride", _c.c_int64 * 3),
                ("rows_per_batch", _c.c_int64), ("batch_stride", _c.c_int64 * 3), ("row0", _c.c_int64 * 3)]


_SIGS = {
    "fpq_version": (_c.c_int, []),
    "fpq_strerror": (_c.c_char_p, [_c.c_int]),
    "fpq_build_tag": (_c.c_char_p, []),
    "fpq_set_option": (_c.c_int, [_c.c_char_p, _c.c_int]),
    "fpq_get_option": (_c.c_int, [_c.c_char_p, _c.POINTER(_c.c_int)]),
    "fpq_option_name": (_c.c_char_p, [_c.c_int]),
    "fpq_table_values": (_c.c_int, [_c.c_int, _c.POINTER(_c.c_float)]),
    "fpq_quant_nearest": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_int64, _c.c_int, _c.c_int,
                                      _c.c_void_p]),
    "fpq_quant_nearest_builtin": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_int64, _c.c_int, _c.c_void_p]),
    "fpq_quant_rows": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_int64, _c.c_int64, _c.c_int, _c.c_int, _c.c_int,
                                   _c.c_void_p]),
    "fpq_attention_blhc": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_int64, _c.c_int64, _c.c_int64,
                                       _c.c_int64, _c.c_int64, _c.c_int64, _c.c_int64, _c.c_int64, _c.c_int64, _c.c_float,
                                       _c.c_void_p]),
    "fpq_gate_residual": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_int64, _c.c_int64, _c.c_int64,
                                      _c.c_void_p]),
    "fpq_kv_cache_step": (_c.c_int, [_c.c_void_p, _c.c_int64, _c.c_int64, _c.c_int64, _c.c_int64, _c.c_int64, _c.c_void_p,
                                      _c.c_void_p, _c.c_int64, _c.c_int64, _c.c_int64, _c.c_int64, _c.c_int64, _c.c_int,
                                      _c.c_void_p]),
    "fpq_quant_rows_argmin": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_int64, _c.c_int64, _c.c_int, _c.c_int, _c.c_int,
                                          _c.c_void_p]),
    "fpq_quant_rows_dual": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_int64, _c.c_int64, _c.c_int, _c.c_int,
                                        _c.c_int, _c.c_int, _c.c_void_p, _c.c_float, _c.c_void_p, _c.c_void_p]),
    "fpq_gelu_quant_rows_dual": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_int64, _c.c_int64, _c.c_int, _c.c_int,
                                             _c.c_void_p, _c.c_void_p]),
    "fpq_quant_rows_codes_fp8": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_int64, _c.c_int64, _c.c_int,
                                             _c.c_int, _c.c_void_p]),
    "fpq_gemm_fp8_rows": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_int, _c.c_void_p, _c.c_void_p, _c.c_int, _c.c_void_p,
                                      _c.c_void_p, _c.c_int64, _c.c_int64, _c.c_int64, _c.c_void_p]),
    "fpq_quant_rows_codes_fp6": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_int64, _c.c_int64, _c.c_int,
                                             _c.c_int, _c.c_void_p]),
    "fpq_gemm_fp6_rows": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_int, _c.c_void_p, _c.c_void_p, _c.c_int, _c.c_void_p,
                                      _c.c_void_p, _c.c_int64, _c.c_int64, _c.c_int64, _c.c_void_p]),
    "fpq_quant_nearest_argmin": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_int64, _c.c_int, _c.c_int,
                                             _c.c_void_p]),
    "fpq_quant_rows_dual_argmin": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_int64, _c.c_int64, _c.c_int, _c.c_int,
                                               _c.c_int, _c.c_void_p, _c.c_float, _c.c_void_p]),
    "fpq_quant_rows_neg_reverse": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_int64, _c.c_int64, _c.c_int, _c.c_int,
                                               _c.c_void_p]),
    "fpq_rotate_quant_rows": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_int64, _c.c_int64, _c.c_int,
                                          _c.c_void_p, _c.POINTER(_c.c_uint32), _c.c_int, _c.c_void_p]),
    "fpq_rotate_quant_rows_codes_mx": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_int64, _c.c_int64, _c.c_int,
                                                   _c.c_void_p, _c.POINTER(_c.c_uint32), _c.c_void_p]),
    "fpq_adaln_rotate_quant_rows_codes_mx": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_int64, _c.c_int64,
                                                         _c.c_int, _c.c_void_p, _c.c_void_p, _c.c_int, _c.c_int64,
                                                         _c.c_float, _c.c_void_p, _c.POINTER(_c.c_uint32), _c.c_void_p]),
    "fpq_adaln_rotate_quant_token_rows": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_int64,
                                                      _c.c_int64, _c.c_int, _c.c_void_p, _c.c_void_p, _c.c_int, _c.c_int64,
                                                      _c.c_float, _c.c_void_p, _c.POINTER(_c.c_uint32), _c.c_int, _c.c_void_p]),
    "fpq_adaln_rotate_quant_token_rows_codes_fp8": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_int64, _c.c_int64,
                                                                _c.c_int, _c.c_void_p, _c.c_void_p, _c.c_int, _c.c_int64,
                                                                _c.c_float, _c.c_void_p, _c.POINTER(_c.c_uint32), _c.c_int,
                                                                _c.c_void_p]),
    "fpq_adaln_rotate_quant_token_rows_codes_fp6": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_int64, _c.c_int64,
                                                                _c.c_int, _c.c_void_p, _c.c_void_p, _c.c_int, _c.c_int64,
                                                                _c.c_float, _c.c_void_p, _c.POINTER(_c.c_uint32), _c.c_int,
                                                                _c.c_void_p]),
    "fpq_adaln_rotate_quant_rows": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_int64, _c.c_int64,
                                                _c.c_int, _c.c_void_p, _c.c_void_p, _c.c_int, _c.c_int64, _c.c_float,
                                                _c.c_void_p, _c.POINTER(_c.c_uint32), _c.c_int, _c.c_void_p]),
    "fpq_quant_rows_multi": (_c.c_int, [_c.c_void_p, _c.c_int, _c.c_int64, _c.c_int, _c.c_int, _c.c_int, _c.c_void_p]),
    "fpq_quant_rows_segments": (_c.c_int, [_c.c_void_p, _c.c_int, _c.c_int64, _c.c_int64, _c.c_int, _c.c_int, _c.c_int,
                                            _c.c_void_p]),
    "fpq_quant_tensor_argmin": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_int64, _c.c_int, _c.c_int,
                                            _c.c_void_p]),
    "fpq_absmax": (_c.c_int, [_c.c_void_p, _c.c_int64, _c.c_int, _c.c_void_p, _c.c_void_p]),
    "fpq_quant_rows_codes": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_int64, _c.c_int64, _c.c_int,
                                         _c.c_int, _c.c_int, _c.c_void_p]),
    "fpq_quant_rows_codes_mx": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_int64, _c.c_int64, _c.c_int,
                                            _c.c_void_p]),
    "fpq_gemm_fp4_mx": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_int, _c.c_void_p, _c.c_void_p,
                                    _c.c_int64, _c.c_int64, _c.c_int64, _c.c_void_p]),
    "fpq_gemm_fp4_mx_ex": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_int, _c.c_void_p, _c.c_void_p,
                                       _c.c_int64, _c.c_int64, _c.c_int64, _c.c_void_p, _c.c_void_p]),
    "fpq_gemm_fp4_gelu_dual": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_int, _c.c_void_p, _c.c_void_p,
                                           _c.c_void_p, _c.c_int64, _c.c_int64, _c.c_int64, _c.c_void_p, _c.c_void_p]),
    "fpq_gemm_fp4_mx_split": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_int, _c.c_void_p, _c.c_int64, _c.c_int64,
                                          _c.c_int64, _c.c_void_p, _c.c_int, _c.c_void_p]),
    "fpq_gemm_fp4_mx_km": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_int, _c.c_void_p, _c.c_void_p,
                                       _c.c_int64, _c.c_int64, _c.c_int64, _c.c_void_p, _c.c_void_p]),
    "fpq_gemm_fp4_gelu_dual_km": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_int, _c.c_void_p, _c.c_void_p,
                                              _c.c_void_p, _c.c_int64, _c.c_int64, _c.c_int64, _c.c_void_p, _c.c_void_p]),
    "fpq_gemm_fp6_rows_km": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_int, _c.c_void_p, _c.c_void_p, _c.c_int, _c.c_void_p,
                                         _c.c_void_p, _c.c_int64, _c.c_int64, _c.c_int64, _c.c_void_p, _c.c_void_p]),
    "fpq_quant_rows_codes_mx_km": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_int64, _c.c_int64, _c.c_int, _c.c_void_p]),
    "fpq_rotate_quant_rows_codes_mx_km": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_int64, _c.c_int64, _c.c_int,
                                                      _c.c_void_p, _c.POINTER(_c.c_uint32), _c.c_void_p]),
    "fpq_adaln_rotate_quant_rows_codes_mx_km": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_int64, _c.c_int64,
                                                            _c.c_int, _c.c_void_p, _c.c_void_p, _c.c_int, _c.c_int64,
                                                            _c.c_float, _c.c_void_p, _c.POINTER(_c.c_uint32), _c.c_void_p]),
    "fpq_quant_rows_codes_fp6_km": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_int64, _c.c_int64, _c.c_int,
                                                _c.c_int, _c.c_void_p]),
    "fpq_adaln_rotate_quant_token_rows_codes_fp6_km": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_int64, _c.c_int64,
                                                                   _c.c_int, _c.c_void_p, _c.c_void_p, _c.c_int, _c.c_int64,
                                                                   _c.c_float, _c.c_void_p, _c.POINTER(_c.c_uint32), _c.c_int,
                                                                   _c.c_void_p]),
    "fpq_scales_to_kmajor": (_c.c_int, [_c.c_void_p, _c.c_int, _c.c_void_p, _c.c_int64, _c.c_int64, _c.c_int, _c.c_void_p]),
    "fpq_codes_to_kmajor": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_int64, _c.c_int64, _c.c_int, _c.c_int, _c.c_void_p]),
    "fpq_gemm_fp8_rows_ex": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_int, _c.c_void_p, _c.c_void_p, _c.c_int, _c.c_void_p,
                                         _c.c_void_p, _c.c_int64, _c.c_int64, _c.c_int64, _c.c_void_p, _c.c_void_p]),
    "fpq_gemm_fp6_rows_ex": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_int, _c.c_void_p, _c.c_void_p, _c.c_int, _c.c_void_p,
                                         _c.c_void_p, _c.c_int64, _c.c_int64, _c.c_int64, _c.c_void_p, _c.c_void_p]),
    "fpq_dequant_rows_codes": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_int64, _c.c_int64, _c.c_int,
                                           _c.c_int, _c.c_int, _c.c_int, _c.c_void_p]),
    "fpq_quant_rows_codes_segments": (_c.c_int, [_c.c_void_p, _c.c_int, _c.c_int64, _c.c_int64, _c.c_int, _c.c_int, _c.c_int,
                                                  _c.c_void_p]),
    "fpq_dequant_rows_codes_segments": (_c.c_int, [_c.c_void_p, _c.c_int, _c.c_int64, _c.c_int64, _c.c_int, _c.c_int, _c.c_int,
                                                    _c.c_int, _c.c_void_p]),
}


def lib() -> ctypes.CDLL:
    """Load (once) and return the C-ABI library.  Raises if it is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  fpqvar_amd has no CPU or eager fallback.")
        l = ctypes.CDLL(LIB_PATH)   # torch is already imported: libamdhip64.so.7 resolves to torch's copy
        for name, (res, args) in _SIGS.items():
            fn = getattr(l, name)
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def use_variant(path: str, expect_tag: Optional[str] = None) -> ctypes.CDLL:
    """Point the ctypes layer at another build of the library (tools/build_variant.sh) - for the A/B and profiling tools
    only.  The compiled binding fpqvar_amd/_native is hard-linked to the stock libfpq_hip.so, so a tool that wants a
    variant measured must also keep the wrappers off that binding: set FPQ_NO_NATIVE=1 BEFORE importing fpqvar_amd
    (quant_utils, rotation and quant_cuda read it at import).  This function refuses to go on otherwise, and returns the
    library's build tag so that the tool can print which build it timed."""
    global _lib
    if os.environ.get("FPQ_NO_NATIVE") != "1":
        raise RuntimeError("use_variant: FPQ_NO_NATIVE=1 must be set before fpqvar_amd is imported - the compiled binding "
                           "would otherwise keep calling the stock library")
    l = ctypes.CDLL(os.path.abspath(path))
    for name, (res, args) in _SIGS.items():
        if hasattr(l, name):
            fn = getattr(l, name)
            fn.restype, fn.argtypes = res, args
    tag = l.fpq_build_tag().decode() if hasattr(l, "fpq_build_tag") else "untagged"
    if expect_tag is not None and tag != expect_tag:
        raise RuntimeError(f"use_variant: {path} carries build tag {tag!r}, expected {expect_tag!r}")
    _lib = l
    return l


def build_tag() -> str:
    return lib().fpq_build_tag().decode()


OPTION_DEFAULT = -2147483648   # FPQ_OPTION_DEFAULT


def set_option(name: str, value: Optional[int]) -> None:
    """fpq_set_option: an experiment switch of the library (tests and A/B tools; include/fpq.h).  None = the built-in choice.
    Process-wide: the compiled binding and the ctypes layer share the one loaded libfpq_hip.so."""
    check(lib().fpq_set_option(name.encode(), OPTION_DEFAULT if value is None else int(value)), f"fpq_set_option({name})")


def get_option(name: str) -> Optional[int]:
    """Current value of a switch, None while it is at the built-in choice."""
    v = _c.c_int(0)
    check(lib().fpq_get_option(name.encode(), _c.byref(v)), f"fpq_get_option({name})")
    return None if v.value == OPTION_DEFAULT else v.value


def option_names() -> list:
    l, out, i = lib(), [], 0
    while True:
        n = l.fpq_option_name(i)
        if n is None:
            return out
        out.append(n.decode())
        i += 1


class option:
    """`with _lib.option("FPQ_NO_HW4", 1): ...` - a switch for the duration of a block, restored afterwards."""

    def __init__(self, name: str, value: Optional[int]):
        self.name, self.value = name, value

    def __enter__(self):
        self.before = get_option(self.name)
        set_option(self.name, self.value)
        return self

    def __exit__(self, *exc):
        set_option(self.name, self.before)
        return False


def check(status: int, what: str) -> None:
    if status != 0:
        msg = lib().fpq_strerror(status).decode()
        raise RuntimeError(f"{what}: fpq error {status}: {msg}")


def dtype_id(dt: torch.dtype) -> int:
    try:
        return _DTYPES[dt]
    except KeyError:
        raise RuntimeError(f"fpqvar_amd: unsupported dtype {dt}") from None


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream_ptr(device: torch.device) -> int:
    """hipStream_t of torch's current stream on `device` (the raw accessor skips building a Stream object: ~1.5 us
    of the ~10 us a small eager call spends on the host)."""
    if _raw_stream is not None:
        return _raw_stream(device.index if device.index is not None else torch.cuda.current_device())
    return torch.cuda.current_stream(device).cuda_stream


class _NoGuard:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


_NO_GUARD = _NoGuard()


def device_guard(device: torch.device):
    """`with device_guard(x.device):` = torch.cuda.device(x.device), minus its cost (2-3 us of the ~11 us a small call
    spends on the host) in the usual case that the tensor already lives on the current device."""
    if device.index is None or device.index == torch.cuda.current_device():
        return _NO_GUARD
    return torch.cuda.device(device)


def require_gpu(t: torch.Tensor, what: str) -> None:
    if not t.is_cuda:
        raise RuntimeError(f"{what}: expected a tensor on the GPU, got device {t.device} "
                           "(fpqvar_amd has no CPU path; the CPU restatement lives in oracle/ for tests only)")


def table_values(name: str) -> torch.Tensor:
    """Host copy of a built-in table as the reference spells it."""
    l = lib()
    tid = TABLE_IDS[name]
    n = l.fpq_table_values(tid, None)
    buf = (_c.c_float * n)()
    l.fpq_table_values(tid, buf)
    return torch.tensor(list(buf), dtype=torch.float32)
