"""Tensor-level wrappers over the C ABI (one HIP launch each).

These are the additive "L1" entry points of SURVEY.md section 8b; the
reference-named functions in :mod:`fpqvar_amd.quant_utils` are thin shims over
them.  Every function here requires GPU tensors and raises otherwise.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from . import _lib
from ._lib import TABLE_IDS, check, dtype_id, lib, require_gpu, stream_ptr, device_guard

try:   # the compiled binding (csrc/quant_cuda_ext.cpp): the per-step calls of a generation go through it when it is built
    from . import _native
except ImportError:   # pragma: no cover - build() always produces it
    _native = None
if __import__("os").environ.get("FPQ_NO_NATIVE") == "1":   # the A/B tools time variant builds of the library through ctypes (_lib.use_variant)
    _native = None


def _contig(x: torch.Tensor) -> torch.Tensor:
    # the reference reshapes (copying when needed) before its kernel; same here
    return x if x.is_contiguous() else x.contiguous()


def quant_nearest(x: torch.Tensor, table: torch.Tensor) -> torch.Tensor:
    """quant_cuda.quant's first output (quant/quant_kernel.cu:11-62)."""
    require_gpu(x, "quant_nearest(x)")
    require_gpu(table, "quant_nearest(table)")
    if x.dtype not in (torch.float32, torch.float64):
        raise RuntimeError(f"quant_nearest: x must be float32 or float64, got {x.dtype}")
    if not x.is_contiguous():
        raise RuntimeError("quant_nearest: x must be contiguous")
    if table.device != x.device:
        raise RuntimeError("quant_nearest: table must live on x's device")
    k = table.numel()
    if k < 1 or k > 256:
        raise RuntimeError(f"quant_nearest: table must hold 1..256 entries, got {k}")
    tab = table.detach().reshape(-1).to(torch.float32).contiguous()
    z = torch.empty_like(x)
    with device_guard(x.device):
        check(lib().fpq_quant_nearest(x.data_ptr(), tab.data_ptr(), z.data_ptr(), x.numel(), k, dtype_id(x.dtype),
                                      stream_ptr(x.device)), "fpq_quant_nearest")
    return z


def quant_nearest_builtin(x: torch.Tensor, table: str) -> torch.Tensor:
    require_gpu(x, "quant_nearest_builtin")
    if x.dtype != torch.float32 or not x.is_contiguous():
        raise RuntimeError("quant_nearest_builtin: x must be contiguous float32")
    z = torch.empty_like(x)
    with device_guard(x.device):
        check(lib().fpq_quant_nearest_builtin(x.data_ptr(), z.data_ptr(), x.numel(), TABLE_IDS[table],
                                              stream_ptr(x.device)), "fpq_quant_nearest_builtin")
    return z


def quant_rows(x: torch.Tensor, table: str, cols: int, out_dtype: Optional[torch.dtype] = None) -> torch.Tensor:
    """Fake-quantize x viewed as [numel/cols, cols], one scale per row; same shape out."""
    require_gpu(x, "quant_rows")
    if x.dtype not in (torch.float16, torch.float32):
        raise RuntimeError(f"quant_rows: x must be float16 or float32, got {x.dtype}")
    out_dtype = x.dtype if out_dtype is None else out_dtype
    n = x.numel()
    if cols <= 0 or n % cols != 0:
        raise RuntimeError(f"quant_rows: numel {n} is not a multiple of the row length {cols}")
    xc = _contig(x)
    out = torch.empty(x.shape, dtype=out_dtype, device=x.device)
    with device_guard(x.device):
        check(lib().fpq_quant_rows(xc.data_ptr(), out.data_ptr(), n // cols, cols, TABLE_IDS[table],
                                   dtype_id(x.dtype), dtype_id(out_dtype), stream_ptr(x.device)), "fpq_quant_rows")
    return out


def quant_rows_multi(xs, table: str, cols: int, out_dtype: Optional[torch.dtype] = None):
    """quant_rows on several tensors of one device and dtype with ONE C-ABI call (one launch for up to 8 fp16 tensors
    with rows of 8 .. 512 elements, fpq_quant_rows_multi): list of results, same shapes."""
    xs = list(xs)
    if not xs:
        return []
    for x in xs:
        require_gpu(x, "quant_rows_multi")
        if x.dtype != xs[0].dtype or x.device != xs[0].device:
            raise RuntimeError("quant_rows_multi: the tensors must share dtype and device")
        if cols <= 0 or x.numel() % cols != 0:
            raise RuntimeError(f"quant_rows_multi: numel {x.numel()} is not a multiple of the row length {cols}")
    if xs[0].dtype not in (torch.float16, torch.float32):
        raise RuntimeError(f"quant_rows_multi: tensors must be float16 or float32, got {xs[0].dtype}")
    out_dtype = xs[0].dtype if out_dtype is None else out_dtype
    xc = [_contig(x) for x in xs]
    outs = [torch.empty(x.shape, dtype=out_dtype, device=x.device) for x in xs]
    segs = (_lib.Segment * len(xs))(*[_lib.Segment(a.data_ptr(), o.data_ptr(), a.numel() // cols) for a, o in zip(xc, outs)])
    with device_guard(xs[0].device):
        check(lib().fpq_quant_rows_multi(segs, len(xs), cols, TABLE_IDS[table], dtype_id(xs[0].dtype), dtype_id(out_dtype),
                                         stream_ptr(xs[0].device)), "fpq_quant_rows_multi")
    return outs


def gate_residual(y: torch.Tensor, gate: torch.Tensor, residual: torch.Tensor) -> torch.Tensor:
    """residual + y.mul(gate) in one launch, bit-identical to the two torch ops (tr/basic_var.py:264,267): y and
    residual fp16 [B, L, C], gate fp16 [B, 1, C] (gamma1 / gamma2 of the AdaLN block)."""
    require_gpu(y, "gate_residual")
    if y.dtype != torch.float16 or gate.dtype != torch.float16 or residual.dtype != torch.float16:
        raise RuntimeError("gate_residual: y, gate and residual must be float16")
    C = y.shape[-1]
    rows = y.numel() // C if C else 0
    g = gate.reshape(-1, C)
    if residual.shape != y.shape or g.shape[0] == 0 or rows % g.shape[0] != 0 or C % 8 != 0:
        raise RuntimeError(f"gate_residual: shapes {tuple(y.shape)} {tuple(gate.shape)} {tuple(residual.shape)} do not fit")
    yc, gc, rc = _contig(y), _contig(g), _contig(residual)
    out = torch.empty_like(yc)
    with device_guard(y.device):
        check(lib().fpq_gate_residual(yc.data_ptr(), gc.data_ptr(), rc.data_ptr(), out.data_ptr(), rows, C,
                                      max(rows // g.shape[0], 1), stream_ptr(y.device)), "fpq_gate_residual")
    return out.view(y.shape)


def attention_blhc(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, scale: float) -> torch.Tensor:
    """fpq_attention_blhc: softmax(q k^T * scale) v for fp16 q [B, Lq, H, 64], k / v [B, Lkv, H, 64] (views with
    contiguous (H, 64) rows are taken as they are) -> [B, Lq, H, 64]; flash_attn_func(q, k, v, softmax_scale=scale)."""
    require_gpu(q, "attention_blhc")
    if q.dtype != torch.float16 or k.dtype != torch.float16 or v.dtype != torch.float16:
        raise RuntimeError("attention_blhc: q, k and v must be float16")
    if q.dim() != 4 or k.dim() != 4 or k.shape != v.shape or q.shape[0] != k.shape[0] or q.shape[2:] != k.shape[2:]:
        raise RuntimeError(f"attention_blhc: expected q [B, Lq, H, c] and k / v [B, Lkv, H, c], got {tuple(q.shape)} {tuple(k.shape)} {tuple(v.shape)}")
    B, Lq, H, c = q.shape
    Lkv = k.shape[1]
    if c != 64:
        raise RuntimeError("attention_blhc: head_dim must be 64")
    if Lkv == 0 and Lq > 0 and B > 0:
        raise RuntimeError("attention_blhc: no keys")

    def rows_ok(t):
        return t.shape[1] == 0 or t.shape[0] == 0 or (t.stride(3) == 1 and t.stride(2) == c and t.stride(0) % 8 == 0
                                                      and t.stride(1) % 8 == 0 and t.data_ptr() % 16 == 0)
    q = q if rows_ok(q) else q.contiguous()
    if not (rows_ok(k) and rows_ok(v) and k.stride() == v.stride()):
        k, v = k.contiguous(), v.contiguous()
    out = torch.empty((B, Lq, H, c), dtype=torch.float16, device=q.device)
    with device_guard(q.device):
        check(lib().fpq_attention_blhc(q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), B, Lq, Lkv, H, c,
                                       q.stride(0), q.stride(1), k.stride(0), k.stride(1), float(scale),
                                       stream_ptr(q.device)), "fpq_attention_blhc")
    return out


def kv_cache_step(cache: torch.Tensor, quant_start: int, quant_stop: int, k: torch.Tensor, v: torch.Tensor,
                  new_start: int, group: int, table: str) -> None:
    """fpq_kv_cache_step: quantize tokens [quant_start, quant_stop) of the fp16 cache [2, B, max_len, H, c] in place
    and copy the new k / v [B, n, H, c] (rows contiguous, any batch / token stride) to tokens new_start.."""
    if _native is not None:   # same checks, same C call, a third of the host time
        return _native.kv_cache_step(cache, quant_start, quant_stop, k, v, new_start, group, TABLE_IDS[table])
    require_gpu(cache, "kv_cache_step")
    if cache.dtype != torch.float16 or k.dtype != torch.float16 or v.dtype != torch.float16:
        raise RuntimeError("kv_cache_step: cache, k and v must be float16")
    if cache.dim() != 5 or cache.shape[0] != 2 or not cache.is_contiguous():
        raise RuntimeError("kv_cache_step: cache must be a contiguous [2, B, max_len, H, c] tensor")
    _, B, max_len, H, c = cache.shape
    if k.shape != v.shape or k.dim() != 4 or k.shape[0] != B or tuple(k.shape[2:]) != (H, c):
        raise RuntimeError(f"kv_cache_step: k / v must be [B, n, H, c] = [{B}, n, {H}, {c}], got {tuple(k.shape)} / {tuple(v.shape)}")
    n = k.shape[1]
    for t in (k, v):
        if n and (t.stride(3) != 1 or t.stride(2) != c):
            raise RuntimeError("kv_cache_step: the (H, c) rows of k / v must be contiguous")
    if n and (k.stride() != v.stride()):
        raise RuntimeError("kv_cache_step: k and v must share their strides")
    with device_guard(cache.device):
        check(lib().fpq_kv_cache_step(cache.data_ptr(), B, max_len, H * c, quant_start, quant_stop, k.data_ptr(),
                                      v.data_ptr(), k.stride(0) if n else 0, k.stride(1) if n else 0, new_start, n, group,
                                      TABLE_IDS[table], stream_ptr(cache.device)), "fpq_kv_cache_step")


def quant_rows_argmin(x: torch.Tensor, table: str, cols: int, clamp3: bool) -> torch.Tensor:
    """The reference's pure-torch quantizers (argmin lookup, float32 result) in one launch."""
    require_gpu(x, "quant_rows_argmin")
    if x.dtype not in (torch.float16, torch.float32):
        raise RuntimeError(f"quant_rows_argmin: x must be float16 or float32, got {x.dtype}")
    n = x.numel()
    if cols <= 0 or n % cols != 0:
        raise RuntimeError(f"quant_rows_argmin: numel {n} is not a multiple of the row length {cols}")
    xc = _contig(x)
    out = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    with device_guard(x.device):
        check(lib().fpq_quant_rows_argmin(xc.data_ptr(), out.data_ptr(), n // cols, cols, TABLE_IDS[table],
                                          dtype_id(x.dtype), int(clamp3), stream_ptr(x.device)),
              "fpq_quant_rows_argmin")
    return out


def quant_rows_neg_reverse(x: torch.Tensor, table: str, cols: int) -> torch.Tensor:
    """Shift-by-|row min| quantizer for the non-positive half (fpq_quant_rows_neg_reverse); x.dtype result."""
    require_gpu(x, "quant_rows_neg_reverse")
    if x.dtype not in (torch.float16, torch.float32):
        raise RuntimeError(f"quant_rows_neg_reverse: x must be float16 or float32, got {x.dtype}")
    n = x.numel()
    if cols <= 0 or n % cols != 0:
        raise RuntimeError(f"quant_rows_neg_reverse: numel {n} is not a multiple of the row length {cols}")
    xc = _contig(x)
    out = torch.empty(x.shape, dtype=x.dtype, device=x.device)
    with device_guard(x.device):
        check(lib().fpq_quant_rows_neg_reverse(xc.data_ptr(), out.data_ptr(), n // cols, cols, TABLE_IDS[table],
                                               dtype_id(x.dtype), stream_ptr(x.device)),
              "fpq_quant_rows_neg_reverse")
    return out


def quant_tensor_argmin(x: torch.Tensor, table: str) -> Tuple[torch.Tensor, torch.Tensor]:
    """One scale for the whole tensor, argmin lookup (fpq_quant_tensor_argmin): (float32 result, 0-dim float32 scale)."""
    require_gpu(x, "quant_tensor_argmin")
    if x.dtype not in (torch.float16, torch.float32):
        raise RuntimeError(f"quant_tensor_argmin: x must be float16 or float32, got {x.dtype}")
    if x.numel() == 0:
        raise RuntimeError("max(): Expected reduction dim to be specified for input.numel() == 0")   # as x.abs().max()
    xc = _contig(x)
    out = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    ws = torch.empty(_lib.TENSOR_WORKSPACE_BYTES // 4 + 1, dtype=torch.int32, device=x.device)   # maxima + the scale
    scale = ws[-1:].view(torch.float32)
    with device_guard(x.device):
        check(lib().fpq_quant_tensor_argmin(xc.data_ptr(), out.data_ptr(), scale.data_ptr(), ws.data_ptr(), xc.numel(),
                                            TABLE_IDS[table], dtype_id(x.dtype), stream_ptr(x.device)),
              "fpq_quant_tensor_argmin")
    return out, scale.reshape(())


def absmax(x: torch.Tensor) -> torch.Tensor:
    """0-dim max|x| in x's dtype (NaN-propagating)."""
    require_gpu(x, "absmax")
    xc = _contig(x)
    buf = torch.empty(2 if x.dtype == torch.float16 else 1, dtype=x.dtype, device=x.device)
    with device_guard(x.device):
        check(lib().fpq_absmax(xc.data_ptr(), xc.numel(), dtype_id(x.dtype), buf.data_ptr(), stream_ptr(x.device)),
              "fpq_absmax")
    return buf[0]


def quant_rows_dual(x: torch.Tensor, neg_table: str, pos_table: str, cols: int,
                    clipping_strength: Optional[float] = None,
                    out_dtype: Optional[torch.dtype] = None) -> torch.Tensor:
    """Dual-format fake-quantize.

    clipping_strength=None: no global clamp (tr/quant_utils.py:577-646).
    clipping_strength=1.0:  the reference's clamp to +-1.0*max|x| (:421-422) is the identity
        unless x holds a NaN, in which case everything becomes zero; reproduced with a NaN
        flag + conditional zero-fill launch instead of a global absmax pass.
    any other number: clamp to +-strength*max|x| first (one extra read pass for the absmax)."""
    require_gpu(x, "quant_rows_dual")
    if x.dtype not in (torch.float16, torch.float32):
        raise RuntimeError(f"quant_rows_dual: x must be float16 or float32, got {x.dtype}")
    out_dtype = x.dtype if out_dtype is None else out_dtype
    n = x.numel()
    if cols <= 0 or n % cols != 0:
        raise RuntimeError(f"quant_rows_dual: numel {n} is not a multiple of the row length {cols}")
    xc = _contig(x)
    out = torch.empty(x.shape, dtype=out_dtype, device=x.device)
    with device_guard(x.device):
        clip_ptr, strength, flag_ptr = None, 1.0, None
        if clipping_strength is not None and float(clipping_strength) == 1.0:
            flag_ptr = _nan_scratch(x.device).data_ptr()
        elif clipping_strength is not None:
            am = absmax(xc)
            clip_ptr, strength = am.data_ptr(), float(clipping_strength)
        status = lib().fpq_quant_rows_dual(xc.data_ptr(), out.data_ptr(), n // cols, cols, TABLE_IDS[neg_table],
                                           TABLE_IDS[pos_table], dtype_id(x.dtype), dtype_id(out_dtype), clip_ptr,
                                           strength, flag_ptr, stream_ptr(x.device))
        if status != 0 and flag_ptr is not None:
            _NAN_SCRATCH.clear()        # a failed launch may have left the words raised: never reuse them
        check(status, "fpq_quant_rows_dual")
    return out


def gelu_quant_rows_dual(y: torch.Tensor, neg_table: str = "e1m2_neg", pos_table: str = "e2m1_pos", cols: int = 128,
                         clipping_strength: Optional[float] = 1.0, return_gelu: bool = False):
    """`quant_rows_dual(F.gelu(y, approximate="tanh"), neg, pos, cols, clipping_strength)` in ONE pass over y
    (fpq_gelu_quant_rows_dual): the reference's `fc2.act_quant(act(fc1_output))` (tr/basic_var.py:120-121; the quantizer bound
    at tr/quant_utils.py:991 - W4A4, groups of 128, strength 1.0 - or :930-931 - W6A6, INT-/E2M3+ per token, no clamp:
    clipping_strength None, cols = the row length) for an fp16 fc1 output.  return_gelu: also the GELU values the quantizer saw
    (quantization bit-exact on those; within one fp16 ulp of torch's GELU on every fp16 input)."""
    if clipping_strength is not None and float(clipping_strength) != 1.0:
        raise RuntimeError("gelu_quant_rows_dual: clipping_strength must be 1.0 (the FP4 pair's default) or None")
    nan_rule = clipping_strength is not None
    if _native is not None:
        out, h = _native.gelu_quant_rows_dual(y, TABLE_IDS[neg_table], TABLE_IDS[pos_table], cols, nan_rule, return_gelu)
        return (out, h) if return_gelu else out
    require_gpu(y, "gelu_quant_rows_dual")
    if y.dtype != torch.float16 or cols <= 0 or cols % 8 != 0 or y.numel() % cols != 0:
        raise RuntimeError(f"gelu_quant_rows_dual: y must be float16 and hold whole rows of {cols} (a multiple of 8) elements, got {y.dtype} {tuple(y.shape)}")
    yc = _contig(y)
    out = torch.empty(y.shape, dtype=torch.float16, device=y.device)
    h = torch.empty(y.shape, dtype=torch.float16, device=y.device) if return_gelu else None
    if yc.numel():
        with device_guard(y.device):
            flag_ptr = _nan_scratch(y.device).data_ptr() if nan_rule else None
            status = lib().fpq_gelu_quant_rows_dual(yc.data_ptr(), out.data_ptr(), None if h is None else h.data_ptr(), yc.numel() // cols, cols,
                                                    TABLE_IDS[neg_table], TABLE_IDS[pos_table], flag_ptr, stream_ptr(y.device))
            if status != 0 and nan_rule:
                _NAN_SCRATCH.clear()
            check(status, "fpq_gelu_quant_rows_dual")
    return (out, h) if return_gelu else out


_NAN_SCRATCH = {}
_NAN_SCRATCH_CAPTURED = []


def _nan_scratch(device: torch.device) -> torch.Tensor:
    """8 zeroed bytes per (device, stream) for fpq_quant_rows_dual's NaN flag: the fix-up launch leaves them zero, so
    one allocation serves every call on that stream (include/fpq.h)."""
    if torch.cuda.is_current_stream_capturing():
        # a graph bakes the pointer in and may be replayed beside eager calls on the same stream: scratch of its own,
        # kept alive with the process (8 bytes per captured call)
        t = torch.zeros(2, dtype=torch.int32, device=device)
        _NAN_SCRATCH_CAPTURED.append(t)
        return t
    key = (device.index, torch.cuda.current_stream(device).cuda_stream)
    t = _NAN_SCRATCH.get(key)
    if t is None:
        t = _NAN_SCRATCH[key] = torch.zeros(2, dtype=torch.int32, device=device)
    return t


def quant_nearest_argmin(x: torch.Tensor, table: torch.Tensor) -> torch.Tensor:
    """grid[argmin |x - grid|] with torch.argmin's tie / NaN rules, any table of <= 256 entries; float32 result."""
    require_gpu(x, "quant_nearest_argmin")
    if x.dtype not in (torch.float16, torch.float32):
        raise RuntimeError(f"quant_nearest_argmin: x must be float16 or float32, got {x.dtype}")
    t = table.detach().to(device=x.device, dtype=torch.float32).reshape(-1).contiguous()
    if t.numel() < 1 or t.numel() > 256:
        raise RuntimeError("quant_nearest_argmin: the table must hold 1..256 entries")
    xc = _contig(x)
    out = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    with device_guard(x.device):
        check(lib().fpq_quant_nearest_argmin(xc.data_ptr(), t.data_ptr(), out.data_ptr(), xc.numel(), t.numel(),
                                             dtype_id(x.dtype), stream_ptr(x.device)), "fpq_quant_nearest_argmin")
    return out


def quant_rows_dual_argmin(x: torch.Tensor, neg_table: str, pos_table: str, cols: int,
                           clipping_strength: float = 1.0) -> torch.Tensor:
    """The reference's pure-torch dual-format quantizer (argmin lookup, float32 result) in two launches: the
    global max|x| its clamp needs, then the rows."""
    require_gpu(x, "quant_rows_dual_argmin")
    if x.dtype not in (torch.float16, torch.float32):
        raise RuntimeError(f"quant_rows_dual_argmin: x must be float16 or float32, got {x.dtype}")
    n = x.numel()
    if cols <= 0 or n % cols != 0:
        raise RuntimeError(f"quant_rows_dual_argmin: numel {n} is not a multiple of the row length {cols}")
    xc = _contig(x)
    out = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    if n == 0:
        return out
    amax = absmax(xc)
    with device_guard(x.device):
        check(lib().fpq_quant_rows_dual_argmin(xc.data_ptr(), out.data_ptr(), n // cols, cols, TABLE_IDS[neg_table],
                                               TABLE_IDS[pos_table], dtype_id(x.dtype), amax.data_ptr(),
                                               float(clipping_strength), stream_ptr(x.device)),
              "fpq_quant_rows_dual_argmin")
    return out


def quant_rows_codes(x: torch.Tensor, table: str, cols: int, pack_nibbles: bool = False
                     ) -> Tuple[torch.Tensor, torch.Tensor]:
    """(codes uint8, scales x.dtype[rows]).  codes are [rows, cols] or, packed, [rows, ceil(cols/2)]."""
    require_gpu(x, "quant_rows_codes")
    n = x.numel()
    if cols <= 0 or n % cols != 0:
        raise RuntimeError("quant_rows_codes: numel is not a multiple of the row length")
    rows = n // cols
    xc = _contig(x)
    ccols = (cols + 1) // 2 if pack_nibbles else cols
    codes = torch.empty((rows, ccols), dtype=torch.uint8, device=x.device)
    scales = torch.empty((rows,), dtype=x.dtype, device=x.device)
    with device_guard(x.device):
        check(lib().fpq_quant_rows_codes(xc.data_ptr(), codes.data_ptr(), scales.data_ptr(), rows, cols,
                                         TABLE_IDS[table], dtype_id(x.dtype), int(pack_nibbles),
                                         stream_ptr(x.device)), "fpq_quant_rows_codes")
    return codes, scales


def dequant_rows_codes(codes: torch.Tensor, scales: torch.Tensor, table: str, cols: int,
                       out_dtype: torch.dtype, pack_nibbles: bool = False) -> torch.Tensor:
    require_gpu(codes, "dequant_rows_codes")
    rows = scales.numel()
    row_bytes = (cols + 1) // 2 if pack_nibbles else cols
    if codes.dtype != torch.uint8 or not codes.is_contiguous():
        raise RuntimeError("dequant_rows_codes: codes must be a contiguous uint8 tensor")
    if scales.device != codes.device or not scales.is_contiguous() or scales.dtype not in (torch.float16, torch.float32):
        raise RuntimeError("dequant_rows_codes: scales must be a contiguous float16 / float32 tensor on the codes' device")
    if cols <= 0 or codes.numel() != rows * row_bytes:
        raise RuntimeError(f"dequant_rows_codes: {codes.numel()} code bytes do not match {rows} rows of {cols} "
                           f"{'nibble-packed ' if pack_nibbles else ''}codes ({rows * row_bytes} bytes)")
    out = torch.empty((rows, cols), dtype=out_dtype, device=codes.device)
    with device_guard(codes.device):
        check(lib().fpq_dequant_rows_codes(codes.data_ptr(), scales.data_ptr(), out.data_ptr(), rows, cols,
                                           TABLE_IDS[table], dtype_id(scales.dtype), dtype_id(out_dtype),
                                           int(pack_nibbles), stream_ptr(codes.device)), "fpq_dequant_rows_codes")
    return out
