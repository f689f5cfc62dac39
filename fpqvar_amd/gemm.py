"""F2: real FP4 arithmetic for the W4A4 per-group E2M1 configuration.

The reference only *simulates* FP4: it de-quantizes and runs an fp16 GEMM
(tr/quant_utils.py:765-767).  Here the quantizer emits hardware E2M1 nibbles + per-group scales and
the product runs on MI355X's block-scaled FP4 matrix cores (`fpq_gemm_fp4_mx`).
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch

import ctypes

from ._lib import GemmEpilogue, check, dtype_id, lib, require_gpu, stream_ptr, device_guard

try:   # the compiled binding (csrc/quant_cuda_ext.cpp)
    from . import _native
except ImportError:   # pragma: no cover - build() always produces it
    _native = None
if __import__("os").environ.get("FPQ_NO_NATIVE") == "1":   # the A/B tools time variant builds of the library through ctypes (_lib.use_variant)
    _native = None

E2M1_LEVELS = (0.0, 0.5, 1.0, 1.5, 2.0, 3.0, 4.0, 6.0)


def _epilogue(name: str, tokens: int, outs: int, gate: Optional[torch.Tensor], residual: Optional[torch.Tensor],
              out: Optional[torch.Tensor], device):
    """The GEMMs' optional fused tail `residual + y * gate` (fpq_gemm_epilogue_t): gate is gamma of the AdaLN block,
    [B, 1, outs] or [B, outs] fp16 with tokens % B == 0; residual is [..., outs] fp16 with `tokens` rows.
    Returns (pointer or None, objects to keep alive until the launch, output tensor)."""
    if gate is None and residual is None:
        return None, (), torch.empty((tokens, outs), dtype=torch.float16, device=device) if out is None else out
    ep, keep = GemmEpilogue(None, None, 1), []
    if gate is not None:
        g = gate.reshape(-1, outs)
        if g.dtype != torch.float16 or g.shape[0] == 0 or tokens % g.shape[0] != 0:
            raise RuntimeError(f"{name}: gate must be float16 [B, outs] with tokens % B == 0, got {tuple(gate.shape)} {gate.dtype}")
        g = g.contiguous()
        ep.gate, ep.rows_per_gate = g.data_ptr(), max(tokens // g.shape[0], 1)
        keep.append(g)
    if residual is not None:
        r = residual.reshape(-1, outs)
        if r.dtype != torch.float16 or r.shape[0] != tokens:
            raise RuntimeError(f"{name}: residual must be float16 with {tokens} rows of {outs}, got {tuple(residual.shape)} {residual.dtype}")
        r = r.contiguous()
        ep.residual = r.data_ptr()
        keep.append(r)
    keep.append(ep)
    return ctypes.byref(ep), keep, torch.empty((tokens, outs), dtype=torch.float16, device=device) if out is None else out


# ---- k-major operand images (include/fpq.h): codes as [K / 128, image rows, 64 | 96] instead of [rows, row bytes] --------
# A 3-D uint8 tensor IS an image, a 2-D one row-major codes: the Linears below pick the entry point from the operand's shape
# (both operands must agree).  Weights: `to_kmajor(codes, bits, dealt=True)` once at load time; activations: the producers'
# `kmajor=True` forms write the image directly (to_kmajor(codes, bits) converts any other producer's output).  The per-group
# scales of an FP4 operand travel with it as an fp32 image [K/128, rows rounded up to 4 | 64] (`to_kmajor_scales`); the FP6
# operands keep their one scale per row.
def to_kmajor(codes: torch.Tensor, code_bits: int, dealt: bool = False) -> torch.Tensor:
    """Row-major operand codes [rows, K/2] (code_bits 4) or [rows, K*3/4] (6) -> the k-major image [K/128, image_rows, 64 | 96]
    (fpq_codes_to_kmajor).  dealt: the weight side's row order, image_rows = rows rounded up to 64."""
    require_gpu(codes, "to_kmajor")
    seg = {4: 64, 6: 96}.get(code_bits)
    if seg is None or codes.dim() != 2 or codes.dtype != torch.uint8 or codes.shape[1] % seg != 0:
        raise RuntimeError("to_kmajor: codes must be uint8 [rows, K/2] (code_bits 4) or [rows, K*3/4] (6) with K % 128 == 0")
    c = codes.contiguous()
    rows, steps = c.shape[0], c.shape[1] // seg
    image = torch.empty((steps, (rows + 63) // 64 * 64 if dealt else rows, seg), dtype=torch.uint8, device=c.device)
    with device_guard(c.device):
        check(lib().fpq_codes_to_kmajor(c.data_ptr(), image.data_ptr(), rows, steps * 128, code_bits, 1 if dealt else 0,
                                        stream_ptr(c.device)), "fpq_codes_to_kmajor")
    return image


def to_kmajor_scales(scales: torch.Tensor, weight_side: bool = False) -> torch.Tensor:
    """Per-group scales [rows, K/128] (fp16 / fp32) of an FP4 operand -> the fp32 k-major scale image [K/128, image_rows]
    (fpq_scales_to_kmajor): image_rows = rows rounded up to 4, or to 64 on the weight side; the padding is zero."""
    require_gpu(scales, "to_kmajor_scales")
    if scales.dim() != 2 or scales.dtype not in (torch.float16, torch.float32):
        raise RuntimeError("to_kmajor_scales: scales must be float16 / float32 [rows, K/128]")
    sc = scales.contiguous()
    rows, groups = sc.shape
    image = torch.empty((groups, (rows + 63) // 64 * 64 if weight_side else (rows + 3) // 4 * 4), dtype=torch.float32, device=sc.device)
    with device_guard(sc.device):
        check(lib().fpq_scales_to_kmajor(sc.data_ptr(), dtype_id(sc.dtype), image.data_ptr(), rows, groups, 1 if weight_side else 0,
                                         stream_ptr(sc.device)), "fpq_scales_to_kmajor")
    return image


def kmajor_mx_scales(rows: int, k: int, device) -> torch.Tensor:
    """an empty k-major scale image for `rows` activation rows ([K/128, rows rounded up to 4] fp32) with its padding zeroed -
    what the *_km producers of FP4 operands write their scales into"""
    pad = (rows + 3) // 4 * 4
    image = torch.empty((k // 128, pad), dtype=torch.float32, device=device)
    if pad != rows:
        image[:, rows:].zero_()
    return image


def _kmajor_pair(name: str, a: torch.Tensor, w: torch.Tensor, seg: int) -> bool:
    """True when both operands are k-major images (3-D), False when both are row-major codes (2-D); anything else is an error."""
    if a.dim() == 3 and w.dim() == 3:
        if a.shape[2] != seg or w.shape[2] != seg or a.shape[0] != w.shape[0] or w.shape[1] % 64 != 0:
            raise RuntimeError(f"{name}: k-major images must be [K/128, rows, {seg}] with the same K and a weight image of a multiple of 64 rows")
        return True
    if a.dim() == 2 and w.dim() == 2:
        return False
    raise RuntimeError(f"{name}: both operands must be row-major codes (2-D) or both k-major images (3-D)")


def quantize_mx(x: torch.Tensor, kmajor: bool = False) -> Tuple[torch.Tensor, torch.Tensor]:
    """x [..., K] fp16/fp32 (K % 128 == 0) -> (codes uint8 [rows, K/2], scales [rows, K/128] in x.dtype).
    kmajor: the activation side's k-major images - codes [K/128, rows, 64] and scales fp32 [K/128, rows rounded up to 4]."""
    require_gpu(x, "quantize_mx")
    if x.dtype not in (torch.float16, torch.float32):
        raise RuntimeError(f"quantize_mx: x must be float16 or float32, got {x.dtype}")
    k = x.shape[-1]
    if k % 128 != 0:
        raise RuntimeError("quantize_mx: the last dimension must be a multiple of 128")
    if kmajor and x.dtype == torch.float32:   # the image-writing quantizer takes fp16 rows (activations); fp32 rows: two steps
        codes, scales = quantize_mx(x)
        return to_kmajor(codes, 4), to_kmajor_scales(scales)
    xc = x.contiguous()
    rows = xc.numel() // k
    codes = torch.empty((k // 128, rows, 64) if kmajor else (rows, k // 2), dtype=torch.uint8, device=x.device)
    scales = kmajor_mx_scales(rows, k, x.device) if kmajor else torch.empty((rows, k // 128), dtype=x.dtype, device=x.device)
    fn = lib().fpq_quant_rows_codes_mx_km if kmajor else lib().fpq_quant_rows_codes_mx
    with device_guard(x.device):
        check(fn(xc.data_ptr(), codes.data_ptr(), scales.data_ptr(), rows, k, dtype_id(x.dtype), stream_ptr(x.device)),
              "fpq_quant_rows_codes_mx_km" if kmajor else "fpq_quant_rows_codes_mx")
    return codes, scales


def dequantize_mx(codes: torch.Tensor, scales: torch.Tensor) -> torch.Tensor:
    """Reference decoder in torch ops (tests / debugging): fp32 [rows, K]."""
    lv = torch.tensor(E2M1_LEVELS + tuple(-v for v in E2M1_LEVELS), dtype=torch.float32, device=codes.device)
    lo, hi = (codes & 0xF).long(), (codes >> 4).long()
    q = torch.stack((lv[lo], lv[hi]), dim=-1).reshape(codes.shape[0], -1)
    return (q.view(codes.shape[0], -1, 128) * scales.float().unsqueeze(-1)).reshape(codes.shape[0], -1)


class _ScaledOperandModule(torch.nn.Module):
    """Base of the matrix-core Linears.  The quantization scales are fp32 in the reference (the weight is quantized in
    fp32, tr/quant_utils.py:828-837); the driver's `var.half()` (evaluate_fp_quant_transform_rotate.py:131) must not
    round them to fp16 - floating-point casts applied to the module (half(), to(dtype), float()) leave `w_scales`
    alone, device moves still apply."""

    def _apply(self, fn, recurse=True):
        keep = self._buffers.get("w_scales")
        super()._apply(fn, recurse)
        moved = self._buffers.get("w_scales")
        if keep is not None and moved is not None and moved.dtype != keep.dtype:
            self._buffers["w_scales"] = keep.to(device=moved.device)
        return self


def _check_operand(what: str, codes: torch.Tensor, scales: torch.Tensor, rows: int, row_bytes: int, n_scales: int,
                   device) -> None:
    """Shapes, dtypes and placement of a (codes, scales) operand BEFORE the kernel sees the pointers: a truncated or
    inconsistent operand is a Python error here, not an out-of-bounds read on the GPU."""
    if codes.dtype != torch.uint8 or not codes.is_contiguous() or codes.device != device:
        raise RuntimeError(f"{what}: codes must be a contiguous uint8 tensor on {device}")
    if codes.numel() != rows * row_bytes:
        raise RuntimeError(f"{what}: codes hold {codes.numel()} bytes, expected {rows} x {row_bytes}")
    if scales.dtype not in (torch.float16, torch.float32) or not scales.is_contiguous() or scales.device != device:
        raise RuntimeError(f"{what}: scales must be a contiguous float16 / float32 tensor on {device}")
    if scales.numel() != n_scales:
        raise RuntimeError(f"{what}: {scales.numel()} scales, expected {n_scales}")



def _check_kmajor_fp4(name: str, a_codes, a_scales, w_codes, w_scales, bias, outs):
    """shapes of a k-major FP4 operand pair -> (tokens, outs, k); `outs` (or the bias) names the Linear's width when it is not
    the weight image's row count (a multiple of 64)"""
    groups, tokens, w_rows = a_codes.shape[0], a_codes.shape[1], w_codes.shape[1]
    if outs is None:
        outs = bias.numel() if bias is not None else w_rows
    if not (w_rows - 64 < outs <= w_rows):
        raise RuntimeError(f"{name}: outs = {outs} does not belong to a weight image of {w_rows} rows")
    dev = a_codes.device
    for what, t, rows in (("activation", a_scales, (tokens + 3) // 4 * 4), ("weight", w_scales, w_rows)):
        if t.dtype != torch.float32 or tuple(t.shape) != (groups, rows) or not t.is_contiguous() or t.device != dev:
            raise RuntimeError(f"{name}({what}): the k-major scale image must be a contiguous float32 [{groups}, {rows}] tensor on {dev}")
    for what, t in (("activation", a_codes), ("weight", w_codes)):
        if t.dtype != torch.uint8 or not t.is_contiguous() or t.device != dev:
            raise RuntimeError(f"{name}({what}): the k-major image must be a contiguous uint8 tensor on {dev}")
    return tokens, outs, groups * 128


def linear_fp4(a_codes: torch.Tensor, a_scales: torch.Tensor, w_codes: torch.Tensor, w_scales: torch.Tensor,
               bias: Optional[torch.Tensor] = None, gate: Optional[torch.Tensor] = None,
               residual: Optional[torch.Tensor] = None, outs: Optional[int] = None) -> torch.Tensor:
    """fp16 [tokens, outs] = dequant(a) @ dequant(w).T + bias on the FP4 matrix cores; with gate / residual the
    AdaLN block's `residual + y.mul(gate)` (tr/basic_var.py:264) is applied in the epilogue, bit-identical to the two
    torch ops on the plain result."""
    if _native is not None:   # same checks, same C calls (fpq_gemm_fp4_mx_ex / fpq_gemm_fp4_mx_km)
        return _native.linear_fp4(a_codes, a_scales, w_codes, w_scales, bias, gate, residual, outs)
    require_gpu(a_codes, "linear_fp4")
    km = _kmajor_pair("linear_fp4", a_codes, w_codes, 64)
    if km:
        tokens, outs, k = _check_kmajor_fp4("linear_fp4", a_codes, a_scales, w_codes, w_scales, bias, outs)
    else:
        tokens, outs, k = a_codes.shape[0], w_codes.shape[0], a_codes.shape[1] * 2
        if w_codes.shape[1] * 2 != k:
            raise RuntimeError("linear_fp4: operand shapes mismatch")
        if a_scales.dtype != torch.float16 or k % 128 != 0:
            raise RuntimeError("linear_fp4: operand shapes / activation scale dtype mismatch")
        _check_operand("linear_fp4(activation)", a_codes, a_scales, tokens, k // 2, tokens * (k // 128), a_codes.device)
        _check_operand("linear_fp4(weight)", w_codes, w_scales, outs, k // 2, outs * (k // 128), a_codes.device)
    ep, keep, out = _epilogue("linear_fp4", tokens, outs, gate, residual, None, a_codes.device)
    b = None if bias is None else bias.detach().to(torch.float16).reshape(-1).contiguous()
    if km and b is not None and b.data_ptr() % 16:
        b = b.clone()
    fn, what = (lib().fpq_gemm_fp4_mx_km, "fpq_gemm_fp4_mx_km") if km else (lib().fpq_gemm_fp4_mx_ex, "fpq_gemm_fp4_mx_ex")
    with device_guard(a_codes.device):
        check(fn(a_codes.data_ptr(), a_scales.data_ptr(), w_codes.data_ptr(), w_scales.data_ptr(), dtype_id(w_scales.dtype),
                 None if b is None else b.data_ptr(), out.data_ptr(), tokens, outs, k, ep, stream_ptr(a_codes.device)), what)
    del keep
    return out


def linear_fp4_qkv_to_cache(a_codes: torch.Tensor, a_scales: torch.Tensor, w_codes: torch.Tensor, w_scales: torch.Tensor,
                            bias: Optional[torch.Tensor], cache_kv: torch.Tensor, pos: int, seq: int) -> torch.Tensor:
    """mat_qkv of an attention block with a split output (fpq_gemm_fp4_mx_split): `qkv = linear_fp4(a, w, bias)` for tokens
    [B * seq] and outs = 3 * C, but only q comes back - fp16 [B, seq, C] - while k and v are written straight into the KV cache
    `cache_kv` [2, B, max_len, H, c] (C = H * c) at token positions pos .. pos + seq: what `qkv.view(B, L, 3, H, c).unbind(2)`
    followed by the cache's copy-in (tr/basic_var.py:173-209; kv_cache.IncrementalKVCache.append) leaves there, without the copy.
    Operands row-major (2-D) or k-major images (3-D) as in linear_fp4."""
    from ._lib import GemmSplit
    require_gpu(a_codes, "linear_fp4_qkv_to_cache")
    km = _kmajor_pair("linear_fp4_qkv_to_cache", a_codes, w_codes, 64)
    dev = a_codes.device
    if cache_kv.dim() != 5 or cache_kv.shape[0] != 2 or cache_kv.dtype != torch.float16 or not cache_kv.is_contiguous() or cache_kv.device != dev:
        raise RuntimeError("linear_fp4_qkv_to_cache: cache_kv must be a contiguous float16 [2, B, max_len, H, c] tensor on the operands' device")
    _, bsz, max_len, heads, hd = cache_kv.shape
    c = heads * hd
    if km:
        tokens, outs, k = _check_kmajor_fp4("linear_fp4_qkv_to_cache", a_codes, a_scales, w_codes, w_scales, bias, 3 * c)
    else:
        tokens, outs, k = a_codes.shape[0], w_codes.shape[0], a_codes.shape[1] * 2
        if w_codes.shape[1] * 2 != k or a_scales.dtype != torch.float16 or k % 128 != 0:
            raise RuntimeError("linear_fp4_qkv_to_cache: operand shapes / activation scale dtype mismatch")
        _check_operand("linear_fp4_qkv_to_cache(activation)", a_codes, a_scales, tokens, k // 2, tokens * (k // 128), dev)
        _check_operand("linear_fp4_qkv_to_cache(weight)", w_codes, w_scales, outs, k // 2, outs * (k // 128), dev)
    if outs != 3 * c or c % 128 != 0 or seq < 1 or tokens != bsz * seq or pos < 0 or pos + seq > max_len:
        raise RuntimeError(f"linear_fp4_qkv_to_cache: {tokens} tokens x {outs} outputs do not fit a cache of [{bsz}, {max_len}, {heads}, {hd}] at {pos} .. {pos + seq}")
    q = torch.empty((bsz, seq, c), dtype=torch.float16, device=dev)
    b = None
    if bias is not None:
        if bias.numel() != outs or bias.device != dev:
            raise RuntimeError("linear_fp4_qkv_to_cache: bias must hold one value per output on the operands' device")
        b = bias.detach().to(torch.float16).reshape(-1).contiguous()
        if b.data_ptr() % 16:
            b = b.clone()
    sp = GemmSplit()
    sp.part_cols, sp.n_parts, sp.rows_per_batch = c, 3, seq
    for p, (t, bstride, row0) in enumerate(((q, seq, 0), (cache_kv[0], max_len, pos), (cache_kv[1], max_len, pos))):
        sp.out[p], sp.row_stride[p], sp.batch_stride[p], sp.row0[p] = t.data_ptr(), c, bstride, row0
    if tokens:
        with device_guard(dev):
            check(lib().fpq_gemm_fp4_mx_split(a_codes.data_ptr(), a_scales.data_ptr(), w_codes.data_ptr(), w_scales.data_ptr(),
                                              dtype_id(w_scales.dtype), None if b is None else b.data_ptr(), tokens, outs, k,
                                              ctypes.byref(sp), 1 if km else 0, stream_ptr(dev)), "fpq_gemm_fp4_mx_split")
    return q


def linear_fp4_gelu_dual(a_codes: torch.Tensor, a_scales: torch.Tensor, w_codes: torch.Tensor, w_scales: torch.Tensor,
                         bias: Optional[torch.Tensor] = None, return_gelu: bool = False, outs: Optional[int] = None):
    """fc1 of the AdaLN block's FFN up to fc2's GEMM in ONE launch (+ the dual quantizer's tiny NaN fix-up launch):
    `fp_quant_e1m2_neg_e2m1_pos_per_group_cuda(F.gelu(linear_fp4(a, w, bias), approximate="tanh"), 4, 128)`
    (tr/basic_var.py:120-121, tr/quant_utils.py:415-452,991) as the epilogue of the FP4 GEMM (fpq_gemm_fp4_gelu_dual):
    fp16 [tokens, outs], outs % 128 == 0.  return_gelu: also the GELU values the quantizer saw - the quantization is
    bit-exact on THOSE, they sit within one fp16 ulp of torch's GELU of the Linear output."""
    if _native is not None:   # same checks, same C calls, the binding's own NaN scratch
        out, h = _native.linear_fp4_gelu_dual(a_codes, a_scales, w_codes, w_scales, bias, return_gelu, outs)
        return (out, h) if return_gelu else out
    require_gpu(a_codes, "linear_fp4_gelu_dual")
    km = _kmajor_pair("linear_fp4_gelu_dual", a_codes, w_codes, 64)
    dev = a_codes.device
    if km:
        tokens, outs, k = _check_kmajor_fp4("linear_fp4_gelu_dual", a_codes, a_scales, w_codes, w_scales, bias, outs)
    else:
        tokens, outs, k = a_codes.shape[0], w_codes.shape[0], a_codes.shape[1] * 2
        if w_codes.shape[1] * 2 != k or a_scales.dtype != torch.float16 or k % 128 != 0:
            raise RuntimeError("linear_fp4_gelu_dual: operand shapes / activation scale dtype mismatch")
        _check_operand("linear_fp4_gelu_dual(activation)", a_codes, a_scales, tokens, k // 2, tokens * (k // 128), dev)
        _check_operand("linear_fp4_gelu_dual(weight)", w_codes, w_scales, outs, k // 2, outs * (k // 128), dev)
    if outs % 128 != 0:
        raise RuntimeError("linear_fp4_gelu_dual: outs must be a multiple of 128")
    out = torch.empty((tokens, outs), dtype=torch.float16, device=dev)
    h = torch.empty((tokens, outs), dtype=torch.float16, device=dev) if return_gelu else None
    b = None
    if bias is not None:
        if bias.numel() != outs or bias.device != dev:
            raise RuntimeError("linear_fp4_gelu_dual: bias must hold one value per output on the operands' device")
        b = bias.detach().to(torch.float16).reshape(-1).contiguous()
        if b.data_ptr() % 16:
            b = b.clone()
    if tokens and outs:
        from .ops import _nan_scratch
        with device_guard(dev):
            flag = _nan_scratch(dev)
            fn = lib().fpq_gemm_fp4_gelu_dual_km if km else lib().fpq_gemm_fp4_gelu_dual
            check(fn(a_codes.data_ptr(), a_scales.data_ptr(), w_codes.data_ptr(), w_scales.data_ptr(),
                     dtype_id(w_scales.dtype), None if b is None else b.data_ptr(), out.data_ptr(),
                     None if h is None else h.data_ptr(), tokens, outs, k, flag.data_ptr(),
                     stream_ptr(dev)), "fpq_gemm_fp4_gelu_dual_km" if km else "fpq_gemm_fp4_gelu_dual")
    return (out, h) if return_gelu else out


class FP4Linear(_ScaledOperandModule):
    """Drop-in for QuantizedLinear in the W4A4 per-group `fp_e2` configuration that runs on the FP4
    matrix cores instead of simulating FP4 in fp16: weights are stored as hardware E2M1 codes + one
    fp32 scale per 128 input channels (4.25 bits per weight instead of 16), the
    activation is quantized to codes on the fly, the product is `fpq_gemm_fp4_mx`.
    Same quantization decisions as the reference (codes * scale == its fake-quantized tensors bit for
    bit); the GEMM itself is more exact than the reference's fp16 GEMM (tolerance-level agreement)."""

    def __init__(self, w_codes, w_scales, bias, in_features, out_features):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.register_buffer("w_codes", w_codes)
        self.register_buffer("w_scales", w_scales)
        self.register_buffer("bias", bias)

    @property
    def kmajor(self) -> bool:
        """the weight is held as a k-major image (to_kmajor(..., dealt=True)): activations must come as images too"""
        return self.w_codes.dim() == 3

    @classmethod
    def from_float(cls, module: torch.nn.Linear, kmajor: bool = False):
        assert isinstance(module, torch.nn.Linear) and module.in_features % 128 == 0 and module.out_features % 8 == 0
        codes, scales = quantize_mx(module.weight.detach().float())
        if kmajor:
            codes, scales = to_kmajor(codes, 4, dealt=True), to_kmajor_scales(scales, weight_side=True)
        bias = None if module.bias is None else module.bias.detach().to(torch.float16)
        return cls(codes, scales, bias, module.in_features, module.out_features)

    @torch.no_grad()
    def forward(self, x, gate=None, residual=None):
        """gate / residual: the AdaLN block's `residual + y.mul(gate)` fused into the GEMM (see linear_fp4)."""
        lead = x.shape[:-1]
        a_codes, a_scales = quantize_mx(x.to(torch.float16).reshape(-1, self.in_features), kmajor=self.kmajor)
        y = linear_fp4(a_codes, a_scales, self.w_codes, self.w_scales, self.bias, gate, residual, outs=self.out_features)
        return y.view(*lead, self.out_features)

    @torch.no_grad()
    def forward_operands(self, a_codes: torch.Tensor, a_scales: torch.Tensor, gate=None, residual=None) -> torch.Tensor:
        """The same product for an activation that already is in operand form - what the fused producers
        `rotation.rotate_quant_mx` / `rotation.adaln_rotate_quant_mx` emit: fp16 [tokens, out_features]."""
        return linear_fp4(a_codes, a_scales, self.w_codes, self.w_scales, self.bias, gate, residual, outs=self.out_features)


class FP4LinearGeluDual(FP4Linear):
    """fc1 of an AdaLN block's FFN in the W4A4 per-group configuration with the FFN's GELU(tanh) AND fc2's dual-format input
    quantizer (`fp_e1m2_neg_e2m1_pos`, tr/quant_utils.py:415-452,991) in the GEMM's epilogue: `forward(x)` returns what
    `fc2.act_quant(act(fc1(x)))` returns in the reference's FFN.forward (tr/basic_var.py:120-121) - the module that follows
    must neither apply the activation nor quantize again (`quant_linear.quantize_VAR(..., real_fp4=True, fuse_ffn=True)` swaps
    the FFN's `act` for an identity and switches fc2's input quantizer off)."""

    @torch.no_grad()
    def forward(self, x):
        lead = x.shape[:-1]
        a_codes, a_scales = quantize_mx(x.to(torch.float16).reshape(-1, self.in_features), kmajor=self.kmajor)
        return linear_fp4_gelu_dual(a_codes, a_scales, self.w_codes, self.w_scales, self.bias, outs=self.out_features).view(*lead, self.out_features)

    @torch.no_grad()
    def forward_operands(self, a_codes: torch.Tensor, a_scales: torch.Tensor) -> torch.Tensor:
        return linear_fp4_gelu_dual(a_codes, a_scales, self.w_codes, self.w_scales, self.bias, outs=self.out_features)


# ---- per-token activations x per-channel weights (W6A6): one scale per row, FP8-coded levels ---------------------
_FP8_TABLES = {"fp6_e2m3": "e2m3", "fp6_e3m2": "e3m2", "fp_e2": "e2m1", "fp_e1": "e1m2", "fp_e3": "e3m0",
               "e2m3": "e2m3", "e3m2": "e3m2", "e2m1": "e2m1", "e1m2": "e1m2", "e3m0": "e3m0"}


def quantize_fp8(x: torch.Tensor, table: str = "e2m3") -> Tuple[torch.Tensor, torch.Tensor]:
    """x [..., K] fp16/fp32 -> (codes uint8 [rows, K]: the level of every element as an OCP E4M3 byte,
    scales [rows] in x.dtype); one scale per row of K elements, e4m3(code) * scale == the fake-quantized value."""
    require_gpu(x, "quantize_fp8")
    if x.dtype not in (torch.float16, torch.float32):
        raise RuntimeError(f"quantize_fp8: x must be float16 or float32, got {x.dtype}")
    from ._lib import TABLE_IDS
    k = x.shape[-1]
    xc = x.contiguous()
    rows = xc.numel() // k
    codes = torch.empty((rows, k), dtype=torch.uint8, device=x.device)
    scales = torch.empty((rows,), dtype=x.dtype, device=x.device)
    with device_guard(x.device):
        check(lib().fpq_quant_rows_codes_fp8(xc.data_ptr(), codes.data_ptr(), scales.data_ptr(), rows, k,
                                             TABLE_IDS[_FP8_TABLES[table]], dtype_id(x.dtype), stream_ptr(x.device)),
              "fpq_quant_rows_codes_fp8")
    return codes, scales


def dequantize_fp8(codes: torch.Tensor, scales: torch.Tensor) -> torch.Tensor:
    """Reference decoder in torch ops (tests / debugging): fp32 [rows, K]."""
    return codes.view(torch.float8_e4m3fn).float() * scales.float().unsqueeze(-1)


def linear_fp8(a_codes: torch.Tensor, a_scales: torch.Tensor, w_codes: torch.Tensor, w_scales: torch.Tensor,
               bias: Optional[torch.Tensor] = None, gate: Optional[torch.Tensor] = None,
               residual: Optional[torch.Tensor] = None) -> torch.Tensor:
    """fp16 [tokens, outs] = dequant(a) @ dequant(w).T + bias on the FP8 matrix cores (row-scaled operands); optional
    fused `residual + y.mul(gate)` as in linear_fp4."""
    require_gpu(a_codes, "linear_fp8")
    if a_codes.dim() != 2 or w_codes.dim() != 2:
        raise RuntimeError("linear_fp8: codes must be [rows, K]")
    tokens, outs, k = a_codes.shape[0], w_codes.shape[0], a_codes.shape[1]
    if w_codes.shape[1] != k:
        raise RuntimeError("linear_fp8: operand shapes mismatch")
    _check_operand("linear_fp8(activation)", a_codes, a_scales, tokens, k, tokens, a_codes.device)
    _check_operand("linear_fp8(weight)", w_codes, w_scales, outs, k, outs, a_codes.device)
    ep, keep, out = _epilogue("linear_fp8", tokens, outs, gate, residual, None, a_codes.device)
    b = None if bias is None else bias.detach().to(torch.float16).reshape(-1).contiguous()
    with device_guard(a_codes.device):
        check(lib().fpq_gemm_fp8_rows_ex(a_codes.data_ptr(), a_scales.data_ptr(), dtype_id(a_scales.dtype), w_codes.data_ptr(),
                                         w_scales.data_ptr(), dtype_id(w_scales.dtype), None if b is None else b.data_ptr(),
                                         out.data_ptr(), tokens, outs, k, ep, stream_ptr(a_codes.device)), "fpq_gemm_fp8_rows_ex")
    del keep
    return out


class FP8Linear(_ScaledOperandModule):
    """Drop-in for QuantizedLinear in the per_channel / per_token configurations (W6A6 `fp6_e2m3` / `fp6_e3m2`,
    run.sh:7) on the FP8 matrix cores: same quantization decisions as the reference (e4m3(code) * scale == its
    fake-quantized tensors), weights stored as one byte per element + one fp32 scale per output channel."""

    def __init__(self, w_codes, w_scales, bias, in_features, out_features, act_table):
        super().__init__()
        self.in_features, self.out_features, self.act_table = in_features, out_features, act_table
        self.register_buffer("w_codes", w_codes)
        self.register_buffer("w_scales", w_scales)
        self.register_buffer("bias", bias)

    @classmethod
    def from_float(cls, module: torch.nn.Linear, weight_fp_type: str = "fp6_e2m3", act_fp_type: str = "fp6_e2m3"):
        assert isinstance(module, torch.nn.Linear) and module.in_features % 128 == 0 and module.out_features % 8 == 0
        codes, scales = quantize_fp8(module.weight.detach().float(), weight_fp_type)
        bias = None if module.bias is None else module.bias.detach().to(torch.float16)
        return cls(codes, scales, bias, module.in_features, module.out_features, act_fp_type)

    @torch.no_grad()
    def forward(self, x, gate=None, residual=None):
        lead = x.shape[:-1]
        a_codes, a_scales = quantize_fp8(x.to(torch.float16).reshape(-1, self.in_features), self.act_table)
        return linear_fp8(a_codes, a_scales, self.w_codes, self.w_scales, self.bias, gate, residual).view(*lead, self.out_features)


# ---- the same with 6-bit packed operands (FP6 E2M3 on both sides: the W6A6 run configuration) ---------------------
def quantize_fp6(x: torch.Tensor, kmajor: bool = False) -> Tuple[torch.Tensor, torch.Tensor]:
    """x [..., K] fp16/fp32 (K % 32 == 0) -> (codes uint8 [rows, K * 3 / 4]: dense 6-bit E2M3 codes,
    scales [rows] in x.dtype); e2m3(code) * scale == fp6_quant_e2m3_per_token_cuda(x).
    kmajor (K % 128 == 0): the codes as the activation side's k-major image [K/128, rows, 96]."""
    require_gpu(x, "quantize_fp6")
    if x.dtype not in (torch.float16, torch.float32):
        raise RuntimeError(f"quantize_fp6: x must be float16 or float32, got {x.dtype}")
    from ._lib import TABLE_IDS
    k = x.shape[-1]
    if k % 32 != 0:
        raise RuntimeError("quantize_fp6: the last dimension must be a multiple of 32")
    xc = x.contiguous()
    rows = xc.numel() // k
    if kmajor and k % 128 != 0:
        raise RuntimeError("quantize_fp6(kmajor=True): the last dimension must be a multiple of 128")
    codes = torch.empty((k // 128, rows, 96) if kmajor else (rows, k * 3 // 4), dtype=torch.uint8, device=x.device)
    scales = torch.empty((rows,), dtype=x.dtype, device=x.device)
    fn = lib().fpq_quant_rows_codes_fp6_km if kmajor else lib().fpq_quant_rows_codes_fp6
    with device_guard(x.device):
        check(fn(xc.data_ptr(), codes.data_ptr(), scales.data_ptr(), rows, k, TABLE_IDS["e2m3"], dtype_id(x.dtype),
                 stream_ptr(x.device)), "fpq_quant_rows_codes_fp6_km" if kmajor else "fpq_quant_rows_codes_fp6")
    return codes, scales


def dequantize_fp6(codes: torch.Tensor, scales: torch.Tensor) -> torch.Tensor:
    """Reference decoder in torch ops (tests / debugging): fp32 [rows, K]."""
    b = codes.reshape(codes.shape[0], -1, 3).to(torch.int32)
    word = b[..., 0] | (b[..., 1] << 8) | (b[..., 2] << 16)
    c = torch.stack((word & 63, (word >> 6) & 63, (word >> 12) & 63, (word >> 18) & 63), dim=-1).reshape(codes.shape[0], -1)
    e, m = (c >> 3) & 3, (c & 7).float()
    mag = torch.where(e == 0, m / 8.0, (1.0 + m / 8.0) * torch.pow(2.0, (e - 1).float()))
    val = torch.where((c & 32) != 0, -mag, mag)
    return val * scales.float().unsqueeze(-1)


def linear_fp6(a_codes: torch.Tensor, a_scales: torch.Tensor, w_codes: torch.Tensor, w_scales: torch.Tensor,
               bias: Optional[torch.Tensor] = None, gate: Optional[torch.Tensor] = None,
               residual: Optional[torch.Tensor] = None) -> torch.Tensor:
    """fp16 [tokens, outs] = dequant(a) @ dequant(w).T + bias on the FP6 matrix cores (row-scaled operands); optional
    fused `residual + y.mul(gate)` as in linear_fp4."""
    require_gpu(a_codes, "linear_fp6")
    km = _kmajor_pair("linear_fp6", a_codes, w_codes, 96)
    if km:
        tokens, outs, k = a_codes.shape[1], w_scales.shape[0], a_codes.shape[0] * 128
        w_rows, row_bytes = (outs + 63) // 64 * 64, a_codes.shape[0] * 96
    else:
        tokens, outs, k = a_codes.shape[0], w_codes.shape[0], a_codes.shape[1] * 4 // 3
        w_rows, row_bytes = outs, a_codes.shape[1]
        if w_codes.shape[1] != a_codes.shape[1] or a_codes.shape[1] % 3 != 0:
            raise RuntimeError("linear_fp6: operand shapes mismatch")
    _check_operand("linear_fp6(activation)", a_codes, a_scales, tokens, row_bytes, tokens, a_codes.device)
    _check_operand("linear_fp6(weight)", w_codes, w_scales, w_rows, row_bytes, outs, a_codes.device)
    ep, keep, out = _epilogue("linear_fp6", tokens, outs, gate, residual, None, a_codes.device)
    b = None if bias is None else bias.detach().to(torch.float16).reshape(-1).contiguous()
    fn, what = (lib().fpq_gemm_fp6_rows_km, "fpq_gemm_fp6_rows_km") if km else (lib().fpq_gemm_fp6_rows_ex, "fpq_gemm_fp6_rows_ex")
    with device_guard(a_codes.device):
        check(fn(a_codes.data_ptr(), a_scales.data_ptr(), dtype_id(a_scales.dtype), w_codes.data_ptr(), w_scales.data_ptr(),
                 dtype_id(w_scales.dtype), None if b is None else b.data_ptr(), out.data_ptr(), tokens, outs, k, ep,
                 stream_ptr(a_codes.device)), what)
    del keep
    return out


class FP6Linear(_ScaledOperandModule):
    """FP8Linear with 6-bit packed operands, for E2M3 activations x E2M3 weights (run.sh:7): 0.75 byte per weight."""

    def __init__(self, w_codes, w_scales, bias, in_features, out_features):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.register_buffer("w_codes", w_codes)
        self.register_buffer("w_scales", w_scales)
        self.register_buffer("bias", bias)

    @property
    def kmajor(self) -> bool:
        return self.w_codes.dim() == 3

    @classmethod
    def from_float(cls, module: torch.nn.Linear, kmajor: bool = False):
        assert isinstance(module, torch.nn.Linear) and module.in_features % 128 == 0 and module.out_features % 8 == 0
        codes, scales = quantize_fp6(module.weight.detach().float())
        if kmajor:
            codes = to_kmajor(codes, 6, dealt=True)
        bias = None if module.bias is None else module.bias.detach().to(torch.float16)
        return cls(codes, scales, bias, module.in_features, module.out_features)

    @torch.no_grad()
    def forward(self, x, gate=None, residual=None):
        lead = x.shape[:-1]
        a_codes, a_scales = quantize_fp6(x.to(torch.float16).reshape(-1, self.in_features), kmajor=self.kmajor)
        return linear_fp6(a_codes, a_scales, self.w_codes, self.w_scales, self.bias, gate, residual).view(*lead, self.out_features)
