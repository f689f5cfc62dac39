"""F4: a packed on-disk / wire format for calibrated weights.

The reference never stores quantized weights: every run re-does ``quantize_VAR`` on the fp32
checkpoint and keeps the de-quantized fp16 tensors (evaluate_fp_quant_transform_rotate.py:59-131).
Here calibration is a one-time artifact: per Linear the file holds the codewords of the sorted,
de-duplicated value table (4-bit: two per byte; 6-bit: four per three bytes), one scale per row of
the quantization view (a 128-group, or an output channel), and in the header the table name, the
row length, the weight's shape, and - for layers calibrated behind the online transforms - the
rotation block size / seed and the name of the GALT smoothing vector stored alongside.

``PackedWeight.dequantize`` reproduces, bit for bit, the tensor ``QuantizedLinear.from_float``
would have produced (`fpq_dequant_rows_codes` computes ``table[code] * scale`` with the same
roundings as the fake-quant kernels), so a model loaded from the packed file generates the same
images; ``PackedWeight.fp4_operands`` re-expresses a per-group E2M1 layer as the hardware nibbles the
FP4 matrix-core GEMM takes (`gemm.linear_fp4`).

Container: safetensors (one file, zero-copy mmap load, no pickle), format tag ``fpqvar-packed/1``.
"""
from __future__ import annotations

import json
from dataclasses import dataclass
from typing import Dict, Mapping, Optional, Tuple

import torch

FORMAT = "fpqvar-packed/1"
FP4_TABLES = ("e2m1", "e1m2", "e3m0")
FP6_TABLES = ("e2m3", "e3m2")


# ---- 6-bit packing: four codes (0..62) -> three bytes, little-endian bit order ------------------
def pack6(codes: torch.Tensor) -> torch.Tensor:
    """uint8 [rows, cols] (cols % 4 == 0, values < 64) -> uint8 [rows, cols * 3 / 4]."""
    if codes.dtype != torch.uint8 or codes.shape[-1] % 4 != 0:
        raise RuntimeError("pack6: need uint8 codes with a row length that is a multiple of 4")
    c = codes.reshape(codes.shape[0], -1, 4).to(torch.int32)
    word = c[..., 0] | (c[..., 1] << 6) | (c[..., 2] << 12) | (c[..., 3] << 18)
    out = torch.stack((word & 0xFF, (word >> 8) & 0xFF, (word >> 16) & 0xFF), dim=-1)
    return out.to(torch.uint8).reshape(codes.shape[0], -1)


def unpack6(packed: torch.Tensor) -> torch.Tensor:
    """Inverse of pack6."""
    if packed.dtype != torch.uint8 or packed.shape[-1] % 3 != 0:
        raise RuntimeError("unpack6: need uint8 data with a row length that is a multiple of 3")
    b = packed.reshape(packed.shape[0], -1, 3).to(torch.int32)
    word = b[..., 0] | (b[..., 1] << 8) | (b[..., 2] << 16)
    out = torch.stack((word & 63, (word >> 6) & 63, (word >> 12) & 63, (word >> 18) & 63), dim=-1)
    return out.to(torch.uint8).reshape(packed.shape[0], -1)


@dataclass
class PackedWeight:
    """One calibrated Linear weight.  codes: packed codewords [rows, bytes_per_row]; scales: [rows]
    (rows = numel / cols, cols = the quantization row length: 128 per-group, in_features per-channel)."""
    codes: torch.Tensor
    scales: torch.Tensor
    table: str
    cols: int
    shape: Tuple[int, ...]
    out_dtype: str = "float16"          # dtype the de-quantized weight has in the reference after .half()
    rotate_block: int = 0               # > 0: calibrated on W @ Q_block(seed), block size
    rotate_seed: int = 0
    smooth: Optional[torch.Tensor] = None   # GALT s (W was divided by it before rotation / quantization)
    bias: Optional[torch.Tensor] = None

    @property
    def bits(self) -> int:
        return 4 if self.table in FP4_TABLES else 6

    def nbytes(self) -> int:
        n = self.codes.numel() + self.scales.numel() * self.scales.element_size()
        for t in (self.smooth, self.bias):
            n += 0 if t is None else t.numel() * t.element_size()
        return n

    def dequantize(self, dtype: Optional[torch.dtype] = None) -> torch.Tensor:
        """The fake-quantized weight, bit-equal to QuantizedLinear.from_float(...).weight after .half()."""
        from . import ops
        dtype = getattr(torch, self.out_dtype) if dtype is None else dtype
        codes = self.codes if self.bits == 4 else unpack6(self.codes)
        w = ops.dequant_rows_codes(codes, self.scales, self.table, self.cols, dtype, pack_nibbles=self.bits == 4)
        return w.view(self.shape)

    def fp4_operands(self) -> Tuple[torch.Tensor, torch.Tensor]:
        """(hardware E2M1 nibbles [out, in/2], scales [out, in/128]) for gemm.linear_fp4."""
        if self.table != "e2m1" or self.cols != 128 or len(self.shape) != 2 or self.shape[1] % 128 != 0:
            raise RuntimeError("fp4_operands: only per-group(128) E2M1 weights map onto the FP4 matrix cores")
        # sorted-table index 0..14 (-6 .. 6, zero at 7) -> OCP nibble: sign bit 3, magnitude index 2:0
        lut = torch.tensor([8 | 7, 8 | 6, 8 | 5, 8 | 4, 8 | 3, 8 | 2, 8 | 1, 0, 1, 2, 3, 4, 5, 6, 7, 0],
                           dtype=torch.uint8, device=self.codes.device)
        lo, hi = lut[(self.codes & 0xF).long()], lut[(self.codes >> 4).long()]
        hw = (lo | (hi << 4)).view(self.shape[0], self.shape[1] // 2)
        return hw, self.scales.view(self.shape[0], self.shape[1] // 128)


def pack_weight(w: torch.Tensor, table: str = "e2m1", cols: int = 128, *, smooth: Optional[torch.Tensor] = None,
                rotate_block: int = 0, rotate_seed: int = 42, bias: Optional[torch.Tensor] = None,
                out_dtype: torch.dtype = torch.float16) -> PackedWeight:
    """Calibrate one weight straight to the packed form (one launch: `fpq_quant_rows_codes`).
    The transforms are applied in the reference's order (W / s, then W @ Q; rotate_model_utils.py,
    transform_model_utils.py) before quantization.  cols = 128 for per-group, w.shape[-1] per-channel."""
    from . import ops, rotation
    if table not in FP4_TABLES + FP6_TABLES:
        raise RuntimeError(f"pack_weight: table {table!r} has no packed form")
    wt = w.detach()
    if smooth is not None:
        wt = rotation.transform_weight(wt, smooth)
    if rotate_block:
        q = rotation.block_random_hadamard_matrix(wt.shape[-1], rotate_block, wt.device, rotate_seed)
        wt = rotation.rotate_weight(wt, q)
    four = table in FP4_TABLES
    codes, scales = ops.quant_rows_codes(wt, table, cols, pack_nibbles=four)
    if not four:
        codes = pack6(codes)
    return PackedWeight(codes, scales, table, cols, tuple(w.shape), str(out_dtype).replace("torch.", ""),
                        rotate_block, rotate_seed if rotate_block else 0,
                        None if smooth is None else smooth.detach().to(torch.float32),
                        None if bias is None else bias.detach())


def save_packed(path: str, layers: Mapping[str, PackedWeight], extra: Optional[Mapping[str, str]] = None) -> int:
    """Write all layers to one safetensors file; returns the number of tensor bytes written."""
    from safetensors.torch import save_file
    tensors: Dict[str, torch.Tensor] = {}
    header: Dict[str, dict] = {}
    for name, p in layers.items():
        tensors[f"{name}.codes"] = p.codes.cpu().contiguous()
        tensors[f"{name}.scales"] = p.scales.cpu().contiguous()
        if p.smooth is not None:
            tensors[f"{name}.smooth"] = p.smooth.cpu().contiguous()
        if p.bias is not None:
            tensors[f"{name}.bias"] = p.bias.cpu().contiguous()
        header[name] = {"table": p.table, "cols": p.cols, "shape": list(p.shape), "out_dtype": p.out_dtype,
                        "rotate_block": p.rotate_block, "rotate_seed": p.rotate_seed}
    meta = {"format": FORMAT, "layers": json.dumps(header)}
    meta.update({k: str(v) for k, v in (extra or {}).items()})
    save_file(tensors, path, metadata=meta)
    return sum(t.numel() * t.element_size() for t in tensors.values())


def _validate_entry(path, name, h, keys, f) -> None:
    """Header against tensor sizes: a truncated or inconsistent file is refused here, before any kernel gets a
    pointer whose extent it would trust."""
    for k in ("table", "cols", "shape", "out_dtype", "rotate_block", "rotate_seed"):
        if k not in h:
            raise RuntimeError(f"{path}: layer {name}: header field {k!r} is missing")
    if h["table"] not in FP4_TABLES + FP6_TABLES or h["out_dtype"] not in ("float16", "float32"):
        raise RuntimeError(f"{path}: layer {name}: unknown table / dtype {h['table']!r} / {h['out_dtype']!r}")
    if f"{name}.codes" not in keys or f"{name}.scales" not in keys:
        raise RuntimeError(f"{path}: layer {name}: codes / scales tensor missing")
    cols, numel = int(h["cols"]), 1
    for d in h["shape"]:
        numel *= int(d)
    if cols <= 0 or numel % cols != 0:
        raise RuntimeError(f"{path}: layer {name}: shape {h['shape']} is not a whole number of rows of {cols}")
    rows = numel // cols
    four = h["table"] in FP4_TABLES
    if not four and cols % 4 != 0:
        raise RuntimeError(f"{path}: layer {name}: 6-bit rows need a multiple of 4 columns")
    row_bytes = (cols + 1) // 2 if four else cols * 3 // 4
    cs, ss = f.get_slice(f"{name}.codes"), f.get_slice(f"{name}.scales")
    n_codes, n_scales = 1, 1
    for d in cs.get_shape():
        n_codes *= int(d)
    for d in ss.get_shape():
        n_scales *= int(d)
    if cs.get_dtype() != "U8" or n_codes != rows * row_bytes:
        raise RuntimeError(f"{path}: layer {name}: codes hold {n_codes} {cs.get_dtype()} values, expected {rows * row_bytes} bytes")
    if ss.get_dtype() not in ("F16", "F32") or n_scales != rows:
        raise RuntimeError(f"{path}: layer {name}: {n_scales} {ss.get_dtype()} scales, expected {rows}")


def load_packed(path: str, device="cpu") -> Dict[str, PackedWeight]:
    from safetensors import safe_open
    out: Dict[str, PackedWeight] = {}
    with safe_open(path, framework="pt", device=str(device)) as f:
        meta = f.metadata() or {}
        if meta.get("format") != FORMAT:
            raise RuntimeError(f"{path}: not a {FORMAT} file (format tag {meta.get('format')!r})")
        header = json.loads(meta["layers"])
        keys = set(f.keys())
        for name, h in header.items():
            _validate_entry(path, name, h, keys, f)
            out[name] = PackedWeight(
                f.get_tensor(f"{name}.codes"), f.get_tensor(f"{name}.scales"), h["table"], int(h["cols"]),
                tuple(h["shape"]), h["out_dtype"], int(h["rotate_block"]), int(h["rotate_seed"]),
                f.get_tensor(f"{name}.smooth") if f"{name}.smooth" in keys else None,
                f.get_tensor(f"{name}.bias") if f"{name}.bias" in keys else None)
    return out
