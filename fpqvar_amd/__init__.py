"""fpqvar_amd - MI355X (gfx950) fake-quantization kernels for FPQVAR's FP4/FP6 path.

Layout:
  csrc/                  hand-written HIP kernels + the C ABI (include/fpq.h) -> libfpq_hip.so
  _lib.py                ctypes loader (fails loudly when the library is missing; there is no CPU path)
  ops.py                 one-launch tensor ops over the C ABI
  quant_utils.py         the reference's function names (tr/quant_utils.py and its older variants) on top of ops
  quant_linear.py        QuantizedLinear / QuantizedLinear_fc2 / quantize_VAR (+ the per-block mixed-format variants)
  rotation.py            Hadamard pieces, transform_model / rotate_model, the fused producers (per-group and per-token)
  kv_cache.py            the reference's KV re-quantization step and the incremental cache
  gemm.py                real low-precision consumers: FP4Linear (W4A4 per-group), FP6Linear / FP8Linear (W6A6 rows)
  packed.py              packed on-disk format for calibrated weights
  calibrate.py, format_search.py, galt.py, generation.py   the sharded (multi-GPU) workloads
"""
from . import _lib, ops, quant_utils  # noqa: F401

__version__ = "0.1.1"
