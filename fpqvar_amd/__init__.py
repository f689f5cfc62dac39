"""fpqvar_amd - MI355X (gfx950) fake-quantization kernels for FPQVAR's FP4/FP6 path.

Layout:
  csrc/fpq_kernels.hip   hand-written HIP kernels + the C ABI (include/fpq.h) -> libfpq_hip.so
  _lib.py                ctypes loader (fails loudly when the library is missing)
  ops.py                 one-launch tensor ops over the C ABI
  quant_utils.py         the reference's function names (tr/quant_utils.py) on top of ops
"""
from . import _lib, ops, quant_utils  # noqa: F401

__version__ = "0.1.0"
