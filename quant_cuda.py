"""Drop-in for the reference's native extension module ``quant_cuda``.

The reference builds ``quant_cuda`` from quant/quant.cpp + quant/quant_kernel.cu and
exports exactly one function (quant/quant.cpp:27-29):

    quant(x: Tensor[N] float32|float64, y: Tensor[K] same device) -> (z, idx)

This module keeps that import name and signature and forwards to the gfx950
kernel behind the C ABI (``fpq_quant_nearest`` in include/fpq.h).  With the repo
root on ``sys.path`` the reference's own ``import quant_cuda`` resolves here.
"""
import torch

from fpqvar_amd import ops as _ops

try:   # the compiled binding (fpqvar_amd/csrc/quant_cuda_ext.cpp): what the reference's pybind module is to its kernel
    from fpqvar_amd import _native
except ImportError:   # pragma: no cover - __graft_entry__.build() always produces it
    _native = None


def _quant_ctypes(x: torch.Tensor, y: torch.Tensor):
    """z[i] = the entry of y nearest to x[i] (last index wins ties; NaN/Inf/out of
    reach -> 0.0).  The second output is all zeros, as in the reference, whose
    kernel never writes it (quant/quant_kernel.cu:18,49): it is returned as a
    zero-stride view of one zero so no memory is filled for it."""
    z = _ops.quant_nearest(x, y)
    idx = torch.zeros((), dtype=x.dtype, device=x.device).expand(x.shape)
    return z, idx


quant = _quant_ctypes if _native is None else _native.quant   # same C entry point (fpq_quant_nearest) either way
