"""Host-side mirror of the reference's FP fake-quant operator layer.

Same function names, argument meaning, dtype rules and assertions as
``models_fp_quant_transform_rotate/quant_utils.py`` (reference, "tr/"), so a
caller can switch ``from models_fp_quant_transform_rotate.quant_utils import ...``
to ``from fpqvar_amd.quant_utils import ...`` unchanged.  Each call is ONE HIP
launch on the current stream instead of ~11 torch ops + the scan kernel.

GPU tensors only: there is deliberately no CPU fallback (the CPU restatement
is test infrastructure under oracle/).
"""
from __future__ import annotations

import torch

from . import ops

try:   # the compiled binding (csrc/quant_cuda_ext.cpp, built by __graft_entry__.build_native): same C entry points, a
    from . import _native   # third of the host time per call; absent -> the ctypes path of ops.py (tests run both)
except ImportError:         # pragma: no cover - build() always produces it
    _native = None
if __import__("os").environ.get("FPQ_NO_NATIVE") == "1":   # the A/B tools time variant builds of the library through ctypes (_lib.use_variant)
    _native = None

# value tables, exactly as tr/quant_utils.py:233-235,458-500 spells them (host tensors)
fp4_e3m0_grid = torch.tensor([-16.0, -8.0, -4.0, -2.0, -1.0, -0.5, -0.25, 0.0, 0.25, 0.5, 1.0, 2.0, 4.0, 8.0, 16.0])
fp4_e2m1_grid = torch.tensor([-6.0, -4.0, -3.0, -2.0, -1.5, -1.0, -0.5, 0.0, 0.5, 1.0, 1.5, 2.0, 3.0, 4.0, 6.0])
fp4_e1m2_grid = torch.tensor([-1.75, -1.5, -1.25, -1.0, -0.75, -0.5, -0.25, 0.0,
                              0.25, 0.5, 0.75, 1.0, 1.25, 1.5, 1.75])


def _e2m3_pos():
    return [m * 0.125 for m in range(8)] + [(1 + m / 8) * 2.0 ** e for e in range(3) for m in range(8)]


def _e3m2_pos():
    return [m * 0.0625 for m in range(4)] + [(1 + m / 4) * 2.0 ** e for e in range(-2, 5) for m in range(4)]


fp6_e2m3_grid = torch.tensor([-v if v else 0.0 for v in reversed(_e2m3_pos())] + _e2m3_pos())
fp6_e3m2_grid = torch.tensor([-v if v else 0.0 for v in reversed(_e3m2_pos())] + _e3m2_pos())
int_neg_grid = torch.tensor([float(-v) if v else 0.0 for v in range(32, -1, -1)])
e2m3_pos_grid = torch.tensor(_e2m3_pos())


# ---- the reference's pure-torch quantizers (tr/quant_utils.py:237-262,285-310,333-358) ----
# argmin lookup (ties to the smaller value), clamp(x, -3, 3) first except e2_per_group,
# float32 result.  QuantizedLinear uses the per_token ones for per_channel weights and
# per_token activations in FP4 even on the GPU (tr/quant_utils.py:699-704,796-807).

def fp_quant_e3_per_token(x, n_bits):
    assert n_bits == 4
    return ops.quant_rows_argmin(x, "e3m0", x.shape[-1], clamp3=True)


def fp_quant_e2_per_token(x, n_bits):
    assert n_bits == 4
    return ops.quant_rows_argmin(x, "e2m1", x.shape[-1], clamp3=True)


def fp_quant_e1_per_token(x, n_bits):
    assert n_bits == 4
    return ops.quant_rows_argmin(x, "e1m2", x.shape[-1], clamp3=True)


def fp_quant_e3_per_group(x, n_bits, group_size=128):
    assert n_bits == 4
    return ops.quant_rows_argmin(x, "e3m0", group_size, clamp3=True)


def fp_quant_e2_per_group(x, n_bits, group_size=128):
    """No +-3 clamp here (commented out in the reference, :302).  The reference divides
    its argument in place through a view; this version leaves x untouched."""
    assert n_bits == 4
    _require_input_viewable(x, group_size)
    return ops.quant_rows_argmin(x, "e2m1", group_size, clamp3=False)


def fp_quant_e1_per_group(x, n_bits, group_size=128):
    assert n_bits == 4
    return ops.quant_rows_argmin(x, "e1m2", group_size, clamp3=True)


# ---- per tensor (BASELINE.json config 1) ----
# search/baseline/plot_weight_distribution_for_motivation.py:285-294: ONE scale = x.abs().max() / 6 (float32, a 0-dim
# tensor), argmin lookup, returns (output float32, scale).  The script only spells the E2M1 one; the E1M2 / E3M0
# twins follow the same four lines with this library's tables.

def fp_quant_e2_per_tensor(x):
    return ops.quant_tensor_argmin(x, "e2m1")


def fp_quant_e1_per_tensor(x):
    return ops.quant_tensor_argmin(x, "e1m2")


def fp_quant_e3_per_tensor(x):
    return ops.quant_tensor_argmin(x, "e3m0")


# ---- FP4, per group of `group_size` consecutive elements (tr/quant_utils.py:265-282,313-330,361-378) ----

def fp_quant_e3_per_group_cuda(x, n_bits, group_size=128):
    assert n_bits == 4
    return ops.quant_rows(x, "e3m0", group_size)


def fp_quant_e2_per_group_cuda(x, n_bits, group_size=128):
    assert n_bits == 4
    return ops.quant_rows(x, "e2m1", group_size)


def fp_quant_e1_per_group_cuda(x, n_bits, group_size=128):
    assert n_bits == 4
    return ops.quant_rows(x, "e1m2", group_size)


if _native is not None:   # the compiled functions carry the same names, defaults and checks
    fp_quant_e3_per_group_cuda = _native.fp_quant_e3_per_group_cuda
    fp_quant_e2_per_group_cuda = _native.fp_quant_e2_per_group_cuda
    fp_quant_e1_per_group_cuda = _native.fp_quant_e1_per_group_cuda


# ---- FP4 asymmetric dual format for the fc2 input (tr/quant_utils.py:415-452) ----

def fp_quant_e1m2_neg_e2m1_pos_per_group_cuda(x, n_bits, group_size=128, clipping_strength=1.0):
    """x <= 0 on E1M2 (scale max|x_neg|/1.75), x > 0 on E2M1 (scale max x_pos/6), after the
    reference's global clamp to +-clipping_strength*max|x| (see ops.quant_rows_dual for how the
    default strength 1.0 avoids the extra pass while keeping the NaN behaviour)."""
    assert n_bits == 4
    return ops.quant_rows_dual(x, "e1m2_neg", "e2m1_pos", group_size, clipping_strength)


if _native is not None:
    fp_quant_e1m2_neg_e2m1_pos_per_group_cuda = _native.fp_quant_e1m2_neg_e2m1_pos_per_group_cuda


def gelu_fp_quant_e1m2_neg_e2m1_pos_per_group_cuda(y, n_bits, group_size=128):
    """`fp_quant_e1m2_neg_e2m1_pos_per_group_cuda(F.gelu(y, approximate="tanh"), n_bits, 128)` - what the reference's FFN feeds fc2's
    GEMM (tr/basic_var.py:120-121: `self.fc2(self.act(self.fc1(x)))`, the quantizer bound at tr/quant_utils.py:991) - in one pass
    over the fc1 output y instead of two (additive, build-defined name: Level 1b of INTEGRATION.md; the GELU is within one fp16
    ulp of torch's on every input, the quantization exact on it)."""
    assert n_bits == 4 and group_size == 128
    return ops.gelu_quant_rows_dual(y, "e1m2_neg", "e2m1_pos")


def gelu_fp4_afpq_per_group_cuda(y, n_bits, group_size=128):
    """`fp4_afpq_per_group_cuda(F.gelu(y, approximate="tanh"), n_bits, 128)` (models_fp_quant/quant_utils.py:498-535 behind the
    FFN's activation) in one pass."""
    assert n_bits == 4 and group_size == 128
    return ops.gelu_quant_rows_dual(y, "e2m1_neg", "e2m1_pos")


def gelu_fp6_quant_int_neg_e2m3_pos_per_token_cuda(y, n_bits):
    """`fp6_quant_int_neg_e2m3_pos_per_token_cuda(F.gelu(y, approximate="tanh"), n_bits)` - fc2's input in the W6A6 run
    (run.sh:7; tr/quant_utils.py:614-646 bound at :930-931) - in one pass over the fc1 output (the layout rule of the
    reference's `.view(-1)` applies: y must be viewable as rows of its last dimension)."""
    assert n_bits == 6
    _require_viewable(y, dual=True)
    return ops.gelu_quant_rows_dual(y, "int_neg", "e2m3_pos", y.shape[-1], None)


def gelu_fp6_quant_int_neg_e2m3_pos_per_group_cuda(y, n_bits, group_size=128):
    """The per-group twin (tr/quant_utils.py:577-611) with the GELU in front, one pass."""
    assert n_bits == 6 and group_size == 128
    return ops.gelu_quant_rows_dual(y, "int_neg", "e2m3_pos", 128, None)


def quantize_to_nearest_grid(x, quant_grid):
    """tr/quant_utils.py:209-230 in one launch (no [N, K] distance tensor): quant_grid[argmin |x - quant_grid|]."""
    return ops.quant_nearest_argmin(x, quant_grid)


def fp_quant_e1m2_neg_e2m1_pos_per_group(x, n_bits, group_size=128, clipping_strength=1.0):
    """tr/quant_utils.py:381-412, the pure-torch twin (argmin lookup on both halves, float32 result; bound by
    models_fp_quant_rotate's QuantizedLinear_fc2).  Its treatment of a group without negatives - every element
    gets -1.75 added before scaling - is the reference's and is reproduced."""
    assert n_bits == 4
    return ops.quant_rows_dual_argmin(x, "e1m2_neg", "e2m1_pos", group_size, clipping_strength)


def fp4_afpq_per_group_cuda(x, n_bits, group_size=128, clipping_strength=1.0):
    """models_fp_quant/quant_utils.py:498-535: like the function above but the negative side
    also uses the E2M1 levels (table [-6 .. 0], scale max|x_neg|/6)."""
    assert n_bits == 4
    return ops.quant_rows_dual(x, "e2m1_neg", "e2m1_pos", group_size, clipping_strength)


if _native is not None:
    fp4_afpq_per_group_cuda = _native.fp4_afpq_per_group_cuda


def fp_neg_reverse_quant_per_group_cuda(x, n_bits, group_size=128):
    """models_fp_quant/quant_utils.py:454-495: x <= 0 is shifted up by |group min|, quantized on E2M1
    and shifted back; x > 0 is quantized on E2M1 as usual."""
    assert n_bits == 4
    return ops.quant_rows_neg_reverse(x, "e2m1", group_size)


# ---- FP6 (tr/quant_utils.py:503-574): output is float16 whatever the input dtype ----

def fp6_quant_e2m3_per_token_cuda(x, n_bits):
    assert n_bits == 6
    if _native is not None and x.is_contiguous():
        return _native.fp6_quant_per_token_contig(x, n_bits, 3)
    _require_viewable(x)
    return ops.quant_rows(x, "e2m3", x.shape[-1], torch.float16)


def fp6_quant_e3m2_per_token_cuda(x, n_bits):
    assert n_bits == 6
    if _native is not None and x.is_contiguous():
        return _native.fp6_quant_per_token_contig(x, n_bits, 4)
    _require_viewable(x)
    return ops.quant_rows(x, "e3m2", x.shape[-1], torch.float16)


def fp6_quant_e2m3_per_group_cuda(x, n_bits, group_size=128):
    assert n_bits == 6
    return ops.quant_rows(x, "e2m3", group_size, torch.float16)


def fp6_quant_e3m2_per_group_cuda(x, n_bits, group_size=128):
    assert n_bits == 6
    return ops.quant_rows(x, "e3m2", group_size, torch.float16)


# ---- FP6 asymmetric dual format (tr/quant_utils.py:577-646) ----

def fp6_quant_int_neg_e2m3_pos_per_group_cuda(x, n_bits, group_size=128):
    assert n_bits == 6
    return ops.quant_rows_dual(x, "int_neg", "e2m3_pos", group_size, None)


if _native is not None:
    fp6_quant_e2m3_per_group_cuda = _native.fp6_quant_e2m3_per_group_cuda
    fp6_quant_e3m2_per_group_cuda = _native.fp6_quant_e3m2_per_group_cuda
    fp6_quant_int_neg_e2m3_pos_per_group_cuda = _native.fp6_quant_int_neg_e2m3_pos_per_group_cuda


def fp6_quant_int_neg_e2m3_pos_per_token_cuda(x, n_bits):
    assert n_bits == 6
    if _native is not None and x.is_contiguous():
        return _native.fp6_quant_int_neg_e2m3_pos_per_token_contig(x, n_bits)
    _require_viewable(x, dual=True)
    return ops.quant_rows_dual(x, "int_neg", "e2m3_pos", x.shape[-1], None)


def _meta(x):
    return torch.empty_strided(x.shape, x.stride(), dtype=x.dtype, device="meta")


def _require_viewable(x, dual=False):
    """The per-token reference functions flatten the RESULT of `x / scale` with .view(-1) (tr/quant_utils.py:510,
    527,633-634).  Whether that works is decided by the strides torch gives that result: a dense but permuted x
    (the BHLc attention layout) keeps its permutation and the view raises; a merely sliced x (k / v taken as
    qkv.unbind(2) views, tr/basic_var.py:187-194) yields a contiguous quotient and the view works.  The same torch
    ops on meta tensors (no memory, no launch) decide it here, with torch's own RuntimeError."""
    if x.is_contiguous():
        return
    xm = _meta(x)
    sm = torch.empty(tuple(x.shape[:-1]) + (1,), dtype=x.dtype, device="meta")
    if dual:   # :614-646 normalises torch.where(x <= 0, x, zeros_like(x))
        xm = torch.where(xm <= 0, xm, torch.zeros_like(xm))
    (xm / sm).view(-1)


def _require_input_viewable(x, group_size):
    """fp_quant_e2_per_group views its ARGUMENT as (-1, group_size) (tr/quant_utils.py:302): torch's own stride rule."""
    if not x.is_contiguous():
        _meta(x).view(-1, group_size)
