"""CPU oracle for the FPQVAR fake-quant hot path.  TEST INFRASTRUCTURE ONLY.

This file is the checker, never the product: only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it.
The product path (``fpqvar_amd``) never routes through here and fails loudly
when the HIP library is missing.

It restates, table-driven, what the reference computes (reference root =
PKU-SEC-Lab/FPQVAR; ``tr/`` = models_fp_quant_transform_rotate/):

* ``nearest_kernel``         - quant/quant_kernel.cu:11-39 (the only native kernel)
* ``per_group_kernel_sem``   - tr/quant_utils.py:265-282, 313-330, 361-378, 537-574
* ``per_token_kernel_sem``   - tr/quant_utils.py:503-534
* ``dual_per_group_kernel_sem`` / ``dual_per_token_kernel_sem``
                             - tr/quant_utils.py:415-452, 577-646
* ``nearest_argmin`` + ``*_argmin_sem`` (the reference's pure-torch "CPU path")
                             - tr/quant_utils.py:209-230, 237-262, 285-310, 333-358, 381-412
* ``per_tensor_argmin_sem``  - search/baseline/plot_weight_distribution_for_motivation.py:286-297
* ``hadamard_block`` / ``sign_vector`` - rotate_utils/hadamard_utils.py:63-99,
                               rotate_utils/rotation_utils.py:69-104

Parity pinning: the reference has no tests or golden vectors.
The native kernel (CUDA) cannot be built in this image (no nvcc).  The oracle
is pinned against the reference's own Python imported from /root/reference
with a stub ``quant_cuda`` whose ``quant`` is ``nearest_kernel`` below; the
resulting input/output vectors are committed under tests/golden/ together with
the generating script (tests/golden/make_golden.py).

All arithmetic is done with torch CPU ops so that dtype promotion and fp16
rounding (compute in fp32, round once to fp16) match what torch does on a GPU
for these elementwise ops.
"""
from __future__ import annotations

import math
from typing import Dict, Tuple

import numpy as np
import torch

# ----------------------------------------------------------------------------
# Value tables (tr/quant_utils.py:233-235, 384-385, 458-500).  Built from the
# format definition instead of literals; tests compare them with the reference's
# literal tensors.
# ----------------------------------------------------------------------------


def _minifloat_pos(ebits: int, mbits: int, bias: int) -> list:
    """Non-negative values of a sign-magnitude minifloat with subnormals and no inf/nan."""
    vals = []
    for e in range(2 ** ebits):
        for m in range(2 ** mbits):
            if e == 0:
                v = (m / 2 ** mbits) * 2.0 ** (1 - bias)
            else:
                v = (1 + m / 2 ** mbits) * 2.0 ** (e - bias)
            vals.append(v)
    return vals


def _sym(pos: list, dup_zero: bool) -> list:
    neg = [(-v if v != 0 else 0.0) for v in reversed(pos)]   # the reference spells both zeros +0
    if not dup_zero:
        neg = neg[:-1]
    return neg + pos


TABLES: Dict[str, torch.Tensor] = {
    # 15 entries, single zero (tr/quant_utils.py:233-235)
    "e3m0": torch.tensor(_sym(_minifloat_pos(3, 0, 3), False), dtype=torch.float32),
    "e2m1": torch.tensor(_sym(_minifloat_pos(2, 1, 1), False), dtype=torch.float32),
    "e1m2": torch.tensor(_sym(_minifloat_pos(1, 2, 1), False), dtype=torch.float32),
    # 64 entries, two zeros (tr/quant_utils.py:458-486)
    "e2m3": torch.tensor(_sym(_minifloat_pos(2, 3, 1), True), dtype=torch.float32),
    "e3m2": torch.tensor(_sym(_minifloat_pos(3, 2, 3), True), dtype=torch.float32),
    # half tables for the asymmetric neg/pos dual formats (:384-385, :488-500)
    "e1m2_neg": torch.tensor(_sym(_minifloat_pos(1, 2, 1), True)[:8], dtype=torch.float32),
    "e2m1_pos": torch.tensor(_minifloat_pos(2, 1, 1), dtype=torch.float32),
    "int_neg": torch.tensor(_sym([float(v) for v in range(33)], True)[:33], dtype=torch.float32),
    "e2m3_pos": torch.tensor(_minifloat_pos(2, 3, 1), dtype=torch.float32),
    # negative half of fp4_afpq_per_group_cuda (models_fp_quant/quant_utils.py:501)
    "e2m1_neg": torch.tensor(_sym(_minifloat_pos(2, 1, 1), True)[:8], dtype=torch.float32),
}

TABLE_IDS = {name: i for i, name in enumerate(
    ["e2m1", "e1m2", "e3m0", "e2m3", "e3m2", "e1m2_neg", "e2m1_pos", "int_neg", "e2m3_pos", "e2m1_neg"])}


def table_absmax(name: str) -> float:
    return float(TABLES[name].abs().max())


# ----------------------------------------------------------------------------
# A1: the native kernel's semantics (quant/quant_kernel.cu:25-37)
# ----------------------------------------------------------------------------


def nearest_kernel(x: torch.Tensor, table: torch.Tensor) -> torch.Tensor:
    """z[i] = table[last j minimising |x[i]-table[j]|], scanning j upward with `<=`.

    best distance starts at 102400.0 and z at 0.0, so NaN, +-Inf and anything
    farther than 102400 from every entry give +0.0.  float64 input is compared
    in float32 (`float x_v = x[idx]`).  Output has x's dtype (float32/float64).
    """
    assert x.dtype in (torch.float32, torch.float64)
    xv = x.detach().to(torch.float32).reshape(-1)
    tab = table.detach().to(torch.float32).reshape(-1)
    best = torch.full_like(xv, 102400.0)
    z = torch.zeros_like(xv)
    for j in range(tab.numel()):
        d = (xv - tab[j]).abs()            # fabsf(x_v - y_shared[i]) in fp32
        take = d <= best                   # NaN compares false
        best = torch.where(take, d, best)
        z = torch.where(take, tab[j].expand_as(z), z)
    return z.to(x.dtype).reshape(x.shape)


def nearest_kernel_index(x: torch.Tensor, table: torch.Tensor) -> torch.Tensor:
    """Winning table index of ``nearest_kernel`` (-1 where nothing was selected).

    The reference never materialises this (its `tensor_idx` output stays zero,
    quant_kernel.cu:18,49); it is the build-defined codeword.
    """
    xv = x.detach().to(torch.float32).reshape(-1)
    tab = table.detach().to(torch.float32).reshape(-1)
    best = torch.full_like(xv, 102400.0)
    idx = torch.full(xv.shape, -1, dtype=torch.int64)
    for j in range(tab.numel()):
        d = (xv - tab[j]).abs()
        take = d <= best
        best = torch.where(take, d, best)
        idx = torch.where(take, torch.full_like(idx, j), idx)
    return idx.reshape(x.shape)


def nearest_closed_form(x: torch.Tensor, table: torch.Tensor) -> torch.Tensor:
    """Closed form used by the HIP kernels: count midpoints of the sorted,
    de-duplicated table that are <= x (ties go to the larger value), and 0.0 for
    NaN/Inf/out-of-reach.  Tests prove it equal to ``nearest_kernel``."""
    xv = x.detach().to(torch.float32).reshape(-1)
    tab = torch.unique(table.detach().to(torch.float32))  # sorted, de-duplicated
    mids = (tab[:-1] + tab[1:]) / 2                        # exact for these tables
    cnt = (xv[:, None] >= mids[None, :]).sum(dim=1)
    z = tab[cnt]
    reach = (xv - tab[0]).abs() <= 102400.0
    reach |= (xv - tab[-1]).abs() <= 102400.0
    z = torch.where(reach, z, torch.zeros_like(z))
    z = torch.where(z == 0, torch.zeros_like(z), z)        # +0.0, never -0.0
    return z.reshape(x.shape)


# ----------------------------------------------------------------------------
# A3/A4/A6/A7: "_cuda" functions = torch ops around the native kernel
# ----------------------------------------------------------------------------


def _rows_kernel_sem(x2d: torch.Tensor, table: torch.Tensor) -> torch.Tensor:
    """Shared body: x2d is [rows, n]; returns the fp32 product q*scale.

    scale = absmax_row / max|table| (in x.dtype; the table max is a 0-dim fp32
    tensor, which does not promote), xn = x / scale (in x.dtype), fp32 cast,
    kernel, q * scale (fp32 * x.dtype -> fp32).
    """
    gmax = table.abs().max()
    scale = x2d.abs().max(dim=-1, keepdim=True)[0] / gmax
    xn = x2d / scale
    q = nearest_kernel(xn.reshape(-1).to(torch.float32), table).view(xn.shape)
    return q * scale


def per_group_kernel_sem(x: torch.Tensor, table_name: str, group_size: int = 128,
                         out_dtype=None) -> torch.Tensor:
    """fp_quant_e{1,2,3}_per_group_cuda (tr/quant_utils.py:265-282,313-330,361-378):
    result cast back to x.dtype.  fp6_quant_*_per_group_cuda (:537-574): result
    always fp16 -> pass out_dtype=torch.float16."""
    table = TABLES[table_name]
    y = _rows_kernel_sem(x.reshape(-1, group_size), table)
    return y.view(x.shape).to(x.dtype if out_dtype is None else out_dtype)


def per_token_kernel_sem(x: torch.Tensor, table_name: str, out_dtype=torch.float16) -> torch.Tensor:
    """fp6_quant_e2m3/e3m2_per_token_cuda (tr/quant_utils.py:503-534): one scale
    per last-dim row, output hard-cast to fp16."""
    table = TABLES[table_name]
    y = _rows_kernel_sem(x.reshape(-1, x.shape[-1]), table)
    return y.view(x.shape).to(out_dtype)


def _dual_rows_kernel_sem(x2d, neg_name, pos_name):
    neg_t, pos_t = TABLES[neg_name], TABLES[pos_name]
    zeros = torch.zeros_like(x2d)
    x_neg = torch.where(x2d <= 0, x2d, zeros)
    x_pos = torch.where(x2d > 0, x2d, zeros)
    s_neg = x_neg.abs().max(dim=-1, keepdim=True)[0] / neg_t.abs().max()
    s_pos = x_pos.abs().max(dim=-1, keepdim=True)[0] / pos_t.abs().max()
    n_neg = (x_neg / s_neg).reshape(-1).to(torch.float32)
    n_pos = (x_pos / s_pos).reshape(-1).to(torch.float32)
    q_neg = nearest_kernel(n_neg, neg_t).view(x2d.shape)
    q_pos = nearest_kernel(n_pos, pos_t).view(x2d.shape)
    return q_neg * s_neg + q_pos * s_pos


def dual_per_group_kernel_sem(x, neg_name="e1m2_neg", pos_name="e2m1_pos", group_size=128,
                              clipping_strength=None) -> torch.Tensor:
    """fp_quant_e1m2_neg_e2m1_pos_per_group_cuda (tr/quant_utils.py:415-452) when
    clipping_strength is a number (global clamp to strength*max|x| first), and
    fp6_quant_int_neg_e2m3_pos_per_group_cuda (:577-611) when it is None."""
    if clipping_strength is not None:
        clip = clipping_strength * x.abs().max()
        x = torch.clamp(x, -clip, clip)
    y = _dual_rows_kernel_sem(x.reshape(-1, group_size), neg_name, pos_name)
    return y.view(x.shape).to(x.dtype)


def neg_reverse_per_group_kernel_sem(x, table_name="e2m1", group_size=128) -> torch.Tensor:
    """fp_neg_reverse_quant_per_group_cuda (models_fp_quant/quant_utils.py:454-495): the
    non-positive half is shifted up by |row min| and quantized on the symmetric table, shifted
    back after de-quantization; the positive half is quantized as usual.  The shift, both scales
    and both quotients live in x.dtype; products, the subtraction and the sum are fp32."""
    table = TABLES[table_name]
    xs = x.reshape(-1, group_size)
    m = xs.min(dim=-1, keepdim=True)[0].abs()
    zeros = torch.zeros_like(xs)
    x_neg = torch.where(xs <= 0, xs, zeros)
    x_pos = torch.where(xs > 0, xs, zeros)
    x_nr = x_neg + m
    s_nr = x_nr.abs().max(dim=-1, keepdim=True)[0] / table.abs().max()
    s_pos = x_pos.abs().max(dim=-1, keepdim=True)[0] / table.abs().max()
    q_nr = nearest_kernel((x_nr / s_nr).reshape(-1).to(torch.float32), table).view(xs.shape)
    q_pos = nearest_kernel((x_pos / s_pos).reshape(-1).to(torch.float32), table).view(xs.shape)
    y = (q_nr * s_nr - m) + q_pos * s_pos
    return y.view(x.shape).to(x.dtype)


def dual_per_token_kernel_sem(x, neg_name="int_neg", pos_name="e2m3_pos") -> torch.Tensor:
    """fp6_quant_int_neg_e2m3_pos_per_token_cuda (tr/quant_utils.py:614-646)."""
    y = _dual_rows_kernel_sem(x.reshape(-1, x.shape[-1]), neg_name, pos_name)
    return y.view(x.shape).to(x.dtype)


# ----------------------------------------------------------------------------
# A9: the reference's pure-torch "CPU path" (argmin semantics)
# ----------------------------------------------------------------------------


def nearest_argmin(x: torch.Tensor, table: torch.Tensor) -> torch.Tensor:
    """quantize_to_nearest_grid (tr/quant_utils.py:209-230): first minimal index
    (ties toward -inf); NaN distance rows resolve to whatever argmin returns."""
    tab = table.to(x.device)
    dist = (x.unsqueeze(-1) - tab).abs()
    return tab[dist.argmin(dim=-1)]


def per_group_argmin_sem(x, table_name, group_size=128, clamp3=False):
    """fp_quant_e{3,1}_per_group (clamp3=True, :250-262,:346-358) and
    fp_quant_e2_per_group (clamp3=False, :298-310; the reference version also
    divides its input in place - this restatement leaves the input untouched)."""
    table = TABLES[table_name]
    if clamp3:
        x = torch.clamp(x, -3, 3)
    x2 = x.reshape(-1, group_size)
    scale = x2.abs().max(dim=-1, keepdim=True)[0] / table.abs().max()
    y = nearest_argmin(x2 / scale, table) * scale
    return y.reshape(x.shape)


def per_token_argmin_sem(x, table_name):
    """fp_quant_e{3,2,1}_per_token (:237-247,:285-295,:333-343): clamp to +-3 first."""
    table = TABLES[table_name]
    x = torch.clamp(x, -3, 3)
    scale = x.abs().max(dim=-1, keepdim=True)[0] / table.abs().max()
    return nearest_argmin(x / scale, table) * scale


def per_tensor_argmin_sem(x, table_name="e2m1"):
    """BASELINE.json config 1: per-tensor scale, argmin lookup
    (search/baseline/plot_weight_distribution_for_motivation.py:286-297)."""
    table = TABLES[table_name]
    scale = x.abs().max() / table.abs().max()
    return nearest_argmin(x / scale, table) * scale


def dual_per_group_argmin_sem(x, neg_name="e1m2_neg", pos_name="e2m1_pos", group_size=128,
                              clipping_strength=1.0):
    """fp_quant_e1m2_neg_e2m1_pos_per_group (:381-412)."""
    neg_t, pos_t = TABLES[neg_name], TABLES[pos_name]
    clip = clipping_strength * x.abs().max()
    x = torch.clamp(x, -clip, clip)
    x2 = x.reshape(-1, group_size)
    zeros = torch.zeros_like(x2)
    x_neg = torch.where(x2 <= 0, x2, zeros)
    x_pos = torch.where(x2 > 0, x2, zeros)
    s_neg = x_neg.abs().max(dim=-1, keepdim=True)[0] / neg_t.abs().max()
    s_pos = x_pos.abs().max(dim=-1, keepdim=True)[0] / pos_t.abs().max()
    q = nearest_argmin(x_neg / s_neg, neg_t) + nearest_argmin(x_pos / s_pos, pos_t)
    return (q * torch.where(x2 <= 0, s_neg, s_pos)).reshape(x.shape)


# ----------------------------------------------------------------------------
# Codewords (build-defined)
# ----------------------------------------------------------------------------


def per_group_codes(x: torch.Tensor, table_name: str, group_size: int = 128
                    ) -> Tuple[torch.Tensor, torch.Tensor]:
    """(codes uint8 [same shape], scales [n_groups] in x.dtype): code = index into
    the sorted de-duplicated table, scale as in per_group_kernel_sem.  All-zero /
    non-finite groups give the code of 0.0."""
    table = TABLES[table_name]
    x2 = x.reshape(-1, group_size)
    scale = x2.abs().max(dim=-1, keepdim=True)[0] / table.abs().max()
    xn = (x2 / scale).reshape(-1).to(torch.float32)
    q = nearest_kernel(xn, table)
    uniq = torch.unique(table)
    codes = torch.searchsorted(uniq, q).to(torch.uint8)
    return codes.view(x.shape), scale.view(-1)


# ----------------------------------------------------------------------------
# A11: rotation matrix pieces
# ----------------------------------------------------------------------------


def sign_vector(n: int = 128, seed: int = 42) -> torch.Tensor:
    """D of random_hadamard_matrix (rotate_utils/hadamard_utils.py:92-99):
    torch.manual_seed(seed); randint(0,2,(n,))*2-1 drawn on the CPU generator."""
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    return (torch.randint(low=0, high=2, size=(n,), generator=g).to(torch.float64) * 2 - 1)


def sylvester(n: int) -> torch.Tensor:
    """H[i,j] = (-1)^popcount(i&j), natural order, fp64."""
    idx = torch.arange(n)
    bits = (idx[:, None] & idx[None, :])
    pop = torch.zeros_like(bits)
    b = bits.clone()
    while b.any():
        pop += b & 1
        b >>= 1
    return (1 - 2 * (pop & 1)).to(torch.float64)


def hadamard_block(n: int = 128, seed: int = 42) -> torch.Tensor:
    """Q_n = diag(D) . H_n / float32(sqrt(n)) in fp64 (hadamard_utils.py:63-99)."""
    d = sign_vector(n, seed)
    c = torch.tensor(n).sqrt()            # float32 sqrt, as the reference does
    return (d[:, None] * sylvester(n)) / c


def block_hadamard(total: int, block: int = 128, seed: int = 42) -> torch.Tensor:
    """block_random_hadamard_matrix (rotation_utils.py:69-104): identical blocks."""
    assert total % block == 0
    q = hadamard_block(block, seed)
    out = torch.zeros(total, total, dtype=torch.float64)
    for i in range(total // block):
        out[i * block:(i + 1) * block, i * block:(i + 1) * block] = q
    return out


def rotate_fp16_reference(x_h: torch.Tensor, q_h: torch.Tensor) -> torch.Tensor:
    """fp64-accumulated x_h @ q_h rounded once to fp16: the yardstick for the
    fused rotate (the reference's own fp16 GEMM has unspecified accumulation
    order, tr/basic_var.py:263,266)."""
    return (x_h.to(torch.float64) @ q_h.to(torch.float64)).to(torch.float16)
