"""Parity of the HIP path (through the C ABI) against the CPU oracle and the
golden vectors.  Everything here needs a real MI355X: run with ``-m gpu``.

Bar: bit-exact dequantized values (all NaNs equal, +0/-0 distinguished).
"""
import os

import numpy as np
import pytest
import torch

from fpqvar_amd import _lib
from oracle import fpq_oracle as orc
from tests.conftest import assert_bits_equal, from_bits

pytestmark = pytest.mark.gpu

KINDS = ("gauss", "heavy", "edge", "weights", "gelu", "inf", "nan")
SYM4 = ("e2m1", "e1m2", "e3m0")
SYM6 = ("e2m3", "e3m2")


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need the GPU"
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def qu():
    import fpqvar_amd.quant_utils as m
    return m


def all_fp16_as_f32():
    return torch.arange(0, 65536, dtype=torch.int32).to(torch.int16).view(torch.float16).to(torch.float32)


def neighbourhoods(tab, width=8):
    uniq = torch.unique(tab)
    pts = torch.cat([uniq, (uniq[:-1] + uniq[1:]) / 2, torch.tensor([102400.0 + float(uniq.abs().max())])])
    nb = []
    for k in range(-width, width + 1):
        nb.append((pts.view(torch.int32) + k).view(torch.float32))
        nb.append(-((pts.view(torch.int32) + k).view(torch.float32)))
    nb = torch.cat(nb)
    return nb[~torch.isnan(nb)]


# ------------------------------------------------------------------ L0: quant_cuda.quant
def test_quant_cuda_module_contract(dev):
    import quant_cuda
    tab = orc.TABLES["e2m1"].to(dev)
    x = torch.tensor([0.25, -0.25, 0.75, 1.25, 1.75, 2.5, 3.5, 5, -5, -2.5, float("nan"), float("inf"), 7],
                     device=dev)
    z, idx = quant_cuda.quant(x, tab)
    want = torch.tensor([0.5, 0, 1, 1.5, 2, 3, 4, 6, -4, -2, 0, 0, 6])
    assert_bits_equal(z, want, "KAT")
    assert z.device == x.device and z.dtype == x.dtype and z.shape == x.shape
    assert idx.shape == x.shape and idx.dtype == x.dtype and not idx.any()
    # input untouched, float64 accepted and compared in float32
    x64 = torch.tensor([0.25 - 1e-12, 0.25, -102407.0, 102406.0], dtype=torch.float64, device=dev)
    z64, _ = quant_cuda.quant(x64, tab.double())
    assert_bits_equal(z64, torch.tensor([0.5, 0.5, 0.0, 6.0], dtype=torch.float64), "f64")
    # error behaviour: RuntimeError, never a silent CPU path
    with pytest.raises(RuntimeError):
        quant_cuda.quant(x.cpu(), tab.cpu())
    with pytest.raises(RuntimeError):
        quant_cuda.quant(x.half(), tab)
    with pytest.raises(RuntimeError):
        quant_cuda.quant(x, torch.zeros(257, device=dev))
    with pytest.raises(RuntimeError):
        quant_cuda.quant(torch.zeros(4, 4, device=dev).t(), tab)
    # empty input
    z0, i0 = quant_cuda.quant(torch.empty(0, device=dev), tab)
    assert z0.numel() == 0 and i0.numel() == 0


@pytest.mark.parametrize("name", list(orc.TABLES))
def test_scan_kernel_vs_oracle(dev, name):
    from fpqvar_amd import ops
    tab = orc.TABLES[name]
    g = torch.Generator().manual_seed(11)
    x = torch.cat([all_fp16_as_f32(), neighbourhoods(tab),
                   (torch.rand(300000, generator=g) * 2 - 1) * float(tab.abs().max()) * 1.3,
                   torch.randn(1000, generator=g) * 1e5])
    got = ops.quant_nearest(x.to(dev), tab.to(dev))          # a table the kernel recognises: closed-form path
    assert_bits_equal(got, orc.nearest_kernel(x, tab), f"recognised table {name}")
    dup = torch.cat([tab, tab[-1:]])                         # same function, one entry more: the literal scan runs
    got = ops.quant_nearest(x.to(dev), dup.to(dev))
    assert_bits_equal(got, orc.nearest_kernel(x, tab), f"scan {name}")
    x64 = x[:70000].double()
    assert_bits_equal(ops.quant_nearest(x64.to(dev), tab.double().to(dev)), orc.nearest_kernel(x[:70000], tab).double(),
                      f"recognised table, float64 {name}")
    # unsorted / arbitrary table: the literal scan must still agree
    perm = tab[torch.randperm(tab.numel(), generator=g)]
    got = ops.quant_nearest(x.to(dev), perm.to(dev))
    assert_bits_equal(got, orc.nearest_kernel(x, perm), f"scan permuted {name}")


@pytest.mark.parametrize("name", list(orc.TABLES))
def test_closed_form_vs_oracle_and_scan(dev, name):
    from fpqvar_amd import ops
    tab = orc.TABLES[name]
    g = torch.Generator().manual_seed(12)
    x = torch.cat([all_fp16_as_f32(), neighbourhoods(tab),
                   (torch.rand(300000, generator=g) * 2 - 1) * float(tab.abs().max()) * 1.3])
    got = ops.quant_nearest_builtin(x.to(dev), name)
    assert_bits_equal(got, orc.nearest_kernel(x, tab), f"closed form {name}")
    # 2^26 random fp32 bit patterns, device scan vs device closed form
    bits = torch.randint(-2**31, 2**31 - 1, (1 << 26,), dtype=torch.int64, device=dev).to(torch.int32)
    xr = bits.view(torch.float32)
    a = ops.quant_nearest_builtin(xr, name)
    b = ops.quant_nearest(xr, torch.cat([tab, tab[-1:]]).to(dev))     # unrecognised spelling of the same table: scan
    assert_bits_equal(a, b, f"closed form vs scan, random bit patterns, {name}")
    assert_bits_equal(ops.quant_nearest(xr, tab.to(dev)), b, f"recognised-table path vs scan, {name}")


# ------------------------------------------------------------------ golden vectors
def _run_named(qu, key, x):
    fam, tab = key.split("/")[0], key.split("/")[1]
    if fam == "per_group_cuda":
        fn = {"e2m1": qu.fp_quant_e2_per_group_cuda, "e1m2": qu.fp_quant_e1_per_group_cuda,
              "e3m0": qu.fp_quant_e3_per_group_cuda, "e2m3": qu.fp6_quant_e2m3_per_group_cuda,
              "e3m2": qu.fp6_quant_e3m2_per_group_cuda}[tab]
        return fn(x, 6 if tab in SYM6 else 4, 128)
    if fam == "per_token_cuda":
        fn = {"e2m3": qu.fp6_quant_e2m3_per_token_cuda, "e3m2": qu.fp6_quant_e3m2_per_token_cuda}[tab]
        return fn(x, 6)
    if fam == "dual_group_cuda":
        if tab.startswith("e2m1_neg"):
            return qu.fp4_afpq_per_group_cuda(x, 4, 128)
        if tab.startswith("e1m2"):
            return qu.fp_quant_e1m2_neg_e2m1_pos_per_group_cuda(x, 4, 128)
        return qu.fp6_quant_int_neg_e2m3_pos_per_group_cuda(x, 6, 128)
    if fam == "dual_group_cuda_clip0.9":
        return qu.fp_quant_e1m2_neg_e2m1_pos_per_group_cuda(x, 4, 128, 0.9)
    if fam == "dual_token_cuda":
        return qu.fp6_quant_int_neg_e2m3_pos_per_token_cuda(x, 6)
    if fam == "neg_reverse_group_cuda":
        return qu.fp_neg_reverse_quant_per_group_cuda(x, 4, 128)
    raise KeyError(key)


@pytest.mark.parametrize("dn", ("f16", "f32"))
@pytest.mark.parametrize("kind", KINDS)
def test_golden_vectors(dev, qu, golden, kind, dn):
    x = from_bits(golden[f"in/{kind}_{dn}"]).to(dev)
    keys = [k[4:] for k in golden.files if k.startswith("out/") and k.endswith(f"/{kind}_{dn}")
            and k.split("/")[1] in ("per_group_cuda", "per_token_cuda", "dual_group_cuda",
                                    "dual_group_cuda_clip0.9", "dual_token_cuda", "neg_reverse_group_cuda")]
    assert len(keys) == 13
    for key in keys:
        want = from_bits(golden[f"out/{key}"])
        x_before = x.clone()
        got = _run_named(qu, key, x)
        assert_bits_equal(got, want, key)
        assert torch.equal(x.view(torch.int16 if dn == "f16" else torch.int32),
                           x_before.view(torch.int16 if dn == "f16" else torch.int32)), "input mutated"
    # KV-cache shapes (tr/basic_var.py:192-200): per-token over 64 channels, per-group 128
    want = from_bits(golden[f"out/kv/e2m3_token64/{kind}_{dn}"])
    assert_bits_equal(qu.fp6_quant_e2m3_per_token_cuda(x.reshape(2, 4, 4, 64), 6), want, "kv e2m3 c=64")
    want = from_bits(golden[f"out/kv/e2m1_group/{kind}_{dn}"])
    assert_bits_equal(qu.fp_quant_e2_per_group_cuda(x, 4), want, "kv e2m1 g=128")


def test_dual_nan_clip_quirk(dev, qu, golden):
    """A NaN anywhere + the reference's global clamp (tr/quant_utils.py:421-422) zeroes the
    whole output.  Default strength 1.0: NaN flag + conditional zero-fill; other strengths:
    the absmax pass.  Also on long rows and at a size that spans many workgroups."""
    from fpqvar_amd import ops
    for dn in ("f16", "f32"):
        x = from_bits(golden[f"in/nan_{dn}"]).to(dev)
        want = from_bits(golden[f"out/dual_group_cuda/e1m2_neg+e2m1_pos/nan_{dn}"])
        assert_bits_equal(qu.fp_quant_e1m2_neg_e2m1_pos_per_group_cuda(x, 4, 128), want, f"nan clip {dn}")
        assert_bits_equal(ops.quant_rows_dual(x, "e1m2_neg", "e2m1_pos", 128, clipping_strength=0.9),
                          from_bits(golden[f"out/dual_group_cuda_clip0.9/e1m2_neg+e2m1_pos/nan_{dn}"]), "clip .9")
    g = torch.Generator().manual_seed(3)
    big = torch.randn(4096, 7680, generator=g).half()
    clean = qu.fp_quant_e1m2_neg_e2m1_pos_per_group_cuda(big.to(dev), 4, 128)
    assert_bits_equal(clean[:64], orc.dual_per_group_kernel_sem(big[:64], "e1m2_neg", "e2m1_pos", 128, 1.0), "clean")
    big[4000, 7001] = float("nan")
    out = qu.fp_quant_e1m2_neg_e2m1_pos_per_group_cuda(big.to(dev), 4, 128)
    assert out.shape == big.shape and not out.any(), "one NaN must zero the whole tensor (torch.clamp semantics)"
    out = ops.quant_rows_dual(big.to(dev), "e1m2_neg", "e2m1_pos", 7680, clipping_strength=1.0)   # long-row kernel
    assert not out.any()
    out = ops.quant_rows_dual(big.to(dev), "int_neg", "e2m3_pos", 128, None)      # no clamp: NaN element -> 0 only
    assert bool(out.any()) and not torch.isnan(out).any()


@pytest.mark.parametrize("tables", (("e1m2_neg", "e2m1_pos"), ("e2m1_neg", "e2m1_pos")))
def test_dual_clip_fast_path(dev, tables):
    """The reference's global clamp to +-strength * max|x| (tr/quant_utils.py:421-422) on the fp16 group-of-128 fast
    path (two packed instructions per pair in front of the dual quantizer): strengths below and above 1, zero, tensors
    with an Inf (bound = inf: no clamp, the row poisons), with a NaN (bound = NaN: every output +0), with -0, ragged
    tile ends - against the oracle; and, through the C ABI with a supplied maximum, NaN ELEMENTS beside a finite bound
    (torch.clamp keeps them, the hardware min / max would not)."""
    from fpqvar_amd import _lib, ops
    from fpqvar_amd.ops import TABLE_IDS, dtype_id, stream_ptr
    neg, pos = tables
    g = torch.Generator().manual_seed(77)
    for rows, kind in ((1000, "gelu"), (4099, "gauss"), (3, "gauss")):
        x = torch.randn(rows, 128 * 5, generator=g) * 1.7
        if kind == "gelu":
            x = torch.nn.functional.gelu(x, approximate="tanh")
        x = x.half()
        x[0, 5] = -0.0
        x[1, :128] = x[1, :128].abs() + 0.1
        x[2, :128] = 0
        for strength in (0.5, 0.9, 1.5, 0.0, 0.999):
            got = ops.quant_rows_dual(x.to(dev), neg, pos, 128, clipping_strength=strength)
            assert_bits_equal(got, orc.dual_per_group_kernel_sem(x, neg, pos, 128, strength), f"{kind} rows={rows} strength={strength}")
        xi = x.clone()
        xi[rows // 2, 130] = float("inf")
        assert_bits_equal(ops.quant_rows_dual(xi.to(dev), neg, pos, 128, clipping_strength=0.7),
                          orc.dual_per_group_kernel_sem(xi, neg, pos, 128, 0.7), f"{kind} with inf")
        xn = x.clone()
        xn[rows - 1, 7] = float("nan")
        out = ops.quant_rows_dual(xn.to(dev), neg, pos, 128, clipping_strength=0.7)
        assert_bits_equal(out, orc.dual_per_group_kernel_sem(xn, neg, pos, 128, 0.7), f"{kind} with nan")
        assert not out.any()
    # a supplied maximum: bound 0.9 * 2.0 beside NaN elements and values beyond the bound
    x = (torch.randn(300, 256, generator=g) * 2.5).half()
    x[5, 9] = float("nan")
    x[5, 200] = float("nan")
    x[299, 255] = float("nan")
    am = torch.tensor([2.0, 0.0], dtype=torch.float16, device=dev)
    xd = x.to(dev)
    out = torch.empty_like(xd)
    rc = _lib.lib().fpq_quant_rows_dual(xd.data_ptr(), out.data_ptr(), x.numel() // 128, 128, TABLE_IDS[neg], TABLE_IDS[pos],
                                        dtype_id(x.dtype), dtype_id(x.dtype), am.data_ptr(), 0.9, None, stream_ptr(dev))
    assert rc == 0
    c = torch.tensor(0.9, dtype=torch.float32) * torch.tensor(2.0)
    want = orc._dual_rows_kernel_sem(torch.clamp(x, -c.half(), c.half()).reshape(-1, 128), neg, pos).view(x.shape).to(x.dtype)
    assert_bits_equal(out, want, "supplied maximum with NaN elements")


# ------------------------------------------------------------------ oracle on seeded inputs
def _inputs(kind, shape, dtype, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(*shape, generator=g)
    if kind == "heavy":
        x = x * torch.exp(0.5 * torch.randn(*shape, generator=g))
    elif kind == "gelu":
        x = torch.nn.functional.gelu(x * 1.5, approximate="tanh")
    elif kind == "weights":
        x = x * 0.02
    return x.to(dtype)


@pytest.mark.parametrize("dtype", (torch.float16, torch.float32))
@pytest.mark.parametrize("kind", ("gauss", "heavy", "weights"))
def test_per_group_vs_oracle(dev, qu, kind, dtype):
    x = _inputs(kind, (512, 1920), dtype, 21)
    x[3, 128:256] = 0                      # an all-zero group
    x[5, :128] = x[5, :128].abs()          # single-sign group
    xd = x.to(dev)
    for name, fn in (("e2m1", qu.fp_quant_e2_per_group_cuda), ("e1m2", qu.fp_quant_e1_per_group_cuda),
                     ("e3m0", qu.fp_quant_e3_per_group_cuda)):
        assert_bits_equal(fn(xd, 4, 128), orc.per_group_kernel_sem(x, name, 128), f"{name} {kind}")
    for name, fn in (("e2m3", qu.fp6_quant_e2m3_per_group_cuda), ("e3m2", qu.fp6_quant_e3m2_per_group_cuda)):
        assert_bits_equal(fn(xd, 6, 128), orc.per_group_kernel_sem(x, name, 128, out_dtype=torch.float16),
                          f"{name} {kind}")


@pytest.mark.parametrize("dtype", (torch.float16, torch.float32))
@pytest.mark.parametrize("group", (64, 32, 256))
def test_other_group_sizes_vs_oracle(dev, qu, group, dtype):
    """BASELINE.json names group_size 64/128; every reference function takes group_size as an argument."""
    x = _inputs("heavy", (300, 1920), dtype, 40 + group)
    x[7, :group] = 0
    xd = x.to(dev)
    assert_bits_equal(qu.fp_quant_e2_per_group_cuda(xd, 4, group), orc.per_group_kernel_sem(x, "e2m1", group), f"e2m1 g={group}")
    assert_bits_equal(qu.fp6_quant_e3m2_per_group_cuda(xd, 6, group),
                      orc.per_group_kernel_sem(x, "e3m2", group, out_dtype=torch.float16), f"e3m2 g={group}")
    assert_bits_equal(qu.fp_quant_e1m2_neg_e2m1_pos_per_group_cuda(xd, 4, group),
                      orc.dual_per_group_kernel_sem(x, "e1m2_neg", "e2m1_pos", group, 1.0), f"dual fp4 g={group}")
    assert_bits_equal(qu.fp6_quant_int_neg_e2m3_pos_per_group_cuda(xd, 6, group),
                      orc.dual_per_group_kernel_sem(x, "int_neg", "e2m3_pos", group, None), f"dual fp6 g={group}")


@pytest.mark.parametrize("dtype", (torch.float16, torch.float32))
@pytest.mark.parametrize("cols", (64, 1920, 2304, 7680, 9216, 1000, 8, 3))
def test_per_token_vs_oracle(dev, qu, cols, dtype):
    x = _inputs("heavy", (37, cols), dtype, 22 + cols)
    xd = x.to(dev)
    for name, fn in (("e2m3", qu.fp6_quant_e2m3_per_token_cuda), ("e3m2", qu.fp6_quant_e3m2_per_token_cuda)):
        assert_bits_equal(fn(xd, 6), orc.per_token_kernel_sem(x, name), f"{name} cols={cols}")
    assert_bits_equal(qu.fp6_quant_int_neg_e2m3_pos_per_token_cuda(xd, 6),
                      orc.dual_per_token_kernel_sem(x), f"dual token cols={cols}")


@pytest.mark.parametrize("dtype", (torch.float16, torch.float32))
@pytest.mark.parametrize("kind", ("gelu", "gauss"))
def test_dual_per_group_vs_oracle(dev, qu, kind, dtype):
    x = _inputs(kind, (256, 7680), dtype, 23)
    x[1, :128] = x[1, :128].abs() + 0.1     # no negatives in the group -> scale_neg = 0
    x[2, :128] = -x[2, :128].abs() - 0.1    # no positives
    x[3, :128] = 0
    xd = x.to(dev)
    assert_bits_equal(qu.fp_quant_e1m2_neg_e2m1_pos_per_group_cuda(xd, 4, 128),
                      orc.dual_per_group_kernel_sem(x, "e1m2_neg", "e2m1_pos", 128, 1.0), f"dual fp4 {kind}")
    assert_bits_equal(qu.fp_quant_e1m2_neg_e2m1_pos_per_group_cuda(xd, 4, 128, 0.8),
                      orc.dual_per_group_kernel_sem(x, "e1m2_neg", "e2m1_pos", 128, 0.8), f"dual fp4 clip {kind}")
    assert_bits_equal(qu.fp6_quant_int_neg_e2m3_pos_per_group_cuda(xd, 6, 128),
                      orc.dual_per_group_kernel_sem(x, "int_neg", "e2m3_pos", 128, None), f"dual fp6 {kind}")


@pytest.mark.parametrize("dtype", (torch.float16, torch.float32))
@pytest.mark.parametrize("kind", ("gelu", "gauss", "heavy"))
def test_neg_reverse_vs_oracle(dev, qu, kind, dtype):
    """fp_neg_reverse_quant_per_group_cuda (models_fp_quant/quant_utils.py:454-495), vector kernel at
    g=128 and other power-of-two groups, scalar kernel at ragged group sizes."""
    from fpqvar_amd import ops
    x = _inputs(kind, (256, 7680), dtype, 29)
    x[1, :128] = x[1, :128].abs() + 0.1     # min > 0: the shift is the smallest positive value
    x[2, :128] = -x[2, :128].abs() - 0.1    # no positives
    x[3, :128] = 0
    x[4, 5] = float("inf")
    x[5, 5] = float("-inf")
    x[6, 5] = float("nan")
    xd = x.to(dev)
    assert_bits_equal(qu.fp_neg_reverse_quant_per_group_cuda(xd, 4, 128),
                      orc.neg_reverse_per_group_kernel_sem(x, "e2m1", 128), f"neg reverse {kind}")
    for g in (32, 512 if dtype == torch.float16 else 256, 60, 7680, 3):
        assert_bits_equal(ops.quant_rows_neg_reverse(xd, "e2m1", g),
                          orc.neg_reverse_per_group_kernel_sem(x, "e2m1", g), f"neg reverse {kind} g={g}")
    assert_bits_equal(ops.quant_rows_neg_reverse(xd, "e2m3", 128),
                      orc.neg_reverse_per_group_kernel_sem(x, "e2m3", 128), f"neg reverse e2m3 {kind}")


def test_exhaustive_fp16_scale_pairs(dev, qu):
    """Every fp16 magnitude as the group maximum against a sweep of element values:
    exercises the scale / normalise roundings far beyond what random data reaches."""
    amax = torch.arange(1, 0x7C00, 7, dtype=torch.int32).to(torch.int16).view(torch.float16)   # group maxima
    g = torch.Generator().manual_seed(5)
    frac = torch.rand(amax.numel(), 127, generator=g) * 2 - 1
    x = torch.cat([amax[:, None], (frac * amax[:, None].float()).half()], dim=1)
    xd = x.to(dev)
    for name, fn in (("e2m1", qu.fp_quant_e2_per_group_cuda), ("e1m2", qu.fp_quant_e1_per_group_cuda),
                     ("e3m0", qu.fp_quant_e3_per_group_cuda)):
        assert_bits_equal(fn(xd, 4, 128), orc.per_group_kernel_sem(x, name, 128), f"scale sweep {name}")
    assert_bits_equal(qu.fp6_quant_e2m3_per_group_cuda(xd, 6, 128),
                      orc.per_group_kernel_sem(x, "e2m3", 128, out_dtype=torch.float16), "scale sweep e2m3")
    assert_bits_equal(qu.fp_quant_e1m2_neg_e2m1_pos_per_group_cuda(xd, 4, 128),
                      orc.dual_per_group_kernel_sem(x, "e1m2_neg", "e2m1_pos", 128, 1.0), "scale sweep dual")


@pytest.mark.parametrize("table", ("e2m1", "e2m1/table", "e1m2", "e3m0", "e2m3", "e2m3/table", "e3m2", "e3m2/table", "e1m2_neg+e2m1_pos",
                                   "int_neg+e2m3_pos"))
def test_every_fp16_pair_fast_path_vs_ieee_path(dev, table, lib_options):
    """EVERY (group maximum, element) pair of finite fp16 values - 1.0e9 per table, both signs - through the fast
    fp16 kernel (one-multiply division, level from the bucket table or - "e2m1", "e2m3", "e3m2": the defaults in groups of
    128 - from the FP4 / FP6 conversion hardware, packed fp16 multiply) and through the generic kernel (IEEE fp32
    division, closed form, fp32 product; selected by asking for a float32 result), which the tests above pin to the
    oracle, and through the reference's own ~11-op torch sequence on this GPU around the literal scan kernel.
    The division of fpq_fast16.h (margin: 0.05 %) and the tie-breaking bias in front of the hardware conversion are
    argued on paper; this is the same statement checked on hardware, exhaustively: the domain is finite.  The fused
    producers (fpq_rotate_mfma.h, fpq_adaln.h) quantize with the same two functions.
    Dual formats: the group holds +max and -max, so both sides' scales sweep every magnitude too."""
    from fpqvar_amd import ops
    if table.endswith("/table"):                   # the bucket-table forms behind the conversion-hardware defaults (round 3: E2M1;
        lib_options("FPQ_NO_HW4", 1)   # round 4: E2M3 / E3M2); switches of the library (fpq_set_option), restored by the fixture
        lib_options("FPQ_NO_HW6", 1)
        table = table.split("/")[0]
    dual = "+" in table
    lead = 2 if dual else 1
    per = 128 - lead
    total = 0
    step = 512
    distinct_out = set()
    for lo in range(0, 0x7C00, step):
        p = torch.arange(lo, min(lo + step, 0x7C00), device=dev, dtype=torch.int64)          # maxima, as bit patterns
        n_val = 2 * (p + 1)                                                                  # signed candidates <= max
        n_grp = (n_val + per - 1) // per
        gm = torch.repeat_interleave(p, n_grp)                                               # max pattern of every group
        first = torch.cumsum(n_grp, 0) - n_grp
        start = (torch.arange(gm.numel(), device=dev) - torch.repeat_interleave(first, n_grp)) * per
        idx = start[:, None] + torch.arange(per, device=dev)[None, :]                        # candidate index per slot
        ok = idx < (2 * (gm + 1))[:, None]
        pat = torch.where(ok, (idx >> 1) | ((idx & 1) << 15), torch.zeros_like(idx))
        head = [gm[:, None]] + ([gm[:, None] | 0x8000] if dual else [])
        x = torch.cat(head + [pat], dim=1).to(torch.int32).to(torch.int16).view(torch.float16)
        assert int((x.view(torch.int16).to(torch.int32) & 0x7FFF).max()) == min(lo + step, 0x7C00) - 1
        if dual:
            neg, pos = table.split("+")
            fast = ops.quant_rows_dual(x, neg, pos, 128, None)
            ieee = ops.quant_rows_dual(x, neg, pos, 128, None, out_dtype=torch.float32).half()
        else:
            fast = ops.quant_rows(x, table, 128, torch.float16)
            ieee = ops.quant_rows(x, table, 128, torch.float32).half()
        bad = fast.view(torch.int16) != ieee.view(torch.int16)
        assert not bool(bad.any()), (table, lo, x[bad][:4].tolist(), fast[bad][:4].tolist(), ieee[bad][:4].tolist())
        # ... and against the reference's own op sequence executed by torch-ROCm in real fp16 arithmetic around the
        # LITERAL scan (the duplicated last entry keeps the kernel from recognising the table)
        if dual:
            tn, tp = orc.TABLES[neg].to(dev), orc.TABLES[pos].to(dev)
            zeros = torch.zeros_like(x)
            x_neg, x_pos = torch.where(x <= 0, x, zeros), torch.where(x > 0, x, zeros)
            s_neg = x_neg.abs().max(dim=-1, keepdim=True)[0] / tn.abs().max()
            s_pos = x_pos.abs().max(dim=-1, keepdim=True)[0] / tp.abs().max()
            q_neg = ops.quant_nearest((x_neg / s_neg).view(-1).to(torch.float32), torch.cat([tn, tn[-1:]])).view(x.shape)
            q_pos = ops.quant_nearest((x_pos / s_pos).view(-1).to(torch.float32), torch.cat([tp, tp[-1:]])).view(x.shape)
            seq = (q_neg * s_neg + q_pos * s_pos).to(torch.float16)
        else:
            tab = orc.TABLES[table].to(dev)
            scale = x.abs().max(dim=-1, keepdim=True)[0] / tab.abs().max()
            q = ops.quant_nearest((x / scale).view(-1).to(torch.float32), torch.cat([tab, tab[-1:]])).view(x.shape)
            seq = (q * scale).to(torch.float16)
        bad = fast.view(torch.int16) != seq.view(torch.int16)
        assert not bool(bad.any()), ("vs torch sequence", table, lo, x[bad][:4].tolist(), fast[bad][:4].tolist(),
                                     seq[bad][:4].tolist())
        total += int(ok.sum())
        if lo % (step * 8) == 0:
            distinct_out.update(torch.unique(fast.view(torch.int16)).tolist()[:4096])
    assert total == sum(2 * (q + 1) for q in range(0x7C00)) == 1_007_713_280
    assert len(distinct_out) > 1000                                   # the sweep really produced a spread of results


@pytest.mark.parametrize("cols", (1920, 2304))
@pytest.mark.parametrize("table", ("e2m3", "e3m2"))
def test_every_fp16_pair_per_token_hw_levels_vs_ieee_path(dev, table, cols):
    """The per-token FP6 quantizer on rows of 1920 (W6A6 activations) takes its levels from the FP6 conversion hardware since
    round 4 (rows16_lut_wave_kernel<HW6>, fp6_levels_hw32: float(xn) + 2^-17 in front of a round-to-nearest-even conversion).
    EVERY (row maximum, element) pair of finite fp16 values through it, through the table form (library switch FPQ_NO_HW6 around
    the call) and through the generic kernel (IEEE division + closed form, selected by asking for a float32 result).
    cols = 2304 (d36): a lane's fifth vector goes through a second conversion with 8 live values."""
    from fpqvar_amd import ops
    per = cols - 1
    total = 0
    step = 1024
    for lo in range(0, 0x7C00, step):
        p = torch.arange(lo, min(lo + step, 0x7C00), device=dev, dtype=torch.int64)          # maxima, as bit patterns
        n_val = 2 * (p + 1)                                                                  # signed candidates <= max
        n_row = (n_val + per - 1) // per
        gm = torch.repeat_interleave(p, n_row)
        first = torch.cumsum(n_row, 0) - n_row
        start = (torch.arange(gm.numel(), device=dev) - torch.repeat_interleave(first, n_row)) * per
        idx = start[:, None] + torch.arange(per, device=dev)[None, :]
        ok = idx < (2 * (gm + 1))[:, None]
        pat = torch.where(ok, (idx >> 1) | ((idx & 1) << 15), torch.zeros_like(idx))
        x = torch.cat([gm[:, None], pat], dim=1).to(torch.int32).to(torch.int16).view(torch.float16)
        fast = ops.quant_rows(x, table, cols, torch.float16)
        ieee = ops.quant_rows(x, table, cols, torch.float32).half()
        bad = fast.view(torch.int16) != ieee.view(torch.int16)
        assert not bool(bad.any()), (table, lo, x[bad][:4].tolist(), fast[bad][:4].tolist(), ieee[bad][:4].tolist())
        with _lib.option("FPQ_NO_HW6", 1):
            tab = ops.quant_rows(x, table, cols, torch.float16)
        assert torch.equal(fast.view(torch.int16), tab.view(torch.int16)), (table, lo, "hardware levels vs table form")
        total += int(ok.sum())
    assert total == sum(2 * (q + 1) for q in range(0x7C00)) == 1_007_713_280
    # rows with non-finite and zero content, ragged row counts (1 .. 9 rows: partial workgroups)
    g = torch.Generator().manual_seed(12)
    e = torch.randn(9, cols, generator=g).half()
    e[1] = 0.0
    e[2, 5] = float("nan")
    e[3, 7] = float("inf")
    e[4, 9] = float("-inf")
    e[5] *= 1e-4
    e[6, ::2] = -0.0
    e[7] = -e[7].abs() * 1e-3
    for rows in (9, 1, 3):
        assert_bits_equal(ops.quant_rows(e[:rows].to(dev), table, cols, torch.float16),
                          orc.per_token_kernel_sem(e[:rows], table), f"{table} edge rows={rows}")


@pytest.mark.parametrize("cols", (1920, 128, 64))
def test_fp6_hardware_levels_full_size_properties(dev, cols):
    """The metric-sized tensor [65536 x 1920] through the FP6 conversion-hardware forms (per token, per group of 128, KV rows of
    64): equal to the table forms element for element, idempotent (a quantized tensor quantizes to itself - the KV path
    relies on it, tr/basic_var.py:186-209), every output a level times its row's scale (checked on oracle slices)."""
    from fpqvar_amd import ops
    g = torch.Generator(device=dev).manual_seed(321)
    x = (torch.randn(65536, 1920, device=dev, generator=g) * torch.exp(0.3 * torch.randn(65536, 1920, device=dev, generator=g))).half()
    for table in ("e2m3", "e3m2"):
        q = ops.quant_rows(x, table, cols, torch.float16)
        with _lib.option("FPQ_NO_HW6", 1):
            qt = ops.quant_rows(x, table, cols, torch.float16)
        assert torch.equal(q.view(torch.int16), qt.view(torch.int16)), f"{table} cols={cols}: hardware levels vs table form"
        assert torch.equal(ops.quant_rows(q, table, cols, torch.float16).view(torch.int16), q.view(torch.int16)), f"{table} cols={cols}: not idempotent"
        for lo in (0, 32768, 65536 - 16):
            want = orc.per_token_kernel_sem(x[lo:lo + 16].cpu().reshape(-1, cols), table).view(16, 1920)
            assert_bits_equal(q[lo:lo + 16], want, f"{table} cols={cols} rows {lo}..")


# ------------------------------------------------------------------ shapes, raggedness, errors
def test_edge_shapes_and_errors(dev, qu):
    from fpqvar_amd import ops
    e = torch.empty(0, 128, dtype=torch.float16, device=dev)
    assert qu.fp_quant_e2_per_group_cuda(e, 4, 128).shape == (0, 128)
    x = torch.randn(4, 30, 128, device=dev).half()
    assert_bits_equal(qu.fp_quant_e2_per_group_cuda(x, 4, 128),
                      orc.per_group_kernel_sem(x.cpu(), "e2m1", 128), "3-d input")
    # non-contiguous input to a per-group function: reshape semantics (copy), like the reference
    xt = torch.randn(256, 64, device=dev).half().t()
    assert_bits_equal(qu.fp_quant_e2_per_group_cuda(xt, 4, 128),
                      orc.per_group_kernel_sem(xt.cpu(), "e2m1", 128), "transposed input")
    # unaligned base pointer -> scalar path
    base = torch.randn(128 * 9 + 1, device=dev).half()
    xu = base[1:]
    assert_bits_equal(qu.fp_quant_e2_per_group_cuda(xu, 4, 128),
                      orc.per_group_kernel_sem(xu.cpu(), "e2m1", 128), "unaligned")
    with pytest.raises(RuntimeError):
        qu.fp_quant_e2_per_group_cuda(torch.randn(100, device=dev).half(), 4, 128)   # numel % 128 != 0
    with pytest.raises(AssertionError):
        qu.fp_quant_e2_per_group_cuda(x, 6, 128)
    with pytest.raises(RuntimeError):
        qu.fp_quant_e2_per_group_cuda(x.cpu(), 4, 128)                                # no CPU fallback
    with pytest.raises(RuntimeError):
        qu.fp6_quant_e2m3_per_token_cuda(torch.randn(8, 4, 64, device=dev).half().permute(1, 0, 2), 6)
    with pytest.raises(RuntimeError):
        ops.quant_rows(x.double(), "e2m1", 128)
    # runs on a side stream, ordered by the stream
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        y = qu.fp_quant_e2_per_group_cuda(x, 4, 128)
    s.synchronize()
    assert_bits_equal(y, orc.per_group_kernel_sem(x.cpu(), "e2m1", 128), "side stream")


# ------------------------------------------------------------------ codewords
@pytest.mark.parametrize("dtype", (torch.float16, torch.float32))
def test_codes_roundtrip(dev, dtype):
    from fpqvar_amd import ops
    x = _inputs("heavy", (64, 1920), dtype, 31)
    x[0, :128] = 0
    xd = x.to(dev)
    for name in SYM4 + SYM6:
        codes, scales = ops.quant_rows_codes(xd, name, 128, pack_nibbles=False)
        want_codes, want_scales = orc.per_group_codes(x, name, 128)
        assert torch.equal(codes.cpu().view(-1), want_codes.view(-1)), f"codes {name}"
        assert_bits_equal(scales, want_scales, f"scales {name}")
        deq = ops.dequant_rows_codes(codes, scales, name, 128, dtype).view(x.shape)
        assert_bits_equal(deq, orc.per_group_kernel_sem(x, name, 128), f"dequant {name}")
    for name in SYM4:
        codes, scales = ops.quant_rows_codes(xd, name, 128, pack_nibbles=True)
        assert codes.shape == (x.numel() // 128, 64)
        deq = ops.dequant_rows_codes(codes, scales, name, 128, dtype, pack_nibbles=True).view(x.shape)
        assert_bits_equal(deq, orc.per_group_kernel_sem(x, name, 128), f"packed dequant {name}")


# ------------------------------------------------------------------ full BASELINE size
def test_full_size_metric_shape(dev, qu):
    """[65536 x 1920] fp16, g=128: the fused kernel against the reference's own op
    sequence executed by torch on this GPU around the L0 scan kernel (itself pinned
    to the oracle above), plus size-independent properties."""
    from fpqvar_amd import ops
    torch.manual_seed(0)
    x = torch.randn(65536, 1920, device=dev).half()
    x[17, 256:384] = 0
    got = qu.fp_quant_e2_per_group_cuda(x, 4, 128)
    tab = orc.TABLES["e2m1"].to(dev)
    want = torch.empty_like(got)
    for lo in range(0, 65536, 8192):      # chunked: the unfused sequence needs ~46 B/elem of scratch
        xs = x[lo:lo + 8192].reshape(-1, 128)
        scale = xs.abs().max(dim=-1, keepdim=True)[0] / tab.abs().max()
        xn = (xs / scale).view(-1).to(torch.float32)
        q = ops.quant_nearest(xn, tab).view(xs.shape)
        want[lo:lo + 8192] = (q * scale).view(8192, 1920).to(torch.float16)
    assert_bits_equal(got, want, "full size vs unfused GPU sequence")
    # the CPU oracle on row slices from both ends, the middle and around the injected all-zero group
    for lo in (0, 16, 32768, 65536 - 64):
        rows = slice(lo, lo + 64)
        assert_bits_equal(got[rows], orc.per_group_kernel_sem(x[rows].cpu(), "e2m1", 128), f"full size rows {lo}.. vs oracle")
    # properties: every output is one of the 15 levels of its group; levels used are sane
    g = got.view(-1, 128).float()
    s = (x.view(-1, 128).abs().max(dim=-1, keepdim=True)[0] / 6.0).float()
    lv = (tab[None, None, :] * s[:8192, :, None]).half().float()
    hit = (g[:8192, :, None] == lv).any(dim=-1)
    assert bool(hit.all()), "output value outside its group's level set"
    assert bool((got.view(-1, 128)[17 * 15 + 2] == 0).all())
    # sign preserved or flushed to +0
    assert bool(((got.float() * x.float()) >= 0).all())


# ------------------------------------------------------------------ sharded calibration on the GPU (world = 1)
def test_calibrate_on_gpu(dev):
    from fpqvar_amd import calibrate as cal
    g = torch.Generator().manual_seed(77)
    shapes = {"b0.qkv": (384, 128), "b0.fc1": (512, 128), "b0.fc2": (128, 512), "b0.proj": (128, 128)}
    w_cpu = {n: torch.randn(*s, generator=g) * 0.02 for n, s in shapes.items()}
    w_gpu = {n: w.to(dev) for n, w in w_cpu.items()}
    want = {n: orc.per_group_kernel_sem(w, "e2m1", 128).to(torch.float16) for n, w in w_cpu.items()}
    got = cal.calibrate_sharded(w_gpu)                        # default HIP quantizer: fp32 in -> fp16 out
    got_codes = cal.calibrate_sharded(w_gpu, exchange="codes")
    for n in shapes:
        assert_bits_equal(got[n], want[n], f"calibrate fp16 {n}")
        assert_bits_equal(got_codes[n], want[n], f"calibrate codes {n}")
    # FP6 per-channel weights (W6A6 run, tr/quant_utils.py:808-815)
    q6 = cal.default_weight_quantizer("per_channel", "fp6_e2m3", 6)
    for n in shapes:
        assert_bits_equal(q6(n, w_gpu[n]), orc.per_token_kernel_sem(w_cpu[n], "e2m3"), f"fp6 per-channel {n}")


# ------------------------------------------------------------------ the reference's pure-torch quantizers (argmin)
@pytest.mark.parametrize("dn", ("f16", "f32"))
@pytest.mark.parametrize("kind", KINDS)
def test_argmin_path_golden(dev, qu, golden, kind, dn):
    x = from_bits(golden[f"in/{kind}_{dn}"]).to(dev)
    fns_g = {"e2m1": qu.fp_quant_e2_per_group, "e1m2": qu.fp_quant_e1_per_group, "e3m0": qu.fp_quant_e3_per_group}
    fns_t = {"e2m1": qu.fp_quant_e2_per_token, "e1m2": qu.fp_quant_e1_per_token, "e3m0": qu.fp_quant_e3_per_token}
    for name in SYM4:
        want = from_bits(golden[f"out/per_group_argmin/{name}/{kind}_{dn}"])
        assert want.dtype == torch.float32
        assert_bits_equal(fns_g[name](x, 4, 128), want, f"argmin group {name} {kind} {dn}")
        want = from_bits(golden[f"out/per_token_argmin/{name}/{kind}_{dn}"])
        assert_bits_equal(fns_t[name](x, 4), want, f"argmin token {name} {kind} {dn}")


@pytest.mark.parametrize("dtype", (torch.float16, torch.float32))
def test_quantize_to_nearest_grid(dev, qu, dtype):
    """The bare argmin lookup for arbitrary (here: every built-in, a permuted and a tiny) table vs torch's own
    distance-tensor formulation on the CPU."""
    g = torch.Generator().manual_seed(71)
    x = torch.cat([all_fp16_as_f32(), torch.randn(5000, generator=g) * 4,
                   torch.tensor([float("nan"), float("inf"), -float("inf"), 0.0, -0.0])]).to(dtype)
    tables = [orc.TABLES[n] for n in orc.TABLES] + [orc.TABLES["e2m3"][torch.randperm(64, generator=g)], torch.tensor([0.25])]
    for tab in tables:
        want = tab[torch.argmin(torch.abs(x.unsqueeze(-1) - tab), dim=-1)]
        assert_bits_equal(qu.quantize_to_nearest_grid(x.to(dev), tab.to(dev)), want, f"argmin lookup K={tab.numel()}")


@pytest.mark.parametrize("dn", ("f16", "f32"))
@pytest.mark.parametrize("kind", KINDS)
def test_dual_argmin_path_golden(dev, qu, golden, kind, dn):
    """fp_quant_e1m2_neg_e2m1_pos_per_group, the pure-torch twin (tr/quant_utils.py:381-412): golden vectors from the
    reference's own function, plus the oracle on a larger tensor with groups that lack one sign."""
    x = from_bits(golden[f"in/{kind}_{dn}"])
    want = from_bits(golden[f"out/dual_group_argmin/e1m2_neg+e2m1_pos/{kind}_{dn}"])
    got = qu.fp_quant_e1m2_neg_e2m1_pos_per_group(x.to(dev), 4, 128)
    assert got.dtype == torch.float32
    assert_bits_equal(got, want, f"dual argmin {kind} {dn}")
    if kind in ("gelu", "gauss"):
        xb = _inputs(kind, (64, 7680), x.dtype, 57)
        xb[1, :128] = xb[1, :128].abs() + 0.1          # no negatives: the reference adds -1.75 to every element
        xb[2, :128] = -xb[2, :128].abs() - 0.1
        xb[3, :128] = 0
        assert_bits_equal(qu.fp_quant_e1m2_neg_e2m1_pos_per_group(xb.to(dev), 4, 128, 0.9),
                          orc.dual_per_group_argmin_sem(xb, "e1m2_neg", "e2m1_pos", 128, 0.9), f"dual argmin big {kind}")


@pytest.mark.parametrize("dtype", (torch.float16, torch.float32))
def test_argmin_path_vs_oracle(dev, qu, dtype):
    x = _inputs("heavy", (128, 1920), dtype, 41)
    x[2, :128] = 0
    # exact ties of the normalised value: group maximum 6 -> scale 1
    tie = torch.tensor([6.0, 0.25, -0.25, 0.75, -0.75, 1.25, -1.25, 1.75, -1.75, 2.5, -2.5, 3.5, -3.5, 5.0, -5.0])
    x[4, :128] = 0
    x[4, :tie.numel()] = tie.to(dtype)
    xd = x.to(dev)
    for name, fg, ft in (("e2m1", qu.fp_quant_e2_per_group, qu.fp_quant_e2_per_token),
                         ("e1m2", qu.fp_quant_e1_per_group, qu.fp_quant_e1_per_token),
                         ("e3m0", qu.fp_quant_e3_per_group, qu.fp_quant_e3_per_token)):
        assert_bits_equal(fg(xd, 4, 128), orc.per_group_argmin_sem(x, name, 128, clamp3=(name != "e2m1")),
                          f"argmin group {name}")
        assert_bits_equal(ft(xd, 4), orc.per_token_argmin_sem(x, name), f"argmin token {name}")
    # the lookup alone on every fp16 value and on fp32 neighbourhoods of the midpoints:
    # one row whose maximum makes the scale exactly 1
    for name in SYM4:
        tab = orc.TABLES[name]
        gmax = float(tab.abs().max())
        v = torch.cat([all_fp16_as_f32(), neighbourhoods(tab)])
        v = v[torch.isfinite(v) & (v.abs() <= gmax)]
        pad = (-v.numel() - 1) % 8
        row = torch.cat([torch.tensor([gmax]), v, torch.zeros(pad)]).unsqueeze(0)
        got = ops_argmin(dev, row, name)
        assert_bits_equal(got, orc.nearest_argmin(row / 1.0, tab) * 1.0, f"argmin lookup {name}")


def ops_argmin(dev, row, name):
    from fpqvar_amd import ops
    return ops.quant_rows_argmin(row.to(dev), name, row.shape[-1], clamp3=False)


# ------------------------------------------------------------------ QuantizedLinear / quantize_VAR mirror
RUN_CFGS = {
    "w4a4": dict(weight_quant="per_group", act_quant="per_group", w_bit=4, a_bit=4, act_quant_sym=True,
                 activation_fp_quant=True, weight_fp_quant=True, act_fp_type="fp_e2", weight_fp_type="fp_e2",
                 fc2_fp_type="fp_e1m2_neg_e2m1_pos"),
    "w6a6": dict(weight_quant="per_channel", act_quant="per_token", w_bit=6, a_bit=6, act_quant_sym=True,
                 activation_fp_quant=True, weight_fp_quant=True, act_fp_type="fp6_e2m3", weight_fp_type="fp6_e2m3",
                 fc2_fp_type="fp6_int_neg_e2m3_pos"),
    "w4a4_tok": dict(weight_quant="per_channel", act_quant="per_token", w_bit=4, a_bit=4, act_quant_sym=True,
                     activation_fp_quant=True, weight_fp_quant=True, act_fp_type="fp_e2", weight_fp_type="fp_e2",
                     fc2_fp_type="fp_e3"),
}


class _FFN(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.fc1 = torch.nn.Linear(128, 256)
        self.fc2 = torch.nn.Linear(256, 128)


class _Attn(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.mat_qkv = torch.nn.Linear(128, 384, bias=False)
        self.proj = torch.nn.Linear(128, 128)


class _Toy(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.ffn = _FFN()
        self.attn = _Attn()


@pytest.mark.parametrize("cname", list(RUN_CFGS))
def test_quantize_var_mirror(dev, golden, cname):
    from fpqvar_amd.quant_linear import QuantizedLinear, QuantizedLinear_fc2, quantize_VAR
    toy = _Toy()
    names = ("ffn.fc1", "ffn.fc2", "attn.mat_qkv", "attn.proj")
    mods = dict(toy.named_modules())
    with torch.no_grad():
        for n in names:
            mods[n].weight.copy_(from_bits(golden[f"ql/w0/{n}"]))
            if mods[n].bias is not None:
                mods[n].bias.copy_(from_bits(golden[f"ql/b0/{n}"]))
    toy = toy.to(dev)
    quantize_VAR(toy, **RUN_CFGS[cname])
    mods = dict(toy.named_modules())
    x = from_bits(golden["ql/x_f32"]).to(dev)
    h = from_bits(golden["ql/h_f32"]).to(dev)
    for n in names:
        m = mods[n]
        assert type(m).__name__ == str(golden[f"ql/{cname}/class/{n}"])
        assert isinstance(m, QuantizedLinear_fc2 if n == "ffn.fc2" else QuantizedLinear)
        assert_bits_equal(m.weight, from_bits(golden[f"ql/{cname}/weight/{n}"]), f"{cname} weight {n}")
        src = h if n == "ffn.fc2" else x
        for dt, dn in ((torch.float16, "f16"), (torch.float32, "f32")):
            assert_bits_equal(m.act_quant(src.to(dt)), from_bits(golden[f"ql/{cname}/act/{n}/{dn}"]),
                              f"{cname} act {n} {dn}")
        # forward as the driver runs it (fp16 module under fp16 autocast); GEMM order is not a contract
        m.weight = m.weight.half()
        if m.bias is not None:
            m.bias = torch.nn.Parameter(m.bias.detach().half(), requires_grad=False)
        with torch.autocast("cuda", dtype=torch.float16):
            y = m(src.half())
        want = from_bits(golden[f"ql/{cname}/fwd_f32/{n}"])
        assert y.dtype == torch.float16
        torch.testing.assert_close(y.float().cpu(), want, rtol=2e-2, atol=2e-2)


# ------------------------------------------------------------------ F1: fused rotate + quant
def _ulp_diff_f16(a, b):
    ai = a.view(torch.int16).to(torch.int32)
    bi = b.view(torch.int16).to(torch.int32)
    ai = torch.where(ai < 0, -(ai & 0x7FFF), ai)
    bi = torch.where(bi < 0, -(bi & 0x7FFF), bi)
    return (ai - bi).abs()


@pytest.mark.parametrize("in_dtype", (torch.float16, torch.float32))
@pytest.mark.parametrize("with_smooth", (False, True))
def test_rotate_quant_fused(dev, in_dtype, with_smooth):
    from fpqvar_amd import rotation as rot
    g = torch.Generator().manual_seed(55)
    x = (torch.randn(300, 1920, generator=g) * torch.exp(0.5 * torch.randn(300, 1920, generator=g))).to(in_dtype)
    x[7, 128:256] = 0
    s = (torch.rand(1920, generator=g) * 1.5 + 0.25) if with_smooth else None
    out, y = rot.rotate_quant(x.to(dev), "e2m1", smooth=s, return_rotated=True)
    out2 = rot.rotate_quant(x.to(dev), "e2m1", smooth=s)
    assert_bits_equal(out2, out, "emit vs no-emit")
    # (1) rotated values: within 1 fp16 ulp of the fp64-accumulated half(x*s) @ half(Q)
    h = (x.float() * s).half() if with_smooth else x.half()
    q_h = rot.block_random_hadamard_matrix(1920, 128, "cpu", 42).float().half()
    y_ref = orc.rotate_fp16_reference(h, q_h)
    d = _ulp_diff_f16(y.cpu(), y_ref)
    assert int(d.max()) <= 1, f"rotated value off by {int(d.max())} ulp"
    assert float((d > 0).float().mean()) < 0.01
    # (2) the quant stage is bit-exact on the rotated values the kernel produced
    assert_bits_equal(out, orc.per_group_kernel_sem(y.cpu(), "e2m1", 128), "quant of rotated")
    # (3) against the reference's op sequence on this GPU (fp16 GEMM under autocast + quant)
    import fpqvar_amd.quant_utils as qu
    with torch.autocast("cuda", dtype=torch.float16):
        x1 = torch.matmul(h.to(dev), rot.block_random_hadamard_matrix(1920, 128, dev, 42).float())
    ref = qu.fp_quant_e2_per_group_cuda(x1, 4, 128)
    agree = float((ref.view(torch.int16) == out.view(torch.int16)).float().mean())
    assert agree > 0.97, f"only {agree:.4f} of the outputs agree with matmul+quant"
    # other tables go through the same kernel
    o3, y3 = rot.rotate_quant(x.to(dev), "e2m3", smooth=s, return_rotated=True)
    assert_bits_equal(o3, orc.per_group_kernel_sem(y3.cpu(), "e2m3", 128, out_dtype=torch.float16), "e2m3 after rotate")
    assert_bits_equal(y3, y, "rotation independent of the table")


@pytest.mark.parametrize("rows,cols", ((1, 128), (3, 384), (17, 1920), (33, 2048), (257, 1152), (40000, 128)))
@pytest.mark.parametrize("in_dtype", (torch.float16, torch.float32))
def test_rotate_quant_tiles(dev, rows, cols, in_dtype):
    """The matrix-core rotation works on tiles of 16 groups per wavefront, persistent over the tensor: tensors of less
    than one tile, a ragged last tile, more tiles than resident wavefronts; non-finite inputs and all-zero groups go
    through the quantizer exactly as the rotated values say (torch's amax keeps NaN; a group with one non-finite input
    has no finite output)."""
    from fpqvar_amd import rotation as rot
    g = torch.Generator().manual_seed(rows * 7 + cols)
    x = (torch.randn(rows, cols, generator=g) * torch.exp(0.7 * torch.randn(rows, cols, generator=g))).to(in_dtype)
    x[0, :128] = 0
    if rows >= 3:
        x[1, 5] = float("nan")
        x[2, 77] = float("inf")
    if rows >= 17:
        x[5, 130 % cols] = float("inf")
        x[5, 140 % cols] = float("-inf")    # a group with both: outputs are +-inf or NaN
        x[16, cols - 1] = float("-inf")
        x[9, :128] = 60000.0                   # finite inputs whose rotated output overflows fp16: inf, as the reference's GEMM
        x[10, :128] = torch.where(torch.arange(128) % 2 == 0, 60000.0, -60000.0).to(x.dtype)
    out, y = rot.rotate_quant(x.to(dev), "e2m1", return_rotated=True)
    assert_bits_equal(rot.rotate_quant(x.to(dev), "e2m1"), out, "emit vs no-emit")
    # the yardstick group by group: in the reference's dense GEMM with the block-diagonal Q (tr/basic_var.py:263) the zero
    # blocks turn one non-finite input into NaN for its whole row (0 * inf); the fused kernels confine it to its group
    q_h = rot.block_random_hadamard_matrix(cols, 128, "cpu", 42).float().half()
    exact = (x.half().view(-1, 128).double() @ q_h[:128, :128].double()).view(rows, cols)
    exact = torch.where(exact.abs() >= 65520.0, exact.sign() * float("inf"), exact)     # what fp16 cannot hold rounds to inf
    assert torch.equal(torch.isfinite(y.cpu()), torch.isfinite(exact)), "non-finite outputs in other places"
    fin = torch.isfinite(exact)
    # error bound: one fp16 rounding + the fp32 accumulation of 128 terms (visible on cancelling outputs of groups
    # with a wide range of magnitudes: the sums are exact in fp32 only while the addends are of like magnitude)
    yf, rf = torch.where(fin, y.cpu().double(), torch.zeros_like(exact)), torch.where(fin, exact, torch.zeros_like(exact))
    ulp = torch.maximum(2.0 ** (torch.floor(torch.log2(rf.abs().clamp_min(2.0 ** -14))) - 10), torch.tensor(2.0 ** -24, dtype=torch.float64))
    l1 = torch.where(torch.isfinite(x), x, torch.zeros_like(x)).double().abs().view(rows, cols // 128, 128).sum(-1, keepdim=True)
    bound = 0.5 * ulp * 1.001 + (2.0 ** -22 * 128 ** -0.5) * l1.expand(-1, -1, 128).reshape(rows, cols)
    worst = float(((yf - rf).abs() / bound).max())
    assert worst <= 1.0, f"rotated value {worst:.2f}x the rounding + accumulation bound"
    assert float(((yf - rf).abs() > 0.5 * ulp * 1.001).double().mean()) < 1e-3
    assert_bits_equal(out, orc.per_group_kernel_sem(y.cpu(), "e2m1", 128), "quant of rotated")
    o6, y6 = rot.rotate_quant(x.to(dev), "e2m3", return_rotated=True)
    assert_bits_equal(y6, y, "rotation independent of the table")
    assert_bits_equal(o6, orc.per_group_kernel_sem(y.cpu(), "e2m3", 128, out_dtype=torch.float16), "e2m3 after rotate")


def test_rotate_butterfly_switch(dev, tmp_path):
    """FPQ_ROT_BUTTERFLY=1 (read once per process, hence a child process) selects the butterfly form of the transform in
    both producers: same contract - quantization exact on the rotated values it produced - and rotated values that agree
    with the matrix-core form up to the last bit of rare elements."""
    import subprocess
    import sys
    from fpqvar_amd import rotation as rot
    script = (
        "import sys, torch\n"
        "sys.path.insert(0, %r)\n"
        "from fpqvar_amd import rotation as rot\n"
        "d = torch.load(sys.argv[1])\n"
        "dev = torch.device('cuda:0')\n"
        "o, y = rot.rotate_quant(d['x'].to(dev), 'e2m1', return_rotated=True)\n"
        "oa, ha, ya = rot.adaln_rotate_quant(d['xa'].to(dev), d['scale'].to(dev), d['shift'].to(dev), 'e2m1', return_intermediates=True)\n"
        "torch.save({'o': o.cpu(), 'y': y.cpu(), 'oa': oa.cpu(), 'ha': ha.cpu(), 'ya': ya.cpu()}, sys.argv[2])\n"
    ) % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    g = torch.Generator().manual_seed(77)
    d = {"x": (torch.randn(65, 1920, generator=g) * torch.exp(0.5 * torch.randn(65, 1920, generator=g))).half(),
         "xa": torch.randn(3, 21, 1920, generator=g).half(),
         "scale": (torch.randn(3, 1, 1920, generator=g) * 0.3).half(), "shift": (torch.randn(3, 1, 1920, generator=g) * 0.3).half()}
    fin, fout = str(tmp_path / "in.pt"), str(tmp_path / "out.pt")
    torch.save(d, fin)
    env = dict(os.environ, FPQ_ROT_BUTTERFLY="1")
    subprocess.run([sys.executable, "-c", script, fin, fout], check=True, env=env, timeout=300)
    b = torch.load(fout)
    o, y = rot.rotate_quant(d["x"].to(dev), "e2m1", return_rotated=True)
    oa, ha, ya = rot.adaln_rotate_quant(d["xa"].to(dev), d["scale"].to(dev), d["shift"].to(dev), "e2m1", return_intermediates=True)
    assert_bits_equal(b["ha"], ha, "modulated rows do not depend on the form of the transform")
    for name, yb, ym in (("rotate", b["y"], y.cpu()), ("adaln", b["ya"], ya.cpu())):
        dd = _ulp_diff_f16(yb, ym)
        assert int(dd.max()) <= 1 and float((dd > 0).float().mean()) < 1e-3, f"{name}: the two forms disagree"
    assert_bits_equal(b["o"], orc.per_group_kernel_sem(b["y"], "e2m1", 128), "butterfly form: quant of rotated")
    assert_bits_equal(b["oa"], orc.per_group_kernel_sem(b["ya"].reshape(-1, 1920), "e2m1", 128).view_as(b["oa"]), "butterfly adaLN: quant of rotated")
    assert_bits_equal(o, orc.per_group_kernel_sem(y.cpu(), "e2m1", 128), "matrix-core form: quant of rotated")


@pytest.mark.parametrize("C", (1920, 2304))
@pytest.mark.parametrize("x_dtype", (torch.float16, torch.float32))
@pytest.mark.parametrize("table", ("e2m3", "e3m2"))
def test_adaln_fp6_hardware_levels_equal_the_table_form(dev, table, x_dtype, C):
    """The adaLN producer's E2M3 / E3M2 value outputs (per group and per token, rows of 13 .. 16 groups) take their levels
    from the FP6 conversion hardware since round 4; the library switch FPQ_NO_HW6 keeps the bucket table: bit-equal,
    including rows with non-finite values, all-zero rows and ragged batch entries; and the quantization is the oracle's on
    the rotated rows the kernel emits."""
    from fpqvar_amd import rotation as rot
    g = torch.Generator().manual_seed(90)
    B, L = 5, 23
    x = (torch.randn(B, L, C, generator=g) * torch.exp(0.5 * torch.randn(B, L, C, generator=g))).to(x_dtype)
    x[0, 1] = 0.0
    x[1, 2, 77] = float("inf")
    x[2, 3, 5] = float("nan")
    x[3, 4] *= 1e-3
    sc = (torch.randn(B, 1, C, generator=g) * 0.3).half().to(dev)
    sh = (torch.randn(B, 1, C, generator=g) * 0.3).half().to(dev)
    sm = (torch.rand(C, generator=g) + 0.5).to(dev)
    xd = x.to(dev)

    def both(fn):
        a = fn()
        with _lib.option("FPQ_NO_HW6", 1):
            b = fn()
        return a, b
    a, b = both(lambda: rot.adaln_rotate_quant(xd, sc, sh, table, smooth=sm))
    assert_bits_equal(a, b, f"adaLN {table} per group: hardware levels vs table")
    a, b = both(lambda: rot.adaln_rotate_quant_token(xd, sc, sh, table, smooth=sm))
    assert_bits_equal(a, b, f"adaLN {table} per token: hardware levels vs table")
    out, h, y = rot.adaln_rotate_quant(xd, sc, sh, table, smooth=sm, return_intermediates=True)   # the emitting (table) form
    assert_bits_equal(rot.adaln_rotate_quant(xd, sc, sh, table, smooth=sm), out, "emit vs no-emit")
    assert_bits_equal(out, orc.per_group_kernel_sem(y.cpu().reshape(-1, C), table, 128, out_dtype=torch.float16).view_as(out),
                      f"adaLN {table}: quantization of the rotated rows vs oracle")


def test_adaln_tail_tiers_switch(dev, tmp_path):
    """FPQ_ADALN_TAIL / FPQ_ADALN_ROWS (read once per process, hence a child process): the last batch entries of the grid cut
    into finer tiers of 8 and 4 rows per workgroup - off by default (profiles/r03_adaln_partition.txt: never faster), but
    the tier decode and the magic-number division of adaln_mfma_kernel ship, so they are covered: results must not depend
    on how rows are cut into workgroups."""
    import subprocess
    import sys
    from fpqvar_amd import rotation as rot
    script = (
        "import sys, torch\n"
        "sys.path.insert(0, %r)\n"
        "from fpqvar_amd import rotation as rot\n"
        "d = torch.load(sys.argv[1])\n"
        "dev = torch.device('cuda:0')\n"
        "res = {}\n"
        "for k in ('x16', 'x32', 'x2304'):\n"
        "    x = d[k].to(dev)\n"
        "    c = x.shape[-1]\n"
        "    res[k] = rot.adaln_rotate_quant(x, d['scale'][..., :c].contiguous().to(dev), d['shift'][..., :c].contiguous().to(dev), 'e2m1',\n"
        "                                    smooth=d['smooth'][:c].contiguous().to(dev)).cpu()\n"
        "    res[k + '_mx'] = tuple(t.cpu() for t in rot.adaln_rotate_quant_mx(x, d['scale'][..., :c].contiguous().to(dev),\n"
        "                                                                      d['shift'][..., :c].contiguous().to(dev)))\n"
        "torch.save(res, sys.argv[2])\n"
    ) % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    g = torch.Generator().manual_seed(78)
    B, L = 7, 53          # batch entries of 53 rows: chunks of 12, 8 and 4 rows all end ragged
    d = {"x16": torch.randn(B, L, 1920, generator=g).half(), "x32": torch.randn(B, L, 1920, generator=g),
         "x2304": torch.randn(B, L, 2304, generator=g).half(),
         "scale": (torch.randn(B, 1, 2304, generator=g) * 0.3).half(), "shift": (torch.randn(B, 1, 2304, generator=g) * 0.3).half(),
         "smooth": torch.rand(2304, generator=g) + 0.5}
    fin, fout = str(tmp_path / "in.pt"), str(tmp_path / "out.pt")
    torch.save(d, fin)
    env = dict(os.environ, FPQ_ADALN_TAIL="120", FPQ_ADALN_ROWS="12")   # tiers: 12 rows, then 8 (3 entries), then 4 (3 entries)
    subprocess.run([sys.executable, "-c", script, fin, fout], check=True, env=env, timeout=300)
    b = torch.load(fout)
    for k in ("x16", "x32", "x2304"):
        x = d[k].to(dev)
        c = x.shape[-1]
        sc, sh, sm = d["scale"][..., :c].contiguous().to(dev), d["shift"][..., :c].contiguous().to(dev), d["smooth"][:c].contiguous().to(dev)
        assert_bits_equal(b[k], rot.adaln_rotate_quant(x, sc, sh, "e2m1", smooth=sm), f"{k}: tiers vs the plain grid")
        codes, scales = rot.adaln_rotate_quant_mx(x, sc, sh)
        assert torch.equal(b[k + "_mx"][0], codes.cpu()) and torch.equal(b[k + "_mx"][1].view(torch.int16), scales.cpu().view(torch.int16)), k


@pytest.mark.parametrize("x_dtype", (torch.float16, torch.float32))
def test_producers_at_full_size_equal_their_slices(dev, x_dtype):
    """BASELINE-size launches ([65536 x 1920]: persistent wavefronts, several passes per wavefront, 100 batch entries of 655
    tokens) against launches on slices of the same tensors: a group's / row's result does not depend on the tile, the
    pass or the workgroup it was computed in; plus the oracle on rows from both ends and the middle."""
    from fpqvar_amd import gemm, rotation as rot
    g = torch.Generator(device=dev).manual_seed(123)
    R, C = 65536, 1920
    x = (torch.randn(R, C, device=dev, generator=g) * 1.3).to(x_dtype)
    out, y = rot.rotate_quant(x, "e2m1", return_rotated=True)
    assert_bits_equal(rot.rotate_quant(x, "e2m1"), out, "emit vs no-emit at full size")
    codes, scales = rot.rotate_quant_mx(x)
    for lo, hi in ((0, 300), (32700, 33111), (R - 257, R)):
        o_s, y_s = rot.rotate_quant(x[lo:hi].contiguous(), "e2m1", return_rotated=True)
        assert_bits_equal(out[lo:hi], o_s, f"rotate_quant rows {lo}:{hi}")
        assert_bits_equal(y[lo:hi], y_s, f"rotated rows {lo}:{hi}")
        assert_bits_equal(o_s, orc.per_group_kernel_sem(y_s.cpu(), "e2m1", 128), f"oracle rows {lo}:{hi}")
        assert_bits_equal(gemm.dequantize_mx(codes[lo:hi].contiguous(), scales[lo:hi].contiguous()).half(), o_s, f"codes rows {lo}:{hi}")
    del out, y, codes, scales
    B, L = 100, 655
    xa = x[:B * L].view(B, L, C)
    scale = (torch.randn(B, 1, C, device=dev, generator=g) * 0.3).half()
    shift = (torch.randn(B, 1, C, device=dev, generator=g) * 0.3).half()
    s = torch.rand(C, device=dev, generator=g) + 0.5
    full = rot.adaln_rotate_quant(xa, scale, shift, "e2m1", smooth=s)
    tok = rot.adaln_rotate_quant_token(xa, scale, shift, "e2m3", smooth=s)
    for b0, b1 in ((0, 2), (49, 51), (98, 100)):
        part, _, y_p = rot.adaln_rotate_quant(xa[b0:b1].contiguous(), scale[b0:b1].contiguous(), shift[b0:b1].contiguous(), "e2m1",
                                              smooth=s, return_intermediates=True)
        assert_bits_equal(full[b0:b1], part, f"adaln entries {b0}:{b1}")
        assert_bits_equal(part, orc.per_group_kernel_sem(y_p.cpu().reshape(-1, C), "e2m1", 128).view_as(part), f"adaln oracle {b0}:{b1}")
        assert_bits_equal(tok[b0:b1], orc.per_token_kernel_sem(y_p.cpu(), "e2m3"), f"adaln per token {b0}:{b1}")


@pytest.mark.parametrize("mod_dtype", (torch.float16, torch.float32))
def test_adaln_two_rows_per_tile_at_c1024(dev, mod_dtype, lib_options):
    """C = 1024 (VAR-d16): two consecutive rows of a batch entry share one matrix-core tile (PAIR2).  Bit for bit against
    the one-row-per-tile kernel - the emitting form always is one, FPQ_ADALN_NO_PAIR2 forces it - for batch entries of
    odd and even length (a lone last row), one-row entries, the BASELINE-like shape, values and FP4 operands, with and
    without a smoothing vector; plus the oracle's quantizer on the emitted rotated rows."""
    from fpqvar_amd import gemm, rotation as rot
    g = torch.Generator().manual_seed(1024)
    C = 1024
    for B, L in ((3, 5), (7, 1), (2, 2), (4, 7), (5, 16), (33, 131), (64, 512)):
        x = (torch.randn(B, L, C, generator=g) * 1.3 + 0.2).half().to(dev)
        x[0, 0] = x[0, 0] * 0 + 3.0                       # a constant row: zero variance
        if L > 2:
            x[B - 1, L - 1, :] = 1000.0 + torch.randn(C, generator=g).half().to(dev) * 0.01   # mean^2 >> var: the centred pass
        sc = (torch.randn(B, 1, C, generator=g) * 0.3).to(mod_dtype).to(dev)
        sh = (torch.randn(B, 1, C, generator=g) * 0.3).to(mod_dtype).to(dev)
        for sm in (None, (torch.rand(C, generator=g) + 0.5).to(dev)):
            got = rot.adaln_rotate_quant(x, sc, sh, "e2m1", smooth=sm)
            codes, scales = rot.adaln_rotate_quant_mx(x, sc, sh, smooth=sm)
            tab = rot.adaln_rotate_quant(x, sc, sh, "e2m3", smooth=sm)
            out_e, h_e, y_e = rot.adaln_rotate_quant(x, sc, sh, "e2m1", smooth=sm, return_intermediates=True)
            assert_bits_equal(got, out_e, f"B={B} L={L}: paired vs emitting (one row per tile)")
            assert_bits_equal(got, orc.per_group_kernel_sem(y_e.cpu().reshape(-1, C), "e2m1", 128).view_as(got), f"B={B} L={L}: oracle on rotated rows")
            assert_bits_equal(gemm.dequantize_mx(codes.view(B * L, -1), scales.view(B * L, -1)).half().view_as(got), got, f"B={B} L={L}: operands")
            lib_options("FPQ_ADALN_NO_PAIR2", 1)
            assert_bits_equal(rot.adaln_rotate_quant(x, sc, sh, "e2m1", smooth=sm), got, f"B={B} L={L}: unpaired values")
            assert_bits_equal(rot.adaln_rotate_quant(x, sc, sh, "e2m3", smooth=sm), tab, f"B={B} L={L}: unpaired, table form")
            c0, s0 = rot.adaln_rotate_quant_mx(x, sc, sh, smooth=sm)
            assert torch.equal(c0, codes) and torch.equal(s0.view(torch.int16), scales.view(torch.int16)), f"B={B} L={L}: unpaired operands"
            lib_options("FPQ_ADALN_NO_PAIR2", None)


@pytest.mark.parametrize("in_dtype", (torch.float16, torch.float32))
@pytest.mark.parametrize("cols", (128, 1024, 1920, 2560, 2688, 7680))
def test_rotate_quant_smoothing_vector_every_width(dev, cols, in_dtype):
    """rotate_quant(x, smooth=s) == rotate_quant(half(float(x) * s)) bit for bit, values and operands: the vector staged in
    LDS (<= 2560 channels) and read from global memory (wider rows), tiles that straddle rows (column chunk = vector
    index mod vectors per row, in 32 bits), several passes per workgroup (a smoothing vector selects a smaller grid),
    ragged last tile, signed zeros."""
    from fpqvar_amd import gemm, rotation as rot
    g = torch.Generator().manual_seed(cols)
    rows = 8192 * 1920 // cols + 3
    x = (torch.randn(rows, cols, generator=g) * 1.1).to(in_dtype)
    x[0, :7] = -0.0
    x[rows - 1, -5:] = 0.0
    s = torch.rand(cols, generator=g) * 1.5 + 0.25
    s[3] = 1.0
    xd, sd = x.to(dev), s.to(dev)
    h = (xd.float() * sd).half()
    want, want_y = rot.rotate_quant(h, "e2m1", return_rotated=True)
    got, got_y = rot.rotate_quant(xd, "e2m1", smooth=sd, return_rotated=True)
    assert_bits_equal(got_y, want_y, f"rotated rows, C={cols}")
    assert_bits_equal(got, want, f"values, C={cols}")
    assert_bits_equal(rot.rotate_quant(xd, "e2m1", smooth=sd), want, f"values (not emitting), C={cols}")
    assert_bits_equal(rot.rotate_quant(xd, "e2m3", smooth=sd), rot.rotate_quant(h, "e2m3"), f"table form, C={cols}")
    c1, s1 = rot.rotate_quant_mx(xd, smooth=sd)
    c0, s0 = rot.rotate_quant_mx(h)
    assert torch.equal(c1, c0) and torch.equal(s1.view(torch.int16), s0.view(torch.int16)), f"operands, C={cols}"


def test_rotate_quant_several_passes_ragged_end(dev):
    """The table-free rotate forms run one pass per workgroup up to 16384 (values) / 8192 (codes) workgroups and several
    beyond: 70001 rows of 1920 = 16407 workgroup-tiles, the last one partial - two passes per workgroup in both forms,
    a grid that does not divide the tiles.  Against launches on slices (one pass) and the oracle."""
    from fpqvar_amd import gemm, rotation as rot
    g = torch.Generator(device=dev).manual_seed(321)
    R, C = 70001, 1920
    x = (torch.randn(R, C, device=dev, generator=g) * 0.7).half()
    out = rot.rotate_quant(x, "e2m1")
    codes, scales = rot.rotate_quant_mx(x)
    for lo, hi in ((0, 129), (34990, 35100), (65500, 65700), (R - 131, R)):
        o_s, y_s = rot.rotate_quant(x[lo:hi].contiguous(), "e2m1", return_rotated=True)
        assert_bits_equal(out[lo:hi], o_s, f"rows {lo}:{hi}")
        assert_bits_equal(o_s, orc.per_group_kernel_sem(y_s.cpu(), "e2m1", 128), f"oracle rows {lo}:{hi}")
        assert_bits_equal(gemm.dequantize_mx(codes[lo:hi].contiguous(), scales[lo:hi].contiguous()).half(), o_s, f"codes rows {lo}:{hi}")


# ------------------------------------------------------------------ KV cache step and format search
def test_kv_cache_step(dev):
    from fpqvar_amd import kv_cache as kv
    g = torch.Generator().manual_seed(61)
    B, H, c = 4, 30, 64
    ck = torch.randn(B, 5, H, c, generator=g).half()      # BLHc (flash layout, tr/basic_var.py:173)
    cv = torch.randn(B, 5, H, c, generator=g).half()
    k = torch.randn(B, 4, H, c, generator=g).half()
    v = torch.randn(B, 4, H, c, generator=g).half()
    for bit, want_fn in ((6, lambda t: orc.per_token_kernel_sem(t, "e2m3")),
                         (4, lambda t: orc.per_group_kernel_sem(t, "e2m1", 128))):
        nk, nv = kv.update_kv_cache(ck.to(dev), cv.to(dev), k.to(dev), v.to(dev), True, bit, 1)
        assert_bits_equal(nk, torch.cat((want_fn(ck), k), 1), f"kv k bit {bit}")
        assert_bits_equal(nv, torch.cat((want_fn(cv), v), 1), f"kv v bit {bit}")
    nk, nv = kv.update_kv_cache(None, None, k.to(dev), v.to(dev), True, 6, 1)
    assert nk.data_ptr() == k.to(dev).data_ptr() or torch.equal(nk.cpu(), k)
    with pytest.raises(RuntimeError):          # permuted (BHLc) cache + kv_bit 6 raises, as in the reference
        kv.quantize_kv(ck.to(dev).permute(0, 2, 1, 3), 6)
    with pytest.raises(AssertionError):
        bad = k.clone()
        bad[0, 0, 0, 0] = float("nan")
        kv.update_kv_cache(ck.to(dev), cv.to(dev), bad.to(dev), v.to(dev), True, 6, 1)


def test_format_search_layer(dev):
    from fpqvar_amd import format_search as fs
    g = torch.Generator().manual_seed(62)
    w = torch.randn(256, 512, generator=g) * 0.05
    xs = [torch.randn(2, 9, 512, generator=g) * torch.exp(torch.randn(2, 9, 512, generator=g)) for _ in range(3)]
    wf, af, losses = fs.search_layer([x.to(dev).half() for x in xs], w.to(dev).half(), fs.FP6_FORMATS)
    # the same loop with the oracle's quantizers in float32 on the CPU
    tab = {"fp6_e2m3": "e2m3", "fp6_e3m2": "e3m2"}
    want = {}
    for a in fs.FP6_FORMATS:
        wq = orc.per_token_kernel_sem(w.half(), tab[a]).float()
        for b in fs.FP6_FORMATS:
            tot = 0.0
            for x in xs:
                xh = x.half()
                ref = xh.float() @ w.half().float().t()
                y = orc.per_token_kernel_sem(xh, tab[b]).float() @ wq.t()
                tot += float(torch.mean((ref - y) ** 2))
            want[(a, b)] = tot
    for key in want:
        assert abs(losses[key] - want[key]) <= 0.05 * want[key] + 1e-6, (key, losses[key], want[key])
    best = min(want, key=want.get)
    assert (wf, af) == best
    # the sample-by-sample loop (the reference's order of operations) gives the same numbers as the batched form
    _, _, loop = fs.search_layer([x.to(dev).half() for x in xs], w.to(dev).half(), fs.FP6_FORMATS, batched=False)
    for key in want:
        assert abs(loop[key] - losses[key]) <= 2e-3 * want[key] + 1e-7, (key, loop[key], losses[key])
    # FP4 twin (search/search_fp4_format.py:782-821): 3 x 3 formats, per-group-128 quantizers on weights and activations
    wf4, af4, l4 = fs.search_layer([x.to(dev).half() for x in xs], w.to(dev).half(), fs.FP4_FORMATS)
    tab4 = {"fp_e1": "e1m2", "fp_e2": "e2m1", "fp_e3": "e3m0"}
    want4 = {}
    for a in fs.FP4_FORMATS:
        wq = orc.per_group_kernel_sem(w.half(), tab4[a], 128).float()
        for b in fs.FP4_FORMATS:
            tot = 0.0
            for x in xs:
                xh = x.half()
                ref = xh.float() @ w.half().float().t()
                y = orc.per_group_kernel_sem(xh, tab4[b], 128).float() @ wq.t()
                tot += float(torch.mean((ref - y) ** 2))
            want4[(a, b)] = tot
    assert set(l4) == set(want4) and len(l4) == 9
    for key in want4:
        assert abs(l4[key] - want4[key]) <= 0.05 * want4[key] + 1e-6, (key, l4[key], want4[key])
    assert (wf4, af4) == min(want4, key=want4.get)
    # ragged calibration sets (different token counts per sample, as the dumps of different scale steps have)
    xr = [torch.randn(2, n, 512, generator=g) for n in (1, 4, 9, 16)]
    _, _, lb = fs.search_layer([x.to(dev).half() for x in xr], w.to(dev).half(), fs.FP4_FORMATS)
    _, _, ll = fs.search_layer([x.to(dev).half() for x in xr], w.to(dev).half(), fs.FP4_FORMATS, batched=False)
    for key in lb:
        assert abs(lb[key] - ll[key]) <= 2e-3 * ll[key] + 1e-7, (key, lb[key], ll[key])


# ------------------------------------------------------------------ hipGraph capture (no alloc / sync inside the ABI)
def test_graph_capture_and_replay(dev, qu):
    from fpqvar_amd import rotation as rot
    g = torch.Generator().manual_seed(71)
    x = torch.randn(400, 1920, generator=g).half().to(dev)
    h = torch.nn.functional.gelu(torch.randn(400, 7680, generator=g)).half().to(dev)
    static_x, static_h = x.clone(), h.clone()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):          # warm-up on the side stream (builds the host-side table cache)
        qu.fp_quant_e2_per_group_cuda(static_x, 4, 128)
        qu.fp_quant_e1m2_neg_e2m1_pos_per_group_cuda(static_h, 4, 128)
        rot.rotate_quant(static_x, "e2m1")
    torch.cuda.current_stream().wait_stream(s)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        y1 = qu.fp_quant_e2_per_group_cuda(static_x, 4, 128)
        y2 = qu.fp_quant_e1m2_neg_e2m1_pos_per_group_cuda(static_h, 4, 128)     # memset + 2 launches
        y3 = qu.fp6_quant_e2m3_per_token_cuda(static_x, 6)
        y4 = rot.rotate_quant(static_x, "e2m1")
    for trial in range(2):
        xn = torch.randn(400, 1920, generator=g).half()
        hn = torch.nn.functional.gelu(torch.randn(400, 7680, generator=g)).half()
        static_x.copy_(xn.to(dev))
        static_h.copy_(hn.to(dev))
        graph.replay()
        torch.cuda.synchronize()
        assert_bits_equal(y1, orc.per_group_kernel_sem(xn, "e2m1", 128), f"graph replay {trial} e2m1")
        assert_bits_equal(y2, orc.dual_per_group_kernel_sem(hn, "e1m2_neg", "e2m1_pos", 128, 1.0), "graph dual")
        assert_bits_equal(y3, orc.per_token_kernel_sem(xn, "e2m3"), "graph token")
        assert_bits_equal(y4, rot.rotate_quant(static_x, "e2m1"), "graph rotate")


@pytest.mark.parametrize("C,L", ((1920, 40), (2304, 17)))
@pytest.mark.parametrize("x_dtype", (torch.float16, torch.float32))
def test_adaln_rotate_quant_fused(dev, C, L, x_dtype):
    """The whole producer (LN, AdaLN modulate, GALT smooth, rotate) + quant in one launch against
    torch's own op chain on this GPU (tr/basic_var.py:263 under fp16 autocast) feeding the fused
    rotate+quant of the previous test."""
    from fpqvar_amd import rotation as rot
    g = torch.Generator().manual_seed(81)
    B = 3
    x = (torch.randn(B, L, C, generator=g) * 2 + 0.3).to(x_dtype).to(dev)
    scale = (torch.randn(B, 1, C, generator=g) * 0.3).half().to(dev)
    shift = (torch.randn(B, 1, C, generator=g) * 0.3).half().to(dev)
    s = (torch.rand(C, generator=g) * 1.5 + 0.25).to(dev)
    out, h, y = rot.adaln_rotate_quant(x, scale, shift, "e2m1", smooth=s, return_intermediates=True)
    assert_bits_equal(rot.adaln_rotate_quant(x, scale, shift, "e2m1", smooth=s), out, "emit vs no-emit")
    # torch's chain, op for op (layer_norm runs in fp32 under autocast)
    ln = torch.nn.functional.layer_norm(x.float(), (C,), eps=1e-6)
    t1 = ln.mul(scale.add(1))
    t32 = (t1 + shift).mul(s)                       # torch's fp32 value just before autocast's cast to fp16
    h_ref = t32.half()
    # LayerNorm's mean / rstd are fp32 reductions whose order is torch's business: allow their
    # rounding (a few 1e-7 relative to the terms BEFORE the `+ shift` cancellation) plus the half
    # ulp of the final fp16 rounding
    ulp16 = torch.maximum(t32.abs(), torch.tensor(2.0 ** -14, device=dev)).log2().floor().exp2() * 2.0 ** -10
    sc1 = scale.add(1).float().abs()
    bound = 0.5 * ulp16 + 4e-6 * (t1.abs() + shift.float().abs()) * s.abs() + 3e-6 * sc1 * s.abs()   # + mean's own rounding
    err = (h.float() - t32).abs()
    assert bool((err <= bound).all()), f"h error {float((err / bound).max()):.2f}x the bound"
    d = _ulp_diff_f16(h.cpu(), h_ref.cpu())
    assert float((d > 0).float().mean()) < 2e-3, "too many fp16 roundings differ from torch's chain"
    # given h, the rest is exactly the fused rotate+quant kernel
    out2, y2 = rot.rotate_quant(h, "e2m1", return_rotated=True)
    assert_bits_equal(y, y2, "rotated")
    assert_bits_equal(out, out2, "quantized")
    assert_bits_equal(out, orc.per_group_kernel_sem(y.cpu(), "e2m1", 128), "quant of rotated")
    # end to end against torch chain + dense GEMM + quant: agreement rate
    import fpqvar_amd.quant_utils as qu
    with torch.autocast("cuda", dtype=torch.float16):
        x1 = torch.matmul(ln.mul(scale.add(1)).add_(shift).mul(s),
                          rot.block_random_hadamard_matrix(C, 128, dev, 42).float())
    ref = qu.fp_quant_e2_per_group_cuda(x1, 4, 128)
    agree = float((ref.view(torch.int16) == out.view(torch.int16)).float().mean())
    assert agree > 0.97, agree
    # fp32 modulation tensors and no smoothing
    o3, h3, _ = rot.adaln_rotate_quant(x, scale.float(), shift.float(), "e2m3", return_intermediates=True)
    h3_ref = ln.mul(scale.float().add(1)).add_(shift.float()).half()
    assert float((_ulp_diff_f16(h3.cpu(), h3_ref.cpu()) > 0).float().mean()) < 2e-3


@pytest.mark.parametrize("x_dtype", (torch.float16, torch.float32))
@pytest.mark.parametrize("L", (1, 2, 5, 65))
def test_adaln_rotate_quant_short_entries(dev, L, x_dtype):
    """Batch entries of 1, 2, 5 tokens (the first scale steps of a generation: idle wavefronts in every workgroup) and of
    65 (a last workgroup with one row), many entries, fp16 and fp32 rows: against the fused rotate+quant on the emitted
    modulated row, and fp32 against fp16 rows of the same values."""
    from fpqvar_amd import rotation as rot
    C, B = 1920, 37
    g = torch.Generator().manual_seed(100 + L)
    x = (torch.randn(B, L, C, generator=g) * 1.5 + 0.1).half().to(x_dtype).to(dev)
    scale = (torch.randn(B, 1, C, generator=g) * 0.3).half().to(dev)
    shift = (torch.randn(B, 1, C, generator=g) * 0.3).half().to(dev)
    out, h, y = rot.adaln_rotate_quant(x, scale, shift, "e2m1", return_intermediates=True)
    assert_bits_equal(rot.adaln_rotate_quant(x, scale, shift, "e2m1"), out, "emit vs no-emit")
    out2, y2 = rot.rotate_quant(h, "e2m1", return_rotated=True)
    assert_bits_equal(y, y2, "rotated")
    assert_bits_equal(out, out2, "quantized")
    if x_dtype == torch.float32:   # the same values as fp16 rows: statistics differ in rounding only
        h16 = rot.adaln_rotate_quant(x.half(), scale, shift, "e2m1", return_intermediates=True)[1]
        assert float((_ulp_diff_f16(h.cpu(), h16.cpu()) > 0).float().mean()) < 2e-3   # (cancelling elements may move by more than one ulp)
        assert float((h.float() - h16.float()).abs().max()) < 2e-3


@pytest.mark.parametrize("C", (128, 512, 1024, 1280, 1536, 2048))
def test_adaln_rotate_quant_widths(dev, C):
    """Rows of 1 .. 16 groups in the matrix-core form of the producer (one row = one 16-group tile, the groups beyond the
    row are zeros in the operand image and fall outside the store range), per group and per token, batch entries of a
    length that leaves a ragged last workgroup: the modulated row equals what the wide rows give, and given that row
    everything downstream is the fused rotate+quant / the per-token quantizer bit for bit."""
    from fpqvar_amd import ops, rotation as rot
    g = torch.Generator().manual_seed(C)
    B, L = 5, 23
    x = (torch.randn(B, L, C, generator=g) * torch.exp(0.4 * torch.randn(B, L, C, generator=g)) + 0.2).half().to(dev)
    scale = (torch.randn(B, 1, C, generator=g) * 0.3).half().to(dev)
    shift = (torch.randn(B, 1, C, generator=g) * 0.3).half().to(dev)
    s = (torch.rand(C, generator=g) * 1.5 + 0.25).to(dev)
    out, h, y = rot.adaln_rotate_quant(x, scale, shift, "e2m1", smooth=s, return_intermediates=True)
    assert_bits_equal(rot.adaln_rotate_quant(x, scale, shift, "e2m1", smooth=s), out, "emit vs no-emit")
    ln = torch.nn.functional.layer_norm(x.float(), (C,), eps=1e-6)
    h_ref = (ln.mul(scale.add(1)) + shift).mul(s).half()
    assert float((_ulp_diff_f16(h.cpu(), h_ref.cpu()) > 0).float().mean()) < 5e-3
    out2, y2 = rot.rotate_quant(h, "e2m1", return_rotated=True)
    assert_bits_equal(y, y2, "rotated")
    assert_bits_equal(out, out2, "quantized")
    assert_bits_equal(out, orc.per_group_kernel_sem(y.cpu(), "e2m1", 128), "quant of rotated")
    tok = rot.adaln_rotate_quant_token(x, scale, shift, "e2m3", smooth=s)
    assert_bits_equal(tok, ops.quant_rows(y, "e2m3", C, torch.float16), "per-token quant of the rotated row")
    assert_bits_equal(tok, orc.per_token_kernel_sem(y.cpu(), "e2m3"), "per-token oracle")


@pytest.mark.parametrize("source", ("qkv_view", "separate", "unaligned"))
@pytest.mark.parametrize("kv_bit", (6, 4))
def test_incremental_kv_equals_requantize_everything(dev, kv_bit, source):
    """10 scale steps of VAR's KV cache: quantizing each entry once (IncrementalKVCache, one fpq_kv_cache_step launch
    per step) reproduces the reference's re-quantize-the-whole-cache loop bit for bit (see
    tests/test_kv_idempotence.py) - with k / v as views of a fused qkv output (the model's case), as separate
    tensors, and through the multi-launch path taken when the views are not 16-byte aligned."""
    from fpqvar_amd import kv_cache as kv
    g = torch.Generator().manual_seed(91)
    B, H, c = 6, 30, 64
    patch = (1, 2, 3, 4, 5, 6, 8, 10, 13, 16)
    inc = kv.IncrementalKVCache(B, sum(p * p for p in patch), H, c, kv_bit, device=dev)
    ck = cv = None
    for pn in patch:
        k = torch.nn.functional.normalize(torch.randn(B, pn * pn, H, c, generator=g), dim=-1).half().to(dev)
        v = torch.randn(B, pn * pn, H, c, generator=g).half().to(dev)
        if source == "qkv_view":
            qkv = torch.stack((torch.zeros_like(k), k, v), dim=2)          # [B, L, 3, H, c] as mat_qkv's output is viewed
            k_in, v_in = qkv[:, :, 1], qkv[:, :, 2]
            assert not k_in.is_contiguous() or pn == 1
        elif source == "unaligned":
            buf = torch.zeros(2, B, pn * pn, H * c + 4, dtype=torch.float16, device=dev)
            buf[..., 4:] = torch.stack((k, v)).view(2, B, pn * pn, H * c)
            k_in, v_in = buf[0, :, :, 4:].view(B, pn * pn, H, c), buf[1, :, :, 4:].view(B, pn * pn, H, c)
            assert k_in.data_ptr() % 16 != 0
        else:
            k_in, v_in = k, v
        ck, cv = kv.update_kv_cache(ck, cv, k, v, True, kv_bit, 1)       # the reference's O(L^2) loop
        ik, iv = inc.append(k_in, v_in)
        assert_bits_equal(ik.contiguous(), ck, f"K step pn={pn}")
        assert_bits_equal(iv.contiguous(), cv, f"V step pn={pn}")


def test_kv_cache_step_argument_checks(dev):
    from fpqvar_amd import ops
    cache = torch.zeros(2, 2, 8, 2, 64, dtype=torch.float16, device=dev)
    k = torch.randn(2, 3, 2, 64, device=dev).half()
    with pytest.raises(RuntimeError):
        ops.kv_cache_step(cache, 0, 0, k, k, 6, 64, "e2m3")              # 6 + 3 > max_len
    with pytest.raises(RuntimeError):
        ops.kv_cache_step(cache, 0, 4, k, k, 3, 64, "e2m3")              # would overwrite rows being quantized
    with pytest.raises(RuntimeError):
        ops.kv_cache_step(cache, 0, 0, k, k, 0, 48, "e2m3")              # rows of 48 halves have no lane mapping
    with pytest.raises(RuntimeError):
        ops.kv_cache_step(cache, 0, 0, k.float(), k.float(), 0, 64, "e2m3")
    ops.kv_cache_step(cache, 0, 0, k, k, 0, 64, "e2m3")                    # copy only
    assert torch.equal(cache[0, :, :3], k) and torch.equal(cache[1, :, :3], k) and not cache[:, :, 3:].any()
    ops.kv_cache_step(cache, 0, 3, k[:, :0], k[:, :0], 3, 64, "e2m3")      # quantize only
    from fpqvar_amd import quant_utils as qu
    assert torch.equal(cache[0, :, :3], qu.fp6_quant_e2m3_per_token_cuda(k, 6))


# ------------------------------------------------------------------ F2: hardware FP4 codes + MFMA GEMM
def test_mx_codes_reproduce_fake_quant(dev, qu):
    from fpqvar_amd import gemm
    g = torch.Generator().manual_seed(101)
    x = (torch.randn(300, 1920, generator=g) * torch.exp(0.5 * torch.randn(300, 1920, generator=g))).half()
    x[3, :128] = 0
    codes, scales = gemm.quantize_mx(x.to(dev))
    assert codes.shape == (300, 960) and scales.shape == (300, 15) and scales.dtype == torch.float16
    deq = gemm.dequantize_mx(codes, scales).half()           # level * scale, product exact in fp32, one rounding
    assert_bits_equal(deq.view(300, 1920), orc.per_group_kernel_sem(x, "e2m1", 128), "fp16 activations")
    w = torch.randn(384, 1920, generator=g) * 0.02
    wc, ws = gemm.quantize_mx(w.to(dev))
    assert ws.dtype == torch.float32
    assert_bits_equal(gemm.dequantize_mx(wc, ws).view(384, 1920), orc.per_group_kernel_sem(w, "e2m1", 128), "fp32 weights")


@pytest.mark.parametrize("T,O,K", ((256, 256, 1920), (1000, 5760, 1920), (130, 1928, 256), (64, 128, 7680),
                                   (20, 6912, 2304), (323, 9216, 2304), (1, 8, 128)))
def test_fp4_gemm(dev, T, O, K):
    from fpqvar_amd import gemm
    import fpqvar_amd.quant_utils as qu
    g = torch.Generator().manual_seed(102 + T)
    x = (torch.randn(T, K, generator=g) * torch.exp(0.3 * torch.randn(T, K, generator=g))).half().to(dev)
    w = (torch.randn(O, K, generator=g) * 0.02).to(dev)
    bias = (torch.randn(O, generator=g) * 0.1).half().to(dev)
    ac, asc = gemm.quantize_mx(x)
    wc, wsc = gemm.quantize_mx(w)
    y = gemm.linear_fp4(ac, asc, wc, wsc, bias)
    assert y.shape == (T, O) and y.dtype == torch.float16
    # exact reference from the decoded operands in float64
    a64 = gemm.dequantize_mx(ac, asc).double()
    w64 = gemm.dequantize_mx(wc, wsc).double()
    ref = a64 @ w64.t() + bias.double()
    err = (y.double() - ref).abs()
    tol = 2.0 ** -10 * ref.abs() + 1e-5 * (a64.abs() @ w64.abs().t()) + 1e-6     # fp16 output rounding + fp32 accumulation
    assert bool((err <= tol).all()), float((err / tol).max())
    # and the reference's own path: fp16 GEMM on the fake-quantized tensors (tr/quant_utils.py:765-767)
    wq16 = qu.fp_quant_e2_per_group_cuda(w, 4, 128).half()
    y_ref = torch.nn.functional.linear(qu.fp_quant_e2_per_group_cuda(x, 4, 128), wq16, bias)
    torch.testing.assert_close(y.float(), y_ref.float(), rtol=2e-2, atol=2e-2 * float(ref.abs().mean()))
    assert gemm.linear_fp4(ac, asc, wc, wsc.half(), None).shape == (T, O)        # fp16 weight scales, no bias


@pytest.mark.parametrize("T,O,K", ((300, 392, 1920), (16384, 8192, 128)))
def test_fp4_gemm_tile_configurations_agree(dev, T, O, K, lib_options):
    """Every tiling of the FP4 GEMM (library switch FPQ_GEMM_CFG): the three LDS-DMA tilings do the same arithmetic per
    element and must agree bit for bit - with bias, gate and residual, on ragged edges - and with the default choice
    ((300, 392): 12 tiles of 128 x 128, the smallest tile is the default; (16384, 8192): 4096 tiles of 256 x 128, past
    the size from which the larger tile is); the register-staged tilings multiply the two scales first and stay within
    the rounding of one fp32 product.  A bias that is not 8-byte aligned: cloned by the Python layers, served by the
    register-staged kernel through the bare C ABI."""
    from fpqvar_amd import gemm
    g = torch.Generator().manual_seed(7 + T)
    x = (torch.randn(T, K, generator=g) * torch.exp(0.3 * torch.randn(T, K, generator=g))).half().to(dev)
    w = (torch.randn(O, K, generator=g) * 0.02).to(dev)
    bias_store = (torch.randn(O + 4, generator=g) * 0.1).half().to(dev)
    bias = bias_store[:O]
    gate = torch.randn(6 if T == 300 else 8, 1, O, generator=g).half().to(dev)     # [B, 1, outs]: 50 / 2048 rows per gate row
    resid = torch.randn(T, O, generator=g).half().to(dev)
    ac, asc = gemm.quantize_mx(x)
    wc, wsc = gemm.quantize_mx(w)

    def run(cfg, b=bias):
        lib_options("FPQ_GEMM_CFG", None if cfg is None else int(cfg))
        plain = gemm.linear_fp4(ac, asc, wc, wsc, b)
        fused = gemm.linear_fp4(ac, asc, wc, wsc, b, gate=gate, residual=resid)
        return plain, fused

    base = run("20")
    for cfg in ("10", "30", None):
        got = run(cfg)
        assert_bits_equal(got[0], base[0], f"cfg {cfg} plain")
        assert_bits_equal(got[1], base[1], f"cfg {cfg} fused")
    scale = float(base[0].float().abs().mean())
    for cfg in ("0", "1", "2"):
        got = run(cfg)
        torch.testing.assert_close(got[0].float(), base[0].float(), rtol=2e-3, atol=2e-3 * scale)
        torch.testing.assert_close(got[1].float(), base[1].float(), rtol=2e-3, atol=2e-3 * (scale + 1))
    odd_bias = bias_store[1:O + 1]                        # 2-byte aligned bias
    assert odd_bias.data_ptr() % 8 == 2
    # the Python layers hand the kernels an aligned copy (round 5): same result as an aligned bias of the same values
    assert_bits_equal(run(None, odd_bias)[0], run(None, odd_bias.clone())[0], "misaligned bias through the binding")
    # ... the bare C ABI serves a bias that is not 8-byte aligned with the register-staged kernel (the LDS-DMA kernel reads
    # the bias four outputs at a time)
    lib_options("FPQ_GEMM_CFG", None)
    raw = torch.empty(T, O, dtype=torch.float16, device=dev)
    rc = _lib.lib().fpq_gemm_fp4_mx_ex(ac.data_ptr(), asc.data_ptr(), wc.data_ptr(), wsc.data_ptr(), _lib.dtype_id(wsc.dtype),
                                       odd_bias.data_ptr(), raw.data_ptr(), T, O, K, None, _lib.stream_ptr(dev))
    assert rc == 0
    assert_bits_equal(raw, run("0", odd_bias.clone())[0], "C ABI: misaligned bias goes to the register-staged kernel")


@pytest.mark.parametrize("kind", ("fp6", "fp8"))
@pytest.mark.parametrize("T,O,K", ((300, 392, 1920), (4100, 520, 256)))
def test_row_scaled_gemm_tile_configurations_agree(dev, kind, T, O, K, lib_options):
    """The two tilings of the FP6 / FP8 row-scaled GEMMs (FPQ_GEMM6_CFG / FPQ_GEMM8_CFG: 128 x 128 and 256 x 128) sum a tile's
    K in the same order and share the epilogue: bit-equal outputs, plain and with bias + gate + residual, on ragged edges;
    4100 tokens is past the size from which the larger FP6 tile is the default.  And against float64 on the decoded operands."""
    from fpqvar_amd import gemm
    g = torch.Generator().manual_seed(11 + T)
    x = (torch.randn(T, K, generator=g) * torch.exp(0.3 * torch.randn(T, K, generator=g))).half().to(dev)
    w = (torch.randn(O, K, generator=g) * 0.02).to(dev)
    bias = (torch.randn(O, generator=g) * 0.1).half().to(dev)
    B = 6 if T == 300 else 4                                     # 50 / 1025 rows per gate row
    gate = torch.randn(B, 1, O, generator=g).half().to(dev)
    resid = torch.randn(T, O, generator=g).half().to(dev)
    quant, lin, env = {"fp6": (gemm.quantize_fp6, gemm.linear_fp6, "FPQ_GEMM6_CFG"),
                       "fp8": (gemm.quantize_fp8, gemm.linear_fp8, "FPQ_GEMM8_CFG")}[kind]
    a, wq = quant(x), quant(w)

    def run(cfg):
        lib_options(env, None if cfg is None else int(cfg))
        return lin(*a, *wq, bias), lin(*a, *wq, bias, gate, resid)

    base = run("0")
    for cfg in ("1", None):
        got = run(cfg)
        assert_bits_equal(got[0], base[0], f"{kind} cfg {cfg} plain")
        assert_bits_equal(got[1], base[1], f"{kind} cfg {cfg} fused")
    assert_bits_equal(base[1].view(B, T // B, O), resid.view(B, T // B, O) + base[0].view(B, T // B, O).mul(gate), f"{kind} fused tail")
    deq = {"fp6": gemm.dequantize_fp6, "fp8": gemm.dequantize_fp8}[kind]
    a64, w64 = deq(*a).double().view(T, K), deq(*wq).double().view(O, K)
    ref = a64 @ w64.t() + bias.double()
    err = (base[0].double() - ref).abs()
    tol = 2.0 ** -10 * ref.abs() + 1e-5 * (a64.abs() @ w64.abs().t()) + 1e-6     # fp16 output rounding + fp32 accumulation
    assert bool((err <= tol).all()), float((err / tol).max())


@pytest.mark.parametrize("kind", ("fp4", "fp6", "fp8"))
def test_gemm_full_size_row_and_column_permutations(dev, kind):
    """The matrix-core GEMMs at the bench shape [65536 x 1920] . [1920 -> 5760], through a property that needs no reference:
    every output element depends on ONE activation row and ONE weight row, and the K-sum order inside a tile is fixed, so
    permuting the activation rows (or the weight rows) permutes the output rows (columns) bit for bit - whatever tile,
    wavefront and lane an element lands in.  Plus a float64 check on a 64 x 64 corner of the result."""
    from fpqvar_amd import gemm
    T, O, K = 65536, 5760, 1920
    g = torch.Generator(device=dev).manual_seed(21)
    x = (torch.randn(T, K, device=dev, generator=g) * torch.exp(0.3 * torch.randn(T, 1, device=dev, generator=g))).half()
    w = torch.randn(O, K, device=dev, generator=g) * 0.02
    bias = (torch.randn(O, device=dev, generator=g) * 0.1).half()
    quant, lin, deq = {"fp4": (gemm.quantize_mx, gemm.linear_fp4, gemm.dequantize_mx), "fp6": (gemm.quantize_fp6, gemm.linear_fp6, gemm.dequantize_fp6),
                       "fp8": (gemm.quantize_fp8, gemm.linear_fp8, gemm.dequantize_fp8)}[kind]
    (ac, asc), (wc, wsc) = quant(x), quant(w)
    del x, w
    y = lin(ac, asc, wc, wsc, bias)
    pt = torch.randperm(T, device=dev, generator=g)
    assert_bits_equal(lin(ac[pt].contiguous(), asc[pt].contiguous(), wc, wsc, bias), y[pt], f"{kind}: activation rows permuted")
    po = torch.randperm(O, device=dev, generator=g)
    assert_bits_equal(lin(ac, asc, wc[po].contiguous(), wsc[po].contiguous(), bias[po].contiguous()), y[:, po], f"{kind}: weight rows permuted")
    rows = torch.tensor([0, 1, 127, 128, 255, 256, 4095, 32767, 32768, 65535] + list(range(1000, 1054)), device=dev)
    cols = torch.tensor([0, 3, 4, 63, 64, 127, 128, 1919, 1920, 5759] + list(range(2000, 2054)), device=dev)
    a64 = deq(ac[rows].contiguous(), asc[rows].contiguous()).double().view(len(rows), K)
    w64 = deq(wc[cols].contiguous(), wsc[cols].contiguous()).double().view(len(cols), K)
    ref = a64 @ w64.t() + bias[cols].double()
    err = (y[rows][:, cols].double() - ref).abs()
    tol = 2.0 ** -10 * ref.abs() + 1e-5 * (a64.abs() @ w64.abs().t()) + 1e-6
    assert bool((err <= tol).all()), float((err / tol).max())


def test_fp4_linear_module(dev, golden):
    """FP4Linear vs the reference-path QuantizedLinear (fake-quant + fp16 GEMM) on the golden toy layer."""
    from fpqvar_amd import gemm
    from fpqvar_amd.quant_linear import QuantizedLinear
    lin = torch.nn.Linear(128, 384, bias=False)
    with torch.no_grad():
        lin.weight.copy_(from_bits(golden["ql/w0/attn.mat_qkv"]))
    lin = lin.to(dev)
    x = torch.randn(3, 50, 128, device=dev).half()
    real = gemm.FP4Linear.from_float(lin)
    fake = QuantizedLinear.from_float(lin, weight_quant="per_group", act_quant="per_group", w_bit=4, a_bit=4,
                                      activation_fp_quant=True, weight_fp_quant=True, act_fp_type="fp_e2",
                                      weight_fp_type="fp_e2")
    # same quantization decisions: decoded weights == the fake-quantized weights, bit for bit
    assert_bits_equal(gemm.dequantize_mx(real.w_codes, real.w_scales).view(384, 128), fake.weight, "weights")
    fake.weight = fake.weight.half()
    y_real, y_fake = real(x), fake(x)
    assert y_real.shape == (3, 50, 384)
    torch.testing.assert_close(y_real.float(), y_fake.float(), rtol=2e-2, atol=2e-3)


def test_more_than_2_31_elements(dev, qu):
    """64-bit indexing: a tensor with more than 2^31 elements (4.3 GB of fp16), checked on slices from
    both ends and across the 2^31 boundary."""
    rows, cols = 1_120_000, 1920                      # 2.15e9 elements
    assert rows * cols > 2 ** 31
    x = torch.empty(rows, cols, dtype=torch.float16, device=dev)
    g = torch.Generator(device=dev).manual_seed(5)
    chunk = 100_000
    for r in range(0, rows, chunk):
        x[r:r + chunk].normal_(generator=g)
    out = qu.fp_quant_e2_per_group_cuda(x, 4, 128)
    boundary = 2 ** 31 // cols
    for r0 in (0, boundary - 2, rows - 64):
        sl = slice(r0, r0 + 64)
        assert_bits_equal(out[sl], orc.per_group_kernel_sem(x[sl].cpu(), "e2m1", 128), f"rows {r0}..")
    del out
    out = qu.fp_quant_e1m2_neg_e2m1_pos_per_group_cuda(x, 4, 128)
    assert_bits_equal(out[rows - 64:], orc.dual_per_group_kernel_sem(x[rows - 64:].cpu(), "e1m2_neg", "e2m1_pos", 128, 1.0),
                      "dual, last rows")
    del out
    out = qu.fp6_quant_e2m3_per_token_cuda(x, 6)
    assert_bits_equal(out[rows - 64:], orc.per_token_kernel_sem(x[rows - 64:].cpu(), "e2m3"), "per-token, last rows")


# ------------------------------------------------------------------ F4: packed on-disk format
def test_packed_file_reproduces_calibration(dev, qu, tmp_path):
    """pack -> save -> load -> dequantize == the tensor the reference keeps after from_float + .half()
    (oracle on CPU), for per-group FP4, per-channel FP6, and a smoothed + rotated layer; the FP4 layer's
    hardware operands give the same GEMM result as quantizing the weight directly."""
    from fpqvar_amd import gemm, packed, rotation as rot
    g = torch.Generator().manual_seed(77)
    w1 = torch.randn(384, 1920, generator=g) * 0.02
    w2 = torch.randn(256, 768, generator=g) * 0.02
    s = torch.rand(1920, generator=g) + 0.5
    layers = {
        "a.fp4": packed.pack_weight(w1.to(dev), "e2m1", 128, bias=torch.randn(384, generator=g).to(dev)),
        "b.fp6": packed.pack_weight(w2.to(dev), "e2m3", 768),
        "c.fp6g": packed.pack_weight(w2.to(dev), "e3m2", 128),
        "d.rot": packed.pack_weight(w1.to(dev), "e2m1", 128, smooth=s.to(dev), rotate_block=128, rotate_seed=42),
    }
    path = str(tmp_path / "m.safetensors")
    nbytes = packed.save_packed(path, layers)
    fp16_bytes = 2 * (2 * w1.numel() + 2 * w2.numel())
    assert nbytes < 0.45 * fp16_bytes
    got = packed.load_packed(path, dev)
    assert_bits_equal(got["a.fp4"].dequantize(), orc.per_group_kernel_sem(w1, "e2m1", 128).half(), "packed fp4")
    assert_bits_equal(got["b.fp6"].dequantize(), orc.per_token_kernel_sem(w2, "e2m3"), "packed fp6 per-channel")
    assert_bits_equal(got["c.fp6g"].dequantize(), orc.per_group_kernel_sem(w2, "e3m2", 128, out_dtype=torch.float16),
                      "packed fp6 per-group")
    qd = rot.block_random_hadamard_matrix(1920, 128, dev, 42)
    wt = rot.rotate_weight(rot.transform_weight(w1.to(dev), s.to(dev)), qd).cpu()
    assert_bits_equal(got["d.rot"].dequantize(), orc.per_group_kernel_sem(wt, "e2m1", 128).half(), "packed rotated")
    assert got["d.rot"].rotate_block == 128 and torch.equal(got["d.rot"].smooth.cpu(), s)
    # hardware operands
    x = torch.randn(256, 1920, generator=g).half().to(dev)
    ac, asc = gemm.quantize_mx(x)
    wc, wsc = got["a.fp4"].fp4_operands()
    direct = gemm.linear_fp4(ac, asc, *gemm.quantize_mx(w1.to(dev)), bias=got["a.fp4"].bias)
    assert_bits_equal(gemm.linear_fp4(ac, asc, wc, wsc, bias=got["a.fp4"].bias), direct, "fp4 operands from the file")


# ------------------------------------------------------------------ F4: GALT on the fused kernels
def test_galt_quantizers_and_objective(dev, golden):
    """The STE quantizers' forwards are bit-equal to the reference's classes on the reference's transformed
    operands; the objective and its gradient (matmuls now on the GPU) agree to GEMM-rounding tolerance."""
    from fpqvar_amd import galt
    x2, w2 = from_bits(golden["galt/fp4/x2_f32"]).to(dev), from_bits(golden["galt/fp4/w2_f32"]).to(dev)
    assert_bits_equal(galt.FPQuant(x2), from_bits(golden["galt/fp4/x2_quant"]), "FPQuant(x2)")
    assert_bits_equal(galt.FPQuant(w2), from_bits(golden["galt/fp4/w2_quant"]), "FPQuant(w2)")
    x2, w2 = from_bits(golden["galt/fp6/x2_f32"]).to(dev), from_bits(golden["galt/fp6/w2_f32"]).to(dev)
    assert_bits_equal(galt.FP6Quant_activation_per_token(x2), from_bits(golden["galt/fp6/x2_quant"]), "fp6 act")
    assert_bits_equal(galt.FP6Quant_weight(w2), from_bits(golden["galt/fp6/w2_quant"]), "fp6 weight")
    assert_bits_equal(galt.FP6Quant_activation(x2), from_bits(golden["galt/fp6/x2_quant_group"]), "fp6 act group")
    x, w, s, q = (from_bits(golden[f"galt/{k}_f32"]).to(dev) for k in ("x", "w", "s", "q"))
    for tag, tol in (("fp4", 2e-3), ("fp6", 2e-2)):
        sp = torch.nn.Parameter(s.clone())
        loss = galt.compute_quant_error(x, w, sp, q, tag)
        loss.backward()
        want_loss = float(from_bits(golden[f"galt/{tag}/loss"]))
        want_grad = from_bits(golden[f"galt/{tag}/grad_s"])
        assert abs(float(loss.detach()) - want_loss) <= tol * want_loss, (tag, float(loss.detach()), want_loss)
        cos = torch.nn.functional.cosine_similarity(sp.grad.cpu().float(), want_grad, dim=0)
        assert float(cos) > 0.99, (tag, float(cos))


def test_galt_loop_equals_torch_argmin_loop(dev, golden):
    """Same loop, same GPU, same matmuls: the fused argmin kernel vs the reference's distance-tensor lookup
    written in torch ops.  Forward values are bit-equal, so the whole AdamW trajectory is identical."""
    from fpqvar_amd import galt
    x, w, _, q = (from_bits(golden[f"galt/{k}_f32"]).to(dev) for k in ("x", "w", "s", "q"))
    grid = orc.TABLES["e2m1"].to(dev)

    def torch_fpquant(t):
        def fwd(v):
            shape = v.shape
            v = v.reshape(-1, 128)
            scale = v.abs().max(dim=-1, keepdim=True)[0] / grid.abs().max()
            v = v / scale
            idx = torch.argmin(torch.abs(v.unsqueeze(-1) - grid), dim=-1)
            return (grid[idx] * scale).view(shape)
        return galt._STE.apply(t, fwd)

    acts = [x[:32], x[32:]]
    la, lb = [], []
    sa = galt.learn_s(acts, w, q, epochs=4, fmt="fp4", log=la)
    sb = galt.learn_s(acts, w, q, epochs=4, fmt="fp4", log=lb, act_quant=torch_fpquant, weight_quant=torch_fpquant)
    assert la == lb
    assert_bits_equal(sa, sb, "learned s")
    assert la[-1] < la[0]


# ------------------------------------------------------------------ F1 -> F2: producers emitting GEMM operands
@pytest.mark.parametrize("x_dtype", (torch.float16, torch.float32))
def test_fused_producers_emit_fp4_operands(dev, x_dtype):
    """rotate_quant_mx / adaln_rotate_quant_mx: level(code) * scale is bit-equal to the value-emitting kernels,
    and the codes feed linear_fp4 with the same result as quantizing the rotated tensor separately."""
    from fpqvar_amd import gemm, rotation as rot
    g = torch.Generator().manual_seed(91)
    B, L, C = 3, 50, 1920
    x = (torch.randn(B, L, C, generator=g) * 2 + 0.3).to(x_dtype).to(dev)
    x[1, 7, 256:384] = 0                      # one all-zero group after LayerNorm? no - but one in the plain rotate case
    s = (torch.rand(C, generator=g) * 1.5 + 0.25).to(dev)
    scale = (torch.randn(B, 1, C, generator=g) * 0.3).half().to(dev)
    shift = (torch.randn(B, 1, C, generator=g) * 0.3).half().to(dev)
    # plain rotate
    x2 = x.reshape(B * L, C)
    want = rot.rotate_quant(x2, "e2m1", smooth=s)
    codes, scales = rot.rotate_quant_mx(x2, smooth=s)
    assert codes.shape == (B * L, C // 2) and scales.shape == (B * L, C // 128) and scales.dtype == torch.float16
    assert_bits_equal(gemm.dequantize_mx(codes, scales).half(), want, "rotate_quant_mx decode")
    # whole producer
    want, _, y = rot.adaln_rotate_quant(x, scale, shift, "e2m1", smooth=s, return_intermediates=True)
    codes, scales = rot.adaln_rotate_quant_mx(x, scale, shift, smooth=s)
    assert_bits_equal(gemm.dequantize_mx(codes, scales).half().view(B, L, C), want, "adaln_rotate_quant_mx decode")
    c2, s2 = gemm.quantize_mx(y.reshape(B * L, C))
    assert_bits_equal(gemm.dequantize_mx(c2, s2), gemm.dequantize_mx(codes, scales), "same operands as quantize_mx(rotated)")
    assert torch.equal(s2, scales)
    w = (torch.randn(256, C, generator=g) * 0.02).to(dev)
    wc, ws = gemm.quantize_mx(w)
    assert_bits_equal(gemm.linear_fp4(codes, scales, wc, ws), gemm.linear_fp4(c2, s2, wc, ws), "GEMM on the emitted codes")
    # a group that rotates to all zeros: scale 0, codes 0
    z = torch.zeros(4, 256, dtype=x_dtype, device=dev)
    cz, sz = rot.rotate_quant_mx(z)
    assert not cz.any() and not sz.any()


@pytest.mark.parametrize("rows,C", ((1, 128), (17, 384), (33, 2048), (257, 1152), (300, 2304), (64, 2560)))
@pytest.mark.parametrize("x_dtype", (torch.float16, torch.float32))
def test_fp4_operand_output_shapes(dev, rows, C, x_dtype):
    """The code-emitting forms over ragged tiles and rows of 1 .. 20 groups (17 .. 20: tile + one chunk per lane):
    level(code) * scale equals the value-emitting kernels bit for bit, scales equal the groups' scales."""
    from fpqvar_amd import gemm, rotation as rot
    g = torch.Generator().manual_seed(rows + C)
    x = (torch.randn(rows, C, generator=g) * torch.exp(0.4 * torch.randn(rows, C, generator=g))).to(x_dtype).to(dev)
    x[0, :128] = 0
    want = rot.rotate_quant(x, "e2m1")
    codes, scales = rot.rotate_quant_mx(x)
    assert codes.shape == (rows, C // 2) and scales.shape == (rows, C // 128)
    assert_bits_equal(gemm.dequantize_mx(codes, scales).half(), want, "rotate_quant_mx decode")
    B = 3 if rows % 3 == 0 else 1
    xa = x.view(B, rows // B, C)
    scale = (torch.randn(B, 1, C, generator=g) * 0.3).half().to(dev)
    shift = (torch.randn(B, 1, C, generator=g) * 0.3).half().to(dev)
    want = rot.adaln_rotate_quant(xa, scale, shift, "e2m1")
    codes, scales = rot.adaln_rotate_quant_mx(xa, scale, shift)
    assert_bits_equal(gemm.dequantize_mx(codes, scales).half().view_as(want), want, "adaln_rotate_quant_mx decode")


def test_quantize_var_real_fp4(dev):
    """quantize_VAR(..., real_fp4=True): fc1 / mat_qkv / proj become FP4Linear, fc2 keeps its dual-format fake quant;
    outputs agree with the default (fake-quant + fp16 GEMM) model to GEMM tolerance."""
    import copy
    from fpqvar_amd import gemm, quant_linear as ql

    class FFN(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.fc1, self.fc2 = torch.nn.Linear(256, 512), torch.nn.Linear(512, 256)

    class Attn(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.mat_qkv, self.proj = torch.nn.Linear(256, 768, bias=False), torch.nn.Linear(256, 256)

    class Toy(torch.nn.Module):
        def __init__(self):
            super().__init__()
            torch.manual_seed(3)
            self.ffn, self.attn = FFN(), Attn()

    cfg = dict(weight_quant="per_group", act_quant="per_group", w_bit=4, a_bit=4, act_quant_sym=True, activation_fp_quant=True,
               weight_fp_quant=True, act_fp_type="fp_e2", weight_fp_type="fp_e2", fc2_fp_type="fp_e1m2_neg_e2m1_pos")
    base = Toy().to(dev)
    fake = ql.quantize_VAR(copy.deepcopy(base), **cfg).half()
    real = ql.quantize_VAR(copy.deepcopy(base), real_fp4=True, **cfg)
    assert isinstance(real.ffn.fc1, gemm.FP4Linear) and isinstance(real.attn.mat_qkv, gemm.FP4Linear)
    assert isinstance(real.attn.proj, gemm.FP4Linear) and type(real.ffn.fc2).__name__ == "QuantizedLinear_fc2"
    x = torch.randn(70, 256, device=dev).half()
    for a, b in ((fake.ffn.fc1, real.ffn.fc1), (fake.attn.mat_qkv, real.attn.mat_qkv), (fake.attn.proj, real.attn.proj)):
        ya, yb = a(x).float(), b(x).float()
        assert float((ya - yb).abs().max()) <= 2e-2 * float(ya.abs().max()) + 1e-3
    with pytest.raises(ValueError):
        ql.quantize_VAR(copy.deepcopy(base), real_fp4=True, **{**cfg, "act_fp_type": "fp_e1"})
    # W6A6 per_channel / per_token -> FP8Linear
    cfg6 = dict(weight_quant="per_channel", act_quant="per_token", w_bit=6, a_bit=6, act_quant_sym=True, activation_fp_quant=True,
                weight_fp_quant=True, act_fp_type="fp6_e2m3", weight_fp_type="fp6_e2m3", fc2_fp_type="fp6_int_neg_e2m3_pos")
    fake6 = ql.quantize_VAR(copy.deepcopy(base), **cfg6).half()
    real6 = ql.quantize_VAR(copy.deepcopy(base), real_fp6=True, **cfg6)
    assert isinstance(real6.ffn.fc1, gemm.FP6Linear) and isinstance(real6.attn.proj, gemm.FP6Linear)
    assert type(real6.ffn.fc2).__name__ == "QuantizedLinear_fc2"
    for a, b in ((fake6.ffn.fc1, real6.ffn.fc1), (fake6.attn.mat_qkv, real6.attn.mat_qkv), (fake6.attn.proj, real6.attn.proj)):
        ya, yb = a(x).float(), b(x).float()
        assert float((ya - yb).abs().max()) <= 2e-2 * float(ya.abs().max()) + 1e-3
    with pytest.raises(ValueError):
        ql.quantize_VAR(copy.deepcopy(base), real_fp6=True, **cfg)


def test_quantize_var_mixed_datatype_variants(dev, qu):
    """The older variants' per-block mixed-format entry points (fq/quant_utils.py:1256-1431, rot/quant_utils.py:982-1066)
    as data: every layer gets the format pair the reference hard-codes for its block, ada_lin[1] is quantized too."""
    from fpqvar_amd import quant_linear as ql

    class Blk(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.ffn, self.attn = torch.nn.Module(), torch.nn.Module()
            self.ffn.fc1, self.ffn.fc2 = torch.nn.Linear(128, 256), torch.nn.Linear(256, 128)
            self.attn.mat_qkv, self.attn.proj = torch.nn.Linear(128, 384, bias=False), torch.nn.Linear(128, 128)
            self.ada_lin = torch.nn.Sequential(torch.nn.SiLU(), torch.nn.Linear(128, 768))

    class Toy(torch.nn.Module):
        def __init__(self):
            super().__init__()
            torch.manual_seed(9)
            self.blocks = torch.nn.ModuleList([Blk() for _ in range(30)])

    def act_name(mod):
        return mod.act_quant.func.__name__

    cfg4 = dict(weight_quant="per_group", act_quant="per_group", w_bit=4, a_bit=4, act_quant_sym=True, activation_fp_quant=True,
                weight_fp_quant=True, act_fp_type="fp_e2", weight_fp_type="fp_e2", fc2_fp_type="fp_e1m2_neg_e2m1_pos")
    base = Toy().to(dev)
    w_fc1_7 = base.blocks[7].ffn.fc1.weight.detach().clone()
    m = ql.quantize_VAR_mixed_fp4_datatype(Toy().to(dev), **cfg4)
    for b in range(30):
        assert act_name(m.blocks[b].ffn.fc1) == ("fp_quant_e2_per_group_cuda" if 6 <= b <= 20 else "fp_quant_e3_per_group_cuda")
        assert act_name(m.blocks[b].attn.mat_qkv) == ("fp_quant_e2_per_group_cuda" if b in (0, 24, 25) else "fp_quant_e3_per_group_cuda")
        assert act_name(m.blocks[b].attn.proj) == "fp_quant_e2_per_group_cuda"
        assert act_name(m.blocks[b].ffn.fc2) == "fp_quant_e1m2_neg_e2m1_pos_per_group_cuda"
        assert type(m.blocks[b].ada_lin[1]).__name__ == "QuantizedLinear"
    assert_bits_equal(m.blocks[7].ffn.fc1.weight, qu.fp_quant_e2_per_group_cuda(w_fc1_7, 4, 128), "mixed fp4 weight")
    m = ql.quantize_VAR_use_different_datatype(Toy().to(dev), **cfg4)
    assert act_name(m.blocks[0].attn.mat_qkv) == "fp_quant_e3_per_group_cuda" and act_name(m.blocks[24].attn.mat_qkv) == "fp_quant_e2_per_group_cuda"
    cfg6 = dict(weight_quant="per_channel", act_quant="per_token", w_bit=6, a_bit=6, act_quant_sym=True, activation_fp_quant=True,
                weight_fp_quant=True, act_fp_type="fp6_e2m3", weight_fp_type="fp6_e2m3", fc2_fp_type="fp6_int_neg_e2m3_pos")
    m = ql.quantize_VAR_mixed_fp6_datatype(Toy().to(dev), **cfg6)
    for b in range(30):
        assert act_name(m.blocks[b].ffn.fc1) == "fp6_quant_e3m2_per_token_cuda" and act_name(m.blocks[b].attn.mat_qkv) == "fp6_quant_e3m2_per_token_cuda"
        assert act_name(m.blocks[b].ffn.fc2) == ("fp6_quant_e2m3_per_token_cuda" if b in (0, 23) else "fp6_quant_e3m2_per_token_cuda")
        assert act_name(m.blocks[b].attn.proj) == ("fp6_quant_e2m3_per_token_cuda" if b >= 2 else "fp6_quant_e3m2_per_token_cuda")
    assert_bits_equal(m.blocks[7].ffn.fc1.weight, qu.fp6_quant_e2m3_per_token_cuda(w_fc1_7, 6), "mixed fp6 weight")
    x = torch.randn(5, 128, device=dev).half()
    assert m.blocks[3].ffn.fc1.half()(x).shape == (5, 256)


def test_fuzz_shapes_tables_alignments(dev):
    """Seeded random sweep over everything the dispatcher looks at - row length (sub-wave, one wave, one workgroup,
    multi-pass, ragged), row count, dtype pair, table, base-pointer misalignment - against the oracle."""
    from fpqvar_amd import ops
    rng = np.random.default_rng(2024)
    col_choices = [1, 2, 3, 7, 8, 16, 24, 31, 64, 100, 128, 136, 256, 384, 512, 520, 1000, 1024, 1920, 2048, 2304, 4104,
                   7680, 9216, 16384, 16392, 20000]
    sym, duals = ("e2m1", "e1m2", "e3m0", "e2m3", "e3m2"), (("e1m2_neg", "e2m1_pos"), ("int_neg", "e2m3_pos"), ("e2m1_neg", "e2m1_pos"))
    for case in range(160):
        cols = int(rng.choice(col_choices))
        rows = int(rng.integers(1, 40 if cols < 4096 else 6))
        dtype = torch.float16 if rng.random() < 0.6 else torch.float32
        off = int(rng.choice([0, 0, 1, 3, 8]))                                  # elements skipped at the front of the allocation
        g = torch.Generator().manual_seed(1000 + case)
        x = torch.randn(rows * cols + off, generator=g) * float(np.exp(rng.normal(0, 1.5)))
        x = x * torch.exp(0.7 * torch.randn(x.shape, generator=g))
        if rng.random() < 0.2:
            x[int(rng.integers(0, x.numel()))] = float(rng.choice([np.inf, -np.inf]))
        xh = x.to(dtype)
        xd = xh.to(dev)[off:].view(rows, cols)
        xc = xh[off:].view(rows, cols)
        if rng.random() < 0.55:
            name = str(rng.choice(sym))
            out_dtype = torch.float16 if (name in SYM6 or rng.random() < 0.3) else dtype
            got = ops.quant_rows(xd, name, cols, out_dtype)
            want = orc._rows_kernel_sem(xc, orc.TABLES[name]).to(out_dtype)
            assert_bits_equal(got, want, f"case {case}: {name} rows={rows} cols={cols} {dtype} -> {out_dtype} off={off}")
        else:
            neg, pos = duals[int(rng.integers(0, 3))]
            got = ops.quant_rows_dual(xd, neg, pos, cols, None)
            want = orc._dual_rows_kernel_sem(xc, neg, pos).to(dtype)
            assert_bits_equal(got, want, f"case {case}: {neg}+{pos} rows={rows} cols={cols} {dtype} off={off}")


# ------------------------------------------------------------------ F2 for W6A6: FP8-coded row-scaled GEMM
@pytest.mark.parametrize("table", ("e2m3", "e3m2", "e2m1"))
@pytest.mark.parametrize("dtype", (torch.float16, torch.float32))
def test_fp8_codes_reproduce_fake_quant(dev, qu, table, dtype):
    from fpqvar_amd import gemm, ops
    x = _inputs("heavy", (70, 1920), dtype, 61)
    x[3] = 0
    x[5, 7] = float("inf")
    xd = x.to(dev)
    codes, scales = gemm.quantize_fp8(xd, table)
    assert codes.shape == (70, 1920) and scales.shape == (70,) and scales.dtype == dtype
    want = ops.quant_rows(xd, table, 1920, torch.float32)                    # fp32 product q * s of the fake quantizer
    got = gemm.dequantize_fp8(codes, scales)
    ok = ~torch.isnan(want)
    assert_bits_equal(got[ok], want[ok], f"fp8 codes {table}")
    assert bool(torch.isnan(got[~ok]).all())
    for cols in (1000, 3):                                                   # ragged rows take the byte-store path
        xr = _inputs("gauss", (9, cols), dtype, 62).to(dev)
        c, s = gemm.quantize_fp8(xr, table)
        assert_bits_equal(gemm.dequantize_fp8(c, s), ops.quant_rows(xr, table, cols, torch.float32), f"ragged {cols}")


@pytest.mark.parametrize("T,O,K", ((256, 256, 1920), (1000, 5760, 1920), (130, 1928, 256), (64, 128, 7680), (20, 6912, 2304), (1, 8, 128)))
def test_fp8_gemm(dev, T, O, K):
    from fpqvar_amd import gemm
    g = torch.Generator().manual_seed(202 + T)
    x = (torch.randn(T, K, generator=g) * torch.exp(0.3 * torch.randn(T, K, generator=g))).half().to(dev)
    w = (torch.randn(O, K, generator=g) * 0.02).to(dev)
    bias = (torch.randn(O, generator=g) * 0.1).half().to(dev)
    ac, asc = gemm.quantize_fp8(x, "e2m3")
    wc, wsc = gemm.quantize_fp8(w, "e2m3")
    y = gemm.linear_fp8(ac, asc, wc, wsc, bias)
    assert y.shape == (T, O) and y.dtype == torch.float16
    a64, w64 = gemm.dequantize_fp8(ac, asc).double(), gemm.dequantize_fp8(wc, wsc).double()
    ref = a64 @ w64.t() + bias.double()
    err = (y.double() - ref).abs()
    tol = 2.0 ** -10 * ref.abs() + 1e-5 * (a64.abs() @ w64.abs().t()) + 1e-6     # fp16 output rounding + fp32 accumulation
    assert bool((err <= tol).all()), float((err / tol).max())
    # against the reference's formulation: fake-quantized fp16 tensors through an fp16 GEMM
    import fpqvar_amd.quant_utils as qu
    ref16 = torch.nn.functional.linear(qu.fp6_quant_e2m3_per_token_cuda(x, 6), qu.fp6_quant_e2m3_per_token_cuda(w, 6), bias)
    assert float((y.float() - ref16.float()).abs().max()) <= 2e-2 * float(ref16.float().abs().max()) + 1e-3


def test_fp8_linear_module(dev):
    from fpqvar_amd import gemm, quant_linear as ql
    torch.manual_seed(4)
    lin = torch.nn.Linear(1920, 640).to(dev)
    x = torch.randn(3, 50, 1920, device=dev).half()
    fp8 = gemm.FP8Linear.from_float(lin)
    cfg = dict(weight_quant="per_channel", act_quant="per_token", w_bit=6, a_bit=6, act_quant_sym=True, activation_fp_quant=True,
               weight_fp_quant=True, act_fp_type="fp6_e2m3", weight_fp_type="fp6_e2m3")
    fake = ql.QuantizedLinear.from_float(lin, **cfg).half()
    ya, yb = fake(x).float(), fp8(x).float()
    assert yb.shape == (3, 50, 640)
    assert float((ya - yb).abs().max()) <= 2e-2 * float(ya.abs().max()) + 1e-3
    # the stored weight decodes to the reference's quantized weight
    assert_bits_equal(gemm.dequantize_fp8(fp8.w_codes, fp8.w_scales).half(), fake.weight, "FP8Linear weight")


# ------------------------------------------------------------------ the same with 6-bit packed operands
@pytest.mark.parametrize("dtype", (torch.float16, torch.float32))
def test_fp6_codes_reproduce_fake_quant(dev, dtype):
    from fpqvar_amd import gemm, ops
    for rows, cols in ((70, 1920), (9, 2304), (5, 7680), (3, 12288), (4, 32)):
        x = _inputs("heavy", (rows, cols), dtype, 63 + cols)
        x[0] = 0
        xd = x.to(dev)
        codes, scales = gemm.quantize_fp6(xd)
        assert codes.shape == (rows, cols * 3 // 4) and scales.shape == (rows,) and scales.dtype == dtype
        assert_bits_equal(gemm.dequantize_fp6(codes, scales), ops.quant_rows(xd, "e2m3", cols, torch.float32), f"fp6 codes {cols}")
        c8, s8 = gemm.quantize_fp8(xd, "e2m3")
        assert_bits_equal(gemm.dequantize_fp6(codes, scales), gemm.dequantize_fp8(c8, s8), "fp6 vs fp8 codes")
    # an unaligned fp16 base pointer takes the generic kernel
    xu = _inputs("gauss", (6 * 256 + 8,), torch.float16, 64).to(dev)[8:].view(6, 256)
    c, s = gemm.quantize_fp6(xu[:, :])
    assert_bits_equal(gemm.dequantize_fp6(c, s), ops.quant_rows(xu, "e2m3", 256, torch.float32), "fp6 generic path")


@pytest.mark.parametrize("T,O,K", ((256, 256, 1920), (1000, 5760, 1920), (130, 1928, 256), (64, 128, 7680), (20, 6912, 2304), (1, 8, 128)))
def test_fp6_gemm(dev, T, O, K):
    from fpqvar_amd import gemm
    g = torch.Generator().manual_seed(302 + T)
    x = (torch.randn(T, K, generator=g) * torch.exp(0.3 * torch.randn(T, K, generator=g))).half().to(dev)
    w = (torch.randn(O, K, generator=g) * 0.02).to(dev)
    bias = (torch.randn(O, generator=g) * 0.1).half().to(dev)
    ac, asc = gemm.quantize_fp6(x)
    wc, wsc = gemm.quantize_fp6(w)
    y = gemm.linear_fp6(ac, asc, wc, wsc, bias)
    assert y.shape == (T, O) and y.dtype == torch.float16
    # same operands through the FP8-coded kernel: both accumulate exact products in fp32 in the same k order per
    # MFMA, only the K-step differs - equal to fp32-accumulation tolerance, and mostly bit-equal
    y8 = gemm.linear_fp8(*gemm.quantize_fp8(x, "e2m3"), *gemm.quantize_fp8(w, "e2m3"), bias)
    a64, w64 = gemm.dequantize_fp6(ac, asc).double(), gemm.dequantize_fp6(wc, wsc).double()
    ref = a64 @ w64.t() + bias.double()
    err = (y.double() - ref).abs()
    tol = 2.0 ** -10 * ref.abs() + 1e-5 * (a64.abs() @ w64.abs().t()) + 1e-6
    assert bool((err <= tol).all()), float((err / tol).max())
    assert float((y.float() - y8.float()).abs().max()) <= 2.0 ** -9 * float(y8.float().abs().max()) + 1e-4


def test_fp6_linear_module(dev):
    from fpqvar_amd import gemm, quant_linear as ql
    torch.manual_seed(5)
    lin = torch.nn.Linear(1920, 640).to(dev)
    x = torch.randn(3, 50, 1920, device=dev).half()
    fp6 = gemm.FP6Linear.from_float(lin)
    cfg = dict(weight_quant="per_channel", act_quant="per_token", w_bit=6, a_bit=6, act_quant_sym=True, activation_fp_quant=True,
               weight_fp_quant=True, act_fp_type="fp6_e2m3", weight_fp_type="fp6_e2m3")
    fake = ql.QuantizedLinear.from_float(lin, **cfg).half()
    ya, yb = fake(x).float(), fp6(x).float()
    assert float((ya - yb).abs().max()) <= 2e-2 * float(ya.abs().max()) + 1e-3
    assert_bits_equal(gemm.dequantize_fp6(fp6.w_codes, fp6.w_scales).half(), fake.weight, "FP6Linear weight")
    assert fp6.w_codes.numel() == 640 * 1920 * 3 // 4


@pytest.mark.parametrize("C", (1920, 2304))
@pytest.mark.parametrize("x_dtype", (torch.float16, torch.float32))
def test_adaln_rotate_quant_per_token(dev, C, x_dtype):
    """The fused producer of the W6A6 configuration: same LayerNorm / modulate / smooth / rotate stages as the
    per-group form (their rotated output is the bit-level reference here), then ONE scale per token row."""
    from fpqvar_amd import gemm, ops, rotation as rot
    g = torch.Generator().manual_seed(83)
    B, L = 3, 37
    x = (torch.randn(B, L, C, generator=g) * 2 + 0.3).to(x_dtype).to(dev)
    scale = (torch.randn(B, 1, C, generator=g) * 0.3).half().to(dev)
    shift = (torch.randn(B, 1, C, generator=g) * 0.3).half().to(dev)
    s = (torch.rand(C, generator=g) * 1.5 + 0.25).to(dev)
    _, _, y = rot.adaln_rotate_quant(x, scale, shift, "e2m1", smooth=s, return_intermediates=True)   # rotated fp16 rows
    for table in ("e2m3", "e3m2"):
        out = rot.adaln_rotate_quant_token(x, scale, shift, table, smooth=s)
        assert out.shape == (B, L, C) and out.dtype == torch.float16
        assert_bits_equal(out, ops.quant_rows(y, table, C, torch.float16), f"per-token quant of the rotated row, {table}")
        assert_bits_equal(out, orc.per_token_kernel_sem(y.cpu(), table), f"oracle, {table}")
        codes, scales = rot.adaln_rotate_quant_token(x, scale, shift, table, smooth=s, emit="fp8")
        c2, s2 = gemm.quantize_fp8(y.reshape(B * L, C), table)
        assert torch.equal(scales, s2) and torch.equal(codes, c2), f"fp8 operands, {table}"
        if table == "e2m3":
            codes6, scales6 = rot.adaln_rotate_quant_token(x, scale, shift, table, smooth=s, emit="fp6")
            c6, s6 = gemm.quantize_fp6(y.reshape(B * L, C))
            assert codes6.shape == (B * L, C * 3 // 4) and torch.equal(scales6, s6) and torch.equal(codes6, c6), "fp6 operands"
        else:
            with pytest.raises(RuntimeError):
                rot.adaln_rotate_quant_token(x, scale, shift, table, smooth=s, emit="fp6")
    with pytest.raises(RuntimeError):
        rot.adaln_rotate_quant_token(torch.zeros(1, 2, 4096, device=dev).half(), torch.zeros(1, 1, 4096, device=dev).half(),
                                     torch.zeros(1, 1, 4096, device=dev).half())


@pytest.mark.parametrize("C", (128, 640, 1024, 2048))
@pytest.mark.parametrize("x_dtype", (torch.float16, torch.float32))
def test_per_token_operand_output_widths(dev, C, x_dtype):
    """E4M3-byte and dense 6-bit operand output of the producer at rows of 1 .. 16 groups (one row = one tile; 5 groups:
    a row of 480 code bytes that is not a multiple of the 16-byte stores' reach within the tile image)."""
    from fpqvar_amd import gemm, rotation as rot
    g = torch.Generator().manual_seed(C + 1)
    B, L = 3, 19
    x = (torch.randn(B, L, C, generator=g) * 2 + 0.3).to(x_dtype).to(dev)
    scale = (torch.randn(B, 1, C, generator=g) * 0.3).half().to(dev)
    shift = (torch.randn(B, 1, C, generator=g) * 0.3).half().to(dev)
    _, _, y = rot.adaln_rotate_quant(x, scale, shift, "e2m1", return_intermediates=True)
    for table in ("e2m3", "e3m2"):
        codes, scales = rot.adaln_rotate_quant_token(x, scale, shift, table, emit="fp8")
        c2, s2 = gemm.quantize_fp8(y.reshape(B * L, C), table)
        assert torch.equal(scales, s2) and torch.equal(codes, c2), f"fp8 operands, {table}"
    codes6, scales6 = rot.adaln_rotate_quant_token(x, scale, shift, "e2m3", emit="fp6")
    c6, s6 = gemm.quantize_fp6(y.reshape(B * L, C))
    assert codes6.shape == (B * L, C * 3 // 4) and torch.equal(scales6, s6) and torch.equal(codes6, c6), "fp6 operands"


def test_rccl_single_rank_group_paths(dev):
    """The collectives the multi-GPU paths use (barrier, all_reduce MAX, all_gather), on a one-rank RCCL group: the
    8-GPU runs are the driver's, this only checks that the nccl backend initialises and the code paths execute here."""
    import socket
    import torch.distributed as dist
    from fpqvar_amd import calibrate as cal, generation as gen
    if dist.is_initialized():
        pytest.skip("a process group already exists in this process")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        dist.barrier()
        t = torch.tensor([1.5], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        assert float(t) == 1.5
        g = torch.Generator().manual_seed(7)
        weights = {f"b{i}.w": (torch.randn(256, 256, generator=g) * 0.02).to(dev) for i in range(3)}
        full = cal.calibrate_sharded(weights, gather=True, exchange="fp16")
        codes = cal.calibrate_sharded(weights, gather=True, exchange="codes")
        for n, w in weights.items():
            want = orc.per_group_kernel_sem(w.cpu(), "e2m1", 128).half()
            assert_bits_equal(full[n], want, f"calibrate_sharded fp16 {n}")
            assert_bits_equal(codes[n], want, f"calibrate_sharded codes {n}")
        total, rate = gen.aggregate_throughput(50, 2.0)
        assert total == 50 and abs(rate - 25.0) < 1e-9
        # the collective itself in the form ShardedCalibration uses at N > 1: in place, the input is this rank's slot
        # of the [world, width] output slab (RCCL must accept the aliasing; at one rank it is the identity)
        sc = cal.ShardedCalibration({n: tuple(w.shape) for n, w in weights.items()}, weights)
        sc.local.quantize()
        before = sc.slab.clone()
        dist.all_gather_into_tensor(sc.slab.view(-1), sc.slab[sc.rank])
        torch.cuda.synchronize()
        assert torch.equal(sc.slab, before)
        for n, w in weights.items():
            assert_bits_equal(sc.views()[n], orc.per_group_kernel_sem(w.cpu(), "e2m1", 128).half(), f"slab view {n}")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("kind", ("fp4", "fp6", "fp8"))
@pytest.mark.parametrize("B,L,O,K", ((4, 75, 1920, 1920), (3, 1, 264, 256), (2, 333, 520, 128), (1, 7, 8, 128)))
def test_gemm_fused_gate_residual_is_bitwise_the_two_torch_ops(dev, kind, B, L, O, K):
    """linear_*(…, gate, residual) == residual + linear_*(…).mul(gate) bit for bit (tr/basic_var.py:264: the AdaLN
    block's gated residual), for gate only, residual only, both, and the residual aliasing nothing / a view."""
    from fpqvar_amd import gemm
    g = torch.Generator().manual_seed(B * 1000 + L)
    x = torch.randn(B * L, K, generator=g).half().to(dev)
    w = (torch.randn(O, K, generator=g) * 0.05).to(dev)
    bias = torch.randn(O, generator=g).half().to(dev)
    gate = (torch.randn(B, 1, O, generator=g) * 0.5).half().to(dev)
    resid = torch.randn(B, L, O, generator=g).half().to(dev)
    quant, lin = {"fp4": (gemm.quantize_mx, gemm.linear_fp4), "fp6": (gemm.quantize_fp6, gemm.linear_fp6),
                  "fp8": (gemm.quantize_fp8, gemm.linear_fp8)}[kind]
    a, wq = quant(x), quant(w)
    y = lin(*a, *wq, bias)
    assert_bits_equal(lin(*a, *wq, bias, gate, resid).view(B, L, O), resid + y.view(B, L, O).mul(gate), f"{kind} gate+residual")
    assert_bits_equal(lin(*a, *wq, bias, gate, None).view(B, L, O), y.view(B, L, O).mul(gate), f"{kind} gate only")
    assert_bits_equal(lin(*a, *wq, bias, None, resid).view(B, L, O), resid + y.view(B, L, O), f"{kind} residual only")
    with pytest.raises(RuntimeError):
        lin(*a, *wq, bias, gate.float(), resid)
    with pytest.raises(RuntimeError):
        lin(*a, *wq, bias, gate, resid[:, :0])


@pytest.mark.parametrize("B,H,Lq,Lkv", ((2, 3, 1, 1), (2, 5, 4, 5), (1, 30, 36, 91), (3, 2, 100, 255), (2, 4, 169, 424),
                                        (1, 2, 256, 680), (1, 3, 130, 64), (2, 1, 33, 129), (1, 9, 7, 2240)))
def test_attention_over_the_cache_matches_sdpa(dev, B, H, Lq, Lkv):
    """fpq_attention_blhc vs an fp32 softmax(q k^T s) v reference on the same fp16 inputs, with q / k / v as the model
    has them (q a view of the qkv output, k / v views of the KV-cache slab) and contiguous.  Tolerance: P is rounded to
    fp16 before P V and the output to fp16, so |err| <= 2e-3 * max|v| is generous; typical 3e-4."""
    from fpqvar_amd import ops
    g = torch.Generator().manual_seed(Lq * 7 + Lkv)
    qkv = torch.randn(B, Lq, 3, H, 64, generator=g).half().to(dev)
    slab = torch.randn(2, B, Lkv + 5, H, 64, generator=g).half().to(dev)
    q = torch.nn.functional.normalize(qkv[:, :, 0].float(), dim=-1).mul(8.0).half()      # VAR: l2-normalised q * learned scale
    qkv[:, :, 0] = q
    q_view, k_view, v_view = qkv[:, :, 0], slab[0, :, :Lkv], slab[1, :, :Lkv]
    for scale in (1.0, 0.125):
        ref = torch.nn.functional.scaled_dot_product_attention(q_view.transpose(1, 2).float(), k_view.transpose(1, 2).float(),
                                                               v_view.transpose(1, 2).float(), scale=scale).transpose(1, 2)
        for qq, kk, vv in ((q_view, k_view, v_view), (q_view.contiguous(), k_view.contiguous(), v_view.contiguous())):
            out = ops.attention_blhc(qq, kk, vv, scale)
            assert out.shape == (B, Lq, H, 64) and out.dtype == torch.float16
            err = (out.float() - ref).abs().max().item()
            assert err <= 2e-3 * float(v_view.abs().max()), (scale, err)


def test_attention_argument_checks_and_graph_capture(dev):
    from fpqvar_amd import ops
    q = torch.randn(2, 5, 3, 64, device=dev).half()
    k = torch.randn(2, 9, 3, 64, device=dev).half()
    with pytest.raises(RuntimeError):
        ops.attention_blhc(q[..., :32], k[..., :32], k[..., :32], 1.0)          # head_dim 32
    with pytest.raises(RuntimeError):
        ops.attention_blhc(q, k[:, :0], k[:, :0], 1.0)                          # no keys
    with pytest.raises(RuntimeError):
        ops.attention_blhc(q.float(), k.float(), k.float(), 1.0)
    with pytest.raises(RuntimeError):
        ops.attention_blhc(q, k, k, 0.0)                                        # FPQ_ERR_ARG from the C ABI: scale must be > 0
    assert ops.attention_blhc(q[:, :0], k, k, 1.0).shape == (2, 0, 3, 64)
    # stream-ordered and capturable: replaying a hipGraph on new inputs reproduces the eager result
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        ops.attention_blhc(q, k, k, 0.5)
    torch.cuda.current_stream().wait_stream(s)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        y = ops.attention_blhc(q, k, k, 0.5)
    q.copy_(torch.randn_like(q))
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(y, ops.attention_blhc(q, k, k, 0.5))


@pytest.mark.parametrize("B,L,C", ((100, 16, 1920), (3, 1, 8), (2, 333, 2304), (1, 5, 64)))
def test_gate_residual_is_bitwise_the_two_torch_ops(dev, B, L, C):
    from fpqvar_amd import ops
    g = torch.Generator().manual_seed(B + L + C)
    y = torch.randn(B, L, C, generator=g).half().to(dev)
    gate = (torch.randn(B, 1, C, generator=g) * 0.5).half().to(dev)
    x = (torch.randn(B, L, C, generator=g) * 3).half().to(dev)
    assert_bits_equal(ops.gate_residual(y, gate, x), x + y.mul(gate), "gate_residual")
    with pytest.raises(RuntimeError):
        ops.gate_residual(y, gate.float(), x)
    with pytest.raises(RuntimeError):
        ops.gate_residual(y, gate, x[:, :0])


def test_attention_fuzz_shapes(dev):
    """Random small (B, H, Lq, Lkv): every remainder of Lq mod 32 / 128 and Lkv mod 64 gets exercised over the seeds."""
    import random
    from fpqvar_amd import ops
    rnd = random.Random(7)
    g = torch.Generator().manual_seed(7)
    for _ in range(40):
        B, H = rnd.randint(1, 3), rnd.randint(1, 9)
        Lq, Lkv = rnd.choice((1, 2, 31, 32, 33, 63, 64, 65, 127, 128, 129, 200, rnd.randint(1, 300))), rnd.randint(1, 400)
        q = torch.randn(B, Lq, H, 64, generator=g).half().to(dev)
        k = torch.randn(B, Lkv, H, 64, generator=g).half().to(dev)
        v = torch.randn(B, Lkv, H, 64, generator=g).half().to(dev)
        ref = torch.nn.functional.scaled_dot_product_attention(q.transpose(1, 2).float(), k.transpose(1, 2).float(),
                                                               v.transpose(1, 2).float(), scale=0.125).transpose(1, 2)
        out = ops.attention_blhc(q, k, v, 0.125)
        err = (out.float() - ref).abs().max().item()
        assert err <= 2e-3 * float(v.abs().max()), (B, H, Lq, Lkv, err)
