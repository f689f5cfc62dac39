"""The oracle against the vectors produced by the reference's own Python
(tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest
import torch

from oracle import fpq_oracle as orc
from tests.conftest import assert_bits_equal, from_bits

KINDS = ("gauss", "heavy", "edge", "weights", "gelu", "inf", "nan")
DTYPES = ("f16", "f32")


def _in(golden, kind, dn):
    return from_bits(golden[f"in/{kind}_{dn}"])


def test_tables_match_reference_literals(golden):
    for name, tab in orc.TABLES.items():
        ref = torch.from_numpy(golden[f"table/{name}"])
        assert_bits_equal(tab, ref, f"table {name}")


def test_kernel_kats():
    # SURVEY.md section 8c golden vectors (1): kernel semantics vs argmin semantics
    x = torch.tensor([0.25, -0.25, 0.75, 1.25, 1.75, 2.5, 3.5, 5, -5, -2.5, float("nan"), float("inf"), 7])
    want_kernel = torch.tensor([0.5, 0, 1, 1.5, 2, 3, 4, 6, -4, -2, 0, 0, 6])
    assert_bits_equal(orc.nearest_kernel(x, orc.TABLES["e2m1"]), want_kernel, "kernel KAT")
    want_argmin = torch.tensor([0, -0.5, 0.5, 1, 1.5, 2, 3, 4, -6, -3, -6, -6, 6])
    got = orc.nearest_argmin(x, orc.TABLES["e2m1"])
    assert torch.equal(got[:10], want_argmin[:10]) and got[12] == 6
    # out of reach of every entry -> 0.0 ; float64 input compared in float32
    far = torch.tensor([102407.0, -102407.0, 102406.0, 1e30, -float("inf")])
    assert_bits_equal(orc.nearest_kernel(far, orc.TABLES["e2m1"]), torch.tensor([0.0, 0.0, 6.0, 0.0, 0.0]), "far")
    x64 = torch.tensor([0.25 - 1e-12, 0.25], dtype=torch.float64)
    assert_bits_equal(orc.nearest_kernel(x64, orc.TABLES["e2m1"]),
                      torch.tensor([0.5, 0.5], dtype=torch.float64), "f64")
    # duplicate zeros of the FP6 tables: last index wins, value is +0.0
    idx = orc.nearest_kernel_index(torch.tensor([0.0, -0.0, 0.01]), orc.TABLES["e2m3"])
    assert idx.tolist() == [32, 32, 32]


@pytest.mark.parametrize("name", list(orc.TABLES))
def test_closed_form_equals_kernel_loop(name):
    tab = orc.TABLES[name]
    # every fp16 value (finite, inf, nan) widened to fp32
    allh = torch.arange(0, 65536, dtype=torch.int32).to(torch.int16).view(torch.float16).to(torch.float32)
    assert_bits_equal(orc.nearest_closed_form(allh, tab), orc.nearest_kernel(allh, tab), f"{name} fp16 sweep")
    # +-8 ulp fp32 neighbourhoods of every entry and midpoint
    uniq = torch.unique(tab)
    pts = torch.cat([uniq, (uniq[:-1] + uniq[1:]) / 2])
    nb = []
    for k in range(-8, 9):
        nb.append((pts.view(torch.int32) + k).view(torch.float32))
        nb.append((-((-pts).view(torch.int32) + k).view(torch.float32)))
    nb = torch.cat(nb)
    nb = nb[torch.isfinite(nb)]
    assert_bits_equal(orc.nearest_closed_form(nb, tab), orc.nearest_kernel(nb, tab), f"{name} fp32 nbhd")
    g = torch.Generator().manual_seed(1)
    r = (torch.rand(200000, generator=g) * 2 - 1) * float(tab.abs().max()) * 1.2
    assert_bits_equal(orc.nearest_closed_form(r, tab), orc.nearest_kernel(r, tab), f"{name} random")


@pytest.mark.parametrize("dn", DTYPES)
@pytest.mark.parametrize("kind", KINDS)
def test_per_group_kernel_sem(golden, kind, dn):
    x = _in(golden, kind, dn)
    for name in ("e2m1", "e1m2", "e3m0"):
        want = from_bits(golden[f"out/per_group_cuda/{name}/{kind}_{dn}"])
        assert_bits_equal(orc.per_group_kernel_sem(x, name, 128), want, f"{name} {kind} {dn}")
    for name in ("e2m3", "e3m2"):
        want = from_bits(golden[f"out/per_group_cuda/{name}/{kind}_{dn}"])
        assert want.dtype == torch.float16
        assert_bits_equal(orc.per_group_kernel_sem(x, name, 128, out_dtype=torch.float16), want,
                          f"{name} {kind} {dn}")


@pytest.mark.parametrize("dn", DTYPES)
@pytest.mark.parametrize("kind", KINDS)
def test_per_token_kernel_sem(golden, kind, dn):
    x = _in(golden, kind, dn)
    for name in ("e2m3", "e3m2"):
        want = from_bits(golden[f"out/per_token_cuda/{name}/{kind}_{dn}"])
        assert_bits_equal(orc.per_token_kernel_sem(x, name), want, f"{name} {kind} {dn}")
    want = from_bits(golden[f"out/kv/e2m3_token64/{kind}_{dn}"])
    assert_bits_equal(orc.per_token_kernel_sem(x.reshape(2, 4, 4, 64), "e2m3"), want, "kv token64")
    want = from_bits(golden[f"out/kv/e2m1_group/{kind}_{dn}"])
    assert_bits_equal(orc.per_group_kernel_sem(x, "e2m1", 128), want, "kv group")


@pytest.mark.parametrize("dn", DTYPES)
@pytest.mark.parametrize("kind", KINDS)
def test_dual_kernel_sem(golden, kind, dn):
    x = _in(golden, kind, dn)
    want = from_bits(golden[f"out/dual_group_cuda/e1m2_neg+e2m1_pos/{kind}_{dn}"])
    assert_bits_equal(orc.dual_per_group_kernel_sem(x, "e1m2_neg", "e2m1_pos", 128, 1.0), want, "dual fp4")
    want = from_bits(golden[f"out/dual_group_cuda_clip0.9/e1m2_neg+e2m1_pos/{kind}_{dn}"])
    assert_bits_equal(orc.dual_per_group_kernel_sem(x, "e1m2_neg", "e2m1_pos", 128, 0.9), want, "dual fp4 clip")
    want = from_bits(golden[f"out/dual_group_cuda/int_neg+e2m3_pos/{kind}_{dn}"])
    assert_bits_equal(orc.dual_per_group_kernel_sem(x, "int_neg", "e2m3_pos", 128, None), want, "dual fp6 group")
    want = from_bits(golden[f"out/dual_group_cuda/e2m1_neg+e2m1_pos/{kind}_{dn}"])
    assert_bits_equal(orc.dual_per_group_kernel_sem(x, "e2m1_neg", "e2m1_pos", 128, 1.0), want, "afpq")
    want = from_bits(golden[f"out/dual_token_cuda/int_neg+e2m3_pos/{kind}_{dn}"])
    assert_bits_equal(orc.dual_per_token_kernel_sem(x, "int_neg", "e2m3_pos"), want, "dual fp6 token")


@pytest.mark.parametrize("dn", DTYPES)
@pytest.mark.parametrize("kind", KINDS)
def test_neg_reverse_kernel_sem(golden, kind, dn):
    x = _in(golden, kind, dn)
    want = from_bits(golden[f"out/neg_reverse_group_cuda/e2m1/{kind}_{dn}"])
    assert_bits_equal(orc.neg_reverse_per_group_kernel_sem(x, "e2m1", 128), want, "neg reverse")


@pytest.mark.parametrize("dn", DTYPES)
@pytest.mark.parametrize("kind", KINDS)
def test_argmin_cpu_path(golden, kind, dn):
    x = _in(golden, kind, dn)
    for name, clamp3 in (("e2m1", False), ("e1m2", True), ("e3m0", True)):
        want = from_bits(golden[f"out/per_group_argmin/{name}/{kind}_{dn}"])
        assert_bits_equal(orc.per_group_argmin_sem(x, name, 128, clamp3), want, f"argmin group {name}")
        want = from_bits(golden[f"out/per_token_argmin/{name}/{kind}_{dn}"])
        assert_bits_equal(orc.per_token_argmin_sem(x, name), want, f"argmin token {name}")
    want = from_bits(golden[f"out/dual_group_argmin/e1m2_neg+e2m1_pos/{kind}_{dn}"])
    assert_bits_equal(orc.dual_per_group_argmin_sem(x), want, "argmin dual")


def test_per_tensor_config1(golden):
    x = from_bits(golden["in/per_tensor_f32"])
    want = from_bits(golden["out/per_tensor_argmin/e2m1"])
    assert_bits_equal(orc.per_tensor_argmin_sem(x, "e2m1"), want, "config 1")


def test_rotation_pieces(golden):
    q = torch.from_numpy(golden["rot/q128_f64"])
    assert torch.equal(orc.hadamard_block(128, 42), q)
    d = orc.sign_vector(128, 42)
    assert d[:16].tolist() == [-1, 1, -1, -1, -1, 1, -1, -1, -1, 1, -1, -1, -1, -1, 1, -1]
    assert d.sum().item() == 20
    assert bool(golden["rot/blocks_identical_1920"]) and bool(golden["rot/block0_equals_q128"])
    assert bool(golden["rot/offdiag_zero_1920"])
    # |Q| constants
    assert abs(q[0, 0].abs().item() - 0.0883883491610206) < 1e-16
    assert q.abs().to(torch.float32)[0, 0].item() == np.float32(0.0883883461356163)
    assert q.abs().to(torch.float16)[0, 0].item() == 0.08837890625


def test_codes_roundtrip():
    g = torch.Generator().manual_seed(3)
    x = torch.randn(4, 256, generator=g).half()
    for name in ("e2m1", "e1m2", "e3m0", "e2m3", "e3m2"):
        codes, scale = orc.per_group_codes(x, name, 128)
        uniq = torch.unique(orc.TABLES[name])
        deq = (uniq[codes.long()].view(-1, 128) * scale.view(-1, 1)).view(x.shape).to(torch.float16)
        assert_bits_equal(deq, orc.per_group_kernel_sem(x, name, 128, out_dtype=torch.float16), name)
