"""The plain-C restatement against the torch restatement and the golden vectors (CPU only).
Two independently written oracles agreeing bit for bit is the cross-check; the C one is
also what bench.py times as the scalar CPU port."""
import pytest
import torch

from oracle import c_oracle as co
from oracle import fpq_oracle as orc
from tests.conftest import assert_bits_equal, from_bits

KINDS = ("gauss", "heavy", "edge", "weights", "gelu", "inf", "nan")


def test_half_conversions_exhaustive():
    allh = torch.arange(0, 65536, dtype=torch.int32).to(torch.int16).view(torch.float16)
    assert_bits_equal(co.h2f(allh), allh.to(torch.float32), "h2f")
    # every fp16 value, its fp32 neighbours and the rounding midpoints between fp16 values
    f = allh.to(torch.float32)
    f = f[torch.isfinite(f)]
    pts = [f]
    for k in (-2, -1, 1, 2):
        pts.append((f.view(torch.int32) + k).view(torch.float32))
    srt = torch.sort(f[f >= 0]).values
    mid = ((srt[:-1].double() + srt[1:].double()) / 2).float()
    pts += [mid, -mid, (mid.view(torch.int32) + 1).view(torch.float32), (mid.view(torch.int32) - 1).view(torch.float32)]
    g = torch.Generator().manual_seed(2)
    pts.append(torch.randn(200000, generator=g) * 100)
    pts.append(torch.tensor([65504.0, 65519.9, 65520.0, 1e9, float("inf"), -float("inf"), 2.0 ** -25, 2.0 ** -24]))
    x = torch.cat(pts)
    x = x[~torch.isnan(x)]
    assert_bits_equal(co.f2h(x), x.to(torch.float16), "f2h")


@pytest.mark.parametrize("name", list(orc.TABLES))
def test_scan(name):
    tab = orc.TABLES[name]
    allh = torch.arange(0, 65536, dtype=torch.int32).to(torch.int16).view(torch.float16).to(torch.float32)
    assert_bits_equal(co.nearest(allh, tab), orc.nearest_kernel(allh, tab), name)


@pytest.mark.parametrize("dn", ("f16", "f32"))
@pytest.mark.parametrize("kind", KINDS)
def test_rows_vs_golden(golden, kind, dn):
    x = from_bits(golden[f"in/{kind}_{dn}"])
    for name in ("e2m1", "e1m2", "e3m0"):
        want = from_bits(golden[f"out/per_group_cuda/{name}/{kind}_{dn}"])
        assert_bits_equal(co.rows(x, orc.TABLES[name], 128), want, f"{name} group")
    for name in ("e2m3", "e3m2"):
        want = from_bits(golden[f"out/per_group_cuda/{name}/{kind}_{dn}"])
        assert_bits_equal(co.rows(x, orc.TABLES[name], 128, out_f16=True), want, f"{name} group")
        want = from_bits(golden[f"out/per_token_cuda/{name}/{kind}_{dn}"])
        assert_bits_equal(co.rows(x, orc.TABLES[name], x.shape[-1], out_f16=True), want, f"{name} token")
    want = from_bits(golden[f"out/dual_group_cuda/int_neg+e2m3_pos/{kind}_{dn}"])
    assert_bits_equal(co.rows_dual(x, orc.TABLES["int_neg"], orc.TABLES["e2m3_pos"], 128), want, "dual fp6")
    if kind != "nan":   # the global clamp quirk (NaN anywhere -> zeros) is outside the row functions
        want = from_bits(golden[f"out/dual_group_cuda/e1m2_neg+e2m1_pos/{kind}_{dn}"])
        assert_bits_equal(co.rows_dual(x, orc.TABLES["e1m2_neg"], orc.TABLES["e2m1_pos"], 128), want, "dual fp4")


def test_rows_vs_python_oracle_random():
    g = torch.Generator().manual_seed(9)
    x = (torch.randn(64, 1920, generator=g) * torch.exp(torch.randn(64, 1920, generator=g))).half()
    for name in ("e2m1", "e2m3"):
        assert_bits_equal(co.rows(x, orc.TABLES[name], 128),
                          orc.per_group_kernel_sem(x, name, 128, out_dtype=torch.float16), name)
    w = torch.randn(32, 1024, generator=g) * 0.02
    assert_bits_equal(co.rows(w, orc.TABLES["e2m1"], 128), orc.per_group_kernel_sem(w, "e2m1", 128), "weights")
