"""Re-quantizing an already fake-quantized row is the identity - proven by exhaustion over every
possible fp16 row maximum and every level, for the two KV-cache modes of the reference
(tr/basic_var.py:192-200: kv_bit 6 = E2M3 per 64-channel row, kv_bit 4 = E2M1 per 128-group) and
the other symmetric tables.  This is what makes the incremental KV path (quantize an entry once,
when it has just entered the cache) bit-identical to the reference's "re-quantize the whole cache
every step".  CPU only (oracle arithmetic = torch fp16 semantics)."""
import pytest
import torch

from oracle import fpq_oracle as orc


@pytest.mark.parametrize("name", ("e2m3", "e2m1", "e1m2", "e3m0", "e3m2"))
def test_requantization_is_identity_exhaustive(name):
    tab = orc.TABLES[name]
    g = tab.abs().max()
    levels = torch.unique(tab.abs())                      # non-negative levels, ascending
    a = torch.arange(1, 0x7C00, dtype=torch.int32).to(torch.int16).view(torch.float16)   # every finite amax > 0
    s = a / g                                             # fp16: scale of the first pass
    # the claim is for NORMAL fp16 scales (row maximum >= g * 2^-14, i.e. ~4e-4): a subnormal scale has
    # too few significant bits for the round trip (e.g. amax = 4 ulp -> scale 1 ulp -> xn = 4, not 6)
    ok = (s >= 2.0 ** -14) & torch.isfinite(s)
    a, s = a[ok], s[ok]
    # first pass output for an element sitting on level L (any element's output is half(L*s) for some L)
    v = (levels[None, :] * s[:, None]).to(torch.float16)  # fp32 product (exact) rounded once, [n_a, n_levels]
    finite = torch.isfinite(v).all(dim=1)
    a, s, v = a[finite], s[finite], v[finite]
    # the row maximum of the first pass's output: the element that carried amax normalises to ~g -> top level
    xn_max = (a / s).to(torch.float32)
    assert torch.equal(orc.nearest_kernel(xn_max, tab), torch.full_like(xn_max, float(g)))
    amax2 = v[:, -1]                                      # = half(g * s)
    s2 = amax2 / g
    assert torch.equal(s2.view(torch.int16), s.view(torch.int16)), "second-pass scale differs from the first"
    # every level maps back to itself under the second pass
    xn = (v / s2[:, None]).to(torch.float32)
    q = orc.nearest_kernel(xn.reshape(-1), tab).view(xn.shape)
    assert torch.equal(q, levels[None, :].expand_as(q)), "a level moved under re-quantization"
    out2 = (q * s2[:, None]).to(torch.float16)
    assert torch.equal(out2.view(torch.int16), v.view(torch.int16))
    # negative side: the table is symmetric except for tie direction; exact levels are never ties
    qn = orc.nearest_kernel((-xn).reshape(-1), tab).view(xn.shape)
    assert torch.equal(qn, torch.where(levels == 0, levels, -levels)[None, :].expand_as(qn))


def test_requantization_identity_on_random_rows():
    gen = torch.Generator().manual_seed(4)
    x = (torch.randn(4096, 64, generator=gen) * torch.exp(torch.randn(4096, 1, generator=gen) * 2)).half()
    x = x[x.abs().max(dim=1).values >= 1e-2]              # rows with a normal fp16 scale (see above)
    x = x[: (x.shape[0] // 2) * 2]
    q1 = orc.per_token_kernel_sem(x, "e2m3")
    assert torch.equal(orc.per_token_kernel_sem(q1, "e2m3").view(torch.int16), q1.view(torch.int16))
    y = x.reshape(-1, 128)
    q1 = orc.per_group_kernel_sem(y, "e2m1", 128)
    assert torch.equal(orc.per_group_kernel_sem(q1, "e2m1", 128).view(torch.int16), q1.view(torch.int16))
