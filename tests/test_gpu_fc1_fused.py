"""The fc1 tail fused into the FP4 GEMM (fpq_gemm_fp4_gelu_dual): GELU(tanh) and fc2's dual E1M2-/E2M1+
per-group quantizer as the GEMM's epilogue (tr/basic_var.py:120-121, tr/quant_utils.py:415-452,991).
Contract, as for the fused producers: the QUANTIZATION is bit-exact on the GELU values the kernel
emits on request - against the oracle's dual quantizer and against the stand-alone HIP quantizer; the GELU values sit
within one fp16 ulp of torch's F.gelu(y, approximate="tanh") of the Linear output y, on EVERY fp16 input (the domain is
finite: all 65536 patterns are pushed through the epilogue); y itself is fpq_gemm_fp4_mx's output bit for bit."""
import pytest
import torch
import torch.nn.functional as Fn

from fpqvar_amd import _lib
from oracle import fpq_oracle as orc
from tests.conftest import assert_bits_equal

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need the GPU"
    return torch.device("cuda:0")


def ordered(h: torch.Tensor) -> torch.Tensor:
    """fp16 bit patterns as integers that grow with the value (-0 and +0 coincide): ulp distances are differences."""
    b = h.view(torch.int16).to(torch.int32) & 0xFFFF
    return torch.where(b >= 0x8000, 0x8000 - b, b)


def ulp_diff(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """|a - b| in fp16 ulps; NaN against NaN counts 0, NaN against a number 2^16."""
    na, nb = torch.isnan(a), torch.isnan(b)
    d = (ordered(a) - ordered(b)).abs()
    d = torch.where(na & nb, torch.zeros_like(d), d)
    return torch.where(na ^ nb, torch.full_like(d, 1 << 16), d)


def operands(dev, T, K, O, seed):
    from fpqvar_amd import gemm
    g = torch.Generator().manual_seed(seed)
    x = (torch.randn(T, K, generator=g) * torch.exp(0.3 * torch.randn(T, K, generator=g))).half().to(dev)
    w = (torch.randn(O, K, generator=g) * 0.05).to(dev)
    bias = (torch.randn(O, generator=g) * 0.3).half().to(dev)
    return gemm.quantize_mx(x), gemm.quantize_mx(w), bias


@pytest.mark.parametrize("T,K,O", [(300, 256, 384), (1, 128, 128), (1000, 1920, 512), (70, 2304, 1152)])
def test_fused_fc1_equals_gemm_gelu_quantizer(dev, T, K, O, lib_options):
    """Every LDS-DMA tiling (64 / 128 / 256 rows x 128 outputs), ragged token counts: the emitted GELU values against torch's
    GELU of the plain GEMM's output (<= 1 fp16 ulp), the quantized result against the stand-alone HIP quantizer on the emitted
    values (bit-exact) and against the oracle's (bit-exact), and the one-tensor form against the two-tensor form."""
    from fpqvar_amd import gemm, ops
    a, w, bias = operands(dev, T, K, O, 11)
    for cfg in (None, 30, 20, 10):
        lib_options("FPQ_GEMM_CFG", cfg)
        for b in (bias, None):
            y = gemm.linear_fp4(*a, *w, b)
            q, h = gemm.linear_fp4_gelu_dual(*a, *w, b, return_gelu=True)
            ref_h = Fn.gelu(y, approximate="tanh")
            d = ulp_diff(h, ref_h)
            assert int(d.max()) <= 1, (cfg, T, K, O, int(d.max()), int((d > 0).sum()))
            assert float((d > 0).float().mean()) < 0.01, "GELU: more than 1 % of the values differ from torch's"
            assert_bits_equal(q, ops.quant_rows_dual(h, "e1m2_neg", "e2m1_pos", 128, 1.0), f"cfg {cfg}: fused vs stand-alone quantizer on the emitted GELU values")
            assert_bits_equal(q.cpu(), orc.dual_per_group_kernel_sem(h.cpu(), "e1m2_neg", "e2m1_pos", 128, 1.0), f"cfg {cfg}: fused vs oracle")
            assert_bits_equal(gemm.linear_fp4_gelu_dual(*a, *w, b), q, f"cfg {cfg}: without the GELU output")
    # float32 weight scales (the reference quantizes weights in fp32) give the same result as their fp16 rounding would not: just run
    wc, ws = w
    q32 = gemm.linear_fp4_gelu_dual(*a, wc, ws.float() if ws.dtype == torch.float16 else ws.half().float(), bias)
    assert q32.shape == (T, O) and q32.dtype == torch.float16


def test_fused_fc1_gelu_on_every_fp16_input(dev):
    """All 65536 fp16 patterns as the Linear output: zero activations (every product is 0) and the pattern as the bias of
    its own output column.  The GELU the epilogue computes against torch's on this GPU: never more than one fp16 ulp apart,
    NaN exactly where torch has NaN (NaN inputs and -inf: 0.5 * -inf * 0)."""
    from fpqvar_amd import gemm
    K, O = 128, 65536
    every = torch.arange(0, 65536, dtype=torch.int32).to(torch.int16).view(torch.float16).to(dev)
    a = (torch.zeros(4, K // 2, dtype=torch.uint8, device=dev), torch.ones(4, 1, dtype=torch.float16, device=dev))
    w = (torch.zeros(O, K // 2, dtype=torch.uint8, device=dev), torch.ones(O, 1, dtype=torch.float32, device=dev))
    y = gemm.linear_fp4(*a, *w, every)
    assert_bits_equal(y[0], torch.where(every == 0, torch.zeros_like(every), every), "the Linear output is the bias (0 + -0 = +0)")
    q, h = gemm.linear_fp4_gelu_dual(*a, *w, every, return_gelu=True)
    ref = Fn.gelu(y, approximate="tanh")
    d = ulp_diff(h, ref)
    worst = int(d.max())
    n_diff = int((d[0] > 0).sum())
    assert worst <= 1, (worst, every[d[0] > 1][:8].tolist(), h[0][d[0] > 1][:8].tolist(), ref[0][d[0] > 1][:8].tolist())
    assert n_diff <= 64, f"{n_diff} of 65536 inputs differ from torch's GELU by one ulp (expected: a handful at fp16 rounding boundaries)"
    assert torch.equal(torch.isnan(h), torch.isnan(ref))
    # a NaN anywhere in h: the reference's global clamp makes the whole result zero (tr/quant_utils.py:421-422)
    assert bool(torch.isnan(h).any()) and not bool(q.any()), "NaN rule: every output must be +0"
    assert q.view(torch.int16).abs().max().item() == 0


def test_fused_fc1_nan_rule_and_scratch(dev):
    """One NaN in the GELU tensor (a NaN bias): every output zero, the 8-byte scratch zero again afterwards - eager and as
    a replayed hipGraph; without a NaN the same buffers give the ordinary result again."""
    from fpqvar_amd import gemm, ops
    a, w, bias = operands(dev, 200, 256, 256, 5)
    clean = gemm.linear_fp4_gelu_dual(*a, *w, bias)
    assert bool(clean.any())
    bad = bias.clone()
    bad[77] = float("nan")
    z = gemm.linear_fp4_gelu_dual(*a, *w, bad)
    assert not bool(z.view(torch.int16).any())
    scratch = ops._nan_scratch(dev)
    torch.cuda.synchronize()
    assert not bool(scratch.any()), "the NaN scratch must be zero again after the fix-up launch"
    assert_bits_equal(gemm.linear_fp4_gelu_dual(*a, *w, bias), clean, "after a NaN call")
    # graph replay: the NaN case twice, then the clean case, on static buffers
    sb = bad.clone()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        gemm.linear_fp4_gelu_dual(*a, *w, sb)
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=s):
            out = gemm.linear_fp4_gelu_dual(*a, *w, sb)
    torch.cuda.current_stream().wait_stream(s)
    for _ in range(2):
        gr.replay()
        torch.cuda.synchronize()
        assert not bool(out.view(torch.int16).any())
    sb.copy_(bias)
    gr.replay()
    torch.cuda.synchronize()
    assert_bits_equal(out, clean, "graph replay without the NaN")


def test_fused_fc1_at_the_model_shape(dev):
    """VAR-d30's fc1 at a late scale step (16900 tokens x 1920 -> 7680): the default tiling of that size, checked through the
    stand-alone quantizer on the emitted values, oracle-checked row slices from both ends, and torch's GELU."""
    from fpqvar_amd import gemm, ops
    a, w, bias = operands(dev, 16900, 1920, 7680, 3)
    q, h = gemm.linear_fp4_gelu_dual(*a, *w, bias, return_gelu=True)
    y = gemm.linear_fp4(*a, *w, bias)
    assert int(ulp_diff(h, Fn.gelu(y, approximate="tanh")).max()) <= 1
    assert_bits_equal(q, ops.quant_rows_dual(h, "e1m2_neg", "e2m1_pos", 128, 1.0), "fused vs stand-alone quantizer")
    for lo in (0, 16900 - 16):
        assert_bits_equal(q[lo:lo + 16].cpu(), orc.dual_per_group_kernel_sem(h[lo:lo + 16].cpu(), "e1m2_neg", "e2m1_pos", 128, 1.0), f"rows {lo}..")


def test_fused_fc1_argument_checks(dev):
    from fpqvar_amd import gemm
    a, w, bias = operands(dev, 8, 128, 128, 1)
    with pytest.raises(RuntimeError):
        gemm.linear_fp4_gelu_dual(a[0], a[1], w[0][:120], w[1][:120])        # outs % 128 != 0
    with pytest.raises(RuntimeError):
        gemm.linear_fp4_gelu_dual(a[0][:, :-8], a[1], *w)                     # truncated operand
    assert gemm.linear_fp4_gelu_dual(a[0][:0], a[1][:0], *w).shape == (0, 128)
    lib = _lib.lib()
    assert lib.fpq_gemm_fp4_gelu_dual(None, None, None, None, 1, None, None, None, 4, 100, 128, None, None) == -3   # shape before pointers
    assert lib.fpq_gemm_fp4_gelu_dual(None, None, None, None, 1, None, None, None, 4, 128, 128, None, None) == -1
    assert lib.fpq_gemm_fp4_gelu_dual(None, None, None, None, 7, None, None, None, 4, 128, 128, None, None) == -2
    assert lib.fpq_gemm_fp4_gelu_dual(None, None, None, None, 1, None, None, None, 0, 128, 128, None, None) == 0


def test_quantize_var_fuses_the_ffn(dev):
    """quantize_VAR(..., real_fp4=True, fuse_ffn=True): the reference-shaped FFN.forward - fc2(act(fc1(x))), tr/basic_var.py:120-121 -
    runs unchanged on the swapped modules and computes what the unfused real_fp4 model computes: fc2's input equals the dual
    quantizer of torch's GELU of the FP4Linear output wherever the two GELUs agree (they differ on 5 of 65536 inputs by one
    ulp, and a one-ulp step can move a group's scale: >= 99.9 % of the elements bit-equal), the FFN output within GEMM tolerance."""
    import copy
    from fpqvar_amd import gemm, quant_linear as ql

    class FFN(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.fc1, self.act, self.fc2 = torch.nn.Linear(256, 1024), torch.nn.GELU(approximate="tanh"), torch.nn.Linear(1024, 256)

        def forward(self, x):
            return self.fc2(self.act(self.fc1(x)))

    class Toy(torch.nn.Module):
        def __init__(self):
            super().__init__()
            torch.manual_seed(4)
            self.ffn = FFN()

    cfg = dict(weight_quant="per_group", act_quant="per_group", w_bit=4, a_bit=4, act_quant_sym=True, activation_fp_quant=True,
               weight_fp_quant=True, act_fp_type="fp_e2", weight_fp_type="fp_e2", fc2_fp_type="fp_e1m2_neg_e2m1_pos")
    base = Toy().to(dev)
    plain = ql.quantize_VAR(copy.deepcopy(base), real_fp4=True, **cfg).half()
    fused = ql.quantize_VAR(copy.deepcopy(base), real_fp4=True, fuse_ffn=True, **cfg).half()
    assert isinstance(fused.ffn.fc1, gemm.FP4LinearGeluDual) and isinstance(fused.ffn.act, torch.nn.Identity)
    assert type(fused.ffn.fc2).__name__ == "QuantizedLinear_fc2" and "epilogue" in repr(fused.ffn.fc2)
    x = torch.randn(3, 50, 256, device=dev).half()
    hq = fused.ffn.fc1(x)
    want = plain.ffn.fc2.act_quant(plain.ffn.act(plain.ffn.fc1(x)))
    assert hq.shape == want.shape == (3, 50, 1024)
    same = (hq.view(torch.int16) == want.view(torch.int16)).float().mean()
    assert float(same) >= 0.999, float(same)
    ya, yb = plain.ffn(x).float(), fused.ffn(x).float()
    assert float((ya - yb).abs().max()) <= 2e-2 * float(ya.abs().max()) + 1e-3
    with pytest.raises(ValueError):
        ql.quantize_VAR(copy.deepcopy(base), real_fp4=True, fuse_ffn=True, **{**cfg, "fc2_fp_type": "fp_e2"})   # not a dual format
    # without real_fp4 (the reference's fake-quant numerics), with the FP6 run's configuration, and with real_fp6: the activation
    # module does GELU + fc2's input quantizer in one pass, fc2 multiplies what it gets
    cfg6 = dict(weight_quant="per_channel", act_quant="per_token", w_bit=6, a_bit=6, act_quant_sym=True, activation_fp_quant=True,
                weight_fp_quant=True, act_fp_type="fp6_e2m3", weight_fp_type="fp6_e2m3", fc2_fp_type="fp6_int_neg_e2m3_pos")
    for c, extra in ((cfg, {}), (cfg6, {}), (cfg6, {"real_fp6": True}), ({**cfg, "fc2_fp_type": "fp4_afpq"}, {})):
        ref = ql.quantize_VAR(copy.deepcopy(base), **extra, **c).half()
        one = ql.quantize_VAR(copy.deepcopy(base), fuse_ffn=True, **extra, **c).half()
        assert isinstance(one.ffn.act, ql.GeluThenFc2Quant) and "one pass" in repr(one.ffn.act) and "one pass" in repr(one.ffn.fc2)
        assert type(one.ffn.fc1) is type(ref.ffn.fc1)
        y1 = ref.ffn.fc1(x)
        want, got = ref.ffn.fc2.act_quant(ref.ffn.act(y1)), one.ffn.act(one.ffn.fc1(x))
        assert float((want.view(torch.int16) == got.view(torch.int16)).float().mean()) >= 0.999, (c["fc2_fp_type"], extra)
        ya, yb = ref.ffn(x).float(), one.ffn(x).float()
        assert float((ya - yb).abs().max()) <= 2e-2 * float(ya.abs().max()) + 1e-3


@pytest.mark.parametrize("shape", [(300, 7680), (1, 128), (5, 37, 1024), (16900, 7680)])
def test_gelu_quant_rows_dual_one_pass(dev, shape):
    """The stand-alone form (fpq_gelu_quant_rows_dual: `fc2.act_quant(act(y))` in one pass over the fc1 output y, for the drop-in
    path whose fc1 stays a torch GEMM): the emitted GELU values within one fp16 ulp of torch's, the quantization bit-exact on
    them against the plain dual quantizer and the oracle, and equal to what the fc1 GEMM's epilogue computes for the same y."""
    from fpqvar_amd import ops, quant_utils as qu
    g = torch.Generator().manual_seed(sum(shape))
    y = (torch.randn(*shape, generator=g) * 1.5).half().to(dev)
    q, h = ops.gelu_quant_rows_dual(y, return_gelu=True)
    assert q.shape == h.shape == y.shape and q.dtype == torch.float16
    d = ulp_diff(h, Fn.gelu(y, approximate="tanh"))
    assert int(d.max()) <= 1 and float((d > 0).float().mean()) < 0.01
    assert_bits_equal(q, ops.quant_rows_dual(h, "e1m2_neg", "e2m1_pos", 128, 1.0), "fused vs the plain dual quantizer on the emitted GELU values")
    if y.numel() <= 1 << 22:
        assert_bits_equal(q.cpu(), orc.dual_per_group_kernel_sem(h.cpu().reshape(-1, 128), "e1m2_neg", "e2m1_pos", 128, 1.0).view(shape), "fused vs oracle")
    assert_bits_equal(qu.gelu_fp_quant_e1m2_neg_e2m1_pos_per_group_cuda(y, 4, 128), q, "quant_utils name, no GELU output")


def test_gelu_quant_rows_dual_every_fp16_input_and_nan_rule(dev):
    """All 65536 fp16 patterns through the stand-alone fused kernel: the same GELU as the GEMM epilogue (<= 1 ulp from torch, NaN
    where torch has NaN), and the NaN rule (the result is zero everywhere, the scratch zero again)."""
    from fpqvar_amd import ops
    every = torch.arange(0, 65536, dtype=torch.int32).to(torch.int16).view(torch.float16).to(dev)
    q, h = ops.gelu_quant_rows_dual(every, return_gelu=True)
    ref = Fn.gelu(every, approximate="tanh")
    d = ulp_diff(h, ref)
    assert int(d.max()) <= 1 and int((d > 0).sum()) <= 64 and torch.equal(torch.isnan(h), torch.isnan(ref))
    assert not bool(q.view(torch.int16).any())
    torch.cuda.synchronize()
    assert not bool(ops._nan_scratch(dev).any())
    finite = every[(every.abs() < 100) & ~torch.isnan(every)]
    finite = finite[: finite.numel() // 128 * 128]
    qf, hf = ops.gelu_quant_rows_dual(finite, return_gelu=True)
    assert bool(qf.any())
    assert_bits_equal(qf, ops.quant_rows_dual(hf, "e1m2_neg", "e2m1_pos", 128, 1.0), "finite inputs")
    with pytest.raises(RuntimeError):
        ops.gelu_quant_rows_dual(every.float())
    with pytest.raises(RuntimeError):
        ops.gelu_quant_rows_dual(every[:100])


@pytest.mark.parametrize("cols,rows", [(7680, 300), (9216, 37), (1920, 64), (128, 4000), (1000, 9)])
def test_gelu_quant_rows_dual_fp6_pairs(dev, cols, rows):
    """The W6A6 run's fc2 input (INT-/E2M3+, tr/quant_utils.py:577-646) with the GELU in front, one pass: per token (one
    workgroup per row of 7680 / 9216 / 1920 / a ragged 1000 elements) and per group of 128 - GELU within one ulp of torch's,
    quantization bit-exact on the emitted values against the plain quantizer and the oracle; no NaN rule (the reference does
    not clamp the FP6 pairs): a NaN poisons its own row only."""
    from fpqvar_amd import ops, quant_utils as qu
    g = torch.Generator().manual_seed(cols + rows)
    y = (torch.randn(rows, cols, generator=g) * 1.5).half().to(dev)
    q, h = ops.gelu_quant_rows_dual(y, "int_neg", "e2m3_pos", cols, None, return_gelu=True)
    d = ulp_diff(h, Fn.gelu(y, approximate="tanh"))
    assert int(d.max()) <= 1
    assert_bits_equal(q, ops.quant_rows_dual(h, "int_neg", "e2m3_pos", cols, None), "one pass vs GELU values + plain quantizer")
    want = orc.dual_per_group_kernel_sem(h.cpu().reshape(-1, cols), "int_neg", "e2m3_pos", cols, None).view(rows, cols) if cols == 128 else \
        orc.dual_per_token_kernel_sem(h.cpu(), "int_neg", "e2m3_pos")
    assert_bits_equal(q.cpu(), want, "one pass vs oracle")
    if cols == 128:
        assert_bits_equal(qu.gelu_fp6_quant_int_neg_e2m3_pos_per_group_cuda(y, 6, 128), q, "quant_utils per-group name")
    elif cols % 8 == 0:
        assert_bits_equal(qu.gelu_fp6_quant_int_neg_e2m3_pos_per_token_cuda(y, 6), q, "quant_utils per-token name")
    yn = y.clone()
    yn[1, 5] = float("nan")
    qn = ops.gelu_quant_rows_dual(yn, "int_neg", "e2m3_pos", cols, None)
    assert_bits_equal(qn[2:], q[2:], "rows after the NaN row are untouched")
    assert_bits_equal(qn[0], q[0], "the row before it too")
