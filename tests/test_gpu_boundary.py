"""Behaviour at the edges of the drop-in boundary (round-1 review items): layouts the reference accepts, float casts
of the matrix-core modules, operand validation.  Needs a real MI355X: run with ``-m gpu``."""
import copy

import pytest
import torch

from oracle import fpq_oracle as orc
from tests.conftest import assert_bits_equal

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need the GPU"
    return torch.device("cuda:0")


@pytest.mark.parametrize("kv_bit", (6, 4))
def test_kv_cache_update_takes_the_references_unbind_views(dev, kv_bit):
    """tr/basic_var.py:187-194 caches `k`, `v` as they come out of `qkv.view(B, L, 3, H, c).unbind(2)` - sliced, not
    dense - and quantizes them at the next step: `k / scale` is contiguous there, so the reference runs; so must we."""
    from fpqvar_amd import kv_cache as kv, quant_utils as qu
    g = torch.Generator().manual_seed(17)
    B, L, H, c = 3, 6, 30, 64
    qkv = torch.randn(B, L, 3 * H * c, generator=g).half()
    _, ck, cv = qkv.to(dev).view(B, L, 3, H, c).unbind(2)
    assert not ck.is_contiguous()
    k = torch.randn(B, 2, H, c, generator=g).half()
    v = torch.randn(B, 2, H, c, generator=g).half()
    ck_cpu, cv_cpu = qkv.view(B, L, 3, H, c).unbind(2)[1:]
    want_fn = (lambda t: orc.per_token_kernel_sem(t.contiguous(), "e2m3")) if kv_bit == 6 else \
              (lambda t: orc.per_group_kernel_sem(t.contiguous(), "e2m1", 128))
    nk, nv = kv.update_kv_cache(ck, cv, k.to(dev), v.to(dev), True, kv_bit, 1)
    assert_bits_equal(nk, torch.cat((want_fn(ck_cpu), k), 1), f"k, kv_bit {kv_bit}")
    assert_bits_equal(nv, torch.cat((want_fn(cv_cpu), v), 1), f"v, kv_bit {kv_bit}")
    # the dual per-token quantizer follows the same rule (tr/quant_utils.py:614-646)
    assert_bits_equal(qu.fp6_quant_int_neg_e2m3_pos_per_token_cuda(ck, 6),
                      orc.dual_per_token_kernel_sem(ck_cpu.contiguous(), "int_neg", "e2m3_pos"), "dual per-token on a view")
    # the permuted (BHLc) layout still raises for the per-token forms, exactly as the reference's .view(-1) does
    bhlc = qkv.to(dev).view(B, L, 3, H, c).permute(2, 0, 3, 1, 4)[1]
    with pytest.raises(RuntimeError):
        kv.quantize_kv(bhlc, 6)
    with pytest.raises(RuntimeError):
        qu.fp6_quant_int_neg_e2m3_pos_per_token_cuda(bhlc, 6)
    # ... and the argmin per-group function views its ARGUMENT: the sliced view raises there (x.view(-1, 128))
    with pytest.raises(RuntimeError):
        qu.fp_quant_e2_per_group(ck, 4, 128)


def test_half_leaves_the_quantization_scales_in_fp32(dev):
    """evaluate_fp_quant_transform_rotate.py:131 calls var.half() after quantize_VAR: the matrix-core Linears keep
    their fp32 weight scales (code * fp32 scale is what the reference's fp32 weight quantization produced)."""
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
            for m in (self.ffn.fc1, self.attn.mat_qkv):
                m.weight.data.mul_(1e-4)          # scales ~1e-6: far below fp16's normal range

    cfg4 = dict(weight_quant="per_group", act_quant="per_group", w_bit=4, a_bit=4, act_quant_sym=True, activation_fp_quant=True,
                weight_fp_quant=True, act_fp_type="fp_e2", weight_fp_type="fp_e2", fc2_fp_type="fp_e1m2_neg_e2m1_pos")
    cfg6 = dict(weight_quant="per_channel", act_quant="per_token", w_bit=6, a_bit=6, act_quant_sym=True, activation_fp_quant=True,
                weight_fp_quant=True, act_fp_type="fp6_e2m3", weight_fp_type="fp6_e2m3", fc2_fp_type="fp6_int_neg_e2m3_pos")
    x = torch.randn(40, 256, device=dev).half()
    for kind, cfg, flag in (("fp4", cfg4, "real_fp4"), ("fp6", cfg6, "real_fp6")):
        base = Toy().to(dev)
        plain = ql.quantize_VAR(copy.deepcopy(base), **{flag: True}, **cfg)
        halved = ql.quantize_VAR(copy.deepcopy(base), **{flag: True}, **cfg).half()
        n_checked = 0
        for a, b in zip(plain.modules(), halved.modules()):
            if isinstance(a, (gemm.FP4Linear, gemm.FP8Linear, gemm.FP6Linear)):
                assert b.w_scales.dtype == torch.float32, kind
                assert torch.equal(a.w_scales, b.w_scales) and torch.equal(a.w_codes, b.w_codes)
                assert b.bias is None or b.bias.dtype == torch.float16
                assert torch.equal(a(x), b(x)), kind
                n_checked += 1
        assert n_checked >= 3, kind
        moved = halved.float().to("cpu")
        for m in moved.modules():
            if isinstance(m, (gemm.FP4Linear, gemm.FP8Linear, gemm.FP6Linear)):
                assert m.w_scales.dtype == torch.float32 and m.w_scales.device.type == "cpu"


def test_code_operands_are_validated_before_the_kernel(dev):
    from fpqvar_amd import gemm, ops, packed
    x = torch.randn(64, 256, device=dev)
    codes, scales = ops.quant_rows_codes(x, "e2m1", 128, pack_nibbles=True)
    ops.dequant_rows_codes(codes, scales, "e2m1", 128, torch.float16, True)
    for bad_codes, bad_scales in ((codes[:-1], scales), (codes, scales[:-1]), (codes.to(torch.int16), scales),
                                  (codes, scales.double()), (codes[:, ::2], scales), (codes, scales.cpu())):
        with pytest.raises(RuntimeError):
            ops.dequant_rows_codes(bad_codes, bad_scales, "e2m1", 128, torch.float16, True)
    with pytest.raises(RuntimeError):
        ops.dequant_rows_codes(codes, scales, "e2m1", 128, torch.float16, False)    # not packed: twice the bytes expected
    p = packed.PackedWeight(codes[:10].contiguous(), scales, "e2m1", 128, (64, 256))
    with pytest.raises(RuntimeError):
        p.dequantize()
    a = gemm.quantize_mx(torch.randn(16, 256, device=dev).half())
    w = gemm.quantize_mx(torch.randn(24, 256, device=dev))
    gemm.linear_fp4(*a, *w)
    with pytest.raises(RuntimeError):
        gemm.linear_fp4(a[0], a[1][:-1], *w)
    with pytest.raises(RuntimeError):
        gemm.linear_fp4(a[0], a[1], w[0], w[1].reshape(-1)[:-1])
    with pytest.raises(RuntimeError):
        gemm.linear_fp4(a[0].cpu(), a[1], *w)
    a8 = gemm.quantize_fp8(torch.randn(16, 256, device=dev).half())
    w8 = gemm.quantize_fp8(torch.randn(24, 256, device=dev))
    gemm.linear_fp8(*a8, *w8)
    with pytest.raises(RuntimeError):
        gemm.linear_fp8(a8[0], a8[1][:-1], *w8)
    with pytest.raises(RuntimeError):
        gemm.linear_fp8(a8[0][:, :128], a8[1], *w8)


def test_calibrate_codes_exchange_rejects_a_custom_quantizer(dev):
    from fpqvar_amd import calibrate as cal
    w = {"a": torch.randn(4, 128, device=dev)}
    with pytest.raises(ValueError):
        cal.calibrate_sharded(w, quantize=lambda n, t: t.half(), exchange="codes")


def test_dual_nan_flag_scratch_cleans_itself(dev):
    """fp_quant_e1m2_neg_e2m1_pos_per_group_cuda's NaN rule (tr/quant_utils.py:421-422: a NaN anywhere zeroes the whole
    result) runs as quantizer + 64-workgroup fix-up with NO memset: the fix-up leaves the scratch zero, so NaN and
    clean calls can alternate on one stream, eagerly and as a replayed graph."""
    from fpqvar_amd import ops, quant_utils as qu
    g = torch.Generator().manual_seed(5)
    clean = torch.nn.functional.gelu(torch.randn(300, 7680, generator=g)).half()
    dirty = clean.clone()
    dirty[123, 4567] = float("nan")
    want_clean = orc.dual_per_group_kernel_sem(clean, "e1m2_neg", "e2m1_pos", 128, 1.0)
    c, d = clean.to(dev), dirty.to(dev)
    for step in range(6):
        if step % 2 == 0:
            assert_bits_equal(qu.fp_quant_e1m2_neg_e2m1_pos_per_group_cuda(c, 4, 128), want_clean, f"clean call {step}")
        else:
            got = qu.fp_quant_e1m2_neg_e2m1_pos_per_group_cuda(d, 4, 128)
            assert not got.any() and not torch.isnan(got).any(), f"NaN call {step}: everything must be +0"
        assert not ops._nan_scratch(dev).any(), "the scratch must be zero again after every call"
    # tiny inputs (fewer elements than fix-up lanes), fp32 input, per-token rows
    t = torch.tensor([[1.0, -2.0, float("nan"), 0.5] * 32], dtype=torch.float16)
    assert not qu.fp_quant_e1m2_neg_e2m1_pos_per_group_cuda(t.to(dev), 4, 128).any()
    assert_bits_equal(qu.fp_quant_e1m2_neg_e2m1_pos_per_group_cuda(clean[:1].float().to(dev), 4, 128),
                      orc.dual_per_group_kernel_sem(clean[:1].float(), "e1m2_neg", "e2m1_pos", 128, 1.0), "fp32 after NaN")
    # a captured call, replayed with clean / dirty / clean contents of its static input
    static_in = c.clone()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        qu.fp_quant_e1m2_neg_e2m1_pos_per_group_cuda(static_in, 4, 128)      # warm-up on the capture stream
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=s):
            static_out = qu.fp_quant_e1m2_neg_e2m1_pos_per_group_cuda(static_in, 4, 128)
    torch.cuda.current_stream().wait_stream(s)
    for step, src in enumerate((c, d, c, d, d, c)):
        static_in.copy_(src)
        graph.replay()
        torch.cuda.synchronize()
        if src is c:
            assert_bits_equal(static_out, want_clean, f"replay {step} (clean)")
        else:
            assert not static_out.any(), f"replay {step} (NaN)"


def test_quant_rows_multi_equals_single_calls(dev):
    """fpq_quant_rows_multi: several tensors, one call; same bits as one fpq_quant_rows per tensor, whatever path it
    takes (one fused launch for <= 8 aligned fp16 tensors, per-tensor launches otherwise)."""
    from fpqvar_amd import ops
    g = torch.Generator().manual_seed(31)
    shapes = [(3, 7, 30, 64), (5, 128), (1, 64), (1000, 64), (2, 3, 1920)]
    for dtype in (torch.float16, torch.float32):
        for cols, table in ((64, "e2m3"), (128, "e2m1"), (1920, "e2m3"), (8, "e2m1")):
            xs = [torch.randn(*s, generator=g).to(dtype) for s in shapes if (torch.Size(s).numel() % cols) == 0]
            if not xs:
                continue
            xs[0].view(-1)[: cols] = 0
            got = ops.quant_rows_multi([x.to(dev) for x in xs], table, cols, torch.float16)
            assert len(got) == len(xs)
            for x, y in zip(xs, got):
                assert y.shape == x.shape
                assert_bits_equal(y, ops.quant_rows(x.to(dev), table, cols, torch.float16), f"multi {dtype} cols={cols}")
                assert_bits_equal(y, orc.per_token_kernel_sem(x.reshape(-1, cols), table).view(x.shape), f"multi vs oracle cols={cols}")
    # more than eight tensors, an unaligned one, an empty list
    many = [torch.randn(4, 128, generator=g).half().to(dev) for _ in range(11)]
    for x, y in zip(many, ops.quant_rows_multi(many, "e2m1", 128)):
        assert_bits_equal(y, ops.quant_rows(x, "e2m1", 128), "eleven tensors")
    odd = torch.randn(4 * 128 + 4, generator=g).half().to(dev)[4:].view(4, 128)
    for x, y in zip([many[0], odd], ops.quant_rows_multi([many[0], odd], "e2m1", 128)):
        assert_bits_equal(y, ops.quant_rows(x, "e2m1", 128), "unaligned member")
    assert ops.quant_rows_multi([], "e2m1", 128) == []
    with pytest.raises(RuntimeError):
        ops.quant_rows_multi([many[0], many[1].float()], "e2m1", 128)
    with pytest.raises(RuntimeError):
        ops.quant_rows_multi([many[0].cpu()], "e2m1", 128)
