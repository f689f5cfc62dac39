"""CPU-only checks of the QuantizedLinear mirror's dispatch and error behaviour
(the arithmetic needs the GPU and lives in tests/test_gpu_parity.py)."""
import pytest
import torch

from fpqvar_amd import quant_utils as qu
from fpqvar_amd.quant_linear import QuantizedLinear, QuantizedLinear_fc2, quantize_VAR


def test_dispatch_tables():
    m = QuantizedLinear(128, 64, act_quant="per_group", a_bit=4, activation_fp_quant=True, act_fp_type="fp_e2")
    assert m.act_quant.func is qu.fp_quant_e2_per_group_cuda and m.act_quant.keywords == {"n_bits": 4, "group_size": 128}
    m = QuantizedLinear(128, 64, act_quant="per_token", a_bit=4, activation_fp_quant=True, act_fp_type="fp_e2")
    assert m.act_quant.func is qu.fp_quant_e2_per_token          # the pure-torch (argmin) variant, as in the reference
    m = QuantizedLinear(128, 64, act_quant="per_token", a_bit=6, activation_fp_quant=True, act_fp_type="fp6_e3m2")
    assert m.act_quant.func is qu.fp6_quant_e3m2_per_token_cuda
    m = QuantizedLinear_fc2(128, 64, act_quant="per_group", a_bit=4, activation_fp_quant=True,
                            act_fp_type="fp_e1m2_neg_e2m1_pos")
    assert m.act_quant.func is qu.fp_quant_e1m2_neg_e2m1_pos_per_group_cuda
    m = QuantizedLinear_fc2(128, 64, act_quant="per_token", a_bit=6, activation_fp_quant=True,
                            act_fp_type="fp6_int_neg_e2m3_pos")
    assert m.act_quant.func is qu.fp6_quant_int_neg_e2m3_pos_per_token_cuda
    assert m.weight.dtype == torch.float16 and m.bias.shape == (1, 64)
    assert "QuantizedLinear_fc2128, 64" in repr(m)


def test_errors():
    with pytest.raises(ValueError, match="Invalid act_quant"):
        QuantizedLinear(8, 8, act_quant="per_banana", activation_fp_quant=True, act_fp_type="fp_e2")
    with pytest.raises(ValueError, match="Unsupported fp_type"):
        QuantizedLinear(8, 8, act_quant="per_group", activation_fp_quant=True, act_fp_type="fp_e1m2_neg_e2m1_pos")
    with pytest.raises(NotImplementedError):
        QuantizedLinear(8, 8, act_quant="per_token", activation_fp_quant=False)     # INT baseline: out of scope
    with pytest.raises(AssertionError):
        QuantizedLinear.from_float(torch.nn.ReLU())
    lin = torch.nn.Linear(128, 64)
    with pytest.raises(RuntimeError, match="GPU"):                                   # no CPU fallback
        QuantizedLinear.from_float(lin, weight_quant="per_group", act_quant="per_group", w_bit=4, a_bit=4,
                                   activation_fp_quant=True, weight_fp_quant=True, act_fp_type="fp_e2",
                                   weight_fp_type="fp_e2")


def test_quantize_var_walk_leaves_other_modules_alone():
    class Block(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.ada_lin = torch.nn.Sequential(torch.nn.SiLU(), torch.nn.Linear(16, 96))
            self.norm = torch.nn.LayerNorm(16)

    blk = Block()
    quantize_VAR(blk, weight_quant="per_group", act_quant="per_group", w_bit=4, a_bit=4, activation_fp_quant=True,
                 weight_fp_quant=True, act_fp_type="fp_e2", weight_fp_type="fp_e2", fc2_fp_type="fp_e2")
    assert isinstance(blk.ada_lin[1], torch.nn.Linear)      # not quantized (tr/quant_utils.py:1147-1155)


def test_viewable_rule_is_torchs_own():
    """The per-token FP6 functions raise exactly when the reference's `(x / scale).view(-1)` would
    (tr/quant_utils.py:508-510): decided with meta tensors, checked here against the real ops on the CPU for the
    layouts of the KV path (tr/basic_var.py:173-194) and a sweep of strided views."""
    import itertools
    import pytest
    import torch
    from fpqvar_amd import quant_utils as qu

    def reference_raises(x, dual):
        t = torch.where(x <= 0, x, torch.zeros_like(x)) if dual else x
        scale = t.abs().max(dim=-1, keepdim=True)[0] / 7.5
        try:
            (t / scale).view(-1)
            return False
        except RuntimeError:
            return True

    def ours_raises(x, dual):
        try:
            qu._require_viewable(x, dual)
            return False
        except RuntimeError:
            return True

    B, L, H, c = 2, 5, 3, 8
    qkv = torch.randn(B, L, 3 * H * c)
    q, k, v = qkv.view(B, L, 3, H, c).unbind(2)             # flash layout: sliced, not dense
    bhlc = qkv.view(B, L, 3, H, c).permute(2, 0, 3, 1, 4)[1]  # BHLc: permuted
    dense_perm = torch.randn(B, L, H, c).transpose(1, 2)
    layouts = {"contiguous": torch.randn(B, L, H, c), "unbind k": k, "unbind v": v, "BHLc of qkv": bhlc,
               "dense permuted": dense_perm, "last-dim strided": torch.randn(4, 16)[:, ::2],
               "row slice": torch.randn(8, 6, 4)[1:7:2], "expanded": torch.randn(1, 4).expand(3, 4),
               "t()": torch.randn(6, 4).t()}
    for name, x in layouts.items():
        for dual in (False, True):
            assert ours_raises(x, dual) == reference_raises(x, dual), (name, dual)
    assert not ours_raises(k, False) and ours_raises(bhlc, False) and ours_raises(dense_perm, False)
    base = torch.randn(6, 5, 4, 3)
    for perm in itertools.permutations(range(4)):
        for sl in (slice(None), slice(0, None, 2)):
            x = base.permute(perm)[sl]
            assert ours_raises(x, False) == reference_raises(x, False), (perm, sl)
    # fp_quant_e2_per_group views its ARGUMENT (tr/quant_utils.py:302)
    for name, x in {"contig": torch.randn(4, 256), "unbind": torch.randn(2, 3, 3 * 256).view(2, 3, 3, 256).unbind(2)[1],
                    "t": torch.randn(128, 4).t()}.items():
        try:
            x.view(-1, 128)
            ref = False
        except RuntimeError:
            ref = True
        try:
            qu._require_input_viewable(x, 128)
            got = False
        except RuntimeError:
            got = True
        assert got == ref, name
