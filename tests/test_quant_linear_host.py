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
