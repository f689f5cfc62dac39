"""Host-side mirror of the reference's quantized Linear wrappers.

``QuantizedLinear`` / ``QuantizedLinear_fc2`` / ``quantize_VAR`` keep the reference's
constructor arguments, string flags and dispatch
(models_fp_quant_transform_rotate/quant_utils.py:649-867, 870-1092, 1095-1167):
activations are fake-quantized on every forward, weights once in ``from_float``;
``forward`` is ``F.linear(act_quant(x), W, b)``.

Only the floating-point formats are wired up (``activation_fp_quant`` /
``weight_fp_quant`` = True): the INT / log2 RTN baselines of the reference are single
fused torch ops already and are out of scope here; asking
for them raises ``NotImplementedError``.
"""
from __future__ import annotations

from functools import partial

import torch
from torch import nn

from . import quant_utils as qu

_GROUP = 128   # hard-coded in every partial(...) of the reference (tr/quant_utils.py:727-735)


def _act_quantizer(act_quant, activation_fp_quant, act_fp_type, a_bit, fc2: bool):
    """tr/quant_utils.py:696-746 (QuantizedLinear) and :917-973 (QuantizedLinear_fc2)."""
    if act_quant not in ("per_token", "per_tensor", "per_group"):
        raise ValueError(f"Invalid act_quant: {act_quant}")
    if not activation_fp_quant or act_quant == "per_tensor":
        raise NotImplementedError("INT / log2 activation quantizers are out of scope (SURVEY.md section 2 row 4)")
    if act_quant == "per_token":
        table = {"fp_e1": qu.fp_quant_e1_per_token, "fp_e2": qu.fp_quant_e2_per_token,
                 "fp_e3": qu.fp_quant_e3_per_token, "fp6_e2m3": qu.fp6_quant_e2m3_per_token_cuda,
                 "fp6_e3m2": qu.fp6_quant_e3m2_per_token_cuda}
        if fc2:
            table["fp6_int_neg_e2m3_pos"] = qu.fp6_quant_int_neg_e2m3_pos_per_token_cuda
        if act_fp_type not in table:
            raise ValueError("Unsupported fp_type.")
        return partial(table[act_fp_type], n_bits=a_bit)
    table = {"fp_e1": qu.fp_quant_e1_per_group_cuda, "fp_e2": qu.fp_quant_e2_per_group_cuda,
             "fp_e3": qu.fp_quant_e3_per_group_cuda, "fp6_e2m3": qu.fp6_quant_e2m3_per_group_cuda,
             "fp6_e3m2": qu.fp6_quant_e3m2_per_group_cuda}
    if fc2:
        table["fp_e1m2_neg_e2m1_pos"] = qu.fp_quant_e1m2_neg_e2m1_pos_per_group_cuda
        table["fp6_int_neg_e2m3_pos"] = qu.fp6_quant_int_neg_e2m3_pos_per_group_cuda
        table["fp4_afpq"] = qu.fp4_afpq_per_group_cuda          # models_fp_quant/quant_utils.py:1040-1041
    if act_fp_type not in table:
        raise ValueError("Unsupported fp_type.")
    return partial(table[act_fp_type], n_bits=a_bit, group_size=_GROUP)


def _quantize_weight(w, weight_quant, weight_fp_quant, weight_fp_type, w_bit):
    """tr/quant_utils.py:794-855.  per_channel FP4 goes through the pure-torch (argmin)
    functions exactly as in the reference; per_group through the `_cuda` ones."""
    if weight_quant == "per_tensor" or not weight_fp_quant:
        raise NotImplementedError("INT weight quantizers are out of scope (SURVEY.md section 2 row 4)")
    if weight_quant == "per_channel":
        table = {"fp_e1": qu.fp_quant_e1_per_token, "fp_e2": qu.fp_quant_e2_per_token,
                 "fp_e3": qu.fp_quant_e3_per_token, "fp6_e2m3": qu.fp6_quant_e2m3_per_token_cuda,
                 "fp6_e3m2": qu.fp6_quant_e3m2_per_token_cuda}
        if weight_fp_type not in table:
            raise ValueError("Unsupported fp_type.")
        return table[weight_fp_type](w, n_bits=w_bit)
    if weight_quant == "per_group":
        table = {"fp_e1": qu.fp_quant_e1_per_group_cuda, "fp_e2": qu.fp_quant_e2_per_group_cuda,
                 "fp_e3": qu.fp_quant_e3_per_group_cuda, "fp6_e2m3": qu.fp6_quant_e2m3_per_group_cuda,
                 "fp6_e3m2": qu.fp6_quant_e3m2_per_group_cuda}
        if weight_fp_type not in table:
            raise ValueError("Unsupported fp_type.")
        return table[weight_fp_type](w, n_bits=w_bit, group_size=_GROUP)
    raise ValueError(f"Invalid weight_quant: {weight_quant}")


class GeluThenFc2Quant(nn.Module):
    """Stands in for an FFN's `act` (GELU, approximate="tanh") when fc2's dual-format input quantizer is fused behind it:
    `forward(y)` = `fc2.act_quant(F.gelu(y, approximate="tanh"))` in ONE pass over the fc1 output (quant_utils.gelu_*), for the
    FFN.forward of the reference (`self.fc2(self.act(self.fc1(x)))`, tr/basic_var.py:120-121) - fc2's own input quantizer is
    switched off by quantize_VAR(..., fuse_ffn=True)."""
    FUSED = {("per_group", "fp_e1m2_neg_e2m1_pos"): lambda y, bits: qu.gelu_fp_quant_e1m2_neg_e2m1_pos_per_group_cuda(y, bits, _GROUP),
             ("per_group", "fp4_afpq"): lambda y, bits: qu.gelu_fp4_afpq_per_group_cuda(y, bits, _GROUP),
             ("per_group", "fp6_int_neg_e2m3_pos"): lambda y, bits: qu.gelu_fp6_quant_int_neg_e2m3_pos_per_group_cuda(y, bits, _GROUP),
             ("per_token", "fp6_int_neg_e2m3_pos"): lambda y, bits: qu.gelu_fp6_quant_int_neg_e2m3_pos_per_token_cuda(y, bits)}

    def __init__(self, act_quant: str, fc2_fp_type: str, a_bit: int):
        super().__init__()
        self.act_quant, self.fc2_fp_type, self.a_bit = act_quant, fc2_fp_type, a_bit
        self.fn = self.FUSED[(act_quant, fc2_fp_type)]

    @torch.no_grad()
    def forward(self, y):
        return self.fn(y.to(torch.float16), self.a_bit)

    def extra_repr(self):
        return f"GELU(tanh) + {self.fc2_fp_type} {self.act_quant} (one pass)"


class QuantizedLinear(nn.Module):
    _FC2 = False

    def __init__(self, in_features, out_features, bias=True, act_quant=None, quantize_output=False,
                 w_bit=8, a_bit=8, act_quant_sym=True, fc2_act_log2_quant=False, activation_fp_quant=False,
                 weight_fp_quant=False, act_fp_type=False, weight_fp_type=False):
        super().__init__()
        self.in_features = in_features
        self.out_features = out_features
        self.w_bit = w_bit
        self.a_bit = a_bit
        self.act_quant_sym = act_quant_sym
        self.fc2_act_log2_quant = fc2_act_log2_quant
        self.activation_fp_quant = activation_fp_quant
        self.weight_fp_quant = weight_fp_quant
        self.act_fp_type = act_fp_type
        self.register_buffer("weight", torch.randn(out_features, in_features, dtype=torch.float16))
        if bias:
            self.register_buffer("bias", torch.zeros((1, out_features), dtype=torch.float16))
        else:
            self.register_buffer("bias", None)
        self.act_quant_name = act_quant
        self.act_quant = _act_quantizer(act_quant, activation_fp_quant, act_fp_type, a_bit, self._FC2)
        if quantize_output:
            self.output_quant_name = self.act_quant_name
            self.output_quant = self.act_quant
        else:
            self.output_quant_name = "None"
            self.output_quant = lambda x: x
        self.weight_quant_name = None

    def to(self, *args, **kwargs):
        super().to(*args, **kwargs)
        self.weight = self.weight.to(*args, **kwargs)
        if self.bias is not None:
            self.bias = self.bias.to(*args, **kwargs)
        return self

    @torch.no_grad()
    def forward(self, x):
        q_x = self.act_quant(x)
        y = torch.functional.F.linear(q_x, self.weight, self.bias)
        return self.output_quant(y)

    @classmethod
    def from_float(cls, module, weight_quant="per_channel", act_quant="per_token", quantize_output=False,
                   w_bit=8, a_bit=8, act_quant_sym=None, fc2_act_log2_quant=False, activation_fp_quant=False,
                   weight_fp_quant=False, act_fp_type=None, weight_fp_type=None):
        assert isinstance(module, torch.nn.Linear)
        new = cls(module.in_features, module.out_features, module.bias is not None, act_quant=act_quant,
                  quantize_output=quantize_output, w_bit=w_bit, a_bit=a_bit, act_quant_sym=act_quant_sym,
                  fc2_act_log2_quant=fc2_act_log2_quant, activation_fp_quant=activation_fp_quant,
                  weight_fp_quant=weight_fp_quant, act_fp_type=act_fp_type, weight_fp_type=weight_fp_type)
        new.weight = _quantize_weight(module.weight.detach(), weight_quant, weight_fp_quant, weight_fp_type, w_bit)
        new.weight_quant_name = weight_quant
        if module.bias is not None:
            new.bias = module.bias
        return new

    def __repr__(self):
        return (f"{type(self).__name__}{self.in_features}, {self.out_features}, bias={self.bias is not None}, "
                f"weight_quant={self.weight_quant_name}, act_quant={self.act_quant_name}, "
                f"output_quant={self.output_quant_name}, w_bit={self.w_bit}, a_bit={self.a_bit}, "
                f"act_quant_sym={self.act_quant_sym}, act_log2_quant={self.fc2_act_log2_quant},"
                f"activation_fp_quant={self.activation_fp_quant}, weight_fp_quant={self.weight_fp_quant}, "
                f"activation_quant_type={self.act_fp_type}")


class QuantizedLinear_fc2(QuantizedLinear):
    """The fc2 variant accepts the asymmetric dual formats for its (post-GELU) input
    (tr/quant_utils.py:917-973)."""
    _FC2 = True


def quantize_VAR(model, weight_quant=None, act_quant=None, quantize_bmm_input=False, w_bit=8, a_bit=8, kv_bit=8,
                 act_quant_sym=None, fc2_act_log2_quant=None, quant_kv=None, activation_fp_quant=False,
                 weight_fp_quant=False, act_fp_type=None, weight_fp_type=None, fc2_fp_type=None, real_fp4=False,
                 real_fp6=False, fuse_ffn=False, kmajor_operands=True):
    """tr/quant_utils.py:1095-1167.  The reference matches its own FFN / SelfAttention
    classes; here a module with Linear children ``fc1``+``fc2`` is an FFN and one with
    ``mat_qkv``+``proj`` is a self-attention block.  As in the reference,
    ``quantize_bmm_input``, ``kv_bit`` and ``quant_kv`` are accepted and ignored, the
    ada_lin Linears stay in full precision, and fc2's input format is ``fc2_fp_type``.

    ``real_fp4`` (additive, default off = the reference's behaviour): in the W4A4 per-group ``fp_e2`` configuration
    fc1 / mat_qkv / proj become ``gemm.FP4Linear`` - same quantization decisions, product on the FP4 matrix cores
    instead of an fp16 GEMM on de-quantized tensors (fc2 keeps its dual-format fake quantization).

    ``fuse_ffn`` (additive; needs an FFN whose ``act`` is GELU(tanh) and a dual-format ``fc2_fp_type``): fc2's input quantizer
    moves in front of fc2 - the FFN's own ``forward`` (``fc2(act(fc1(x)))``, tr/basic_var.py:120-121) stays as it is.
    With ``real_fp4`` and ``fp_e1m2_neg_e2m1_pos``: fc1 becomes ``gemm.FP4LinearGeluDual`` (GELU and the quantizer in the fc1
    GEMM's epilogue) and ``act`` an identity; otherwise (the fake-quant default, ``real_fp6``, the FP6 / AFPQ dual pairs) ``act``
    becomes ``GeluThenFc2Quant`` - GELU and the quantizer in one pass over the fc1 output.  fc2 multiplies its already
    quantized input either way.

    ``kmajor_operands`` (with ``real_fp4`` / ``real_fp6``'s FP4 / FP6 Linears; default on): weights are held, and activations
    quantized, as k-major operand images (include/fpq.h) - the layout the GEMMs' LDS-DMA engine reads in contiguous 1 KiB
    pieces; results are bit-identical to the row-major form, the GEMMs 4 - 30 % faster."""
    fp4_ok = (real_fp4 and weight_quant == "per_group" and act_quant == "per_group" and w_bit == 4 and a_bit == 4
              and activation_fp_quant and weight_fp_quant and act_fp_type == "fp_e2" and weight_fp_type == "fp_e2")
    if real_fp4 and not fp4_ok:
        raise ValueError("real_fp4 needs weight_quant = act_quant = 'per_group', w_bit = a_bit = 4, fp_e2 on both sides")
    # ``real_fp6`` (additive, default off): in the W6A6 per_channel / per_token configuration (run.sh:7) fc1 / mat_qkv /
    # proj become ``gemm.FP8Linear`` - levels stored as E4M3 bytes, one scale per row, product on the FP8 matrix cores.
    fp6_ok = (real_fp6 and weight_quant == "per_channel" and act_quant == "per_token" and w_bit == 6 and a_bit == 6
              and activation_fp_quant and weight_fp_quant and act_fp_type in ("fp6_e2m3", "fp6_e3m2")
              and weight_fp_type in ("fp6_e2m3", "fp6_e3m2"))
    if real_fp6 and not fp6_ok:
        raise ValueError("real_fp6 needs weight_quant='per_channel', act_quant='per_token', w_bit = a_bit = 6, fp6 formats")
    common = dict(weight_quant=weight_quant, act_quant=act_quant, w_bit=w_bit, a_bit=a_bit,
                  activation_fp_quant=activation_fp_quant, weight_fp_quant=weight_fp_quant,
                  weight_fp_type=weight_fp_type)
    def plain(lin, **kw):
        if fp4_ok and lin.in_features % 128 == 0 and lin.out_features % 8 == 0:
            from .gemm import FP4Linear
            return FP4Linear.from_float(lin, kmajor=kmajor_operands)
        if fp6_ok and lin.in_features % 128 == 0 and lin.out_features % 8 == 0:
            from .gemm import FP6Linear, FP8Linear
            if weight_fp_type == "fp6_e2m3" and act_fp_type == "fp6_e2m3":
                return FP6Linear.from_float(lin, kmajor=kmajor_operands)   # 6-bit packed operands
            return FP8Linear.from_float(lin, weight_fp_type, act_fp_type)   # mixed / E3M2: E4M3-coded levels
        return QuantizedLinear.from_float(lin, **kw)

    for _, m in list(model.named_modules()):
        fc1, fc2 = getattr(m, "fc1", None), getattr(m, "fc2", None)
        qkv, proj = getattr(m, "mat_qkv", None), getattr(m, "proj", None)
        if isinstance(fc1, nn.Linear) and isinstance(fc2, nn.Linear):
            act = getattr(m, "act", None)
            gelu_tanh = isinstance(act, nn.GELU) and getattr(act, "approximate", "none") == "tanh"
            in_gemm = (fuse_ffn and gelu_tanh and fp4_ok and fc2_fp_type == "fp_e1m2_neg_e2m1_pos"
                       and fc1.in_features % 128 == 0 and fc1.out_features % 128 == 0)
            one_pass = (fuse_ffn and gelu_tanh and not in_gemm and activation_fp_quant and (act_quant, fc2_fp_type) in GeluThenFc2Quant.FUSED
                        and fc1.out_features % (128 if act_quant == "per_group" else 8) == 0)
            if fuse_ffn and not (in_gemm or one_pass):
                raise ValueError("fuse_ffn needs an FFN with act = GELU(approximate='tanh') and a dual-format fc2_fp_type "
                                 "(fp_e1m2_neg_e2m1_pos / fp4_afpq per group, fp6_int_neg_e2m3_pos per group or per token)")
            m.fc2 = QuantizedLinear_fc2.from_float(fc2, act_quant_sym=False, fc2_act_log2_quant=fc2_act_log2_quant,
                                                   act_fp_type=fc2_fp_type, **common)
            if in_gemm:
                from .gemm import FP4LinearGeluDual
                m.fc1 = FP4LinearGeluDual.from_float(fc1, kmajor=kmajor_operands)
                m.act = nn.Identity()
                m.fc2.act_quant = lambda t: t                      # its input arrives quantized from fc1's epilogue
                m.fc2.act_quant_name = "in fc1's epilogue"
            else:
                m.fc1 = plain(fc1, act_quant_sym=act_quant_sym, act_fp_type=act_fp_type, **common)
                if one_pass:
                    m.act = GeluThenFc2Quant(act_quant, fc2_fp_type, a_bit)
                    m.fc2.act_quant = lambda t: t                  # ... from the activation module in front of it
                    m.fc2.act_quant_name = "behind the GELU (one pass)"
        elif isinstance(qkv, nn.Linear) and isinstance(proj, nn.Linear):
            m.mat_qkv = plain(qkv, act_quant_sym=act_quant_sym, act_fp_type=act_fp_type, **common)
            m.proj = plain(proj, act_quant_sym=act_quant_sym, act_fp_type=act_fp_type, **common)
    return model


# ---- per-block mixed formats (the older variants' quantize_VAR_* functions, as data) --------------------------
def _block_index(name: str) -> int:
    """'blocks.<i>.ffn' -> i, as the reference does (int(name.split('.')[1]))."""
    return int(name.split(".")[1])


def quantize_VAR_mixed(model, layer_formats, weight_quant=None, act_quant=None, w_bit=8, a_bit=8, act_quant_sym=None,
                       fc2_act_log2_quant=None, activation_fp_quant=False, weight_fp_quant=False,
                       ada_lin_formats=None):
    """quantize_VAR with a format pair per (block, layer): ``layer_formats(block_idx, layer)`` returns
    ``(act_fp_type, weight_fp_type)`` for layer in {"fc1", "fc2", "mat_qkv", "proj"}.  ``ada_lin_formats``:
    None leaves the AdaLN Linear in full precision (as tr/ does), a pair quantizes ``ada_lin[1]`` (as the fq/ and
    rot/ mixed variants do).  Module matching is duck-typed as in quantize_VAR."""
    common = dict(weight_quant=weight_quant, act_quant=act_quant, w_bit=w_bit, a_bit=a_bit,
                  activation_fp_quant=activation_fp_quant, weight_fp_quant=weight_fp_quant)
    for name, m in list(model.named_modules()):
        fc1, fc2 = getattr(m, "fc1", None), getattr(m, "fc2", None)
        qkv, proj = getattr(m, "mat_qkv", None), getattr(m, "proj", None)
        ada = getattr(m, "ada_lin", None)
        if isinstance(fc1, nn.Linear) and isinstance(fc2, nn.Linear):
            b = _block_index(name)
            a, w = layer_formats(b, "fc1")
            m.fc1 = QuantizedLinear.from_float(fc1, act_quant_sym=act_quant_sym, act_fp_type=a, weight_fp_type=w, **common)
            a, w = layer_formats(b, "fc2")
            m.fc2 = QuantizedLinear_fc2.from_float(fc2, act_quant_sym=False, fc2_act_log2_quant=fc2_act_log2_quant,
                                                   act_fp_type=a, weight_fp_type=w, **common)
        elif isinstance(qkv, nn.Linear) and isinstance(proj, nn.Linear):
            b = _block_index(name)
            a, w = layer_formats(b, "mat_qkv")
            m.mat_qkv = QuantizedLinear.from_float(qkv, act_quant_sym=act_quant_sym, act_fp_type=a, weight_fp_type=w, **common)
            a, w = layer_formats(b, "proj")
            m.proj = QuantizedLinear.from_float(proj, act_quant_sym=act_quant_sym, act_fp_type=a, weight_fp_type=w, **common)
        if ada_lin_formats is not None and isinstance(ada, nn.Sequential) and len(ada) > 1 and isinstance(ada[1], nn.Linear):
            a, w = ada_lin_formats
            ada[1] = QuantizedLinear.from_float(ada[1], act_quant_sym=act_quant_sym, act_fp_type=a, weight_fp_type=w, **common)
    return model


def quantize_VAR_mixed_fp4_datatype(model, weight_quant=None, act_quant=None, quantize_bmm_input=False, w_bit=8,
                                    a_bit=8, kv_bit=8, act_quant_sym=None, fc2_act_log2_quant=None, quant_kv=None,
                                    activation_fp_quant=False, weight_fp_quant=False, act_fp_type=None,
                                    weight_fp_type=None, fc2_fp_type=None):
    """models_fp_quant/quant_utils.py:1256-1341: fc1 is E2M1 in blocks 6-20 and E3M0 elsewhere, mat_qkv E2M1 in
    blocks 0, 24, 25 and E3M0 elsewhere (activations; weights always E2M1); proj, fc2 and ada_lin[1] take the
    caller's formats."""
    fc1_e2, qkv_e2 = set(range(6, 21)), {0, 24, 25}

    def fmt(b, layer):
        if layer == "fc1":
            return ("fp_e2" if b in fc1_e2 else "fp_e3", "fp_e2")
        if layer == "mat_qkv":
            return ("fp_e2" if b in qkv_e2 else "fp_e3", "fp_e2")
        if layer == "fc2":
            return (fc2_fp_type, weight_fp_type)
        return (act_fp_type, weight_fp_type)

    return quantize_VAR_mixed(model, fmt, weight_quant, act_quant, w_bit, a_bit, act_quant_sym, fc2_act_log2_quant,
                              activation_fp_quant, weight_fp_quant, ada_lin_formats=(act_fp_type, weight_fp_type))


def quantize_VAR_use_different_datatype(model, weight_quant=None, act_quant=None, quantize_bmm_input=False, w_bit=8,
                                        a_bit=8, kv_bit=8, act_quant_sym=None, fc2_act_log2_quant=None, quant_kv=None,
                                        activation_fp_quant=False, weight_fp_quant=False, act_fp_type=None,
                                        weight_fp_type=None, fc2_fp_type=None):
    """models_fp_quant_rotate/quant_utils.py:982-1066: as the mixed FP4 variant, with mat_qkv E2M1 in blocks 24, 25 only."""
    fc1_e2, qkv_e2 = set(range(6, 21)), {24, 25}

    def fmt(b, layer):
        if layer == "fc1":
            return ("fp_e2" if b in fc1_e2 else "fp_e3", "fp_e2")
        if layer == "mat_qkv":
            return ("fp_e2" if b in qkv_e2 else "fp_e3", "fp_e2")
        if layer == "fc2":
            return (fc2_fp_type, weight_fp_type)
        return (act_fp_type, weight_fp_type)

    return quantize_VAR_mixed(model, fmt, weight_quant, act_quant, w_bit, a_bit, act_quant_sym, fc2_act_log2_quant,
                              activation_fp_quant, weight_fp_quant, ada_lin_formats=(act_fp_type, weight_fp_type))


def quantize_VAR_mixed_fp6_datatype(model, weight_quant=None, act_quant=None, quantize_bmm_input=False, w_bit=8,
                                    a_bit=8, kv_bit=8, act_quant_sym=None, fc2_act_log2_quant=None, quant_kv=None,
                                    activation_fp_quant=False, weight_fp_quant=False, act_fp_type=None,
                                    weight_fp_type=None, fc2_fp_type=None):
    """models_fp_quant/quant_utils.py:1344-1431: weights always E2M3; activations E3M2 for fc1 and mat_qkv, for fc2
    E2M3 in blocks 0 and 23 (E3M2 elsewhere), for proj E2M3 in blocks 2-29 (E3M2 in 0, 1); ada_lin[1] E2M3 / E2M3."""
    def fmt(b, layer):
        if layer in ("fc1", "mat_qkv"):
            return ("fp6_e3m2", "fp6_e2m3")
        if layer == "fc2":
            return ("fp6_e2m3" if b in (0, 23) else "fp6_e3m2", "fp6_e2m3")
        return ("fp6_e2m3" if 2 <= b <= 29 else "fp6_e3m2", "fp6_e2m3")

    return quantize_VAR_mixed(model, fmt, weight_quant, act_quant, w_bit, a_bit, act_quant_sym, fc2_act_log2_quant,
                              activation_fp_quant, weight_fp_quant, ada_lin_formats=("fp6_e2m3", "fp6_e2m3"))
