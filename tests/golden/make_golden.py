"""Generate tests/golden/*.npz by IMPORTING the reference's own Python.

Run in the build container only (needs /root/reference):

    python tests/golden/make_golden.py

The reference cannot run as shipped: `quant_cuda` is a CUDA extension (no nvcc
here) and `dist.py` / a top-level `quant_utils.py` are absent from its repo
(SURVEY.md section 0).  Two stub modules are injected before the import:

* ``quant_cuda.quant(x, y)`` = ``oracle.fpq_oracle.nearest_kernel`` - our
  restatement of quant/quant_kernel.cu:25-37 - plus the all-zero second output.
* ``dist`` with ``get_device``/``initialized``.

Everything else (scales, dtype promotion, reshapes, neg/pos split, clamp, the
argmin "CPU path", Hadamard construction) is executed by the reference code
itself.  The outputs are stored as raw bit patterns next to their inputs.
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")

from oracle import fpq_oracle as orc  # noqa: E402

_qc = types.ModuleType("quant_cuda")
_qc.quant = lambda x, y: (orc.nearest_kernel(x, y), torch.zeros_like(x))
sys.modules["quant_cuda"] = _qc
_dist = types.ModuleType("dist")
_dist.get_device = lambda: "cpu"
_dist.initialized = lambda: False
sys.modules["dist"] = _dist
sys.modules.setdefault("quant_utils", types.ModuleType("quant_utils"))

import models_fp_quant_transform_rotate.quant_utils as qu  # noqa: E402
import models_fp_quant_transform_rotate.basic_var as bv    # noqa: E402
import models_fp_quant.quant_utils as fqu                  # noqa: E402  (older variant: adds fp4_afpq)
from rotate_utils import hadamard_utils as hu              # noqa: E402
from rotate_utils import rotation_utils as ru              # noqa: E402


def bits(t: torch.Tensor) -> np.ndarray:
    t = t.detach().contiguous()
    if t.dtype == torch.float16:
        return t.view(torch.int16).numpy().view(np.uint16).copy()
    if t.dtype == torch.float32:
        return t.view(torch.int32).numpy().view(np.uint32).copy()
    if t.dtype == torch.float64:
        return t.view(torch.int64).numpy().view(np.uint64).copy()
    raise TypeError(t.dtype)


def make_inputs(kind: str, dtype, rows: int, cols: int, seed: int) -> torch.Tensor:
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(rows, cols, generator=g)
    if kind == "gauss":
        pass
    elif kind == "heavy":                       # pre-rotation-like outliers
        x = x * torch.exp(0.5 * torch.randn(rows, cols, generator=g))
    elif kind == "gelu":                        # fc2 input
        x = torch.nn.functional.gelu(x * 1.5, approximate="tanh")
    elif kind == "weights":
        x = x * 0.02
    elif kind == "edge":
        # group 0 all zero, group 1 all positive, group 2 all negative, group 3 one
        # huge outlier, group 4 tiny (scale underflows in fp16), group 5 exact ties
        x[0, 0:128] = 0.0
        x[0, 128:256] = x[0, 128:256].abs() + 0.01
        x[1, 0:128] = -x[1, 0:128].abs() - 0.01
        x[1, 128] = 1000.0
        x[2, 0:128] = x[2, 0:128] * 1e-7
        tie = torch.tensor([6.0, 0.25, -0.25, 0.75, -0.75, 1.25, -1.25, 1.75, -1.75, 2.5, -2.5,
                            3.5, -3.5, 5.0, -5.0, 0.0, -0.0, 6.0, -6.0, 4.5, 0.125, -0.125])
        x[2, 128:128 + tie.numel()] = tie
        x[3, 0:128] = x[3, 0:128] * 3e-8
    elif kind == "inf":                         # +-inf poison their group (0*inf), no NaN inputs
        x[0, 5] = float("inf")
        x[1, 130] = -float("inf")
        x[2, 7] = float("inf")
        x[2, 9] = -float("inf")
    elif kind == "nan":                         # NaN inputs (the global clip of A5 sees them too)
        x[0, 5] = float("nan")
        x[1, 130] = float("nan")
        x[3, 200] = float("inf")
    else:
        raise ValueError(kind)
    return x.to(dtype)


def main():
    out = {}
    # ---- 1. tables, literally as the reference spells them -----------------------
    ref_tables = {
        "e3m0": qu.fp4_e3m0_grid, "e2m1": qu.fp4_e2m1_grid, "e1m2": qu.fp4_e1m2_grid,
        "e2m3": qu.fp6_e2m3_grid, "e3m2": qu.fp6_e3m2_grid,
        "int_neg": qu.int_neg_grid, "e2m3_pos": qu.e2m3_pos_grid,
        "e1m2_neg": torch.tensor([-1.75, -1.5, -1.25, -1.0, -0.75, -0.5, -0.25, 0.0]),
        "e2m1_pos": torch.tensor([0.0, 0.5, 1.0, 1.5, 2.0, 3.0, 4.0, 6.0]),
        "e2m1_neg": torch.tensor([-6.0, -4.0, -3.0, -2.0, -1.5, -1.0, -0.5, 0.0]),   # fq/quant_utils.py:501
    }
    # the two 8-entry tables are locals of tr/quant_utils.py:418-419; check the
    # literals above against the function by feeding exact table values through it
    probe = torch.cat([ref_tables["e1m2_neg"] * (6.0 / 6.0), ref_tables["e2m1_pos"]])
    probe = torch.cat([probe, torch.zeros(128 - probe.numel())]).reshape(1, 128)
    probe[0, 0] = -1.75
    probe[0, 15] = 6.0
    got = qu.fp_quant_e1m2_neg_e2m1_pos_per_group_cuda(probe.clone(), 4, 128)
    assert torch.equal(got, probe), "dual-format literals do not match tr/quant_utils.py:418-419"
    for k, v in ref_tables.items():
        out[f"table/{k}"] = v.to(torch.float32).numpy().copy()

    # ---- 2. per-function vectors ---------------------------------------------------
    P = 4
    sym_group = {"e2m1": qu.fp_quant_e2_per_group_cuda, "e1m2": qu.fp_quant_e1_per_group_cuda,
                 "e3m0": qu.fp_quant_e3_per_group_cuda}
    fp6_group = {"e2m3": qu.fp6_quant_e2m3_per_group_cuda, "e3m2": qu.fp6_quant_e3m2_per_group_cuda}
    fp6_token = {"e2m3": qu.fp6_quant_e2m3_per_token_cuda, "e3m2": qu.fp6_quant_e3m2_per_token_cuda}
    cases = []
    seed = 100
    for dtype, dn in ((torch.float16, "f16"), (torch.float32, "f32")):
        for kind in ("gauss", "heavy", "edge", "weights", "gelu", "inf", "nan"):
            seed += 1
            x = make_inputs(kind, dtype, 8, 256, seed)
            out[f"in/{kind}_{dn}"] = bits(x)
            for name, fn in sym_group.items():
                cases.append((f"per_group_cuda/{name}/{kind}_{dn}", fn(x.clone(), P, 128)))
            for name, fn in fp6_group.items():
                cases.append((f"per_group_cuda/{name}/{kind}_{dn}", fn(x.clone(), 6, 128)))
            for name, fn in fp6_token.items():
                cases.append((f"per_token_cuda/{name}/{kind}_{dn}", fn(x.clone(), 6)))
            cases.append((f"dual_group_cuda/e1m2_neg+e2m1_pos/{kind}_{dn}",
                          qu.fp_quant_e1m2_neg_e2m1_pos_per_group_cuda(x.clone(), P, 128)))
            cases.append((f"dual_group_cuda_clip0.9/e1m2_neg+e2m1_pos/{kind}_{dn}",
                          qu.fp_quant_e1m2_neg_e2m1_pos_per_group_cuda(x.clone(), P, 128, 0.9)))
            cases.append((f"dual_group_cuda/int_neg+e2m3_pos/{kind}_{dn}",
                          qu.fp6_quant_int_neg_e2m3_pos_per_group_cuda(x.clone(), 6, 128)))
            cases.append((f"dual_token_cuda/int_neg+e2m3_pos/{kind}_{dn}",
                          qu.fp6_quant_int_neg_e2m3_pos_per_token_cuda(x.clone(), 6)))
            cases.append((f"dual_group_cuda/e2m1_neg+e2m1_pos/{kind}_{dn}",
                          fqu.fp4_afpq_per_group_cuda(x.clone(), P, 128)))
            cases.append((f"neg_reverse_group_cuda/e2m1/{kind}_{dn}",
                          fqu.fp_neg_reverse_quant_per_group_cuda(x.clone(), P, 128)))
            # the pure-torch CPU path (A9)
            cases.append((f"per_group_argmin/e2m1/{kind}_{dn}", qu.fp_quant_e2_per_group(x.clone(), P, 128)))
            cases.append((f"per_group_argmin/e1m2/{kind}_{dn}", qu.fp_quant_e1_per_group(x.clone(), P, 128)))
            cases.append((f"per_group_argmin/e3m0/{kind}_{dn}", qu.fp_quant_e3_per_group(x.clone(), P, 128)))
            cases.append((f"per_token_argmin/e2m1/{kind}_{dn}", qu.fp_quant_e2_per_token(x.clone(), P)))
            cases.append((f"per_token_argmin/e1m2/{kind}_{dn}", qu.fp_quant_e1_per_token(x.clone(), P)))
            cases.append((f"per_token_argmin/e3m0/{kind}_{dn}", qu.fp_quant_e3_per_token(x.clone(), P)))
            cases.append((f"dual_group_argmin/e1m2_neg+e2m1_pos/{kind}_{dn}",
                          qu.fp_quant_e1m2_neg_e2m1_pos_per_group(x.clone(), P, 128)))
            # KV-cache quantizers as basic_var.py spells them (tr/basic_var.py:50-87)
            cases.append((f"kv/e2m1_group/{kind}_{dn}", bv.fp_quant_e2_per_group_cuda(x.clone(), P)))
            kv = x.reshape(2, 4, 4, 64)
            cases.append((f"kv/e2m3_token64/{kind}_{dn}", bv.fp6_quant_e2m3_per_token_cuda(kv.clone(), 6)))
    for key, val in cases:
        out[f"out/{key}"] = bits(val)
        out[f"dtype/{key}"] = np.array(str(val.dtype))

    # ---- 3. config 1: per-tensor E2M1 via the argmin path ---------------------------
    # (search/baseline/plot_weight_distribution_for_motivation.py:286-297 is a script
    #  with top-level side effects; its 4-line body is executed here through the
    #  reference's quantize_to_nearest_grid)
    x = make_inputs("gauss", torch.float32, 8, 256, 7)
    out["in/per_tensor_f32"] = bits(x)
    grid = qu.fp4_e2m1_grid
    scale = x.abs().max() / grid.abs().max()
    out["out/per_tensor_argmin/e2m1"] = bits(qu.quantize_to_nearest_grid(x / scale, grid) * scale)

    # ---- 3b. QuantizedLinear / QuantizedLinear_fc2 / quantize_VAR through the reference classes --
    class Toy(torch.nn.Module):
        def __init__(self):
            super().__init__()
            torch.manual_seed(5)
            self.ffn = bv.FFN(in_features=128, hidden_features=256, fused_if_available=False)
            self.attn = bv.SelfAttention(block_idx=0, embed_dim=128, num_heads=2, flash_if_available=False)

    run_cfgs = {
        # README.md:33 / run.sh:4  (W4A4)
        "w4a4": dict(weight_quant="per_group", act_quant="per_group", w_bit=4, a_bit=4, act_quant_sym=True,
                     activation_fp_quant=True, weight_fp_quant=True, act_fp_type="fp_e2", weight_fp_type="fp_e2",
                     fc2_fp_type="fp_e1m2_neg_e2m1_pos"),
        # run.sh:7 (W6A6)
        "w6a6": dict(weight_quant="per_channel", act_quant="per_token", w_bit=6, a_bit=6, act_quant_sym=True,
                     activation_fp_quant=True, weight_fp_quant=True, act_fp_type="fp6_e2m3",
                     weight_fp_type="fp6_e2m3", fc2_fp_type="fp6_int_neg_e2m3_pos"),
        # per_channel / per_token FP4: the pure-torch (argmin) functions even for "GPU" runs
        "w4a4_tok": dict(weight_quant="per_channel", act_quant="per_token", w_bit=4, a_bit=4, act_quant_sym=True,
                         activation_fp_quant=True, weight_fp_quant=True, act_fp_type="fp_e2",
                         weight_fp_type="fp_e2", fc2_fp_type="fp_e3"),
    }
    gx = torch.Generator().manual_seed(77)
    xin = torch.randn(6, 128, generator=gx)
    hid = torch.nn.functional.gelu(torch.randn(6, 256, generator=gx) * 1.5, approximate="tanh")
    out["ql/x_f32"] = bits(xin)
    out["ql/h_f32"] = bits(hid)
    toy0 = Toy()
    for lname in ("ffn.fc1", "ffn.fc2", "attn.mat_qkv", "attn.proj"):
        mod = dict(toy0.named_modules())[lname]
        out[f"ql/w0/{lname}"] = bits(mod.weight.detach())
        if mod.bias is not None:
            out[f"ql/b0/{lname}"] = bits(mod.bias.detach())
    for cname, cfg in run_cfgs.items():
        toy = Toy()
        qu.quantize_VAR(toy, **cfg)
        for lname in ("ffn.fc1", "ffn.fc2", "attn.mat_qkv", "attn.proj"):
            mod = dict(toy.named_modules())[lname]
            out[f"ql/{cname}/weight/{lname}"] = bits(mod.weight.detach())
            out[f"ql/{cname}/class/{lname}"] = np.array(type(mod).__name__)
            src = hid if lname == "ffn.fc2" else xin
            for dtype, dn in ((torch.float16, "f16"), (torch.float32, "f32")):
                out[f"ql/{cname}/act/{lname}/{dn}"] = bits(mod.act_quant(src.to(dtype).clone()))
            # forward as the driver runs it: module cast to half, fp16 input (CPU fp16 linear here;
            # only compared approximately, GEMM accumulation order is not a contract)
            mod.weight = mod.weight.half()
            if mod.bias is not None:
                mod.bias = torch.nn.Parameter(mod.bias.detach().half(), requires_grad=False)
            with torch.autocast('cpu', dtype=torch.float16):   # the driver runs generation under fp16 autocast
                y = mod.forward(src.half().clone())
            out[f"ql/{cname}/fwd_f32/{lname}"] = bits(y.float())

    # ---- 4. rotation pieces ----------------------------------------------------------
    q128 = hu.random_hadamard_matrix(128, "cpu", 42)            # fp64
    out["rot/q128_f64"] = q128.numpy().copy()
    qb = ru.block_random_hadamard_matrix(1920, 128, "cpu", 42) if hasattr(ru, "block_random_hadamard_matrix") else None
    if qb is not None:
        blocks_equal = all(torch.equal(qb[i * 128:(i + 1) * 128, i * 128:(i + 1) * 128], qb[:128, :128])
                           for i in range(15))
        out["rot/blocks_identical_1920"] = np.array(blocks_equal)
        out["rot/block0_equals_q128"] = np.array(torch.equal(qb[:128, :128].to(torch.float64), q128))
        off = qb.clone()
        for i in range(15):
            off[i * 128:(i + 1) * 128, i * 128:(i + 1) * 128] = 0
        out["rot/offdiag_zero_1920"] = np.array(bool((off == 0).all()))

    # ---- 5. GALT objective and its STE gradient (learnable_transformation/*.py) --------------------
    import learnable_transformation.learnable_transformation_mat_qkv_fp4 as g4   # noqa: E402
    import learnable_transformation.learnable_transformation_mat_qkv_fp6 as g6   # noqa: E402
    gg = torch.Generator().manual_seed(91)
    gx_ = torch.randn(64, 256, generator=gg) * torch.exp(0.5 * torch.randn(64, 256, generator=gg))
    gw_ = torch.randn(48, 256, generator=gg) * 0.05
    gs_ = torch.rand(256, generator=gg) + 0.5
    gq_ = ru.block_random_hadamard_matrix(256, 128, "cpu", 42).to(torch.float32)
    out["galt/x_f32"], out["galt/w_f32"], out["galt/s_f32"], out["galt/q_f32"] = bits(gx_), bits(gw_), bits(gs_), bits(gq_)
    for tag, mod in (("fp4", g4), ("fp6", g6)):
        sp = torch.nn.Parameter(gs_.clone())
        loss = mod.compute_quant_error_v1(gx_, gw_, sp, gq_)
        loss.backward()
        out[f"galt/{tag}/loss"] = bits(loss.detach().float().reshape(1))
        out[f"galt/{tag}/grad_s"] = bits(sp.grad.float())
        # the quantizers' forward on the exact transformed operands (bit-level parity target)
        x2 = torch.matmul(gx_ * gs_, gq_)
        w2 = torch.matmul(gw_ / gs_, gq_)
        out[f"galt/{tag}/x2_f32"], out[f"galt/{tag}/w2_f32"] = bits(x2), bits(w2)
        if tag == "fp4":
            out["galt/fp4/x2_quant"] = bits(mod.FPQuant.apply(x2))
            out["galt/fp4/w2_quant"] = bits(mod.FPQuant.apply(w2))
        else:
            out["galt/fp6/x2_quant"] = bits(mod.FP6Quant_activation_per_token.apply(x2))
            out["galt/fp6/w2_quant"] = bits(mod.FP6Quant_weight.apply(w2))
            out["galt/fp6/x2_quant_group"] = bits(mod.FP6Quant_activation.apply(x2))

    # ---- 6. model-level preprocessing: transform_model then rotate_model (block mode) on a toy model -------------
    import learnable_transformation.transform_model_utils as tmu   # noqa: E402

    class _Blk(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.attn, self.ffn = torch.nn.Module(), torch.nn.Module()
            self.attn.mat_qkv = torch.nn.Linear(128, 384, bias=False)
            self.ffn.fc1 = torch.nn.Linear(128, 64)

    class _ToyVAR(torch.nn.Module):
        def __init__(self):
            super().__init__()
            torch.manual_seed(21)
            self.C = 128
            self.blocks = torch.nn.ModuleList([_Blk() for _ in range(2)])

    toy = _ToyVAR()
    gsm = torch.Generator().manual_seed(22)
    s_qkv = [torch.rand(128, generator=gsm) + 0.5 for _ in range(2)]
    s_fc1 = [torch.rand(128, generator=gsm) + 0.5 for _ in range(2)]
    for i in range(2):
        out[f"prep/w0/qkv{i}"], out[f"prep/w0/fc1{i}"] = bits(toy.blocks[i].attn.mat_qkv.weight.detach()), bits(toy.blocks[i].ffn.fc1.weight.detach())
        out[f"prep/s/qkv{i}"], out[f"prep/s/fc1{i}"] = bits(s_qkv[i]), bits(s_fc1[i])
    tmu.transform_model(toy, s_qkv, s_fc1)
    ru.rotate_model(toy, "cpu", True)
    for i in range(2):
        out[f"prep/w1/qkv{i}"], out[f"prep/w1/fc1{i}"] = bits(toy.blocks[i].attn.mat_qkv.weight.detach()), bits(toy.blocks[i].ffn.fc1.weight.detach())

    # ---- 7. full-width (non-block) rotation: the literal Hadamard tables, Q for VAR's widths, rotate_model(False) ----
    import hashlib
    for k in (12, 20, 28, 36, 40, 60, 108, 140, 52, 156, 172):
        had, kk = hu.get_hadK(k)
        assert kk == k
        out[f"had/table/{k}"] = had.numpy().astype(np.int8)
    for n in (1280, 1536, 1920, 2304):                     # VAR-d20 / d24 / d30 / d36
        q = hu.random_hadamard_matrix(n, "cpu", 42)
        out[f"had/q_sha256/{n}"] = np.frombuffer(hashlib.sha256(q.numpy().tobytes()).digest(), dtype=np.uint8).copy()
        out[f"had/q_corner/{n}"] = q[:3, :64].numpy().copy()
        out[f"had/q_lastrow/{n}"] = q[-1].numpy().copy()

    class _BlkF(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.attn, self.ffn = torch.nn.Module(), torch.nn.Module()
            self.attn.mat_qkv = torch.nn.Linear(240, 72, bias=False)
            self.ffn.fc1 = torch.nn.Linear(240, 40)

    class _ToyVARF(torch.nn.Module):
        def __init__(self):
            super().__init__()
            torch.manual_seed(23)
            self.C = 240                                     # 60 * 4: the had60 branch of get_hadK
            self.blocks = torch.nn.ModuleList([_BlkF() for _ in range(2)])

    toyf = _ToyVARF()
    for i in range(2):
        out[f"prep_full/w0/qkv{i}"], out[f"prep_full/w0/fc1{i}"] = bits(toyf.blocks[i].attn.mat_qkv.weight.detach()), bits(toyf.blocks[i].ffn.fc1.weight.detach())
    ru.rotate_model(toyf, "cpu", False)
    for i in range(2):
        out[f"prep_full/w1/qkv{i}"], out[f"prep_full/w1/fc1{i}"] = bits(toyf.blocks[i].attn.mat_qkv.weight.detach()), bits(toyf.blocks[i].ffn.fc1.weight.detach())

    np.savez_compressed(os.path.join(HERE, "reference_vectors.npz"), **out)
    print("wrote", os.path.join(HERE, "reference_vectors.npz"), len(out), "arrays")


if __name__ == "__main__":
    main()
