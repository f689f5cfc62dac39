#!/usr/bin/env python3
"""One W4A4 AdaLN transformer block of VAR-d30 (C=1920, 30 heads, hidden 7680; tr/basic_var.py:253-269),
random weights, three implementations of everything around the attention core:

  R  the reference's op sequence on this GPU: LayerNorm / modulate / smooth as torch ops, dense fp16
     GEMM with the block-diagonal Q, the ~11-op fake-quant bodies around the scan kernel, fp16 Linears
  F  this repo's fused fake-quant path: adaln_rotate_quant + fused per-group / dual quantizers,
     fp16 Linears on the de-quantized tensors (same arithmetic contract as the reference)
  Q  as F but mat_qkv / proj / fc1 run on the FP4 matrix cores (FP4Linear)

Reports the time per block and the agreement of F and Q with R - and attributes the F / R difference: at the two
producer sites the rotation is also computed EXACTLY (the fp64 product of the same fp16 operands, rounded to fp16 once),
and a block E runs on those exact rotations.  F-vs-E and R-vs-E say whose rounding moves the result: the fused kernels'
(one fp32 accumulation + one rounding) or the reference GEMM's (fp16 tensor-core GEMM with c_h folded into the operands).
Synthetic data; no KV cache (one scale step of `rows` = B*L tokens).
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.nn.functional as Fn  # noqa: E402

from fpqvar_amd import gemm, ops, quant_utils as qu, rotation as rot  # noqa: E402

C, H, HID = 1920, 30, 7680


def unfused_sym(x, grid):
    xs = x.reshape(-1, 128)
    scale = xs.abs().max(dim=-1, keepdim=True)[0] / grid.abs().max()
    xn = (xs / scale).view(-1).to(torch.float32)
    z = ops.quant_nearest(xn, grid)
    torch.zeros_like(xn)
    return (z.view(xs.shape) * scale).view(x.shape).to(x.dtype)


def unfused_dual(x, gneg, gpos):
    clip = 1.0 * x.abs().max()
    x = torch.clamp(x, -clip, clip)
    xs = x.reshape(-1, 128)
    zeros = torch.zeros_like(xs)
    xn_, xp_ = torch.where(xs <= 0, xs, zeros), torch.where(xs > 0, xs, zeros)
    sn = xn_.abs().max(dim=-1, keepdim=True)[0] / gneg.abs().max()
    sp = xp_.abs().max(dim=-1, keepdim=True)[0] / gpos.abs().max()
    a = (xn_ / sn).view(-1).to(torch.float32)
    b = (xp_ / sp).view(-1).to(torch.float32)
    qa, qb = ops.quant_nearest(a, gneg), ops.quant_nearest(b, gpos)
    torch.zeros_like(a), torch.zeros_like(b)
    return (qa.view(xs.shape) * sn + qb.view(xs.shape) * sp).view(x.shape).to(x.dtype)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=100)
    ap.add_argument("--tokens", type=int, default=256)
    ap.add_argument("--residual", default="fp32", choices=("fp32", "fp16"),
                    help="dtype of the residual stream: fp32 is what the reference's autocast run carries (tr/var.py:209 adds an "
                         "fp32 position embedding to the fp16 word embedding; tr/basic_var.py:264,267 keep x fp32)")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    B, L = args.batch, args.tokens
    x = torch.randn(B, L, C, device=dev)
    if args.residual == "fp16":
        x = x.half()
    mod = [(torch.randn(B, 1, C, device=dev) * 0.2).half() for _ in range(6)]   # gamma1, gamma2, scale1, scale2, shift1, shift2
    gamma1, gamma2, scale1, scale2, shift1, shift2 = mod
    s_qkv = torch.rand(C, device=dev) + 0.5
    s_fc1 = torch.rand(C, device=dev) + 0.5
    q64 = rot.block_random_hadamard_matrix(C, 128, dev, 42)
    q32 = q64.float()

    def lin_w(o, i, smooth=None, rotate=False):
        w = torch.randn(o, i, device=dev) * 0.02
        if smooth is not None:
            w = rot.transform_weight(w, smooth)
        if rotate:
            w = rot.rotate_weight(w, q64)
        return w

    w_qkv, w_proj = lin_w(3 * C, C, s_qkv, True), lin_w(C, C)
    w_fc1, w_fc2 = lin_w(HID, C, s_fc1, True), lin_w(C, HID)
    wq = {n: qu.fp_quant_e2_per_group_cuda(w, 4, 128).half() for n, w in
          (("qkv", w_qkv), ("proj", w_proj), ("fc1", w_fc1), ("fc2", w_fc2))}
    fp4 = {n: gemm.quantize_mx(w) for n, w in (("qkv", w_qkv), ("proj", w_proj), ("fc1", w_fc1))}
    grid = qu.fp4_e2m1_grid.to(dev)
    gneg = torch.tensor([-1.75, -1.5, -1.25, -1.0, -0.75, -0.5, -0.25, 0.0], device=dev)
    gpos = torch.tensor([0.0, 0.5, 1.0, 1.5, 2.0, 3.0, 4.0, 6.0], device=dev)

    def attention(qkv):
        qkv = qkv.view(B, L, 3, H, C // H)
        q, k, v = qkv.unbind(2)
        o = Fn.scaled_dot_product_attention(q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2))
        return o.transpose(1, 2).reshape(B, L, C)

    def block_ref(x):
        with torch.autocast("cuda", dtype=torch.float16):
            x1 = torch.matmul(Fn.layer_norm(x, (C,), eps=1e-6).mul(scale1.add(1)).add_(shift1).mul(s_qkv), q32)
            a = attention(Fn.linear(unfused_sym(x1, grid), wq["qkv"]))
            a = Fn.linear(unfused_sym(a, grid), wq["proj"])
            x = x + a.mul(gamma1)
            x2 = torch.matmul(Fn.layer_norm(x, (C,), eps=1e-6).mul(scale2.add(1)).add_(shift2).mul(s_fc1), q32)
            h = Fn.gelu(Fn.linear(unfused_sym(x2, grid), wq["fc1"]), approximate="tanh")
            f = Fn.linear(unfused_dual(h, gneg, gpos), wq["fc2"])
            return x + f.mul(gamma2)

    def block_fused(x):
        a = attention(Fn.linear(rot.adaln_rotate_quant(x, scale1, shift1, "e2m1", smooth=s_qkv), wq["qkv"]))
        a = Fn.linear(qu.fp_quant_e2_per_group_cuda(a, 4, 128), wq["proj"])
        x = x + a.mul(gamma1)
        h = Fn.gelu(Fn.linear(rot.adaln_rotate_quant(x, scale2, shift2, "e2m1", smooth=s_fc1), wq["fc1"]), approximate="tanh")
        f = Fn.linear(qu.fp_quant_e1m2_neg_e2m1_pos_per_group_cuda(h, 4, 128), wq["fc2"])
        return x + f.mul(gamma2)

    def fp4_linear(y, name):
        ac, asc = gemm.quantize_mx(y.reshape(-1, y.shape[-1]))
        return gemm.linear_fp4(ac, asc, *fp4[name]).view(*y.shape[:-1], -1)

    def block_fp4(x):
        ac, asc = rot.adaln_rotate_quant_mx(x, scale1, shift1, smooth=s_qkv)      # producer -> GEMM operands, one launch
        a = attention(gemm.linear_fp4(ac, asc, *fp4["qkv"]).view(B, L, 3 * C))
        a = fp4_linear(a, "proj")
        x = x + a.mul(gamma1)
        ac, asc = rot.adaln_rotate_quant_mx(x, scale2, shift2, smooth=s_fc1)
        h = Fn.gelu(gemm.linear_fp4(ac, asc, *fp4["fc1"]).view(B, L, HID), approximate="tanh")
        f = Fn.linear(qu.fp_quant_e1m2_neg_e2m1_pos_per_group_cuda(h, 4, 128), wq["fc2"])
        return x + f.mul(gamma2)

    q16d = q32.half().double()          # the operand the reference's autocast GEMM sees: half(Q)

    def exact_rotate(h16):
        """half( exact product of the fp16 operands ): what both implementations approximate"""
        return (h16.double().reshape(-1, C) @ q16d).half().view(h16.shape)

    def ref_modulated(x, scale, shift, s):
        with torch.autocast("cuda", dtype=torch.float16):
            return Fn.layer_norm(x, (C,), eps=1e-6).mul(scale.add(1)).add_(shift).mul(s).half()   # the cast autocast applies in front of the GEMM

    def block_exact(x):
        """F's structure with the rotation replaced by the exact one (fp64 GEMM, one rounding), same quantizers and Linears"""
        _, h1, _ = rot.adaln_rotate_quant(x, scale1, shift1, "e2m1", smooth=s_qkv, return_intermediates=True)
        a = attention(Fn.linear(qu.fp_quant_e2_per_group_cuda(exact_rotate(h1), 4, 128), wq["qkv"]))
        a = Fn.linear(qu.fp_quant_e2_per_group_cuda(a, 4, 128), wq["proj"])
        x = x + a.mul(gamma1)
        _, h2, _ = rot.adaln_rotate_quant(x, scale2, shift2, "e2m1", smooth=s_fc1, return_intermediates=True)
        h = Fn.gelu(Fn.linear(qu.fp_quant_e2_per_group_cuda(exact_rotate(h2), 4, 128), wq["fc1"]), approximate="tanh")
        f = Fn.linear(qu.fp_quant_e1m2_neg_e2m1_pos_per_group_cuda(h, 4, 128), wq["fc2"])
        return x + f.mul(gamma2)

    def site_report(x, scale, shift, s):
        """rotated values and their E2M1 codes at one producer site: each implementation against the exact rotation of ITS OWN
        modulated rows (so only the rotation + rounding is compared), and the two modulated rows against each other"""
        out_f, h_f, y_f = rot.adaln_rotate_quant(x, scale, shift, "e2m1", smooth=s, return_intermediates=True)
        h_r = ref_modulated(x, scale, shift, s)
        with torch.autocast("cuda", dtype=torch.float16):
            y_r = torch.matmul(h_r, q32)
        e_f, e_r = exact_rotate(h_f), exact_rotate(h_r)

        def neq(a, b):
            return float((a.view(torch.int16) != b.view(torch.int16)).float().mean())

        def codes_neq(a, b):   # the fake-quantized values stand for the codes (same scale <=> same code)
            qa, qb = qu.fp_quant_e2_per_group_cuda(a, 4, 128), qu.fp_quant_e2_per_group_cuda(b, 4, 128)
            return float((qa.view(torch.int16) != qb.view(torch.int16)).float().mean())
        ulp = lambda a, b: float(((a.float() - b.float()).abs() / (b.float().abs().clamp_min(2.0 ** -14).log2().floor().exp2() * 2.0 ** -10)).max())
        return {"modulated_rows_F_vs_R_differing": round(neq(h_f, h_r), 6),
                "rotated_F_vs_exact_differing": round(neq(y_f, e_f), 6), "rotated_F_vs_exact_max_ulp": round(ulp(y_f, e_f), 2),
                "rotated_R_vs_exact_differing": round(neq(y_r, e_r), 6), "rotated_R_vs_exact_max_ulp": round(ulp(y_r, e_r), 2),
                "quantized_F_vs_quantized_exact_differing": round(codes_neq(y_f, e_f), 6),
                "quantized_R_vs_quantized_exact_differing": round(codes_neq(y_r, e_r), 6),
                "quantized_F_vs_quantized_R_differing": round(codes_neq(y_f, y_r), 6)}

    def timed(fn, n=5):
        fn(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn(x)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    yr, yf, yq, ye = block_ref(x).float(), block_fused(x).float(), block_fp4(x).float(), block_exact(x).float()
    delta = (yr - x.float())

    def rel(a, b=None):
        return float((a - (yr if b is None else b)).norm() / delta.norm())     # error relative to what the block adds to the residual

    res = {"rows": B * L, "residual": args.residual, "R_reference_sequence_ms": round(timed(block_ref, 3), 3), "F_fused_fake_quant_ms": round(timed(block_fused), 3),
           "Q_fp4_matrix_cores_ms": round(timed(block_fp4), 3), "F_vs_R_rel_err_of_block_update": round(rel(yf), 5),
           "Q_vs_R_rel_err_of_block_update": round(rel(yq), 5),
           # attribution: E = the block on EXACT rotations (fp64 product of the same fp16 operands, one rounding)
           "F_vs_E_rel_err_of_block_update": round(rel(yf, ye), 5), "R_vs_E_rel_err_of_block_update": round(rel(yr, ye), 5),
           "site_mat_qkv_input": site_report(x, scale1, shift1, s_qkv),
           "site_fc1_input": site_report(x, scale2, shift2, s_fc1)}
    # the same block without any quantization, for scale
    def block_fp16(x):
        with torch.autocast("cuda", dtype=torch.float16):
            x1 = torch.matmul(Fn.layer_norm(x, (C,), eps=1e-6).mul(scale1.add(1)).add_(shift1).mul(s_qkv), q32)
            a = Fn.linear(attention(Fn.linear(x1, wq["qkv"])), wq["proj"])
            x = x + a.mul(gamma1)
            x2 = torch.matmul(Fn.layer_norm(x, (C,), eps=1e-6).mul(scale2.add(1)).add_(shift2).mul(s_fc1), q32)
            return x + Fn.linear(Fn.gelu(Fn.linear(x2, wq["fc1"]), approximate="tanh"), wq["fc2"]).mul(gamma2)
    res["unquantized_act_fp16_block_ms"] = round(timed(block_fp16), 3)
    res["W4A4_noise_R_vs_unquantized_act_rel"] = round(rel(block_fp16(x).float()), 5)   # scale for the two errors above
    print(json.dumps(res))


if __name__ == "__main__":
    main()
