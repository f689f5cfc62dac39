#!/usr/bin/env python3
"""Row-major operand codes against k-major images (include/fpq.h), piece by piece, at the row counts of a VAR-d30 generation
batch (B = 100 rows per token, C = 1920): the three activation producers and the GEMMs they feed, us per launch (best of 5
alternating bursts of 20).  usage: ab_kmajor.py [w4a4|w6a6]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from fpqvar_amd import gemm, rotation as rot  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "w4a4"
W6 = cfg == "w6a6"
dev = torch.device("cuda:0")
torch.manual_seed(0)
B, C = 100, 1920
PN = (1, 2, 3, 4, 5, 6, 8, 10, 13, 16)


def burst(fn, n=20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def ab(f_rows, f_km):
    best = [1e9, 1e9]
    for _ in range(5):
        for i, f in enumerate((f_rows, f_km)):
            f()
            best[i] = min(best[i], burst(f))
    return best


bits = 6 if W6 else 4
quant = gemm.quantize_fp6 if W6 else gemm.quantize_mx
lin = gemm.linear_fp6 if W6 else gemm.linear_fp4
w = {n: quant(torch.randn(o, C, device=dev) * 0.02) for n, o in (("qkv", 3 * C), ("proj", C), ("fc1", 4 * C))}
wk = {n: (gemm.to_kmajor(c, bits, dealt=True), s if W6 else gemm.to_kmajor_scales(s, weight_side=True)) for n, (c, s) in w.items()}
sm = torch.rand(C, device=dev) + 0.5
print(f"# {cfg}: us per launch, row-major / k-major (ratio); d30 batch of {B} rows per token")
print("# tokens   adaLN producer          quantizer (proj input)   GEMM qkv                GEMM proj               GEMM fc1" + ("" if W6 else " (+GELU+dual)"))
tot = {k: [0.0, 0.0] for k in ("prod", "quant", "qkv", "proj", "fc1")}
for pn in PN:
    L = pn * pn
    x = torch.randn(B, L, C, device=dev).half()
    sc = (torch.randn(B, 1, C, device=dev) * 0.2).half()
    a2 = torch.randn(B * L, C, device=dev).half()
    if W6:
        prod = lambda km: rot.adaln_rotate_quant_token(x, sc, sc, "e2m3", smooth=sm, emit="fp6", kmajor=km)
    else:
        prod = lambda km: rot.adaln_rotate_quant_mx(x, sc, sc, smooth=sm, kmajor=km)
    r = {"prod": ab(lambda: prod(False), lambda: prod(True)), "quant": ab(lambda: quant(a2), lambda: quant(a2, kmajor=True))}
    ar, ak = prod(False), prod(True)
    r["qkv"] = ab(lambda: lin(*ar, *w["qkv"]), lambda: lin(*ak, *wk["qkv"]))
    r["proj"] = ab(lambda: lin(*ar, *w["proj"]), lambda: lin(*ak, *wk["proj"]))
    if W6:
        r["fc1"] = ab(lambda: lin(*ar, *w["fc1"]), lambda: lin(*ak, *wk["fc1"]))
    else:
        r["fc1"] = ab(lambda: gemm.linear_fp4_gelu_dual(*ar, *w["fc1"]), lambda: gemm.linear_fp4_gelu_dual(*ak, *wk["fc1"]))
    for k in tot:
        tot[k][0] += r[k][0]
        tot[k][1] += r[k][1]
    print(f"{B * L:7d}   " + "   ".join(f"{r[k][0]:7.1f} /{r[k][1]:7.1f} ({r[k][0] / r[k][1]:.2f})" for k in ("prod", "quant", "qkv", "proj", "fc1")))
print("    sum   " + "   ".join(f"{tot[k][0]:7.1f} /{tot[k][1]:7.1f} ({tot[k][0] / tot[k][1]:.2f})" for k in ("prod", "quant", "qkv", "proj", "fc1")))
per_block = [2 * tot["prod"][i] + tot["quant"][i] + tot["qkv"][i] + tot["proj"][i] + tot["fc1"][i] for i in (0, 1)]
print(f"# per block (two adaLN producers, one quantizer, three GEMMs): {per_block[0]:.0f} -> {per_block[1]:.0f} us; x 30 blocks: {per_block[0] * 30 / 1e3:.1f} -> {per_block[1] * 30 / 1e3:.1f} ms")
