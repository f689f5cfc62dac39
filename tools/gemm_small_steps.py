#!/usr/bin/env python3
"""FP4 GEMM at the token counts of the ten scale steps (VAR-d30: 2 x 50 x pn^2 rows; the three Linears fed by per-group FP4
activations), per tile configuration (the library switches FPQ_GEMM_CFG / FPQ_GEMM6_CFG / FPQ_GEMM8_CFG, set through fpq_set_option).
usage: gemm_small_steps.py [fp4|fp6|fp8] [kmajor] [cfg ...]      kmajor: the operands as k-major images (include/fpq.h; fp4 / fp6)"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fpqvar_amd import _lib, gemm

kind = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1] in ("fp4", "fp6", "fp8") else "fp4"
KM = "kmajor" in sys.argv[1:]
cfgs = [a for a in sys.argv[1:] if a not in ("fp4", "fp6", "fp8", "kmajor")] or {"fp4": ["default", "20", "30"], "fp6": ["default", "0", "1"], "fp8": ["default", "0", "1"]}[kind]
quant, linear, env = {"fp4": (gemm.quantize_mx, gemm.linear_fp4, "FPQ_GEMM_CFG"), "fp6": (gemm.quantize_fp6, gemm.linear_fp6, "FPQ_GEMM6_CFG"),
                      "fp8": (gemm.quantize_fp8, gemm.linear_fp8, "FPQ_GEMM8_CFG")}[kind]
dev = torch.device("cuda:0")
torch.manual_seed(0)
K = 1920
tot = {c: 0.0 for c in cfgs}
for O in (5760, 1920, 7680):
    w = torch.randn(O, K, device=dev) * 0.02
    wc, wsc = quant(w)
    if KM:
        wc = gemm.to_kmajor(wc, 4 if kind == "fp4" else 6, dealt=True)
        if kind == "fp4":
            wsc = gemm.to_kmajor_scales(wsc, weight_side=True)
    for pn in (1, 2, 3, 4, 5, 6, 8, 10, 13, 16):
        T = 100 * pn * pn
        x = torch.randn(T, K, device=dev).half()
        ac, asc = quant(x, kmajor=True) if KM else quant(x)
        row = []
        for c in cfgs:
            _lib.set_option(env, None if c == "default" else int(c))
            for _ in range(5):
                linear(ac, asc, wc, wsc)
            torch.cuda.synchronize()
            best = 1e9
            for _ in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20):
                    linear(ac, asc, wc, wsc)
                e1.record()
                torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / 20 * 1e3)
            row.append(best)
            tot[c] += best
        print(f"O={O:5d} T={T:6d} " + "  ".join(f"{c}: {t:7.1f} us" for c, t in zip(cfgs, row)), flush=True)
print("sum over the thirty calls: " + "  ".join(f"{c}: {t:8.1f} us" for c, t in tot.items()))
