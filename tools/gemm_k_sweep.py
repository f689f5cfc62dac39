#!/usr/bin/env python3
"""FP4 GEMM time against the number of 128-groups (K / 128) at fixed [tokens x outs]: time = fixed part per tile
(prologue + epilogue) + groups x per-group part.  usage: gemm_k_sweep.py [fp4|fp6|fp8] [tokens outs]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fpqvar_amd import gemm

kind = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1] in ("fp4", "fp6", "fp8") else "fp4"
rest = [a for a in sys.argv[1:] if a not in ("fp4", "fp6", "fp8")]
T, O = (int(a) for a in rest[:2]) if len(rest) > 1 else (65536, 5760)
quant, linear = {"fp4": (gemm.quantize_mx, gemm.linear_fp4), "fp6": (gemm.quantize_fp6, gemm.linear_fp6),
                 "fp8": (gemm.quantize_fp8, gemm.linear_fp8)}[kind]
dev = torch.device("cuda:0")
torch.manual_seed(0)
rows = []
for K in (640, 1280, 1920, 2560, 3840):
    x = torch.randn(T, K, device=dev).half()
    w = torch.randn(O, K, device=dev) * 0.02
    ac, asc = quant(x)
    wc, wsc = quant(w)
    del x, w
    for _ in range(5):
        linear(ac, asc, wc, wsc)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            linear(ac, asc, wc, wsc)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 10)
    rows.append((K // 128, best))
    print(f"{kind} K={K:5d} groups={K // 128:3d}  {best * 1e3:8.1f} us  {2 * T * O * K / best / 1e12:7.3f} PFLOP/s", flush=True)
n = len(rows)
sx = sum(g for g, _ in rows); sy = sum(t for _, t in rows)
sxx = sum(g * g for g, _ in rows); sxy = sum(g * t for g, t in rows)
b = (n * sxy - sx * sy) / (n * sxx - sx * sx)
a = (sy - b * sx) / n
print(f"fit: {a * 1e3:.1f} us fixed + {b * 1e3:.2f} us per group; per-group part alone = {2 * T * O * 128 / b / 1e12:.3f} PFLOP/s; "
      f"fixed part = {a / b:.1f} groups' worth")
