#!/usr/bin/env python3
"""FP4 GEMM kernel only, mat_qkv shape, per tile configuration (the library switch FPQ_GEMM_CFG, set through fpq_set_option)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from fpqvar_amd import _lib, gemm  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
T, K, O = 65536, 1920, 5760
x = torch.randn(T, K, device=dev).half()
w = torch.randn(O, K, device=dev) * 0.02
ac, asc = gemm.quantize_mx(x)
wc, wsc = gemm.quantize_mx(w)
ref = None
for cfg in (sys.argv[1:] or ["0", "3", "4", "5"]):
    _lib.set_option("FPQ_GEMM_CFG", int(cfg))
    y = gemm.linear_fp4(ac, asc, wc, wsc)
    if ref is None:
        ref = y
    same = float((y.float() - ref.float()).abs().max() / ref.float().abs().max())
    for _ in range(20):
        gemm.linear_fp4(ac, asc, wc, wsc)
    torch.cuda.synchronize()
    ms = float("inf")
    for _ in range(3):          # best of three bursts (the clock state of the box moves single bursts by 5-10 %)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            gemm.linear_fp4(ac, asc, wc, wsc)
        e1.record()
        torch.cuda.synchronize()
        ms = min(ms, e0.elapsed_time(e1) / 20)
    print(f"cfg {cfg}: {ms:.4f} ms  {2.0 * T * K * O / ms / 1e9:.0f} TFLOP/s  max_rel_diff_vs_first={same:.2e}", flush=True)
