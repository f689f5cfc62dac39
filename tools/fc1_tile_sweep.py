#!/usr/bin/env python3
"""The fused fc1 GEMM (gemm.linear_fp4_gelu_dual) at the token counts of the ten scale steps per LDS-DMA tiling (library switch
FPQ_GEMM_CFG: 30 = 64 x 128, 20 = 128 x 128, 10 = 256 x 128 tiles) - what the default selection in fpq_gemm_fp4_gelu_dual rests on.
usage: fc1_tile_sweep.py [d30|d36] [kmajor]      kmajor: operands as k-major images (include/fpq.h)"""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from fpqvar_amd import _lib, gemm  # noqa: E402

KM = "kmajor" in sys.argv[1:]
model = ([a for a in sys.argv[1:] if a != "kmajor"] or ["d30"])[0]
C, B, pns = (1920, 100, (1, 2, 3, 4, 5, 6, 8, 10, 13, 16)) if model == "d30" else (2304, 20, (1, 2, 3, 4, 6, 9, 13, 18, 24, 32))
dev = torch.device("cuda:0")
torch.manual_seed(0)
w = gemm.quantize_mx(torch.randn(4 * C, C, device=dev) * 0.02)
if KM:
    w = (gemm.to_kmajor(w[0], 4, dealt=True), gemm.to_kmajor_scales(w[1], weight_side=True))
bias = (torch.randn(4 * C, device=dev) * 0.1).half()


def graph_us(fn, calls):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(calls):
                fn()
    torch.cuda.current_stream().wait_stream(s)
    g.replay()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / calls)
    return statistics.median(ts)


print(f"# {model}: fused fc1 [{C} -> {4 * C}], us per call by tiling (hipGraph, median of 5); default = the library's choice")
print(f"{'tokens':>7} {'default':>9} {'64x128':>9} {'128x128':>9} {'256x128':>9}")
tot = {k: 0.0 for k in ("default", 30, 20, 10)}
for T in [B * p * p for p in pns]:
    a = gemm.quantize_mx(torch.randn(T, C, device=dev).half(), kmajor=KM)
    row = []
    for cfg in (None, 30, 20, 10):
        _lib.set_option("FPQ_GEMM_CFG", cfg)
        t = graph_us(lambda: gemm.linear_fp4_gelu_dual(*a, *w, bias), 20 if T <= 10000 else 5)
        row.append(t)
        tot["default" if cfg is None else cfg] += t
    _lib.set_option("FPQ_GEMM_CFG", None)
    print(f"{T:7d} " + " ".join(f"{t:9.1f}" for t in row), flush=True)
print("sum     " + " ".join(f"{tot[k]:9.1f}" for k in ("default", 30, 20, 10)) + "   (the default column is measured first at every size: it carries the clocks settling)")
