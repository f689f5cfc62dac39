#!/usr/bin/env python3
"""A/B in ONE process: fc1 of a W4A4 AdaLN block with its tail fused into the FP4 GEMM
(gemm.linear_fp4_gelu_dual: one launch + the NaN fix-up launch) against the round-4 form - the GEMM, torch's GELU, the dual
E1M2-/E2M1+ quantizer (two launches) - at the token counts of the ten scale steps of VAR-d30 (C = 1920 -> 7680) and VAR-d36 512
(C = 2304 -> 9216), and at the metric's 65536 rows.  Each form: N calls captured in one hipGraph, median of 7 replays.
usage: ab_fc1.py [d30|d36]"""
import json
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.nn.functional as Fn  # noqa: E402

from fpqvar_amd import _lib, gemm, quant_utils as qu  # noqa: E402

model = sys.argv[1] if len(sys.argv) > 1 else "d30"
C, B, pns = (1920, 100, (1, 2, 3, 4, 5, 6, 8, 10, 13, 16)) if model == "d30" else (2304, 20, (1, 2, 3, 4, 6, 9, 13, 18, 24, 32))
dev = torch.device("cuda:0")
torch.manual_seed(0)
w = gemm.quantize_mx(torch.randn(4 * C, C, device=dev) * 0.02)
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
    for _ in range(2):
        g.replay()
    ts = []
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / calls)
    return statistics.median(ts)


rows = [B * p * p for p in pns] + [65536]
tot_f = tot_u = 0.0
print(f"# library {_lib.build_tag()}, {model}: fc1 [{C} -> {4 * C}] + GELU + dual quantizer; us per call (hipGraph, median of 7)")
print(f"{'tokens':>7} {'fused':>9} {'3 launches':>11} {'gemm':>9} {'gelu':>9} {'dual':>9}  ratio")
for T in rows:
    a = gemm.quantize_mx(torch.randn(T, C, device=dev).half())
    calls = 20 if T <= 10000 else 5
    y = gemm.linear_fp4(*a, *w, bias)
    h = Fn.gelu(y, approximate="tanh")
    fused = graph_us(lambda: gemm.linear_fp4_gelu_dual(*a, *w, bias), calls)
    unf = graph_us(lambda: qu.fp_quant_e1m2_neg_e2m1_pos_per_group_cuda(Fn.gelu(gemm.linear_fp4(*a, *w, bias), approximate="tanh"), 4, 128), calls)
    t_g = graph_us(lambda: gemm.linear_fp4(*a, *w, bias), calls)
    t_a = graph_us(lambda: Fn.gelu(y, approximate="tanh"), calls)
    t_q = graph_us(lambda: qu.fp_quant_e1m2_neg_e2m1_pos_per_group_cuda(h, 4, 128), calls)
    print(f"{T:7d} {fused:9.1f} {unf:11.1f} {t_g:9.1f} {t_a:9.1f} {t_q:9.1f}  {unf / fused:5.2f}", flush=True)
    if T != 65536:
        tot_f += fused
        tot_u += unf
    del a, y, h
    torch.cuda.empty_cache()
print(json.dumps({"model": model, "sum_over_the_ten_steps_us": {"fused": round(tot_f, 1), "three_launches": round(tot_u, 1)},
                  "per_batch_ms_over_%d_blocks" % (C // 64): {"fused": round(tot_f * (C // 64) / 1e3, 2), "three_launches": round(tot_u * (C // 64) / 1e3, 2)}}))
