#!/usr/bin/env python3
"""W6A6 (per-token activations x per-channel weights, FP6 E2M3) act-quant + Linear for VAR-d30's mat_qkv / fc1 / proj
shapes at 65536 rows: reference formulation (fused fake-quant + fp16 F.linear on de-quantized tensors) vs the
FP8-coded path (quantize to E4M3 bytes + fpq_gemm_fp8_rows) and the 6-bit packed path (fpq_gemm_fp6_rows)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from fpqvar_amd import gemm, quant_utils as qu  # noqa: E402


def timed(fn, n=20):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    best = float("inf")
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n)
    return best


def main():
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    res = {}
    for name, (T, K, O) in {"mat_qkv": (65536, 1920, 5760), "fc1": (65536, 1920, 7680), "proj": (65536, 1920, 1920)}.items():
        x = torch.randn(T, K, device=dev).half()
        w = torch.randn(O, K, device=dev) * 0.02
        wq16 = qu.fp6_quant_e2m3_per_token_cuda(w, 6)
        wc, wsc = gemm.quantize_fp8(w, "e2m3")
        flops = 2.0 * T * K * O
        t_ref = timed(lambda: torch.nn.functional.linear(qu.fp6_quant_e2m3_per_token_cuda(x, 6), wq16))
        t_gemm16 = timed(lambda: torch.nn.functional.linear(x, wq16))
        ac, asc = gemm.quantize_fp8(x, "e2m3")
        t_q = timed(lambda: gemm.quantize_fp8(x, "e2m3"))
        t_g8 = timed(lambda: gemm.linear_fp8(ac, asc, wc, wsc))
        a6, s6 = gemm.quantize_fp6(x)
        w6, ws6 = gemm.quantize_fp6(w)
        t_q6 = timed(lambda: gemm.quantize_fp6(x))
        t_g6 = timed(lambda: gemm.linear_fp6(a6, s6, w6, ws6))
        res[name] = {"T,K,O": [T, K, O], "quantize_to_fp6_codes_ms": round(t_q6, 3), "fp6_gemm_ms": round(t_g6, 3),
                     "fp6_gemm_TFLOPs": round(flops / t_g6 / 1e9, 1), "fp6_path_total_ms": round(t_q6 + t_g6, 3), "ref_fakequant_plus_fp16_gemm_ms": round(t_ref, 3), "fp16_gemm_only_ms": round(t_gemm16, 3),
                     "fp16_gemm_TFLOPs": round(flops / t_gemm16 / 1e9, 1), "quantize_to_fp8_codes_ms": round(t_q, 3),
                     "fp8_gemm_ms": round(t_g8, 3), "fp8_gemm_TFLOPs": round(flops / t_g8 / 1e9, 1),
                     "fp8_path_total_ms": round(t_q + t_g8, 3)}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
