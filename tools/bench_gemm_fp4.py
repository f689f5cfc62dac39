#!/usr/bin/env python3
"""F2 timing: act-quant + Linear for VAR-d30's mat_qkv / fc1 / fc2 shapes at 65536 rows -
reference path (fused fake-quant + fp16 F.linear on de-quantized tensors) vs FP4 MFMA path
(quantize to hardware codes + fpq_gemm_fp4_mx)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from fpqvar_amd import gemm, quant_utils as qu  # noqa: E402


def timed(fn, n=10):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def main():
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    res = {}
    for name, (T, K, O) in {"mat_qkv": (65536, 1920, 5760), "fc1": (65536, 1920, 7680), "proj": (65536, 1920, 1920)}.items():
        x = torch.randn(T, K, device=dev).half()
        w = torch.randn(O, K, device=dev) * 0.02
        wq16 = qu.fp_quant_e2_per_group_cuda(w, 4, 128).half()
        wc, wsc = gemm.quantize_mx(w)
        flops = 2.0 * T * K * O
        t_ref = timed(lambda: torch.nn.functional.linear(qu.fp_quant_e2_per_group_cuda(x, 4, 128), wq16))
        t_gemm16 = timed(lambda: torch.nn.functional.linear(x, wq16))
        ac, asc = gemm.quantize_mx(x)
        t_q = timed(lambda: gemm.quantize_mx(x))
        t_g4 = timed(lambda: gemm.linear_fp4(ac, asc, wc, wsc))
        res[name] = {"T,K,O": [T, K, O], "ref_fakequant_plus_fp16_gemm_ms": round(t_ref, 3), "fp16_gemm_only_ms": round(t_gemm16, 3),
                     "fp16_gemm_TFLOPs": round(flops / t_gemm16 / 1e9, 1), "quantize_to_codes_ms": round(t_q, 3),
                     "fp4_gemm_ms": round(t_g4, 3), "fp4_gemm_TFLOPs": round(flops / t_g4 / 1e9, 1),
                     "fp4_path_total_ms": round(t_q + t_g4, 3)}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
