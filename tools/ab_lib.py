#!/usr/bin/env python3
"""A/B two builds of libfpq_hip.so in ONE process on ONE box (box-to-box and run-to-run variance is +-7 %, larger than
most kernel changes): alternates the two libraries over three rounds per shape and prints every burst.

    python tools/ab_lib.py /path/libA.so /path/libB.so [fp4|fp6|fp8|quant|attn]
FPQ_AB_EPI=1: the GEMMs with bias, gate (one row per 256 tokens) and an in-place residual (the *_ex entry points).
Build the variants with the flags of __graft_entry__.HIP_FLAGS into files outside fpqvar_amd/ (on the GPU box the
libraries must travel inside the repo snapshot, e.g. under tools/ab/ - git-ignored)."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from fpqvar_amd import gemm  # noqa: E402
from fpqvar_amd._lib import F16, TABLE_IDS, GemmEpilogue, dtype_id, stream_ptr  # noqa: E402

V = ctypes.c_void_p


def burst(fn, n=20):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    pa, pb = sys.argv[1], sys.argv[2]
    what = sys.argv[3] if len(sys.argv) > 3 else "fp4"
    libs = {"A": ctypes.CDLL(os.path.abspath(pa)), "B": ctypes.CDLL(os.path.abspath(pb))}
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    sp = stream_ptr(dev)
    if what == "attn":      # attention over the KV cache at generation shapes (B, H, Lq, Lkv)
        for B, H, Lq, Lkv in ((100, 30, 64, 155), (100, 30, 100, 255), (100, 30, 169, 424), (100, 30, 256, 680), (20, 36, 1024, 2240)):
            q = torch.nn.functional.normalize(torch.randn(B, Lq, H, 64, device=dev), dim=-1).mul(8).half()
            k = torch.nn.functional.normalize(torch.randn(B, Lkv, H, 64, device=dev), dim=-1).half()
            v = torch.randn(B, Lkv, H, 64, device=dev).half()
            out = torch.empty_like(q)

            def call(lib):
                f = lib.fpq_attention_blhc
                f.argtypes = [V, V, V, V] + [ctypes.c_int64] * 9 + [ctypes.c_float, V]
                return lambda: f(q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), B, Lq, Lkv, H, 64, q.stride(0),
                                 q.stride(1), k.stride(0), k.stride(1), 1.0, sp)
            res = {"A": [], "B": []}
            for _ in range(3):
                for n in "AB":
                    res[n].append(round(burst(call(libs[n])), 4))
            print((B, H, Lq, Lkv), res, flush=True)
        return
    for T, K, O in ((65536, 1920, 5760), (65536, 1920, 7680), (65536, 1920, 1920), (25600, 1920, 5760)):
        x = torch.randn(T, K, device=dev).half()
        w = torch.randn(O, K, device=dev) * 0.02
        out = torch.empty(T, O, dtype=torch.float16, device=dev)
        with_epi = bool(os.environ.get("FPQ_AB_EPI")) and what in ("fp4", "fp6", "fp8")
        if with_epi:
            bias = (torch.randn(O, device=dev) * 0.1).half()
            gate = torch.randn(T // 256, O, device=dev).half()
            out.normal_()
            ep = GemmEpilogue(gate.data_ptr(), out.data_ptr(), 256)

            def call(lib, what=what):
                if what == "fp4":
                    f = lib.fpq_gemm_fp4_mx_ex
                    f.argtypes = [V, V, V, V, ctypes.c_int, V, V, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, V, V]
                    return lambda: f(ac.data_ptr(), asc.data_ptr(), wc.data_ptr(), wsc.data_ptr(), dtype_id(wsc.dtype), bias.data_ptr(),
                                     out.data_ptr(), T, O, K, ctypes.byref(ep), sp)
                f = lib.fpq_gemm_fp6_rows_ex if what == "fp6" else lib.fpq_gemm_fp8_rows_ex
                f.argtypes = [V, V, ctypes.c_int, V, V, ctypes.c_int, V, V, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, V, V]
                return lambda: f(ac.data_ptr(), asc.data_ptr(), dtype_id(asc.dtype), wc.data_ptr(), wsc.data_ptr(),
                                 dtype_id(wsc.dtype), bias.data_ptr(), out.data_ptr(), T, O, K, ctypes.byref(ep), sp)
            ac, asc = {"fp4": gemm.quantize_mx, "fp6": gemm.quantize_fp6, "fp8": gemm.quantize_fp8}[what](x)
            wc, wsc = {"fp4": gemm.quantize_mx, "fp6": gemm.quantize_fp6, "fp8": gemm.quantize_fp8}[what](w)
        elif what == "fp4":
            ac, asc = gemm.quantize_mx(x)
            wc, wsc = gemm.quantize_mx(w)

            def call(lib):
                f = lib.fpq_gemm_fp4_mx
                f.argtypes = [V, V, V, V, ctypes.c_int, V, V, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, V]
                return lambda: f(ac.data_ptr(), asc.data_ptr(), wc.data_ptr(), wsc.data_ptr(), dtype_id(wsc.dtype), None,
                                 out.data_ptr(), T, O, K, sp)
        elif what == "fp6":
            ac, asc = gemm.quantize_fp6(x)
            wc, wsc = gemm.quantize_fp6(w)

            def call(lib):
                f = lib.fpq_gemm_fp6_rows
                f.argtypes = [V, V, ctypes.c_int, V, V, ctypes.c_int, V, V, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, V]
                return lambda: f(ac.data_ptr(), asc.data_ptr(), dtype_id(asc.dtype), wc.data_ptr(), wsc.data_ptr(),
                                 dtype_id(wsc.dtype), None, out.data_ptr(), T, O, K, sp)
        elif what == "fp8":
            ac, asc = gemm.quantize_fp8(x)
            wc, wsc = gemm.quantize_fp8(w)

            def call(lib):
                f = lib.fpq_gemm_fp8_rows
                f.argtypes = [V, V, ctypes.c_int, V, V, ctypes.c_int, V, V, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, V]
                return lambda: f(ac.data_ptr(), asc.data_ptr(), dtype_id(asc.dtype), wc.data_ptr(), wsc.data_ptr(),
                                 dtype_id(wsc.dtype), None, out.data_ptr(), T, O, K, sp)
        else:   # the headline fake-quant kernel on [T, K]
            o16 = torch.empty_like(x)

            def call(lib):
                f = lib.fpq_quant_rows
                f.argtypes = [V, V, ctypes.c_int64, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int, V]
                return lambda: f(x.data_ptr(), o16.data_ptr(), x.numel() // 128, 128, TABLE_IDS["e2m1"], F16, F16, sp)
        res = {"A": [], "B": []}
        for _ in range(3):
            for n in "AB":
                res[n].append(round(burst(call(libs[n])), 4))
        print((T, K, O), res, flush=True)


if __name__ == "__main__":
    main()
