#!/usr/bin/env python3
"""Stock library against a GEMM-unit variant (tools/build_variant.sh --gemm <name> -D...), same process, alternating bursts:
the FP6 / FP8 row-scaled GEMMs and the FP4 per-group GEMM (plain and with the fused fc1 tail) at the bench shapes; every
variant result must be bit-equal to the stock one (ragged shapes included).
usage: ab_gemm_variant.py tools/ab/lib<name>.so [kmajor] [timing-only]      kmajor: the FP4 / FP6 operands as k-major images (include/fpq.h)"""
import os
import sys

os.environ["FPQ_NO_NATIVE"] = "1"   # the compiled binding is linked to the stock library
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from fpqvar_amd import _lib, gemm  # noqa: E402

stock = _lib.lib()
variant_path = sys.argv[1]
KM = "kmajor" in sys.argv[2:]
dev = torch.device("cuda:0")
torch.manual_seed(0)


def operands(T, K, O):
    x = torch.randn(T, K, device=dev).half()
    w = torch.randn(O, K, device=dev) * 0.02
    b = (torch.randn(O, device=dev) * 0.1).half()
    return x, w, b


def cases(T, K, O):
    x, w, b = operands(T, K, O)
    a6, w6 = gemm.quantize_fp6(x, kmajor=KM), gemm.quantize_fp6(w)
    a8, w8 = gemm.quantize_fp8(x), gemm.quantize_fp8(w)
    a4, w4 = gemm.quantize_mx(x, kmajor=KM), gemm.quantize_mx(w)
    if KM:
        w6, w4 = (gemm.to_kmajor(w6[0], 6, dealt=True), w6[1]), (gemm.to_kmajor(w4[0], 4, dealt=True), gemm.to_kmajor_scales(w4[1], weight_side=True))
    out = {
        "fp6": lambda: gemm.linear_fp6(*a6, *w6, bias=b),
        "fp8": lambda: gemm.linear_fp8(*a8, *w8, bias=b),
        "fp4": lambda: gemm.linear_fp4(*a4, *w4, bias=b),
    }
    if O % 128 == 0:
        out["fp4_fc1"] = lambda: gemm.linear_fp4_gelu_dual(*a4, *w4, bias=b)
    return out


def burst(fn, n=20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def use(l):
    _lib._lib = l


variant = _lib.use_variant(variant_path)
use(stock)
print(f"# stock {os.path.basename(stock._name)} against {os.path.basename(variant_path)}, {'k-major images' if KM else 'row-major codes'}; ms = best of 5 alternating bursts of 20")
for T, K, O in ((65536, 1920, 5760), (65536, 1920, 7680), (16900, 1920, 1920), (301, 1920, 392)):
    cs = cases(T, K, O)
    for name, fn in cs.items():
        use(stock)
        ref = fn()
        use(variant)
        got = fn()
        ref = ref if isinstance(ref, tuple) else (ref,)
        got = got if isinstance(got, tuple) else (got,)
        same = all(torch.equal(r, g) for r, g in zip(ref, got))
        best = {"stock": 1e9, "variant": 1e9}
        if T >= 16900:
            for _ in range(5):
                for tag, l in (("stock", stock), ("variant", variant)):
                    use(l)
                    fn()
                    best[tag] = min(best[tag], burst(fn))
            print(f"{name:8s} [{T} x {K}] -> {O}: stock {best['stock']:.4f} ms, variant {best['variant']:.4f} ms ({best['stock'] / best['variant']:.3f} x), bit-equal {same}")
        else:
            print(f"{name:8s} [{T} x {K}] -> {O}: bit-equal {same}")
        assert same or "timing-only" in sys.argv[2:], name   # timing-only: a variant that is wrong on purpose (cost of a phase)
