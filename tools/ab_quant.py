#!/usr/bin/env python3
"""A/B two builds of libfpq_hip.so on the quantizer / producer kernels in ONE process (box-to-box variance is larger
than most kernel changes): the Python wrappers are pointed at library A and B alternately, three rounds per case,
rotating inputs (cold HBM).

    python tools/ab_quant.py tools/ab/libA.so tools/ab/libB.so [case ...]
"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["FPQ_NO_NATIVE"] = "1"   # the compiled binding is linked to the stock library: every case goes through ctypes here
import torch  # noqa: E402

from fpqvar_amd import _lib, ops, rotation as rot  # noqa: E402


def load(path):
    l = ctypes.CDLL(os.path.abspath(path))
    for name, (res, args) in _lib._SIGS.items():
        if hasattr(l, name):
            fn = getattr(l, name)
            fn.restype, fn.argtypes = res, args
    return l


def burst(fn, n=20):
    for _ in range(8):
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
    libs = {"A": load(sys.argv[1]), "B": load(sys.argv[2])}
    for n, l in libs.items():
        print(f"{n}: {sys.argv[1 if n == 'A' else 2]}  build tag {l.fpq_build_tag().decode() if hasattr(l, 'fpq_build_tag') else 'untagged'}", flush=True)
    want = sys.argv[3:]
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(0)
    R, C = 65536, 1920
    k = [0]

    def rotating(ts):
        def nxt():
            k[0] += 1
            return ts[k[0] % len(ts)]
        return nxt

    cases = {}
    x16 = rotating([torch.randn(R, C, device=dev, generator=g).half() for _ in range(4)])
    cases["sym_e2m1_g128_f16"] = lambda: ops.quant_rows(x16(), "e2m1", 128)
    cases["sym_e2m3_token_f16_1920"] = lambda: ops.quant_rows(x16(), "e2m3", C, torch.float16)
    cases["rotate_quant_e2m1"] = lambda: rot.rotate_quant(x16(), "e2m1")
    cases["rotate_quant_mx"] = lambda: rot.rotate_quant_mx(x16())
    B, L = 100, 655
    scale = (torch.randn(B, 1, C, device=dev, generator=g) * 0.3).half()
    shift = (torch.randn(B, 1, C, device=dev, generator=g) * 0.3).half()
    s = torch.rand(C, device=dev, generator=g) + 0.5
    cases["adaln_rotate_quant"] = lambda: rot.adaln_rotate_quant(x16()[:B * L].view(B, L, C), scale, shift, "e2m1", smooth=s)
    cases["adaln_token_fp8"] = lambda: rot.adaln_rotate_quant_token(x16()[:B * L].view(B, L, C), scale, shift, "e2m3", smooth=s, emit="fp8")
    cases["adaln_token_fp6"] = lambda: rot.adaln_rotate_quant_token(x16()[:B * L].view(B, L, C), scale, shift, "e2m3", smooth=s, emit="fp6")
    cases["adaln_token_e2m3"] = lambda: rot.adaln_rotate_quant_token(x16()[:B * L].view(B, L, C), scale, shift, "e2m3", smooth=s)
    x32r = None

    def x32():   # the residual stream is fp32 in the model (tr/var.py:209)
        nonlocal x32r
        if x32r is None:
            x32r = rotating([torch.randn(B, L, C, device=dev, generator=g) for _ in range(3)])
        return x32r()
    xr32 = rotating([torch.randn(R // 2, C, device=dev, generator=g) for _ in range(3)])
    cases["rotate_quant_x32"] = lambda: rot.rotate_quant(xr32(), "e2m1")
    cases["adaln_rotate_quant_x32"] = lambda: rot.adaln_rotate_quant(x32(), scale, shift, "e2m1", smooth=s)
    cases["adaln_token_e2m3_x32"] = lambda: rot.adaln_rotate_quant_token(x32(), scale, shift, "e2m3", smooth=s)
    cases["adaln_codes_mx_x32"] = lambda: rot.adaln_rotate_quant_mx(x32(), scale, shift, smooth=s)
    cases["adaln_codes_mx"] = lambda: rot.adaln_rotate_quant_mx(x16()[:B * L].view(B, L, C), scale, shift, smooth=s)
    big = None

    def fc2():
        nonlocal big
        if big is None:
            big = rotating([torch.nn.functional.gelu(torch.randn(R, 4 * C, device=dev, generator=g), approximate="tanh").half()
                            for _ in range(2)])
        return big()
    cases["dual_fp4_g128_7680"] = lambda: ops.quant_rows_dual(fc2(), "e1m2_neg", "e2m1_pos", 128, 1.0)
    cases["dual_fp6_g128_7680"] = lambda: ops.quant_rows_dual(fc2(), "int_neg", "e2m3_pos", 128, None)
    cases["dual_fp6_token_7680"] = lambda: ops.quant_rows_dual(fc2(), "int_neg", "e2m3_pos", 4 * C, None)
    cases["sym_e2m3_token_f16_7680"] = lambda: ops.quant_rows(fc2(), "e2m3", 4 * C, torch.float16)
    w32 = rotating([torch.randn(R // 2, C, device=dev, generator=g) * 0.02 for _ in range(3)])
    cases["weights_g128_f32_to_f16"] = lambda: ops.quant_rows(w32(), "e2m1", 128, torch.float16)
    cases["weights_channel_f32_to_f16"] = lambda: ops.quant_rows(w32(), "e2m3", C, torch.float16)
    # the clocks of an idle GPU take tens of milliseconds of sustained load to settle: warm up first, and alternate
    # the order of A and B between rounds so that neither always runs on the warmer chip
    _lib._lib = libs["A"]
    warm = cases["sym_e2m1_g128_f16"]
    for _ in range(4000):
        warm()
    torch.cuda.synchronize()
    for name, fn in cases.items():
        if want and name not in want:
            continue
        res = {"A": [], "B": []}
        for rnd in range(4):
            for n in ("AB" if rnd % 2 == 0 else "BA"):
                _lib._lib = libs[n]
                res[n].append(round(burst(fn, 40) * 1e3, 1))
        a, b = min(res["A"]), min(res["B"])
        print(f"{name:32s} A {res['A']}  B {res['B']}  us   B/A = {b / a:.3f}", flush=True)


if __name__ == "__main__":
    main()
