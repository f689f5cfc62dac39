#!/usr/bin/env python3
"""Survey of the hot-path entry points over the shapes of the whole VAR family (d16 .. d36: C = 1024 .. 2304, hidden 4 C,
KV rows of 64) and their optional arguments - the point is to find an instantiation that is far off the others (round 3:
rotate_quant with a smoothing vector had never been timed and ran at 0.22 of 8 TB/s).  One line per case: us per call
(bursts of 60 behind 6 untimed launches, three inputs in turn, best of 3) and the fraction of 8 TB/s at the case's
algorithmic bytes.      python tools/survey_shapes.py [substring ...]  ->  stdout, gpurun_out/survey_shapes.json"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from fpqvar_amd import ops, quant_utils as qu, rotation as rot  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(7)
ROWS = 32768   # tokens per case (the large steps of a batch)
W4 = 0.5 + 2.0 / 128


def inputs(rows, cols, dtype, n=3, gelu=False):
    xs = [torch.randn(rows, cols, device=dev, generator=g) for _ in range(n)]
    if gelu:
        xs = [torch.nn.functional.gelu(x, approximate="tanh") for x in xs]
    return [x.to(dtype) for x in xs]


def timed(fn, xs):
    k = [0]

    def call():
        k[0] += 1
        return fn(xs[k[0] % len(xs)])
    best = 1e9
    for _ in range(3):
        for _ in range(6):
            call()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(60):
            call()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 60 * 1e3)
    return best


def cases():
    for depth, C in ((16, 1024), (20, 1280), (24, 1536), (30, 1920), (36, 2304)):
        B = 64
        L = ROWS // B
        sm = torch.rand(C, device=dev, generator=g) + 0.5
        sc16 = (torch.randn(B, 1, C, device=dev, generator=g) * 0.3).half()
        sh16 = (torch.randn(B, 1, C, device=dev, generator=g) * 0.3).half()
        for xdt, xb in ((torch.float16, 2), (torch.float32, 4)):
            tag = f"d{depth} C={C} {'fp16' if xb == 2 else 'fp32'} rows"
            xs = inputs(ROWS, C, xdt)
            xa = [x.view(B, L, C) for x in xs]
            n = ROWS * C
            yield f"{tag}: rotate_quant e2m1", lambda x: rot.rotate_quant(x, "e2m1"), xs, n * (xb + 2)
            yield f"{tag}: rotate_quant e2m1 + smooth", lambda x: rot.rotate_quant(x, "e2m1", smooth=sm), xs, n * (xb + 2)
            yield f"{tag}: rotate_quant -> codes", lambda x: rot.rotate_quant_mx(x), xs, n * (xb + W4)
            yield f"{tag}: rotate_quant -> codes + smooth", lambda x: rot.rotate_quant_mx(x, smooth=sm), xs, n * (xb + W4)
            yield f"{tag}: adaln e2m1", lambda x: rot.adaln_rotate_quant(x, sc16, sh16, "e2m1"), xa, n * (xb + 2)
            yield f"{tag}: adaln e2m1 + smooth", lambda x: rot.adaln_rotate_quant(x, sc16, sh16, "e2m1", smooth=sm), xa, n * (xb + 2)
            yield f"{tag}: adaln e2m1, fp32 modulation", lambda x: rot.adaln_rotate_quant(x, sc16.float(), sh16.float(), "e2m1", smooth=sm), xa, n * (xb + 2)
            yield f"{tag}: adaln -> codes + smooth", lambda x: rot.adaln_rotate_quant_mx(x, sc16, sh16, smooth=sm), xa, n * (xb + W4)
            yield f"{tag}: adaln per token e2m3 + smooth", lambda x: rot.adaln_rotate_quant_token(x, sc16, sh16, "e2m3", smooth=sm), xa, n * (xb + 2)
            yield f"{tag}: adaln per token -> fp8 + smooth", lambda x: rot.adaln_rotate_quant_token(x, sc16, sh16, "e2m3", smooth=sm, emit="fp8"), xa, n * (xb + 1)
            yield f"{tag}: adaln per token -> fp6 + smooth", lambda x: rot.adaln_rotate_quant_token(x, sc16, sh16, "e2m3", smooth=sm, emit="fp6"), xa, n * (xb + 0.75)
            if xb == 2:
                for t in ("e2m1", "e1m2", "e3m0", "e2m3", "e3m2"):
                    yield f"{tag}: per group {t}", (lambda x, t=t: ops.quant_rows(x, t, 128, torch.float16)), xs, n * 4
                for t in ("e2m3", "e3m2"):
                    yield f"{tag}: per token {t}", (lambda x, t=t: ops.quant_rows(x, t, C, torch.float16)), xs, n * 4
                kv = [x.view(-1, 64) for x in xs]
                yield f"{tag}: KV rows of 64, e2m3", lambda x: ops.quant_rows(x, "e2m3", 64, torch.float16), kv, n * 4
                yield f"{tag}: KV rows of 64, e2m1", lambda x: ops.quant_rows(x, "e2m1", 64, torch.float16), kv, n * 4
            del xs, xa
        hs = inputs(ROWS, 4 * C, torch.float16, n=2, gelu=True)
        n = ROWS * 4 * C
        tag = f"d{depth} hidden {4 * C} fp16"
        yield f"{tag}: dual e1m2-/e2m1+ per group", lambda x: ops.quant_rows_dual(x, "e1m2_neg", "e2m1_pos", 128, 1.0), hs, n * 4
        yield f"{tag}: dual e1m2-/e2m1+ per group, clip 0.9", lambda x: ops.quant_rows_dual(x, "e1m2_neg", "e2m1_pos", 128, 0.9), hs, n * 4
        yield f"{tag}: dual int-/e2m3+ per group", lambda x: ops.quant_rows_dual(x, "int_neg", "e2m3_pos", 128, None), hs, n * 4
        yield f"{tag}: dual int-/e2m3+ per token", lambda x: ops.quant_rows_dual(x, "int_neg", "e2m3_pos", 4 * C, None), hs, n * 4
        yield f"{tag}: per token e2m3", lambda x: ops.quant_rows(x, "e2m3", 4 * C, torch.float16), hs, n * 4
        yield f"{tag}: per group e2m1", lambda x: ops.quant_rows(x, "e2m1", 128, torch.float16), hs, n * 4
        del hs
        ws = inputs(3 * C, C, torch.float32)
        n = 3 * C * C
        yield f"d{depth} weight [{3 * C} x {C}] fp32: per group e2m1 -> fp16", lambda x: ops.quant_rows(x, "e2m1", 128, torch.float16), ws, n * 6
        yield f"d{depth} weight [{3 * C} x {C}] fp32: per channel e2m3 -> fp16", lambda x: ops.quant_rows(x, "e2m3", C, torch.float16), ws, n * 6


def other_cases():
    """The entry points that are not producers or row quantizers of an activation, at the d30 shape."""
    from fpqvar_amd import gemm
    C = 1920
    n = ROWS * C
    x16 = inputs(ROWS, C, torch.float16)
    x32 = inputs(ROWS, C, torch.float32)
    tab = torch.tensor([-6.0, -4.0, -3.0, -2.0, -1.5, -1.0, -0.5, 0.0, 0.5, 1.0, 1.5, 2.0, 3.0, 4.0, 6.0], device=dev)
    odd = torch.tensor([-5.0, -2.5, -1.0, -0.3, 0.0, 0.2, 0.7, 1.9, 4.4], device=dev)
    yield "quant_cuda.quant fp32, the E2M1 table (recognised)", lambda x: ops.quant_nearest(x.view(-1), tab), x32, n * 8
    yield "quant_cuda.quant fp32, a 9-entry table of its own (literal scan)", lambda x: ops.quant_nearest(x.view(-1), odd), x32, n * 8
    yield "per tensor argmin e2m1 fp32 (BASELINE config 1 form)", lambda x: ops.quant_tensor_argmin(x, "e2m1"), x32, n * 12
    yield "per group argmin e2m1 fp32 (pure-torch semantics)", lambda x: ops.quant_rows_argmin(x, "e2m1", 128, False), x32, n * 8
    yield "per token argmin e2m1 fp32, clamp3", lambda x: ops.quant_rows_argmin(x, "e2m1", C, True), x32, n * 8
    yield "per group argmin e2m1 fp16 -> fp32", lambda x: ops.quant_rows_argmin(x, "e2m1", 128, False), x16, n * 6
    yield "dual argmin e1m2-/e2m1+ fp16 -> fp32", lambda x: ops.quant_rows_dual_argmin(x, "e1m2_neg", "e2m1_pos", 128, 1.0), x16, n * 6
    yield "neg_reverse per group e2m1 fp16", lambda x: ops.quant_rows_neg_reverse(x, "e2m1", 128), x16, n * 4
    yield "nearest argmin, 15-entry grid, fp32", lambda x: ops.quant_nearest_argmin(x.view(-1), tab), x32, n * 8
    yield "per group e2m1 fp32 -> fp32", lambda x: ops.quant_rows(x, "e2m1", 128), x32, n * 8
    yield "per group e2m1 fp16 -> fp32 (mixed)", lambda x: ops.quant_rows(x, "e2m1", 128, torch.float32), x16, n * 6
    yield "per token e2m3 fp32 -> fp16", lambda x: ops.quant_rows(x, "e2m3", C, torch.float16), x32, n * 6
    yield "per token e2m3 fp32 -> fp32", lambda x: ops.quant_rows(x, "e2m3", C, torch.float32), x32, n * 8
    yield "rows of 1000 (ragged) e2m1 fp16", lambda x: ops.quant_rows(x.view(-1)[:32000 * 1000].view(-1, 1000), "e2m1", 1000), x16, 32000 * 1000 * 4
    yield "codes: generic byte codes + scales, e2m1 g128", lambda x: ops.quant_rows_codes(x, "e2m1", 128), x16, n * (2 + 1 + 2 / 128)
    yield "codes: nibble-packed + scales, e2m1 g128", lambda x: ops.quant_rows_codes(x, "e2m1", 128, True), x16, n * (2 + W4)
    yield "codes: FP4 GEMM operands (quantize_mx)", lambda x: gemm.quantize_mx(x), x16, n * (2 + W4)
    yield "codes: E4M3 bytes per token (quantize_fp8)", lambda x: gemm.quantize_fp8(x), x16, n * 3
    yield "codes: dense 6-bit per token (quantize_fp6)", lambda x: gemm.quantize_fp6(x), x16, n * 2.75
    cm = [gemm.quantize_mx(x) for x in x16]
    yield "decode FP4 operands (dequantize_mx)", lambda cs: gemm.dequantize_mx(*cs), cm, n * (W4 + 4)
    k16 = [x.view(-1, 64) for x in x16]
    yield "K and V in one launch (quant_rows_multi, rows of 64, e2m3)", lambda x: ops.quant_rows_multi([x, x], "e2m3", 64, torch.float16), k16, 2 * n * 4
    yield "absmax fp16", lambda x: ops.absmax(x), x16, n * 2
    yield "rotate_quant returning the rotated rows too", lambda x: rot.rotate_quant(x, "e2m1", return_rotated=True), x16, n * 6


def main():
    want = sys.argv[1:]
    res = {}
    import itertools
    for name, fn, xs, nbytes in itertools.chain(other_cases(), cases()):
        if want and not any(w in name for w in want):
            continue
        try:
            us = timed(fn, xs)
            res[name] = {"us": round(us, 1), "frac_of_8TBps": round(nbytes / us / 8e6, 3)}
            flag = "   <-- slow" if res[name]["frac_of_8TBps"] < 0.5 and nbytes > 5e7 else ""
            print(f"{name:70s} {us:8.1f} us  {res[name]['frac_of_8TBps']:.3f}{flag}", flush=True)
        except Exception as e:   # an unsupported combination is a finding too
            res[name] = {"error": repr(e)[:160]}
            print(f"{name:70s} ERROR {e!r}"[:200], flush=True)
        torch.cuda.empty_cache()
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(res, open("gpurun_out/survey_shapes.json", "w"), indent=1)


if __name__ == "__main__":
    main()
