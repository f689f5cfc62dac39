#!/usr/bin/env python3
"""Time any of the row quantizers at a given shape with rotating buffers (cold HBM), HIP-event timing.
usage: bench_ops.py [case ...]   (no argument: all cases)"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from fpqvar_amd import ops  # noqa: E402

R = 65536
CASES = {
    # name: (rows, cols, dtype, bytes per element, fn(x))
    "sym_e2m1_g128_f16": (R, 1920, torch.float16, 4, lambda x: ops.quant_rows(x, "e2m1", 128)),
    "sym_e2m1_g128_f16_7680": (R, 7680, torch.float16, 4, lambda x: ops.quant_rows(x, "e2m1", 128)),
    "sym_e2m1_g128_f16_16384x7680": (R // 4, 7680, torch.float16, 4, lambda x: ops.quant_rows(x, "e2m1", 128)),
    "sym_e2m3_token_f16_16384x7680": (R // 4, 7680, torch.float16, 4, lambda x: ops.quant_rows(x, "e2m3", 7680, torch.float16)),
    "dual_fp4_g128_f16_7680": (R, 7680, torch.float16, 4, lambda x: ops.quant_rows_dual(x, "e1m2_neg", "e2m1_pos", 128, 1.0)),
    "dual_fp6_g128_f16_7680": (R, 7680, torch.float16, 4, lambda x: ops.quant_rows_dual(x, "int_neg", "e2m3_pos", 128, None)),
    "dual_fp6_token_f16_7680": (R, 7680, torch.float16, 4, lambda x: ops.quant_rows_dual(x, "int_neg", "e2m3_pos", 7680, None)),
    "dual_fp4_token_f16_7680": (R, 7680, torch.float16, 4, lambda x: ops.quant_rows_dual(x, "e1m2_neg", "e2m1_pos", 7680, None)),
    "sym_e2m3_token_f16_7680": (R, 7680, torch.float16, 4, lambda x: ops.quant_rows(x, "e2m3", 7680, torch.float16)),
    "sym_e2m3_token_f16_1920": (R, 1920, torch.float16, 4, lambda x: ops.quant_rows(x, "e2m3", 1920, torch.float16)),
    "sym_e2m3_token_f32in_1920": (R // 2, 1920, torch.float32, 6, lambda x: ops.quant_rows(x, "e2m3", 1920, torch.float16)),
    "sym_e2m3_token_f32in_7680": (R // 4, 7680, torch.float32, 6, lambda x: ops.quant_rows(x, "e2m3", 7680, torch.float16)),
    "sym_e2m3_g128_f32in": (R // 2, 1920, torch.float32, 6, lambda x: ops.quant_rows(x, "e2m3", 128, torch.float16)),
    "sym_e2m1_g128_f32": (R // 2, 1920, torch.float32, 8, lambda x: ops.quant_rows(x, "e2m1", 128)),
    "argmin_e2m1_g128_f32": (R // 2, 1920, torch.float32, 8, lambda x: ops.quant_rows_argmin(x, "e2m1", 128, False)),
    "argmin_e2m1_token_f32_1920": (R // 2, 1920, torch.float32, 8, lambda x: ops.quant_rows_argmin(x, "e2m1", 1920, True)),
    "neg_reverse_g128_f16_7680": (R, 7680, torch.float16, 4, lambda x: ops.quant_rows_neg_reverse(x, "e2m1", 128)),
    "nearest_scan_e2m1_f32": (R // 2, 1920, torch.float32, 8, None),
}


def main():
    dev = torch.device("cuda:0")
    names = sys.argv[1:] or list(CASES)
    res = {}
    for name in names:
        rows, cols, dtype, bpe, fn = CASES[name]
        g = torch.Generator(device=dev).manual_seed(3)
        n = rows * cols
        nbuf = max(2, int(1.2e9 // (n * bpe)))           # > 1 GB in flight: nothing survives in the 256 MiB MALL
        xs = [torch.nn.functional.gelu(torch.randn(rows, cols, device=dev, generator=g), approximate="tanh").to(dtype)
              for _ in range(nbuf)]
        if fn is None:
            tab = torch.tensor([-6.0, -4.0, -3.0, -2.0, -1.5, -1.0, -0.5, 0.0, 0.5, 1.0, 1.5, 2.0, 3.0, 4.0, 6.0], device=dev)
            fn = lambda x: ops.quant_nearest(x.view(-1), tab)     # noqa: E731
        for i in range(10):
            fn(xs[i % nbuf])
        torch.cuda.synchronize()
        iters = 40
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(iters):
            fn(xs[i % nbuf])
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / iters
        res[name] = {"ms": round(ms, 4), "GBps": round(n * bpe / ms / 1e6, 1), "frac_of_8TBps": round(n * bpe / ms / 8e9, 3)}
        print(name, res[name], flush=True)
        del xs
    print(json.dumps(res))


if __name__ == "__main__":
    main()
