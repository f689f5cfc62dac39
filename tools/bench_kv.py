#!/usr/bin/env python3
"""KV-cache quantization over one VAR-d30 256^2 generation (10 scale steps, B=50 x2 for CFG, 30 heads
x 64): (A) the reference's op sequence - re-quantize the whole cache with ~11 torch ops + scan kernel,
(B) re-quantize the whole cache with the fused kernel, (C) IncrementalKVCache (each entry once).
Times only the cache maintenance (quant + cat / copies), synthetic k/v."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from fpqvar_amd import kv_cache as kv, ops, quant_utils as qu  # noqa: E402

PATCH = (1, 2, 3, 4, 5, 6, 8, 10, 13, 16)
B, H, c = 100, 30, 64


def unfused_token(x, grid):
    scale = x.abs().max(dim=-1, keepdim=True)[0] / grid.abs().max()
    xn = (x / scale).view(-1).to(torch.float32)
    z = ops.quant_nearest(xn, grid)
    torch.zeros_like(xn)
    return (z.view(x.shape) * scale).to(torch.float16)


def main():
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    ks = [torch.nn.functional.normalize(torch.randn(B, p * p, H, c, device=dev), dim=-1).half() for p in PATCH]
    vs = [torch.randn(B, p * p, H, c, device=dev).half() for p in PATCH]
    grid = qu.fp6_e2m3_grid.to(dev)
    res = {}
    for kv_bit in (6,):
        def run_a():
            ck = cv = None
            for k, v in zip(ks, vs):
                if ck is None:
                    ck, cv = k, v
                else:
                    ck, cv = torch.cat((unfused_token(ck, grid), k), 1), torch.cat((unfused_token(cv, grid), v), 1)
            return ck

        def run_b():
            ck = cv = None
            for k, v in zip(ks, vs):
                ck, cv = kv.update_kv_cache(ck, cv, k, v, True, kv_bit, 1, check_finite=False)
            return ck

        def run_c():
            inc = kv.IncrementalKVCache(B, sum(p * p for p in PATCH), H, c, kv_bit, device=dev)
            out = None
            for k, v in zip(ks, vs):
                out, _ = inc.append(k, v)
            return out

        def timed(fn, n=5):
            fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / n * 1e3

        a, b_, c_ = run_a(), run_b(), run_c()
        same_ab = bool(torch.equal(a.view(torch.int16), b_.view(torch.int16)))
        same_bc = bool(torch.equal(b_.view(torch.int16), c_.contiguous().view(torch.int16)))
        res[f"kv_bit{kv_bit}"] = {"A_reference_sequence_ms": round(timed(run_a, 2), 3), "B_fused_requantize_all_ms": round(timed(run_b), 3),
                                  "C_incremental_ms": round(timed(run_c), 3), "A_equals_B": same_ab, "B_equals_C": same_bc}
    print(json.dumps({"workload": "KV cache maintenance, one block, 10 steps, [100, 680, 30, 64] fp16 K and V", **res}))


if __name__ == "__main__":
    main()
