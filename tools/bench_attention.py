#!/usr/bin/env python3
"""Attention over the KV cache at the ten scale steps of a generation batch: torch SDPA (AOTriton flash kernel on this
image) vs fpq_attention_blhc, same fp16 inputs in the reference's [B, L, H, c] layout.
    python tools/bench_attention.py [--model d30-256|d36-512]"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from fpqvar_amd import ops  # noqa: E402

MODELS = {"d30-256": (30, (1, 2, 3, 4, 5, 6, 8, 10, 13, 16), 100), "d36-512": (36, (1, 2, 3, 4, 6, 9, 13, 18, 24, 32), 20)}


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n * 1e3)
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="d30-256", choices=tuple(MODELS))
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    H, patch, B = MODELS[args.model]
    c, cum, steps, tot = 64, 0, [], [0.0, 0.0]
    for pn in patch:
        Lq = pn * pn
        cum += Lq
        q = F.normalize(torch.randn(B, Lq, H, c, device=dev), dim=-1).mul(8).half()
        k = F.normalize(torch.randn(B, cum, H, c, device=dev), dim=-1).half()
        v = torch.randn(B, cum, H, c, device=dev).half()
        t_sdpa = timeit(lambda: F.scaled_dot_product_attention(q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2), scale=1.0))
        t_fpq = timeit(lambda: ops.attention_blhc(q, k, v, 1.0))
        ref = F.scaled_dot_product_attention(q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2), scale=1.0).transpose(1, 2)
        err = (ops.attention_blhc(q, k, v, 1.0).float() - ref.float()).abs().max().item()
        fl = 4.0 * B * H * c * Lq * cum
        steps.append({"Lq": Lq, "Lkv": cum, "sdpa_us": round(t_sdpa, 1), "fpq_us": round(t_fpq, 1),
                      "sdpa_TFLOPs": round(fl / t_sdpa / 1e6, 1), "fpq_TFLOPs": round(fl / t_fpq / 1e6, 1), "max_abs_diff": round(err, 5)})
        tot[0] += t_sdpa
        tot[1] += t_fpq
        print(steps[-1], flush=True)
    print(json.dumps({"workload": f"attention over the KV cache, VAR-{args.model}, B={B}, H={H}, c=64, 10 steps", "steps": steps,
                      "sum_sdpa_us": round(tot[0], 1), "sum_fpq_us": round(tot[1], 1), "speedup": round(tot[0] / tot[1], 2)}))


if __name__ == "__main__":
    main()
