#!/usr/bin/env python3
"""adaln_rotate_quant at the d30 generation shape ([100 x 655 x 1920] fp16), rotating inputs, HIP-event timing.
usage: bench_adaln.py [C=1920]   env FPQ_ADALN_V1=1: first-generation kernel, FPQ_ADALN_ROWS=n: rows per workgroup"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from fpqvar_amd import rotation as rot  # noqa: E402

C = int(sys.argv[1]) if len(sys.argv) > 1 else 1920
B, L = 100, 655
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
xs = [torch.randn(B, L, C, device=dev, generator=g).half() for _ in range(4)]
scale = (torch.randn(B, 1, C, device=dev, generator=g) * 0.3).half()
shift = (torch.randn(B, 1, C, device=dev, generator=g) * 0.3).half()
s = torch.rand(C, device=dev, generator=g) + 0.5
k = [0]


def fn():
    k[0] += 1
    return rot.adaln_rotate_quant(xs[k[0] % 4], scale, shift, "e2m1", smooth=s)


for _ in range(10):
    fn()
torch.cuda.synchronize()
best = 1e9
for _ in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fn()
    e1.record()
    torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1) / 20)
n = B * L * C
print(json.dumps({"case": f"adaln_rotate_quant fp16 [{B}x{L}x{C}]", "v1": bool(os.environ.get("FPQ_ADALN_V1")),
                  "rows_per_wg": os.environ.get("FPQ_ADALN_ROWS", "16"), "ms": round(best, 4),
                  "GBps": round(n * 4 / best / 1e6, 1), "frac_of_8TBps": round(n * 4 / best / 1e6 / 8000, 3)}))
