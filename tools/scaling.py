#!/usr/bin/env python3
"""Time against size for the hot kernels: t = fixed + per_row * rows.  The fixed part (launch ramp, the drain of the last
workgroups) and the steady-state rate are what a single-size number mixes.
usage: scaling.py [case ...]     cases: sym rotate adaln adaln32 token"""
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fpqvar_amd import ops, rotation as rot

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
C, L = 1920, 655
want = sys.argv[1:] or ["sym", "rotate", "adaln", "adaln32"]


def timed(fn, n=40):
    for _ in range(60):
        fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n * 1e3)
    return best


res = {}
for case in want:
    pts = []
    for B in (25, 50, 100, 200, 400):
        rows = B * L
        bpe = 6 if case == "adaln32" else 4
        xs = [torch.randn(B, L, C, device=dev, generator=g) for _ in range(3)]
        if case != "adaln32":
            xs = [x.half() for x in xs]
        scale = (torch.randn(B, 1, C, device=dev, generator=g) * 0.3).half()
        shift = (torch.randn(B, 1, C, device=dev, generator=g) * 0.3).half()
        s = torch.rand(C, device=dev, generator=g) + 0.5
        k = [0]

        def nxt():
            k[0] += 1
            return xs[k[0] % 3]
        fn = {"sym": lambda: ops.quant_rows(nxt().view(-1, C), "e2m1", 128),
              "rotate": lambda: rot.rotate_quant(nxt().view(-1, C), "e2m1"),
              "adaln": lambda: rot.adaln_rotate_quant(nxt(), scale, shift, "e2m1", smooth=s),
              "adaln32": lambda: rot.adaln_rotate_quant(nxt(), scale, shift, "e2m1", smooth=s),
              "token": lambda: rot.adaln_rotate_quant_token(nxt(), scale, shift, "e2m3", smooth=s)}[case]
        t = timed(fn)
        pts.append((rows, t, rows * C * bpe / t / 1e6 / 8))
        del xs
    (r0, t0, _), (r1, t1, _) = pts[-2], pts[-1]
    slope = (t1 - t0) / (r1 - r0)
    fixed = t1 - slope * r1
    bpe = 6 if case == "adaln32" else 4
    res[case] = {"points_rows_us_frac": [(r, round(t, 1), round(f, 3)) for r, t, f in pts], "fixed_us": round(fixed, 1),
                 "steady_frac_of_8TBps": round(C * bpe / slope / 1e6 / 8, 3)}
    print(case, json.dumps(res[case]), flush=True)
