#!/usr/bin/env python3
"""Time adaln_rotate_quant (fp16 rows, e2m1) through a given build of the library: time_lib.py <lib.so> [B L C]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["FPQ_NO_NATIVE"] = "1"   # the compiled binding is linked to the stock library (the values case measured stock whatever was passed)
import torch
from fpqvar_amd import _lib, rotation as rot
_lib.use_variant(sys.argv[1])
dev = torch.device("cuda:0")
B, L, C = (int(a) for a in sys.argv[2:5]) if len(sys.argv) > 4 else (100, 655, 1920)
xs = [torch.randn(B, L, C, device=dev).half() for _ in range(3)]
if os.environ.get("TIME_X32"):
    xs = [x.float() for x in xs]
sc = (torch.randn(B, 1, C, device=dev) * 0.3).half(); sh = (torch.randn(B, 1, C, device=dev) * 0.3).half()
sm = (torch.rand(C, device=dev) + 0.5) if os.environ.get("TIME_SMOOTH") else None
k = 0
def run():
    global k; k += 1
    case = os.environ.get("TIME_CASE", "values")
    if case == "codes":      # packed E2M1 codes + fp16 group scales
        return rot.adaln_rotate_quant_mx(xs[k % 3], sc, sh, smooth=sm)
    if case in ("fp8", "fp6"):   # per-token operand outputs
        return rot.adaln_rotate_quant_token(xs[k % 3], sc, sh, "e2m3", smooth=sm, emit=case)
    if case == "token":
        return rot.adaln_rotate_quant_token(xs[k % 3], sc, sh, "e2m3", smooth=sm)
    return rot.adaln_rotate_quant(xs[k % 3], sc, sh, "e2m1", smooth=sm)
for _ in range(200): run()
torch.cuda.synchronize()
best = 1e9
for _ in range(4):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): run()
    e1.record(); torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1) / 50 * 1e3)
if os.environ.get("TIME_SUSTAIN"):   # the same call for seconds on end: does the time drift once the chip is warm?
    series = []
    for _ in range(int(os.environ["TIME_SUSTAIN"])):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200): run()
        e1.record(); torch.cuda.synchronize()
        series.append(round(e0.elapsed_time(e1) / 200 * 1e3, 1))
    print("sustained, 200 launches per figure:", series, flush=True)
tag = " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith(("FPQ_", "TIME_CASE", "TIME_X32")))
print(f"{os.path.basename(sys.argv[1]):22s} tag={_lib.build_tag():10s} {tag:24s} [{B}x{L}x{C}] {best:7.1f} us  frac {B*L*C*4/best/1e6/8:.3f}", flush=True)
