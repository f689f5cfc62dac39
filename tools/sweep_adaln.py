#!/usr/bin/env python3
"""adaln_rotate_quant at [65500 x 1920] for the library in use (FPQ_ADALN_* / FPQ_ROT_* environment read at first launch).
usage: sweep_adaln.py [fp32|fp16] [group|token] [B L C]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fpqvar_amd import rotation as rot

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
dt = sys.argv[1] if len(sys.argv) > 1 else "fp32"
mode = sys.argv[2] if len(sys.argv) > 2 else "group"
B, L, C = (int(a) for a in sys.argv[3:6]) if len(sys.argv) > 5 else (100, 655, 1920)
xs = [torch.randn(B, L, C, device=dev, generator=g) for _ in range(3)]
if dt == "fp16":
    xs = [x.half() for x in xs]
scale = (torch.randn(B, 1, C, device=dev, generator=g) * 0.3).half()
shift = (torch.randn(B, 1, C, device=dev, generator=g) * 0.3).half()
s = torch.rand(C, device=dev, generator=g) + 0.5
k = 0
def run():
    global k
    k += 1
    if mode == "token":
        return rot.adaln_rotate_quant_token(xs[k % 3], scale, shift, "e2m3", smooth=s)
    return rot.adaln_rotate_quant(xs[k % 3], scale, shift, "e2m1", smooth=s)
for _ in range(200):
    run()
torch.cuda.synchronize()
res = []
for _ in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        run()
    e1.record()
    torch.cuda.synchronize()
    res.append(e0.elapsed_time(e1) / 50 * 1e3)
bpe = 6 if dt == "fp32" else 4
tag = " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("FPQ_"))
print(f"{dt} {mode} [{B}x{L}x{C}] {tag or 'default':30s} " + " ".join(f"{r:6.1f}" for r in res) + f"  us   min {min(res):.1f}  frac {B*L*C*bpe/min(res)/1e6/8:.3f}")
