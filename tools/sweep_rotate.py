#!/usr/bin/env python3
"""rotate_quant at the headline shape for the library in use (FPQ_ROT_* environment read at first launch)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fpqvar_amd import rotation as rot

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
xs = [torch.randn(65536, 1920, device=dev, generator=g).half() for _ in range(4)]
sm = (torch.rand(1920, device=dev, generator=g) + 0.5) if os.environ.get("SWEEP_SMOOTH") else None
if os.environ.get("SWEEP_X32"):
    xs = [x[:32768].float() for x in xs]
k = 0
def run():
    global k
    k += 1
    if os.environ.get("SWEEP_CODES"):   # the operand-emitting form: packed E2M1 codes + fp16 group scales (2.53 B per element)
        return rot.rotate_quant_mx(xs[k % 4], smooth=sm)
    return rot.rotate_quant(xs[k % 4], "e2m1", smooth=sm)
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
tag = " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("FPQ_ROT"))
print(f"{tag or 'default':40s} " + " ".join(f"{r:6.1f}" for r in res) + f"  us   min {min(res):.1f}  frac {65536*1920*(2 + 0.5 + 2 / 128 if os.environ.get('SWEEP_CODES') else 4)/min(res)/1e6/8:.3f}")
