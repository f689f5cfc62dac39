#!/usr/bin/env python3
"""Time per call against the number of launches per burst (profiles/r03_burst_lead.txt): the tensors, calls and burst
protocol of bench.py's other_kernels; JUNK=n / CHURN=GiB reproduce bench.py's allocation history first."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fpqvar_amd import rotation as rot, ops
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
ROWS, COLS = 65536, 1920
if os.environ.get("JUNK"):   # the allocations bench.py holds by the time it reaches its secondary kernels
    junk = [torch.empty(ROWS * COLS, dtype=torch.half, device=dev) for _ in range(int(os.environ["JUNK"]))]
if os.environ.get("CHURN"):  # ... and a large block allocated and returned to the caching allocator
    big = torch.empty(int(float(os.environ["CHURN"]) * 2**30), dtype=torch.uint8, device=dev); del big
xs = [torch.randn(ROWS, COLS, device=dev, generator=g).half() for _ in range(3)]
print("x data_ptr % 2MiB:", [hex(x.data_ptr() % (1 << 21)) for x in xs], flush=True)
B, L = 100, 655
xa = [xs[i][:B * L].view(B, L, COLS) for i in range(3)]
scale = (torch.randn(B, 1, COLS, device=dev, generator=g) * 0.3).half()
shift = (torch.randn(B, 1, COLS, device=dev, generator=g) * 0.3).half()
s = torch.rand(COLS, device=dev, generator=g) + 0.5
k = [0]
def nxt(l):
    k[0] += 1
    return l[k[0] % len(l)]
fns = {"adaln": lambda: rot.adaln_rotate_quant(nxt(xa), scale, shift, "e2m1", smooth=s),
       "rotate": lambda: rot.rotate_quant(nxt(xs), "e2m1"),
       "sym": lambda: ops.quant_rows(nxt(xs), "e2m1", 128)}
def burst(fn, iters):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for name, fn in fns.items():
    for _ in range(300): fn()
    torch.cuda.synchronize()
    for iters in (20, 100):
        print(name, iters, [round(burst(fn, iters), 1) for _ in range(8)], flush=True)
# per-launch times inside one burst of 40 (events between launches)
fn = fns["adaln"]
for rep in range(2):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(41)]
    ev[0].record()
    for i in range(40):
        fn(); ev[i + 1].record()
    torch.cuda.synchronize()
    print("per-launch:", [round(ev[i].elapsed_time(ev[i + 1]) * 1e3, 1) for i in range(40)], flush=True)
