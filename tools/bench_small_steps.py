#!/usr/bin/env python3
"""The small-step regime of a VAR-d30 generation batch (tr/var.py:175: rows = 100 * pn^2, pn = 1, 2, 3, ...): what one
quantizer / producer call costs when the tensor is tiny.  Three clocks per (op, rows):
  eager_us   host loop over the Python wrapper (launch-bound: Python + ctypes + hipLaunchKernel), wall clock / call
  graph_us   the same calls captured in ONE hipGraph and replayed: GPU time per call incl. the inter-kernel gaps
  kernels    launches per call
and the floor: an empty-range launch of the same library (rows = 1) replayed the same way.
"eager_ctypes_us" is the same call through the Python + ctypes path of fpqvar_amd.ops (round 2's only binding); "eager_us" goes
through quant_utils, i.e. the compiled binding fpqvar_amd._native where it covers the function.
usage: bench_small_steps.py > profiles/r03_small_steps.json"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from fpqvar_amd import ops, quant_utils as qu, rotation as rot  # noqa: E402

dev = torch.device("cuda:0")
C, HID, B = 1920, 7680, 100
PN = (1, 2, 3, 4, 5, 6, 8, 10, 13, 16)
g = torch.Generator(device=dev).manual_seed(0)
N = 50


def graph_time(fn):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=s):
            for _ in range(N):
                fn()
    torch.cuda.current_stream().wait_stream(s)
    for _ in range(3):
        gr.replay()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        gr.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / N)
    return best


def eager_time(fn):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 200 * 1e6


res = {"note": "VAR-d30 256x256, B = 50 with CFG (100 conditioned rows per token); us per call", "steps": []}
tiny = torch.randn(1, 128, device=dev, generator=g).half()
res["launch_floor_graph_us"] = round(graph_time(lambda: qu.fp_quant_e2_per_group_cuda(tiny, 4, 128)), 2)
res["launch_floor_eager_us"] = round(eager_time(lambda: qu.fp_quant_e2_per_group_cuda(tiny, 4, 128)), 2)
res["launch_floor_eager_ctypes_us"] = round(eager_time(lambda: ops.quant_rows(tiny, "e2m1", 128)), 2)
res["compiled_binding"] = getattr(qu, "_native", None) is not None
scale = (torch.randn(B, 1, C, device=dev, generator=g) * 0.3).half()
shift = (torch.randn(B, 1, C, device=dev, generator=g) * 0.3).half()
smooth = torch.rand(C, device=dev, generator=g) + 0.5
for pn in PN:
    L = pn * pn
    rows = B * L
    x = torch.randn(B, L, C, device=dev, generator=g).half()
    hid = torch.nn.functional.gelu(torch.randn(rows, HID, device=dev, generator=g), approximate="tanh").half()
    ops_ = {
        "act_quant_e2m1_g128 (proj input)": (lambda: qu.fp_quant_e2_per_group_cuda(x, 4, 128), 1,
                                              lambda: ops.quant_rows(x, "e2m1", 128)),
        "adaln_rotate_quant (mat_qkv / fc1 input)": (lambda: rot.adaln_rotate_quant(x, scale, shift, "e2m1", smooth=smooth), 1, None),
        "dual_fp4_g128 (fc2 input, default clip)": (lambda: qu.fp_quant_e1m2_neg_e2m1_pos_per_group_cuda(hid, 4, 128), 2,
                                                     lambda: ops.quant_rows_dual(hid, "e1m2_neg", "e2m1_pos", 128, 1.0)),
    }
    step = {"pn": pn, "rows": rows}
    for name, (fn, launches, slow) in ops_.items():
        step[name] = {"eager_us": round(eager_time(fn), 2), "graph_us": round(graph_time(fn), 2), "kernels": launches}
        if slow is not None:
            step[name]["eager_ctypes_us"] = round(eager_time(slow), 2)
    res["steps"].append(step)
print(json.dumps(res, indent=1))
