#!/usr/bin/env python3
"""Where a generation batch's time goes BY SCALE STEP: the ten hipGraphs of path Q (or F / R) replayed one at a time, HIP events
around each (median of 5).  usage: step_times.py [model] [path]"""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from fpqvar_amd import var_block  # noqa: E402

model = sys.argv[1] if len(sys.argv) > 1 else "d30-256"
path = sys.argv[2] if len(sys.argv) > 2 else "Q"
gb = var_block.GenerationBatch(model, "w4a4", device="cuda:0")
with var_block.tuned_torch_gemms():
    gb.run_eager(path)
    graphs, keep = gb.capture(path)
    gb.replay(graphs)
    per = [[] for _ in graphs]
    for _ in range(5):
        for i, g in enumerate(graphs):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            g.replay()
            e1.record()
            torch.cuda.synchronize()
            per[i].append(e0.elapsed_time(e1))
ms = [statistics.median(p) for p in per]
tot = sum(ms)
print(f"# {gb.describe()}, path {path}: ms per scale step (one hipGraph each, {gb.depth} blocks), share of the batch")
for pn, t in zip(gb.patch_nums, ms):
    print(f"  {pn:2d} x {pn:2d} = {gb.B * pn * pn:6d} rows  {t:8.3f} ms  {100 * t / tot:5.1f} %   {1e3 * t / gb.depth:7.1f} us per block")
print(f"  sum {tot:.2f} ms")
