#!/usr/bin/env python3
"""The format search of BASELINE config 4 as a measured workload (reference: search/search_fp6_format.py:589-608 and its
FP4 twin search/search_fp4_format.py:782-821): one VAR-d30 block's mat_qkv layer, 100 calibration samples of
[2, pn^2, 1920] (the ten scale steps, ten samples each), w [5760 x 1920].
Three ways, same quantizer semantics:
  reference_sequence  the reference's loop with its quantizer as it is written - ~11 torch ops around the scan kernel per
                      call, sample by sample, one host sync per sample and pair
  loop_fused          the same loop with this library's one-launch quantizers (format_search.search_layer(batched=False))
  batched             all samples of the layer in one quantizer launch per format, one GEMM per pair, one sync per layer
usage: bench_format_search.py > profiles/r03_format_search.json"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from fpqvar_amd import format_search as fs, ops, quant_utils as qu  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
C, OUT = 1920, 5760
PN = (1, 2, 3, 4, 5, 6, 8, 10, 13, 16)
xs = [(torch.randn(2, pn * pn, C, device=dev, generator=g) * torch.exp(0.5 * torch.randn(2, pn * pn, C, device=dev, generator=g))).half()
      for pn in PN for _ in range(10)]
w = (torch.randn(OUT, C, device=dev, generator=g) * 0.02).half()
GRIDS = {"fp6_e2m3": qu.fp6_e2m3_grid, "fp6_e3m2": qu.fp6_e3m2_grid, "fp_e1": qu.fp4_e1m2_grid, "fp_e2": qu.fp4_e2m1_grid,
         "fp_e3": qu.fp4_e3m0_grid}


def reference_quantizer(fmt):
    """tr/quant_utils.py:503-517 (per token) / :313-330 (per group 128), op for op, around the literal scan kernel."""
    grid = GRIDS[fmt]

    def per_token(x):
        t = grid.to(x.device)
        scale = x.abs().max(dim=-1, keepdim=True)[0] / t.abs().max()
        xn = (x / scale).view(-1).to(torch.float32)
        q = ops.quant_nearest(xn, torch.cat([t, t[-1:]]).type_as(xn))     # duplicated last entry: the literal K-step scan
        torch.zeros_like(xn)
        return (q.view(x.shape) * scale).to(torch.float16)

    def per_group(x):
        t = grid.to(x.device)
        xs_ = x.reshape(-1, 128)
        scale = xs_.abs().max(dim=-1, keepdim=True)[0] / t.abs().max()
        xn = (xs_ / scale).view(-1).to(torch.float32)
        q = ops.quant_nearest(xn, torch.cat([t, t[-1:]]).type_as(xn))
        torch.zeros_like(xn)
        return (q.view(xs_.shape) * scale).view(x.shape).to(x.dtype)
    return per_token if fmt.startswith("fp6") else per_group


def reference_sequence(formats):
    losses = {}
    for wf in formats:
        wq = reference_quantizer(wf)(w)
        for af in formats:
            qa = reference_quantizer(af)
            loss = 0.0
            for x in xs:
                y_fp = torch.matmul(x, w.T)
                y_q = torch.matmul(qa(x), wq.T)
                loss += torch.mean((y_fp - y_q) ** 2).item()               # compute_quant_error: one sync per sample
            losses[(wf, af)] = loss
    return losses


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best * 1e3


res = {"layer": f"VAR-d30 mat_qkv [{OUT} x {C}], {len(xs)} samples [2, pn^2, {C}], {sum(x.shape[1] * 2 for x in xs)} rows", "ms_per_block_layer": {}}
for name, formats in (("fp6 2x2", fs.FP6_FORMATS), ("fp4 3x3", fs.FP4_FORMATS)):
    r = {"reference_sequence": round(timed(lambda: reference_sequence(formats), 2), 2),
         "loop_fused": round(timed(lambda: fs.search_layer(xs, w, formats, batched=False)), 2),
         "batched": round(timed(lambda: fs.search_layer(xs, w, formats)), 2)}
    a, b = reference_sequence(formats), fs.search_layer(xs, w, formats)[2]
    r["max_rel_loss_diff_batched_vs_reference_sequence"] = float(max(abs(a[k] - b[k]) / a[k] for k in a))
    r["same_winner"] = min(a, key=a.get) == min(b, key=b.get)
    res["ms_per_block_layer"][name] = r
print(json.dumps(res, indent=1))
