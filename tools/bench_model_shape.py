#!/usr/bin/env python3
"""Model-shaped sweep: the activation-quant calls of ONE W4A4 VAR-d30
256x256 generation batch (B=50, CFG doubles it to 100): 10 scale steps x 30 blocks x
{mat_qkv, proj, fc1: E2M1 g=128 on [rows,1920]; fc2: E1M2-/E2M1+ g=128 on [rows,7680]}.
Times the fused kernels (eager and replayed from one hipGraph per step) against the
reference's unfused op sequence on the same GPU.  Synthetic activations.
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from fpqvar_amd import ops, quant_utils as qu  # noqa: E402

PATCH = (1, 2, 3, 4, 5, 6, 8, 10, 13, 16)
BLOCKS, B2, C = 30, 100, 1920


def unfused_sym(x, grid):
    xs = x.reshape(-1, 128)
    scale = xs.abs().max(dim=-1, keepdim=True)[0] / grid.abs().max()
    xn = (xs / scale).view(-1).to(torch.float32)
    z = ops.quant_nearest(xn, grid)
    torch.zeros_like(xn)
    return (z.view(xs.shape) * scale).view(x.shape).to(x.dtype)


def unfused_dual(x, gneg, gpos):
    clip = 1.0 * x.abs().max()
    x = torch.clamp(x, -clip, clip)
    xs = x.reshape(-1, 128)
    zeros = torch.zeros_like(xs)
    xn_, xp_ = torch.where(xs <= 0, xs, zeros), torch.where(xs > 0, xs, zeros)
    sn = xn_.abs().max(dim=-1, keepdim=True)[0] / gneg.abs().max()
    sp = xp_.abs().max(dim=-1, keepdim=True)[0] / gpos.abs().max()
    a = (xn_ / sn).view(-1).to(torch.float32)
    b = (xp_ / sp).view(-1).to(torch.float32)
    qa, qb = ops.quant_nearest(a, gneg), ops.quant_nearest(b, gpos)
    torch.zeros_like(a), torch.zeros_like(b)
    return (qa.view(xs.shape) * sn + qb.view(xs.shape) * sp).view(x.shape).to(x.dtype)


def main():
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    grid = qu.fp4_e2m1_grid.to(dev)
    gneg = torch.tensor([-1.75, -1.5, -1.25, -1.0, -0.75, -0.5, -0.25, 0.0], device=dev)
    gpos = torch.tensor([0.0, 0.5, 1.0, 1.5, 2.0, 3.0, 4.0, 6.0], device=dev)
    per_step = []
    tot = {"fused_eager_ms": 0.0, "fused_graph_ms": 0.0, "unfused_ms": 0.0, "elements": 0}
    for pn in PATCH:
        rows = B2 * pn * pn
        x = torch.randn(rows, C, device=dev).half()
        h = torch.nn.functional.gelu(torch.randn(rows, 4 * C, device=dev)).half()

        def block_fused():
            qu.fp_quant_e2_per_group_cuda(x, 4, 128)
            qu.fp_quant_e2_per_group_cuda(x, 4, 128)
            qu.fp_quant_e2_per_group_cuda(x, 4, 128)
            qu.fp_quant_e1m2_neg_e2m1_pos_per_group_cuda(h, 4, 128)

        def block_unfused():
            unfused_sym(x, grid), unfused_sym(x, grid), unfused_sym(x, grid)
            unfused_dual(h, gneg, gpos)

        def timed(fn, reps):
            fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / reps * 1e3

        t_eager = timed(block_fused, BLOCKS)
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            block_fused()
        torch.cuda.current_stream().wait_stream(s)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            for _ in range(BLOCKS):
                block_fused()
        t_graph = timed(graph.replay, 3) / BLOCKS
        t_unf = timed(block_unfused, 3)
        elems = rows * C * 3 + rows * 4 * C
        per_step.append({"pn": pn, "rows": rows, "fused_eager_us_per_block": round(t_eager * 1e3, 1),
                         "fused_graph_us_per_block": round(t_graph * 1e3, 1),
                         "unfused_us_per_block": round(t_unf * 1e3, 1)})
        tot["fused_eager_ms"] += t_eager * BLOCKS
        tot["fused_graph_ms"] += t_graph * BLOCKS
        tot["unfused_ms"] += t_unf * BLOCKS
        tot["elements"] += elems * BLOCKS
    tot = {k: (round(v, 2) if isinstance(v, float) else v) for k, v in tot.items()}
    tot["fused_graph_Gelem_s"] = round(tot["elements"] / tot["fused_graph_ms"] / 1e6, 1)
    tot["unfused_Gelem_s"] = round(tot["elements"] / tot["unfused_ms"] / 1e6, 1)
    print(json.dumps({"workload": "act-quant calls of one VAR-d30 256^2 W4A4 batch (B=50, cfg x2)", "total": tot,
                      "per_step": per_step}))


if __name__ == "__main__":
    main()
