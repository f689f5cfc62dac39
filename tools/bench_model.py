#!/usr/bin/env python3
"""Model-shaped end to end (BASELINE.json configs 3 / 5): the transformer part of one generation
batch through fpqvar_amd/var_block.py - the reference's op sequence (R), the fused fake-quant launches (F), the matrix-core
path (Q; --unfused-fc1: with fc1's GELU and fc2's input quantizer as separate launches, the round-4 form) - eager and as
one hipGraph per scale step.  bench.py's `generation` records come from the same module."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from fpqvar_amd import _lib, var_block  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="d30-256", choices=tuple(var_block.MODELS),
                    help="d30-256: VAR-d30 256x256 (C = 1920); d36-512: VAR-d36 512x512 (C = 2304, 2240 tokens)")
    ap.add_argument("--depth", type=int, default=None, help="run fewer blocks than the model has")
    ap.add_argument("--batch", type=int, default=None)
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--paths", default="F,Q,R")
    ap.add_argument("--config", default="w4a4", choices=("w4a4", "w6a6"),
                    help="w4a4: run.sh line 4 (per-group fp_e2, fc2 dual FP4); w6a6: run.sh line 10 (per-token / per-channel fp6_e2m3, fc2 dual FP6)")
    ap.add_argument("--no-graphs", action="store_true")
    ap.add_argument("--unfused-fc1", action="store_true", help="Q path: GEMM, GELU and the dual quantizer as three launches")
    ap.add_argument("--sdpa-in-f", action="store_true", help="path F with torch's SDPA instead of fpq_attention_blhc (as rounds 1 - 4 timed it)")
    ap.add_argument("--row-major-operands", action="store_true", help="Q path: row-major code tensors instead of k-major images (the form of rounds 1 - 4)")
    ap.add_argument("--qkv-copy-in", action="store_true", help="Q path: mat_qkv writes one [tokens, 3 C] tensor and the cache copies k / v in (the form before fpq_gemm_fp4_mx_split)")
    ap.add_argument("--tuned-gemms", action="store_true", help="torch's own GEMMs with the recorded TunableOp selections (var_block.tuned_torch_gemms)")
    args = ap.parse_args()
    torch.manual_seed(0)
    gb = var_block.GenerationBatch(args.model, args.config, depth=args.depth, batch_rows=args.batch, device="cuda:0",
                                   fused_fc1=not args.unfused_fc1, sdpa_in_f=args.sdpa_in_f, kmajor=not args.row_major_operands,
                                   qkv_to_cache=not args.qkv_copy_in)
    res = {"workload": gb.describe(), "depth": gb.depth, "batch_rows": gb.B, "fc1_epilogue_fused": gb.fused_fc1, "kmajor_operands": gb.kmajor, "qkv_to_cache": gb.qkv_to_cache,
           "library": _lib.build_tag()}
    paths = args.paths.split(",")
    import contextlib
    ctx = var_block.tuned_torch_gemms() if args.tuned_gemms else contextlib.nullcontext()
    with ctx as tg:
        res["torch_gemms_tuned"] = bool(args.tuned_gemms and tg.active)
        time_paths(args, gb, paths, res)
    if "R" in paths:
        for pth in ("F", "Q"):
            if pth in paths:
                res[f"speedup_{pth}_vs_R"] = round(res["R_ms_per_batch"] / res[f"{pth}_ms_per_batch"], 2)
    for pth in paths:
        res[f"images_per_s_{pth}"] = round((gb.B // 2) / (res[f"{pth}_ms_per_batch"] / 1e3), 1)
    print(json.dumps(res))


def time_paths(args, gb, paths, res):
    for path in paths:
        gb.run_eager(path)                                   # warm-up (allocator, kernel load)
        res[f"{path}_ms_per_batch"] = round(min(gb.run_eager(path) for _ in range(args.reps)), 1)
        torch.cuda.empty_cache()
        if args.no_graphs:
            continue
        try:
            graphs, keep = gb.capture(path)
            gb.replay(graphs)
            res[f"{path}_ms_per_batch_hipgraph"] = round(min(gb.replay(graphs) for _ in range(args.reps + 1)), 1)
            del graphs, keep
        except Exception as e:      # extra information only
            res[f"{path}_ms_per_batch_hipgraph"] = f"error: {str(e)[:120]}"
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
