#!/usr/bin/env python3
"""Weight-calibration throughput (BASELINE.json config 4): all Linear weights of
VAR-d30 (or d36 / d16) quantized per-group FP4 from fp32, sharded over N GPUs, one
all-gather of the fp16 results.

    python tools/bench_calib.py [--depth 30] [--iters 5] [--exchange fp16|codes]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P tools/bench_calib.py --depth 30

Weights are synthetic (randn*0.02, no checkpoints exist offline); every rank only
materialises the layers it owns.  Prints one JSON line on rank 0 with Gelem/s including
and excluding the gather.
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from fpqvar_amd import calibrate as cal  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--depth", type=int, default=30)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--exchange", default="fp16")
    args = ap.parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)
    shapes = cal.var_linear_shapes(args.depth)
    sizes = [(n, o * i) for n, (o, i) in shapes.items()]
    plan = cal.partition(sizes, world)
    mine = set(plan[rank])
    torch.manual_seed(1000 + rank)
    weights = {}
    for n, (o, i) in shapes.items():      # non-owned layers: shape only (meta), never read
        weights[n] = torch.randn(o, i, device=dev) * 0.02 if n in mine else torch.empty(o, i, device="meta")
    total = sum(s for _, s in sizes)

    def run(gather):
        w = weights if not gather else {n: (t if n in mine else torch.empty(t.shape, device=dev)) for n, t in weights.items()}
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        t0 = time.perf_counter()
        for _ in range(args.iters):
            cal.calibrate_sharded(w, gather=gather, exchange=args.exchange)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        dt = (time.perf_counter() - t0) / args.iters
        if world > 1:
            t = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t)
        return dt

    cal.calibrate_sharded({n: weights[n] for n in mine} if world == 1 else weights, gather=False)   # warm-up
    t_local = run(False)
    t_full = run(True) if world > 1 else t_local
    if rank == 0:
        print(json.dumps({"workload": f"VAR-d{args.depth} all-Linear weight calibration, fp32 -> per-group(128) E2M1 -> fp16",
                          "elements": total, "n_gpus": world, "exchange": args.exchange,
                          "quantize_only_ms": round(t_local * 1e3, 3), "with_all_gather_ms": round(t_full * 1e3, 3),
                          "Gelem_s_quantize_only": round(total / t_local / 1e9, 2),
                          "Gelem_s_with_gather": round(total / t_full / 1e9, 2)}))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
