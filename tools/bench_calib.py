#!/usr/bin/env python3
"""Weight-calibration throughput (BASELINE.json config 4): all Linear weights of VAR-d30 (or d36 / d16) quantized
per-group FP4 from fp32 -> fp16, sharded over N GPUs, one all_gather_into_tensor of the fp16 results.

    python tools/bench_calib.py [--depth 30] [--iters 5] [--per-layer]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P tools/bench_calib.py --depth 30

Weights are synthetic (randn*0.02, no checkpoints exist offline); every rank only materialises the layers it owns.
Three forms are timed: ONE launch over the rank's segment table (LocalShard), the same with the in-place all-gather
(ShardedCalibration.run), and - for comparison - one fpq_quant_rows call per layer.  One JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from fpqvar_amd import calibrate as cal, ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--depth", type=int, default=30)
    ap.add_argument("--iters", type=int, default=5)
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
    mine = cal.plan_owners(shapes, world)[rank]
    torch.manual_seed(1000 + rank)
    own = {n: torch.randn(*shapes[n], device=dev) * 0.02 for n in mine}
    total = sum(o * i for o, i in shapes.values())

    def timed(fn):
        fn()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        best = float("inf")
        for _ in range(3):
            t0 = time.perf_counter()
            for _ in range(args.iters):
                fn()
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            best = min(best, (time.perf_counter() - t0) / args.iters)
        if world > 1:
            t = torch.tensor([best], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            best = float(t)
        return best

    sc = cal.ShardedCalibration(shapes, own)
    t_local = timed(sc.local.quantize)
    t_full = timed(sc.run) if world > 1 else t_local
    outs = {n: torch.empty(shapes[n], dtype=torch.float16, device=dev) for n in mine}
    t_layers = timed(lambda: [ops.quant_rows(own[n], "e2m1", 128, torch.float16) for n in mine])
    del outs
    if rank == 0:
        per_gpu_bytes = total * 6 / world
        print(json.dumps({"workload": f"VAR-d{args.depth} all-Linear weight calibration, fp32 -> per-group(128) E2M1 -> fp16",
                          "elements": total, "n_gpus": world,
                          "one_launch_ms": round(t_local * 1e3, 3), "with_all_gather_ms": round(t_full * 1e3, 3),
                          "per_layer_launches_ms": round(t_layers * 1e3, 3),
                          "one_launch_GBps_per_gpu": round(per_gpu_bytes / t_local / 1e9, 1),
                          "one_launch_frac_of_8TBps": round(per_gpu_bytes / t_local / 8e12, 3),
                          "Gelem_s_quantize_only": round(total / t_local / 1e9, 2),
                          "Gelem_s_with_gather": round(total / t_full / 1e9, 2)}))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
