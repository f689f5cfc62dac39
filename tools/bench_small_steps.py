#!/usr/bin/env python3
"""BASELINE configs 3 and 5 at the row counts they actually run (tr/var.py:175: ten scale steps, rows = 2 B pn^2): what the
four quantizer calls of a W4A4 AdaLN block cost per step, and the time-weighted fraction of 8 TB/s over the ten steps.
The measurement itself is bench.generation_steps (the bench line carries its summary as config3_steps / config5_steps);
this tool writes the full record.

  bench_small_steps.py --model d30|d36-512 --rows fp16|fp32 --mode rotating|resident [--eager] > profiles/r04_steps_<model>.json

--eager adds the host-side cost of one call through the Python wrappers (compiled binding where it covers the function)
and through ctypes: wall clock per call of a 200-call loop, launch-bound at small steps."""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if "--lib" in sys.argv:
    os.environ["FPQ_NO_NATIVE"] = "1"   # a variant build is reached through ctypes only (fpqvar_amd._lib.use_variant)
import torch  # noqa: E402

import bench  # noqa: E402


def stamp():
    """What was measured: the commit and the hash of the library that ran."""
    import hashlib
    from fpqvar_amd import _lib
    try:
        head = subprocess.run(["git", "-C", ROOT, "rev-parse", "HEAD"], capture_output=True, text=True).stdout.strip()
    except Exception:
        head = ""
    with open(_lib.LIB_PATH, "rb") as f:
        sha = hashlib.sha256(f.read()).hexdigest()
    return {"git_head": head or os.environ.get("FPQ_GIT_HEAD", "unknown (no .git on the GPU box: see FPQ_GIT_HEAD)"),
            "libfpq_hip_sha256": sha}


def eager_us(fn, n=200):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


def eager(dev, model, rows_dtype):
    from fpqvar_amd import ops, quant_utils as qu, rotation as rot
    m = bench.STEP_MODELS[model]
    C, B = m["C"], m["B"]
    g = torch.Generator(device=dev).manual_seed(0)
    scale = (torch.randn(B, 1, C, device=dev, generator=g) * 0.3).half()
    shift = (torch.randn(B, 1, C, device=dev, generator=g) * 0.3).half()
    smooth = torch.rand(C, device=dev, generator=g) + 0.5
    out = []
    # the "Q" path's calls at the first step (the compiled binding covers them since round 4): operand-emitting producers,
    # per-token producer with its two code forms, the KV-cache step, the FP4 GEMM
    from fpqvar_amd import gemm, kv_cache
    L0 = m["pn"][0] ** 2
    x0 = torch.randn(B, L0, C, device=dev, generator=g)
    x0 = x0 if rows_dtype == "fp32" else x0.half()
    heads = C // 64
    cache = kv_cache.IncrementalKVCache(B, 4096, heads, 64, 6, device=dev)
    kk = torch.randn(B, L0, heads, 64, device=dev, generator=g).half()
    a_codes, a_scales = gemm.quantize_mx(torch.randn(B * L0, C, device=dev, generator=g).half())
    w_codes, w_scales = gemm.quantize_mx(torch.randn(C, C, device=dev, generator=g) * 0.02)

    def kv_step():
        if cache.len + L0 > 4096:
            cache.len = cache._prev = 0
        cache.append(kk, kk)
    q_path = {"rotate_quant_mx": lambda: rot.rotate_quant_mx(x0.view(-1, C), smooth=smooth),
              "adaln_rotate_quant_mx": lambda: rot.adaln_rotate_quant_mx(x0, scale, shift, smooth=smooth),
              "adaln_rotate_quant_token": lambda: rot.adaln_rotate_quant_token(x0, scale, shift, "e2m3", smooth=smooth),
              "adaln_rotate_quant_token_fp8": lambda: rot.adaln_rotate_quant_token(x0, scale, shift, "e2m3", smooth=smooth, emit="fp8"),
              "adaln_rotate_quant_token_fp6": lambda: rot.adaln_rotate_quant_token(x0, scale, shift, "e2m3", smooth=smooth, emit="fp6"),
              "kv_cache_step": kv_step,
              "linear_fp4": lambda: gemm.linear_fp4(a_codes, a_scales, w_codes, w_scales)}
    out.append({"q_path_eager_us_at_the_first_step": {k: round(eager_us(f), 2) for k, f in q_path.items()}, "rows": B * L0})
    for pn in m["pn"]:
        L = pn * pn
        x = torch.randn(B, L, C, device=dev, generator=g)
        x = x if rows_dtype == "fp32" else x.half()
        a = torch.randn(B * L, C, device=dev, generator=g).half()
        hid = torch.nn.functional.gelu(torch.randn(B * L, 4 * C, device=dev, generator=g), approximate="tanh").half()
        out.append({"pn": pn, "rows": B * L,
                    "adaln_eager_us": round(eager_us(lambda: rot.adaln_rotate_quant(x, scale, shift, "e2m1", smooth=smooth)), 2),
                    "act_eager_us": round(eager_us(lambda: qu.fp_quant_e2_per_group_cuda(a, 4, 128)), 2),
                    "act_eager_ctypes_us": round(eager_us(lambda: ops.quant_rows(a, "e2m1", 128)), 2),
                    "dual_eager_us": round(eager_us(lambda: qu.fp_quant_e1m2_neg_e2m1_pos_per_group_cuda(hid, 4, 128)), 2),
                    "dual_eager_ctypes_us": round(eager_us(lambda: ops.quant_rows_dual(hid, "e1m2_neg", "e2m1_pos", 128, 1.0)), 2)})
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="d30", choices=sorted(bench.STEP_MODELS))
    ap.add_argument("--rows", default="fp32", choices=("fp16", "fp32"))
    ap.add_argument("--mode", default="rotating", choices=("rotating", "resident"))
    ap.add_argument("--ops", default=None, help="comma-separated subset of adaln,act,dual")
    ap.add_argument("--eager", action="store_true")
    ap.add_argument("--lib", default=None, help="a variant build of the library (tools/build_variant.sh) instead of the stock one")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    if a.lib:
        from fpqvar_amd import _lib
        _lib.LIB_PATH = os.path.abspath(a.lib)
        _lib.use_variant(a.lib)
    res = dict(stamp())
    if a.lib:
        res["build_tag"] = _lib.build_tag()
    res.update(bench.generation_steps(dev, a.model, a.rows, a.mode, only_ops=a.ops.split(",") if a.ops else None))
    if a.eager:
        res["eager"] = eager(dev, a.model, a.rows)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
