#!/usr/bin/env python3
"""Rotate + quant at the metric shape, three ways on one MI355X (extra measurement, not bench.py):
  A  reference sequence: fp16 GEMM with the dense block-diagonal Q + the ~11 torch ops + L0 scan kernel
  B  fp16 GEMM + fused quant kernel
  C  fused rotate (FWHT-128 butterfly) + quant kernel
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from fpqvar_amd import ops, quant_utils as qu, rotation as rot  # noqa: E402

ROWS, C = 65536, 1920


def timeit(fn, iters=10):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


def main():
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    xs = [torch.randn(ROWS, C, device=dev).half() for _ in range(3)]
    q = rot.block_random_hadamard_matrix(C, 128, dev, 42).float()
    qh = q.half()
    grid = qu.fp4_e2m1_grid.to(dev)
    k = [0]

    def nxt():
        k[0] += 1
        return xs[k[0] % len(xs)]

    def seq_a():
        x1 = torch.matmul(nxt(), qh)
        xs_ = x1.reshape(-1, 128)
        scale = xs_.abs().max(dim=-1, keepdim=True)[0] / grid.abs().max()
        xn = (xs_ / scale).view(-1).to(torch.float32)
        z = ops.quant_nearest(xn, grid)
        torch.zeros_like(xn)
        return (z.view(xs_.shape) * scale).view(x1.shape).to(x1.dtype)

    def seq_b():
        return qu.fp_quant_e2_per_group_cuda(torch.matmul(nxt(), qh), 4, 128)

    def seq_c():
        return rot.rotate_quant(nxt(), "e2m1")

    res = {"shape": [ROWS, C], "A_gemm_plus_unfused_quant_ms": timeit(seq_a, 3), "B_gemm_plus_fused_quant_ms": timeit(seq_b),
           "C_fused_rotate_quant_ms": timeit(seq_c, 20), "gemm_only_ms": timeit(lambda: torch.matmul(nxt(), qh))}
    res["C_GBps_at_4B_per_elem"] = ROWS * C * 4 / (res["C_fused_rotate_quant_ms"] * 1e-3) / 1e9
    # the complete producer (tr/basic_var.py:263): LN, AdaLN modulate, smooth, rotate, quant
    B, L = 100, ROWS // 100
    xb = [torch.randn(B, L, C, device=dev).half() for _ in range(3)]
    scale = (torch.randn(B, 1, C, device=dev) * 0.3).half()
    shift = (torch.randn(B, 1, C, device=dev) * 0.3).half()
    s = torch.rand(C, device=dev) + 0.5

    def nxb():
        k[0] += 1
        return xb[k[0] % len(xb)]

    def chain_ref():
        x = nxb()
        with torch.autocast("cuda", dtype=torch.float16):
            ln = torch.nn.functional.layer_norm(x, (C,), eps=1e-6)
            x1 = torch.matmul(ln.mul(scale.add(1)).add_(shift).mul(s), q)
        xs_ = x1.reshape(-1, 128)
        sc = xs_.abs().max(dim=-1, keepdim=True)[0] / grid.abs().max()
        xn = (xs_ / sc).view(-1).to(torch.float32)
        z = ops.quant_nearest(xn, grid)
        torch.zeros_like(xn)
        return (z.view(xs_.shape) * sc).view(x1.shape).to(x1.dtype)

    res["D_torch_producer_chain_gemm_unfused_quant_ms"] = timeit(chain_ref, 3)
    res["E_fused_adaln_rotate_quant_ms"] = timeit(lambda: rot.adaln_rotate_quant(nxb(), scale, shift, "e2m1", smooth=s), 20)
    res["E_rows"] = B * L
    res["E_GBps_at_4B_per_elem"] = B * L * C * 4 / (res["E_fused_adaln_rotate_quant_ms"] * 1e-3) / 1e9
    print(json.dumps(res))


if __name__ == "__main__":
    main()
