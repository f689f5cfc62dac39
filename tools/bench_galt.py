#!/usr/bin/env python3
"""One GALT optimisation step at VAR-d30 mat_qkv shape (learnable_transformation_mat_qkv_fp4.py:267-304):
x [tokens, 1920] fp32 calibration activations, W [5760, 1920], Q block-Hadamard, AdamW on s.
Compares the fused STE quantizer (one launch) with the reference's distance-tensor argmin in torch ops."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from fpqvar_amd import galt, rotation as rot  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    tokens = int(sys.argv[1]) if len(sys.argv) > 1 else 25600
    x = torch.randn(tokens, 1920, device=dev) * torch.exp(0.5 * torch.randn(tokens, 1920, device=dev))
    w = torch.randn(5760, 1920, device=dev) * 0.02
    q = rot.block_random_hadamard_matrix(1920, 128, dev, 42).float()
    grid = torch.tensor([-6.0, -4.0, -3.0, -2.0, -1.5, -1.0, -0.5, 0.0, 0.5, 1.0, 1.5, 2.0, 3.0, 4.0, 6.0], device=dev)

    def torch_fpquant(t):
        def fwd(v):
            shape = v.shape
            v = v.reshape(-1, 128)
            scale = v.abs().max(dim=-1, keepdim=True)[0] / grid.abs().max()
            v = v / scale
            idx = torch.argmin(torch.abs(v.unsqueeze(-1) - grid), dim=-1)
            return (grid[idx] * scale).view(shape)
        return galt._STE.apply(t, fwd)

    def steps(n, **kw):
        s = torch.nn.Parameter(torch.ones(1920, device=dev))
        opt = torch.optim.AdamW([s], lr=0.01)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            loss = galt.compute_quant_error(x, w, s, q, "fp4", **kw)
            loss.backward()
            opt.step()
            opt.zero_grad()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3, float(loss.detach())

    steps(2)
    fused_ms, lf = steps(10)
    steps(1, act_quant=torch_fpquant, weight_quant=torch_fpquant)
    ref_ms, lr = steps(3, act_quant=torch_fpquant, weight_quant=torch_fpquant)
    print(json.dumps({"tokens": tokens, "fused_step_ms": round(fused_ms, 3), "torch_argmin_step_ms": round(ref_ms, 3),
                      "loss_fused": lf, "loss_torch": lr, "peak_mem_GB": round(torch.cuda.max_memory_allocated() / 1e9, 2)}))


if __name__ == "__main__":
    main()
