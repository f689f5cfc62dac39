import os, sys
sys.path.insert(0, os.getcwd())
import torch
from fpqvar_amd import _lib, gemm
dev = torch.device("cuda:0"); torch.manual_seed(0)
K = 1920
for T in (25600, 65536):
    for O in (1920, 5760, 7680):
        x = torch.randn(T, K, device=dev).half(); w = torch.randn(O, K, device=dev) * 0.02
        a = gemm.quantize_mx(x, kmajor=True); wc, ws = gemm.quantize_mx(w)
        wk = (gemm.to_kmajor(wc, 4, dealt=True), gemm.to_kmajor_scales(ws, weight_side=True))
        row = []
        for cfg in (10, 20):
            _lib.set_option("FPQ_GEMM_CFG", cfg)
            best = 1e9
            for _ in range(4):
                gemm.linear_fp4(*a, *wk, outs=O)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20): gemm.linear_fp4(*a, *wk, outs=O)
                e1.record(); torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / 20 * 1e3)
            row.append(best)
        print(f"T={T} O={O}: 256x128 {row[0]:.1f} us, 128x128 {row[1]:.1f} us ({row[0]/row[1]:.3f})")
