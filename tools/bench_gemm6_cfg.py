import os, sys, torch
sys.path.insert(0, os.getcwd())
from fpqvar_amd import _lib, gemm
dev = torch.device("cuda:0"); torch.manual_seed(0)
T, K, O = 65536, 1920, 5760
x = torch.randn(T, K, device=dev).half(); w = torch.randn(O, K, device=dev) * 0.02
ac, asc = gemm.quantize_fp6(x); wc, wsc = gemm.quantize_fp6(w)
ref = None
for cfg in ("0", "1", "2"):
    _lib.set_option("FPQ_GEMM6_CFG", int(cfg))
    y = gemm.linear_fp6(ac, asc, wc, wsc)
    ref = y if ref is None else ref
    for _ in range(10): gemm.linear_fp6(ac, asc, wc, wsc)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): gemm.linear_fp6(ac, asc, wc, wsc)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 20)
    print(cfg, round(best, 4), "ms", round(2.0 * T * K * O / best / 1e9), "TFLOP/s", bool(torch.equal(y, ref)))
