#!/usr/bin/env python3
"""Run ONE kernel a few times (for rocprofv3 --pmc passes).
usage: prof_one.py {adaln|adaln32|adaln_codes|rotate|rotate_smooth|rotate_codes|dual|dual6|token6|group6|sym|calib|channel|gemm|gemm6|gemm8|fc1|gemm_km|gemm6_km|fc1_km}   (_km: operands as k-major images, include/fpq.h)   (FPQ_ADALN_V1=1: the round-1 adaLN kernel)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if os.environ.get("FPQ_PROF_LIB"):
    os.environ["FPQ_NO_NATIVE"] = "1"   # the compiled binding is linked to the stock library: keep the wrappers on ctypes
import torch  # noqa: E402

from fpqvar_amd import _lib, ops, quant_utils as qu, rotation as rot  # noqa: E402

if os.environ.get("FPQ_PROF_LIB"):   # a variant build of the library (tools/build_variant.sh), for A/B profiling
    _lib.use_variant(os.environ["FPQ_PROF_LIB"])
    print("library:", os.environ["FPQ_PROF_LIB"], "build tag:", _lib.build_tag(), file=sys.stderr, flush=True)

which = sys.argv[1] if len(sys.argv) > 1 else "sym"
dev = torch.device("cuda:0")
torch.manual_seed(0)
C = 1920
_k = [0]


def nxt(ts):   # three inputs in turn: a single 252 MB tensor would be served partly from the 256 MiB Infinity Cache
    _k[0] += 1
    return ts[_k[0] % len(ts)]


if which in ("adaln", "adaln32"):
    B, L = 100, 655
    xs = [torch.randn(B, L, C, device=dev) for _ in range(3)]
    if which == "adaln":
        xs = [t.half() for t in xs]
    scale = (torch.randn(B, 1, C, device=dev) * 0.3).half()
    shift = (torch.randn(B, 1, C, device=dev) * 0.3).half()
    s = torch.rand(C, device=dev) + 0.5
    fn = lambda: rot.adaln_rotate_quant(nxt(xs), scale, shift, "e2m1", smooth=s)
elif which == "adaln_codes":    # the producer writing packed E2M1 codes + one fp16 scale per group (2.53 B per element)
    B, L = 100, 655
    xs = [torch.randn(B, L, C, device=dev).half() for _ in range(3)]
    scale = (torch.randn(B, 1, C, device=dev) * 0.3).half()
    shift = (torch.randn(B, 1, C, device=dev) * 0.3).half()
    fn = lambda: rot.adaln_rotate_quant_mx(nxt(xs), scale, shift)
elif which == "rotate_codes":
    xs = [torch.randn(65536, C, device=dev).half() for _ in range(3)]
    fn = lambda: rot.rotate_quant_mx(nxt(xs))
elif which in ("gemm", "gemm_km"):
    from fpqvar_amd import gemm
    x = torch.randn(65536, C, device=dev).half()
    w = torch.randn(5760, C, device=dev) * 0.02
    ac, asc = gemm.quantize_mx(x, kmajor=which == "gemm_km")
    wc, wsc = gemm.quantize_mx(w)
    if which == "gemm_km":
        wc, wsc = gemm.to_kmajor(wc, 4, dealt=True), gemm.to_kmajor_scales(wsc, weight_side=True)
    fn = lambda: gemm.linear_fp4(ac, asc, wc, wsc)
elif which in ("fc1", "fc1_km"):   # fc1 with GELU and fc2's dual quantizer in the GEMM's epilogue (fpq_gemm_fp4_gelu_dual), VAR-d30's shape
    from fpqvar_amd import gemm
    x = torch.randn(65536, C, device=dev).half()
    w = torch.randn(7680, C, device=dev) * 0.02
    bias = (torch.randn(7680, device=dev) * 0.1).half()
    ac, asc = gemm.quantize_mx(x, kmajor=which == "fc1_km")
    wc, wsc = gemm.quantize_mx(w)
    if which == "fc1_km":
        wc, wsc = gemm.to_kmajor(wc, 4, dealt=True), gemm.to_kmajor_scales(wsc, weight_side=True)
    fn = lambda: gemm.linear_fp4_gelu_dual(ac, asc, wc, wsc, bias)
elif which in ("gemm6", "gemm8", "gemm6_km"):   # the row-scaled GEMMs of the W6A6 configuration (6-bit packed / E4M3 bytes)
    from fpqvar_amd import gemm
    x = torch.randn(65536, C, device=dev).half()
    w = torch.randn(5760, C, device=dev) * 0.02
    quant, lin = (gemm.quantize_fp6, gemm.linear_fp6) if which != "gemm8" else (gemm.quantize_fp8, gemm.linear_fp8)
    ac, asc = quant(x, kmajor=True) if which == "gemm6_km" else quant(x)
    wc, wsc = quant(w)
    if which == "gemm6_km":
        wc = gemm.to_kmajor(wc, 6, dealt=True)
    fn = lambda: lin(ac, asc, wc, wsc)
elif which == "calib":
    from fpqvar_amd import calibrate as cal
    shapes = cal.var_linear_shapes(30)
    names = list(shapes)[:16]                      # four blocks of VAR-d30: 177 M fp32 weights
    w = {n: torch.randn(*shapes[n], device=dev) * 0.02 for n in names}
    shard = cal.LocalShard(w, shapes)
    fn = shard.quantize
elif which == "channel":
    x = torch.randn(32768, C, device=dev) * 0.02
    fn = lambda: ops.quant_rows(x, "e2m3", C, torch.float16)
elif which == "token6":   # per-token E2M3 on rows of 1920 (the W6A6 activations): one wavefront per row, 2 x 512-bucket table in LDS
    xs = [torch.randn(65536, C, device=dev).half() for _ in range(3)]
    fn = lambda: ops.quant_rows(nxt(xs), "e2m3", C, torch.float16)
elif which == "group6":   # per-group(128) E2M3
    xs = [torch.randn(65536, C, device=dev).half() for _ in range(3)]
    fn = lambda: ops.quant_rows(nxt(xs), "e2m3", 128, torch.float16)
elif which == "dual6":
    x = torch.nn.functional.gelu(torch.randn(65536, 4 * C, device=dev)).half()
    fn = lambda: ops.quant_rows_dual(x, "int_neg", "e2m3_pos", 128, None)
elif which == "rotate":
    xs = [torch.randn(65536, C, device=dev).half() for _ in range(3)]
    fn = lambda: rot.rotate_quant(nxt(xs), "e2m1")
elif which == "rotate_smooth":   # with a GALT smoothing vector: the SMOOTH = 1 instantiation (one resident workgroup fewer per CU)
    xs = [torch.randn(65536, C, device=dev).half() for _ in range(3)]
    s = torch.rand(C, device=dev) + 0.5
    fn = lambda: rot.rotate_quant(nxt(xs), "e2m1", smooth=s)
elif which == "dual":
    x = torch.nn.functional.gelu(torch.randn(65536, C, device=dev)).half()
    fn = lambda: ops.quant_rows_dual(x, "e1m2_neg", "e2m1_pos", 128, None)
else:
    xs = [torch.randn(65536, C, device=dev).half() for _ in range(3)]
    fn = lambda: qu.fp_quant_e2_per_group_cuda(nxt(xs), 4, 128)
for _ in range(6):
    fn()
torch.cuda.synchronize()
