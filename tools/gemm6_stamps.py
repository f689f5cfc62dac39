#!/usr/bin/env python3
"""Where a wavefront's time goes in the FP6 row-scaled GEMM: needs tools/ab/libstamps6.so (tools/build_variant.sh --gemm stamps6
-DFPQ_GEMM6_STAMPS), whose kernel sums s_memtime differences per phase of a K step and wavefront.
usage: gemm6_stamps.py tools/ab/libstamps6.so [kmajor] [tokens outs k]      kmajor: operands as k-major images (include/fpq.h)"""
import ctypes
import os
import statistics
import sys

os.environ["FPQ_NO_NATIVE"] = "1"   # the compiled binding is linked to the stock library
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from fpqvar_amd import _lib, gemm  # noqa: E402

lib = _lib.use_variant(sys.argv[1])   # (a GEMM-unit variant keeps the quantizer unit's build tag: the stamp entry point identifies it)
assert hasattr(lib, "fpq_debug_gemm6_stamp_buffer"), "not a -DFPQ_GEMM6_STAMPS build"
lib.fpq_debug_gemm6_stamp_buffer.restype, lib.fpq_debug_gemm6_stamp_buffer.argtypes = ctypes.c_int, [ctypes.c_void_p]
KM = "kmajor" in sys.argv[2:]
nums = [a for a in sys.argv[2:] if a != "kmajor"]
T, O, K = (int(a) for a in nums[:3]) if len(nums) >= 3 else (65536, 5760, 1920)
dev = torch.device("cuda:0")
torch.manual_seed(0)
a = gemm.quantize_fp6(torch.randn(T, K, device=dev).half(), kmajor=KM)
w = gemm.quantize_fp6(torch.randn(O, K, device=dev) * 0.02)
if KM:
    w = (gemm.to_kmajor(w[0], 6, dealt=True), w[1])
n_wg = 8 * ((O // 128 + 7) // 8) * ((T + 255) // 256)
buf = torch.zeros(n_wg * 4 * 8, dtype=torch.int64, device=dev)
for _ in range(20):
    gemm.linear_fp6(*a, *w)
torch.cuda.synchronize()
assert lib.fpq_debug_gemm6_stamp_buffer(buf.data_ptr()) == 0
buf.zero_()
gemm.linear_fp6(*a, *w)
torch.cuda.synchronize()
st = buf.view(-1, 8).cpu()
st = st[st[:, 7] == 1].double()
names = ["wait for the stage (s_waitcnt vmcnt(0))", "barrier", "issue the next stage's LDS-DMA pieces", "fragment reads + 32 MFMAs", "prologue (scale tiles, first stage issue)", "epilogue (scales, bias, stores)"]
tot = st[:, :6].sum(dim=1)
steps = int(st[0, 6])
print(f"# FP6 GEMM [{T} x {K}] . [{K} -> {O}], {'k-major images' if KM else 'row-major codes'}, library {_lib.build_tag()}: {st.shape[0]} wavefronts with work, {steps} K steps each; s_memtime cycles per wavefront (median), share of its lifetime")
for k, n in enumerate(names):
    per = st[:, k] / (steps if k < 4 else 1)
    print(f"  {n:48s} {statistics.median(per.tolist()):9.0f} cycles per {'step' if k < 4 else 'tile'}   {100 * float((st[:, k] / tot).mean()):5.1f} %")
print(f"  wavefront lifetime (median) {statistics.median(tot.tolist()):.0f} cycles; MFMA pipe time inside a step: 32 x 16 = 512 cycles")
