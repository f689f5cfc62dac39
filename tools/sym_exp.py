"""Experiment driver behind profiles/r03_e2m3_grid.txt: E2M3 quantizers under grid variations.  The environment hooks it
sets (FPQ_SYM_BIGTAB="U,cap", FPQ_WAVE_CAP, FPQ_BLOCK_RPB) existed only in a temporary diagnostic build of fpq_kernels.hip
(three getenv lines in fpq_quant_rows / launch_fast16_block); the measured choices are the defaults now, so against the
regular library every line of this script prints the same figure.      python tools/sym_exp.py <libfpq_hip.so>"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["FPQ_NO_NATIVE"] = "1"
import torch
from fpqvar_amd import _lib, ops
_lib.use_variant(sys.argv[1])
dev = torch.device("cuda:0")
xs = [torch.randn(65536, 1920, device=dev).half() for _ in range(4)]
hs = [torch.randn(16384, 7680, device=dev).half() for _ in range(4)]
k = [0]
def t(fn):
    best = 1e9
    for _ in range(3):
        for _ in range(10): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(100): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 100 * 1e3)
    return best
def nxt(l):
    k[0] += 1
    return l[k[0] % 4]
for cap in (None, "4096", "8192", "12288", "16384"):
    if cap: os.environ["FPQ_WAVE_CAP"] = cap
    print("wave cap", cap, "per token e2m3 [65536x1920]: %.1f us   dual int-/e2m3+ per token: n/a" % t(lambda: ops.quant_rows(nxt(xs), "e2m3", 1920, torch.float16)), flush=True)
for rpb in (None, "1", "2", "3", "4"):
    if rpb: os.environ["FPQ_BLOCK_RPB"] = rpb
    print("block rows per workgroup", rpb, "per token e2m3 [16384x7680]: %.1f us   dual int-/e2m3+ per token: %.1f us" % (
        t(lambda: ops.quant_rows(nxt(hs), "e2m3", 7680, torch.float16)), t(lambda: ops.quant_rows_dual(nxt(hs), "int_neg", "e2m3_pos", 7680, None))), flush=True)
