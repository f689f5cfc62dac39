"""Time the weight-calibration forms of VAR-d30 at world size 1 (profiles/r03_calib_codes_n1.txt)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fpqvar_amd import calibrate as cal, ops
dev = torch.device("cuda:0")
shapes = cal.var_linear_shapes(30)
torch.manual_seed(0)
w = {n: torch.randn(*s, device=dev) * 0.02 for n, s in shapes.items()}
def t(fn, it=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(it): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / it * 1e3
sc = cal.ShardedCalibration(shapes, w)
print("fp16 slab, one launch: %.2f ms" % t(sc.run))
print("codes exchange path (world 1: quantize to codes + decode): %.2f ms" % t(lambda: cal.calibrate_sharded(w, exchange="codes")))
cc = cal.ShardedCodesCalibration(w)
print("  prebuilt ShardedCodesCalibration.run() (two launches): %.2f ms" % t(cc.run))
names = list(w)
def q_only():
    return [ops.quant_rows_codes(w[n], "e2m1", 128, pack_nibbles=True) for n in names]
print("  quantize to codes, 120 launches: %.2f ms" % t(q_only))
cs = q_only()
def d_only():
    return [ops.dequant_rows_codes(c.view(-1, 64), s.reshape(-1), "e2m1", 128, torch.float16, True) for c, s in cs]
print("  decode, 120 launches: %.2f ms" % t(d_only))
big = w[names[2]]
print("  one layer", names[2], tuple(big.shape), "quant codes %.1f us, decode %.1f us" % (
    t(lambda: ops.quant_rows_codes(big, "e2m1", 128, pack_nibbles=True), 50) * 1e3,
    t(lambda: ops.dequant_rows_codes(cs[2][0].view(-1, 64), cs[2][1].reshape(-1), "e2m1", 128, torch.float16, True), 50) * 1e3))
