import ctypes, os, sys
sys.path.insert(0, "/root/repo")
import torch
from fpqvar_amd import _lib, ops
l = ctypes.CDLL(os.path.abspath(sys.argv[1]))
for name, (res, args) in _lib._SIGS.items():
    if hasattr(l, name):
        fn = getattr(l, name); fn.restype, fn.argtypes = res, args
_lib._lib = l
dev = torch.device("cuda:0")
xs = [torch.randn(65536, 1920, device=dev).half() for _ in range(4)]
k = [0]
def run(table, cols):
    k[0] += 1
    return ops.quant_rows(xs[k[0] % 4], table, cols, torch.float16)
def t(table, cols):
    best = 1e9
    for _ in range(3):
        for _ in range(10): run(table, cols)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(100): run(table, cols)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 100 * 1e3)
    return best
for cfg in (None, "2,10240", "2,12288", "2,15360", "2,16384", "2,20480", "2,24576"):
    if cfg: os.environ["FPQ_SYM_BIGTAB"] = cfg
    print(cfg, "e2m3 g128: %.1f us   e2m3 kv64: %.1f us   e2m1 g128 (unaffected): %.1f" % (t("e2m3", 128), t("e2m3", 64), t("e2m1", 128)), flush=True)
