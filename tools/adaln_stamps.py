#!/usr/bin/env python3
"""Where a wavefront's row time goes in the adaLN producer: needs a library built with -DFPQ_ADALN_STAMPS
(tools/build_variant.sh stamps -DFPQ_ADALN_STAMPS), whose kernel sums s_memtime differences per phase and wavefront.
usage: adaln_stamps.py tools/ab/libstamps.so [fp16|fp32] [B L C] [cold]
"cold": the stamped launch works on a tensor the caches have not seen (a 512 MiB write in front of it), as a launch of a
hipGraph over rotating inputs does (bench.generation_steps); default: the 30th launch on the same tensor."""
import ctypes
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fpqvar_amd import _lib, rotation as rot

lib = ctypes.CDLL(os.path.abspath(sys.argv[1]))
for name, (res, args) in _lib._SIGS.items():
    if hasattr(lib, name):
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
dt = sys.argv[2] if len(sys.argv) > 2 else "fp16"
B, L, C = (int(a) for a in sys.argv[3:6]) if len(sys.argv) > 5 else (100, 655, 1920)
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn(B, L, C, device=dev, generator=g)
x = x.half() if dt == "fp16" else x
scale = (torch.randn(B, C, device=dev, generator=g) * 0.3).half()
shift = (torch.randn(B, C, device=dev, generator=g) * 0.3).half()
s = torch.rand(C, device=dev, generator=g) + 0.5
cold = len(sys.argv) > 6 and sys.argv[6] == "cold"
x_cold = torch.randn(B, L, C, device=dev, generator=g).to(x.dtype) if cold else None
flush = torch.empty(1 << 29, dtype=torch.uint8, device=dev) if cold else None
out = torch.empty(B, L, C, dtype=torch.float16, device=dev)
stamps = torch.zeros(16 * (1 << 19), dtype=torch.int64, device=dev)
mask = rot._mask_arg(None)
for it in range(30):
    if it == 29:
        stamps.zero_()
        if cold:
            flush.zero_()
            x = x_cold
    _lib.check(lib.fpq_adaln_rotate_quant_rows(x.data_ptr(), out.data_ptr(), None, stamps.data_ptr(), B * L, C,
                                               _lib.dtype_id(x.dtype), scale.data_ptr(), shift.data_ptr(), _lib.F16, L, 1e-6,
                                               s.data_ptr(), mask, _lib.TABLE_IDS["e2m1"], _lib.stream_ptr(dev)), "stamps")
torch.cuda.synchronize()
st = stamps.view(-1, 16).cpu()
st = st[st[:, 8] > 0].double()
rows = st[:, 8].sum()
names = ["between rows / prologue", "wait for the row (vmcnt 0)", "statistics + rstd", "modulate -> image (+ prefetch issue)",
         "MFMA + butterfly + round", "group max + scale", "divide + level + dequant", "store tile"]
tot = st[:, :8].sum()
print(f"{dt} [{B}x{L}x{C}]: {int(rows)} rows on {st.shape[0]} wavefronts; {tot / rows:.0f} cycles per row and wavefront")
for k, n in enumerate(names):
    print(f"  {n:40s} {st[:, k].sum() / rows:8.0f} cycles per row  {100 * st[:, k].sum() / tot:5.1f} %")

# ---- residency timeline: wavefronts alive per CU over the launch (s_memrealtime: 10 ns ticks, chip-wide) ----
import collections
t0, t1 = st[:, 10], st[:, 11]
hw, xcc = st[:, 12].long(), st[:, 13].long() & 0xF
cu = (hw >> 8) & 0xF
sh = (hw >> 12) & 0x1
se = (hw >> 13) & 0x7
simd = (hw >> 4) & 0x3
cu_key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
start, end = float(t0.min()), float(t1.max())
print(f"launch: {(end - start) * 0.01:.1f} us from the first wavefront's start to the last one's end; wavefront lifetime "
      f"mean {float((t1 - t0).mean()) * 0.01:.1f} us, min {float((t1 - t0).min()) * 0.01:.1f}, max {float((t1 - t0).max()) * 0.01:.1f}")
print(f"distinct CUs seen: {len(set(cu_key.tolist()))}; wavefronts per CU: min {min(collections.Counter(cu_key.tolist()).values())} "
      f"max {max(collections.Counter(cu_key.tolist()).values())}")
# when the wavefronts start, get their first row, and end (us from the first start; percentiles over the wavefronts)
if st.shape[1] > 14 and float(st[:, 14].max()) > 0:
    def pct(v):
        q = torch.quantile((v - start) * 0.01, torch.tensor([0.0, 0.1, 0.25, 0.5, 0.75, 0.9, 1.0], dtype=torch.float64))
        return " ".join(f"{float(x):5.1f}" for x in q)
    print("percentiles over the wavefronts      min   10%   25%   50%   75%   90%   max   (us after the first wavefront's start)")
    print("  start                            " + pct(t0))
    print("  first row has arrived            " + pct(st[:, 14]))
    print("  end                              " + pct(t1))
    late = t0 > (t0.min() + 100)   # started more than 1 us after the first: a later generation
    print(f"  wavefronts of later generations: {int(late.sum())} of {st.shape[0]}; their wait for the first row: "
          f"median {float(((st[:, 14] - t0)[late]).median()) * 0.01 if late.any() else 0:.1f} us against "
          f"{float(((st[:, 14] - t0)[~late]).median()) * 0.01:.1f} us for the first generation")
nb = 20
edges = [start + (end - start) * k / nb for k in range(nb + 1)]
print("resident wavefronts per CU (average over the chip) by twentieth of the launch:")
row = []
for k in range(nb):
    a, b = edges[k], edges[k + 1]
    overlap = (torch.clamp(t1, max=b) - torch.clamp(t0, min=a)).clamp_min(0).sum()
    row.append(float(overlap) / (b - a) / 256)
print("  " + " ".join(f"{v:4.1f}" for v in row))
mid = start + 0.5 * (end - start)
alive = ((t0 <= mid) & (t1 >= mid))
per_cu = collections.Counter(cu_key[alive].tolist())
vals = sorted(per_cu.values())
print(f"at mid-launch: {int(alive.sum())} wavefronts alive on {len(per_cu)} CUs; per CU min {vals[0]} median {vals[len(vals) // 2]} max {vals[-1]}")
per_simd = collections.Counter((cu_key[alive] * 4 + simd[alive]).tolist())
sv = sorted(per_simd.values())
print(f"               per SIMD min {sv[0]} median {sv[len(sv) // 2]} max {sv[-1]} ({len(per_simd)} SIMDs with at least one)")
