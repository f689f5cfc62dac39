#!/usr/bin/env python3
"""Compare the variants of tools/probe/gelu_probe.hip with torch's F.gelu(x, approximate="tanh") on this GPU, all 65536 fp16 inputs.
usage: gelu_probe_check.py gelu_probe.bin"""
import sys

import numpy as np
import torch

NAMES = ["0 torch order, fma, devlib tanh restated", "1 torch order, no fma", "2 torch order, fma, tanhf()", "3 exp branch only, bare exp2",
         "4 sigmoid + snap (torch u)", "5 sigmoid, no snap (torch u)", "6 sigmoid + snap, 3-op argument", "7 exp branch only, hi/lo exp2"]
raw = np.fromfile(sys.argv[1], dtype=np.uint16).reshape(-1, 65536)
x = torch.arange(0, 65536, dtype=torch.int32).to(torch.int16).view(torch.float16).cuda()
ref = torch.nn.functional.gelu(x, approximate="tanh").cpu()


def ordered(h):
    b = h.view(torch.int16).to(torch.int32) & 0xFFFF
    return torch.where(b >= 0x8000, 0x8000 - b, b)


xs = x.cpu().float()
for v in range(raw.shape[0]):
    got = torch.from_numpy(raw[v].view(np.int16).copy()).view(torch.float16)
    nan_ok = bool(torch.equal(torch.isnan(got), torch.isnan(ref)))
    fin = ~torch.isnan(ref) & ~torch.isnan(got)
    d = (ordered(got) - ordered(ref)).abs()[fin]
    hist = {int(k): int((d == k).sum()) for k in torch.unique(d)}
    bad = fin.nonzero().flatten()[d > 1]
    where = xs[bad]
    print(f"variant {NAMES[v]}: NaN pattern {'equal' if nan_ok else 'DIFFERS'}; ulp histogram {hist}; "
          f"> 1 ulp on {bad.numel()} inputs" + (f" in x = [{float(where.min()):.4g}, {float(where.max()):.4g}]" if bad.numel() else ""))
