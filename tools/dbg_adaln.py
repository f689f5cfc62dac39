import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fpqvar_amd import rotation as rot
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(81)
B, L, C = 3, 40, 1920
x = (torch.randn(B, L, C, generator=g) * 2 + 0.3).half().to(dev)
scale = (torch.randn(B, 1, C, generator=g) * 0.3).half().to(dev)
shift = (torch.randn(B, 1, C, generator=g) * 0.3).half().to(dev)
s = (torch.rand(C, generator=g) * 1.5 + 0.25).to(dev)
for sm in (None, s):
    out, h, y = rot.adaln_rotate_quant(x, scale, shift, "e2m1", smooth=sm, return_intermediates=True)
    ln = torch.nn.functional.layer_norm(x.float(), (C,), eps=1e-6)
    t32 = (ln.mul(scale.add(1)) + shift)
    if sm is not None:
        t32 = t32.mul(sm)
    err = (h.float() - t32).abs()
    print("smooth", sm is not None, "max err", float(err.max()), "mean err", float(err.mean()))
    e = err.view(B * L, C)
    print(" per-row max (first 20 rows):", [round(float(v), 4) for v in e.max(dim=1)[0][:20]])
    print(" per-vector-slot max (c=0..3):", [round(float(e.view(B*L, 30, 64)[:, 8*c:8*c+8 if False else None].max()), 4) for c in range(1)])
    ev = e.view(B * L, 240, 8)
    print(" by vector index /64:", [round(float(ev[:, 64*c:64*c+64].max()), 4) for c in range(4)])
    print(" by element in vector:", [round(float(ev[:, :, k].max()), 4) for k in range(8)])
    # ratio h / t32 for a few
    r = (h.float() / t32).view(B*L, C)
    print(" ratio row0 first 8:", [round(float(v), 4) for v in r[0, :8]], " row1:", [round(float(v), 4) for v in r[1, :8]])
    hm = h.float().view(B*L, C)
    print(" implied: mean(ratio) per row first 6:", [round(float(v), 4) for v in r.median(dim=1)[0][:6]])
