#!/usr/bin/env python3
"""Debug aid for the matrix-core rotate: emit / no-emit outputs against the oracle, mismatch positions."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fpqvar_amd import rotation as rot
from oracle import fpq_oracle as orc

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(55)
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 300
x = (torch.randn(rows, 1920, generator=g) * torch.exp(0.5 * torch.randn(rows, 1920, generator=g))).half()
out, y = rot.rotate_quant(x.to(dev), "e2m1", return_rotated=True)
out2 = rot.rotate_quant(x.to(dev), "e2m1")
q_h = rot.block_random_hadamard_matrix(1920, 128, "cpu", 42).float().half()
y_ref = orc.rotate_fp16_reference(x, q_h)
yc = y.cpu()
print("y vs ref: mismatching", int((yc.view(torch.int16) != y_ref.view(torch.int16)).sum()), "of", yc.numel())
want = orc.per_group_kernel_sem(yc, "e2m1", 128)
want_ref = orc.per_group_kernel_sem(y_ref, "e2m1", 128)
for name, o in (("emit", out), ("noemit", out2)):
    oc = o.cpu()
    bad = (oc.view(torch.int16) != want.view(torch.int16)).reshape(-1)
    bad_ref = (oc.view(torch.int16) != want_ref.view(torch.int16)).reshape(-1)
    print(name, "vs quant(y_emitted):", int(bad.sum()), " vs quant(y_ref):", int(bad_ref.sum()))
    if bad.any():
        idx = bad.nonzero().reshape(-1)
        grp = idx // 128
        print("  first idx", idx[:8].tolist(), "groups hit", int(grp.unique().numel()), "of", yc.numel() // 128)
        print("  in-group positions hist (by 8):", torch.bincount((idx % 128) // 8, minlength=16).tolist())
        print("  tile-group hist (group % 32):", torch.bincount(grp.unique() % 32, minlength=32).tolist())

# ---- sign matrix probe: row b has a one at in-group position b of every group
xs = torch.zeros(128, 1920)
for b in range(128):
    xs[b, b::128] = 1.0
xs = xs.half()
_, ys = rot.rotate_quant(xs.to(dev), "e2m1", return_rotated=True)
ys = ys.cpu().float()
ref = orc.rotate_fp16_reference(xs, q_h).float()
bad = (ys != ref)
print("one-hot probe: wrong entries", int(bad.sum()), "of", bad.numel())
if bad.any():
    bb, cc = bad.nonzero(as_tuple=True)
    print("  wrong inputs b:", sorted(set(bb.tolist()))[:40])
    print("  wrong outputs o (mod 128):", sorted(set((cc % 128).tolist()))[:64])
    print("  wrong group idx:", sorted(set((cc // 128).tolist())))
    print("  sample", [(int(b), int(c), float(ys[b, c]), float(ref[b, c])) for b, c in list(zip(bb.tolist(), cc.tolist()))[:10]])
# magnitude of the random-input errors
d = (yc.float() - y_ref.float()).abs()
rel = d / y_ref.float().abs().clamp_min(1e-6)
print("random input: max abs err", float(d.max()), "max rel", float(rel.max()), "median rel of wrong", float(rel[rel > 0].median()))
