#!/usr/bin/env python3
"""fpq_kv_cache_step at the ten scale steps of a d30 batch (B = 100, H = 30, c = 64, FP6 E2M3 per head): us per call (hipGraph of 20
calls on rotating caches, median), bytes moved (quantized entries read + written, new entries read + written), fraction of 8 TB/s."""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from fpqvar_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
B, H, c = 100, 30, 64
pns = (1, 2, 3, 4, 5, 6, 8, 10, 13, 16)
max_len = sum(p * p for p in pns)
caches = [torch.randn(2, B, max_len, H, c, device=dev).half() for _ in range(3)]
pos, prev = 0, 0
print("# tokens(new)  quantized   us per call   MB moved   fraction of 8 TB/s")
for p in pns:
    n = p * p
    kv = [torch.randn(B, n, 3, H, c, device=dev).half() for _ in range(3)]
    it = [0]

    def call():
        i = it[0] % 3
        it[0] += 1
        ops.kv_cache_step(caches[i], prev, pos, kv[i][:, :, 1], kv[i][:, :, 2], pos, 64, "e2m3")

    call()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(21):
            call()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / 21)
    us = statistics.median(ts)
    mb = (2 * (pos - prev) + 2 * n) * B * H * c * 2 * 2 / 1e6
    print(f"{B * n:8d} {B * (pos - prev):10d} {us:12.1f} {mb:10.1f} {mb / us / 8e3 * 1e3 / 1e3:10.3f}")
    prev, pos = pos, pos + n
