#!/usr/bin/env python3
"""F3 groundwork: the reference re-quantizes the WHOLE KV cache at every
step (tr/basic_var.py:186-209).  Quantizing each entry once when it first enters the cache would be
O(L) instead of O(L^2) traffic - but only equals the reference if re-quantization is idempotent.
This tool measures how far from idempotent it is, per mode, on synthetic K/V (unit-norm keys like
the attn_l2_norm path and Gaussian values): fraction of rows / elements that change when an already
quantized cache is quantized again, and the size of the change.
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from fpqvar_amd import kv_cache as kv  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    B, L, H, c = 100, 680, 30, 64
    res = {}
    for name, t in (("k_unit_norm", torch.nn.functional.normalize(torch.randn(B, L, H, c, device=dev), dim=-1).half()),
                    ("v_gauss", torch.randn(B, L, H, c, device=dev).half())):
        for bit, rowlen in ((6, 64), (4, 128)):
            q1 = kv.quantize_kv(t, bit)
            cur, changed_rows, changed_elems, max_rel = q1, [], [], 0.0
            for it in range(9):                    # the oldest entry is re-quantized up to 9 times (10 scale steps)
                nxt = kv.quantize_kv(cur, bit)
                diff = nxt.view(torch.int16) != cur.view(torch.int16)
                changed_elems.append(float(diff.float().mean()))
                changed_rows.append(float(diff.view(-1, rowlen).any(dim=1).float().mean()))
                d = (nxt.float() - cur.float()).abs() / cur.float().abs().clamp_min(1e-6)
                max_rel = max(max_rel, float(d[diff].max()) if diff.any() else 0.0)
                cur = nxt
            drift = (cur.float() - q1.float()).abs() / q1.float().abs().clamp_min(1e-6)
            res[f"{name}/kv_bit{bit}"] = {
                "rows_changed_by_2nd_pass": changed_rows[0], "elements_changed_by_2nd_pass": changed_elems[0],
                "rows_changed_by_10th_pass": changed_rows[-1], "max_relative_step": max_rel,
                "elements_differing_after_10_vs_1_pass": float((cur.view(torch.int16) != q1.view(torch.int16)).float().mean()),
                "max_relative_drift_after_10_passes": float(drift.max())}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
