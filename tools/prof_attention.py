#!/usr/bin/env python3
"""A few launches of fpq_attention_blhc at the largest d30 step, for rocprofv3 --pmc runs
(e.g. --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE, or SQ_INSTS_VALU SQ_INSTS_MFMA ...)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from fpqvar_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
B, H, Lq, Lkv = 100, 30, 256, 680
q = F.normalize(torch.randn(B, Lq, H, 64, device=dev), dim=-1).mul(8).half()
k = F.normalize(torch.randn(B, Lkv, H, 64, device=dev), dim=-1).half()
v = torch.randn(B, Lkv, H, 64, device=dev).half()
for _ in range(3):
    ops.attention_blhc(q, k, v, 1.0)
torch.cuda.synchronize()
