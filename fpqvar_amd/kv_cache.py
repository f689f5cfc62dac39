"""KV-cache fake quantization as the reference does it inside SelfAttention.forward
(models_fp_quant_transform_rotate/basic_var.py:186-209): at every step after the first
the WHOLE cached K and V are re-quantized before the new k / v are appended -
kv_bit 6: FP6-E2M3, one scale per (token, head) row of head_dim (=64) channels;
kv_bit 4: FP4-E2M1 on consecutive groups of 128 elements of the flattened cache.
Both are single launches of the fused kernels (8 or 16 lanes own a row).
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from . import quant_utils as qu


def quantize_kv(t: torch.Tensor, kv_bit: int) -> torch.Tensor:
    """basic_var.py:193-200.  kv_bit 6 needs a contiguous cache (the reference's
    x.view(-1) raises on the permuted BHLc layout too); the result is fp16."""
    if kv_bit == 6:
        return qu.fp6_quant_e2m3_per_token_cuda(t, kv_bit)
    if kv_bit == 4:
        return qu.fp_quant_e2_per_group_cuda(t, kv_bit)
    raise NotImplementedError


def update_kv_cache(cached_k: Optional[torch.Tensor], cached_v: Optional[torch.Tensor], k: torch.Tensor,
                    v: torch.Tensor, quant_KV: bool, kv_bit: int, dim_cat: int, check_finite: bool = True
                    ) -> Tuple[torch.Tensor, torch.Tensor]:
    """One caching step: returns the new (cached_k, cached_v), which are also the k / v
    attention runs on.  `check_finite` keeps the reference's asserts on the new k / v
    (they synchronise the host, as they do in the reference)."""
    if cached_k is None:
        return k, v
    if quant_KV:
        cached_k = quantize_kv(cached_k, kv_bit)
        cached_v = quantize_kv(cached_v, kv_bit)
        if check_finite:
            assert not torch.isnan(k).any(), "Tensor contains NaN values!"
            assert not torch.isinf(k).any(), "Tensor contains inf values!"
            assert not torch.isnan(v).any(), "Tensor contains NaN values!"
            assert not torch.isinf(v).any(), "Tensor contains inf values!"
    return torch.cat((cached_k, k), dim=dim_cat), torch.cat((cached_v, v), dim=dim_cat)
