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

from . import ops, quant_utils as qu


def quantize_kv(t: torch.Tensor, kv_bit: int) -> torch.Tensor:
    """basic_var.py:193-200.  kv_bit 6 needs a contiguous cache (the reference's
    x.view(-1) raises on the permuted BHLc layout too); the result is fp16."""
    if kv_bit == 6:
        return qu.fp6_quant_e2m3_per_token_cuda(t, kv_bit)
    if kv_bit == 4:
        return qu.fp_quant_e2_per_group_cuda(t, kv_bit)
    raise NotImplementedError


def quantize_kv_pair(k: torch.Tensor, v: torch.Tensor, kv_bit: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """quantize_kv on the cached K and the cached V (basic_var.py:193-200 quantizes them back to back): one launch for
    the pair when both are fp16 (fpq_quant_rows_multi), same results and the same layout rule as the single calls."""
    if k.dtype == v.dtype == torch.float16 and k.device == v.device and kv_bit in (4, 6):
        if kv_bit == 6:
            qu._require_viewable(k)
            qu._require_viewable(v)
            a, b = ops.quant_rows_multi([k, v], "e2m3", k.shape[-1], torch.float16) if k.shape[-1] == v.shape[-1] else \
                (quantize_kv(k, 6), quantize_kv(v, 6))
        else:
            a, b = ops.quant_rows_multi([k, v], "e2m1", 128)
        return a, b
    return quantize_kv(k, kv_bit), quantize_kv(v, kv_bit)


def update_kv_cache(cached_k: Optional[torch.Tensor], cached_v: Optional[torch.Tensor], k: torch.Tensor,
                    v: torch.Tensor, quant_KV: bool, kv_bit: int, dim_cat: int, check_finite: bool = True
                    ) -> Tuple[torch.Tensor, torch.Tensor]:
    """One caching step: returns the new (cached_k, cached_v), which are also the k / v
    attention runs on.  `check_finite` keeps the reference's asserts on the new k / v
    (they synchronise the host, as they do in the reference)."""
    if cached_k is None:
        return k, v
    if quant_KV:
        cached_k, cached_v = quantize_kv_pair(cached_k, cached_v, kv_bit)
        if check_finite:
            assert not torch.isnan(k).any(), "Tensor contains NaN values!"
            assert not torch.isinf(k).any(), "Tensor contains inf values!"
            assert not torch.isnan(v).any(), "Tensor contains NaN values!"
            assert not torch.isinf(v).any(), "Tensor contains inf values!"
    return torch.cat((cached_k, k), dim=dim_cat), torch.cat((cached_v, v), dim=dim_cat)


class IncrementalKVCache:
    """F3: the same K / V the reference's re-quantize-everything loop produces,
    with every cache entry quantized exactly ONCE.

    Why this is exact: re-quantizing an already fake-quantized row returns it unchanged (same scale,
    same levels) whenever the row's fp16 scale is a normal number, i.e. max|row| >= ~4e-4 - proven by
    exhaustion over every fp16 row maximum and every level in tests/test_kv_idempotence.py.  The
    reference quantizes the cache BEFORE appending the new k / v (tr/basic_var.py:192-209), so at
    step t the entries of step t-1 are quantized for the first time and all older ones are
    re-quantized to themselves; here only the former happens.  Rows below that magnitude (not seen
    with unit-norm keys / O(1) values) may differ in their last bits.

    Layout: flash layout [B, L, H, c] (`dim_cat` = 1), as the reference's published KV runs use; with
    kv_bit 4 a 128-group then never straddles tokens as long as H*c is a multiple of 128.
    Buffers are allocated once for `max_len` tokens; `append` returns views of the filled prefix.
    """

    def __init__(self, batch: int, max_len: int, heads: int, head_dim: int, kv_bit: int,
                 dtype=torch.float16, device="cuda"):
        assert kv_bit in (4, 6)
        assert kv_bit == 6 or (heads * head_dim) % 128 == 0
        self.kv_bit = kv_bit
        self.kv = torch.empty(2, batch, max_len, heads, head_dim, dtype=dtype, device=device)   # K and V in one slab
        self.k, self.v = self.kv[0], self.kv[1]
        self.len = 0
        self._prev = 0              # entries [_prev, len) were appended by the last step and are still unquantized
        # one launch per step (fpq_kv_cache_step) when the rows fit the fused kernels' lanes
        group = head_dim if kv_bit == 6 else 128
        self._group = group if (dtype == torch.float16 and group in (8, 16, 32, 64, 128, 256, 512)) else None

    @torch.no_grad()
    def commit_written(self, n: int) -> Tuple[torch.Tensor, torch.Tensor]:
        """`append` for k / v that the producer has ALREADY written into the cache's slots [len, len + n) (the qkv GEMM with a
        split output, gemm.linear_fp4_qkv_to_cache(..., cache.kv, cache.len, n)): quantizes the previous step's entries - one launch,
        nothing copied - and returns the same views."""
        assert self.len + n <= self.k.shape[1], "IncrementalKVCache: max_len exceeded"
        if self.len > self._prev:
            a, b = self._prev, self.len
            if self._group is not None:
                empty = self.kv[0, :, :0]
                ops.kv_cache_step(self.kv, a, b, empty, empty, self.len, self._group, "e2m3" if self.kv_bit == 6 else "e2m1")
            else:
                self.k[:, a:b].copy_(quantize_kv(self.k[:, a:b].contiguous(), self.kv_bit))
                self.v[:, a:b].copy_(quantize_kv(self.v[:, a:b].contiguous(), self.kv_bit))
        self._prev = self.len
        self.len += n
        return self.k[:, :self.len], self.v[:, :self.len]

    @torch.no_grad()
    def append(self, k: torch.Tensor, v: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """k, v: [B, n, H, c] (views of a fused qkv output are fine).  Returns the K / V attention runs on at this
        step: everything older than the previous step's entries as quantized before, the previous step's entries
        quantized now, the new ones as they are - what the reference's quantize-then-concatenate produces."""
        n = k.shape[1]
        assert self.len + n <= self.k.shape[1], "IncrementalKVCache: max_len exceeded"
        fused = (self._group is not None and k.dtype == torch.float16 and v.dtype == torch.float16
                 and k.stride() == v.stride() and k.stride(3) == 1 and k.stride(2) == k.shape[3]
                 and k.stride(0) % 8 == 0 and k.stride(1) % 8 == 0 and k.data_ptr() % 16 == 0 and v.data_ptr() % 16 == 0)
        if fused:
            ops.kv_cache_step(self.kv, self._prev, self.len, k, v, self.len, self._group, "e2m3" if self.kv_bit == 6 else "e2m1")
        else:
            if self.len > self._prev:             # what the reference's quantize-the-cache does NEW work on
                a, b = self._prev, self.len
                self.k[:, a:b].copy_(quantize_kv(self.k[:, a:b].contiguous(), self.kv_bit))
                self.v[:, a:b].copy_(quantize_kv(self.v[:, a:b].contiguous(), self.kv_bit))
            self.k[:, self.len:self.len + n].copy_(k)
            self.v[:, self.len:self.len + n].copy_(v)
        self._prev = self.len
        self.len += n
        return self.k[:, :self.len], self.v[:, :self.len]
