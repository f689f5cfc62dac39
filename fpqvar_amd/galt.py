"""GALT (learnable per-channel smoothing `s`) on the fused kernels - the training-like workload of the
reference (learnable_transformation/learnable_transformation_{mat_qkv,fc1}_{fp4,fp6}.py; SURVEY.md
section 8f, F4).

The reference minimises  mean((x W^T - Q_a(x*s @ Q) Q_w(W/s @ Q)^T)^2)  over `s` with AdamW(lr 0.01),
passing gradients straight through the quantizers (STE).  Its quantizers are

  * FP4 scripts  : ``FPQuant`` - the pure-torch argmin lookup, per-group 128, float32 end to end
                   (..._mat_qkv_fp4.py:75-100), which materialises a [N, 15] distance tensor per call;
  * FP6 scripts  : ``FP6Quant_activation_per_token`` / ``FP6Quant_weight`` - the ``_cuda`` op sequence
                   on the E2M3 table with an fp16 result (..._mat_qkv_fp6.py:245-296).

Here the forward of each is ONE launch (`fpq_quant_rows_argmin` / `fpq_quant_rows`) wrapped in an
autograd Function whose backward is the reference's identity; everything else (the two transforms, the
three matmuls, AdamW) stays in torch, exactly as there.  Blocks are independent, so `learn_blocks_sharded`
gives block b to rank b % world and ends with one all-gather of the learned vectors.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence

import torch
import torch.distributed as dist

from . import ops


class _STE(torch.autograd.Function):
    """y = fn(x) forward, dL/dx = dL/dy backward (the reference's `grad_output.clone()`)."""

    @staticmethod
    def forward(ctx, x, fn):
        return fn(x.detach())

    @staticmethod
    def backward(ctx, grad_output):
        return grad_output.clone(), None


def FPQuant(x: torch.Tensor, n_bits: int = 4, group_size: int = 128) -> torch.Tensor:
    """..._mat_qkv_fp4.py:75-100: per-group E2M1 through the argmin lookup, float32 result."""
    assert n_bits == 4
    return _STE.apply(x, lambda t: ops.quant_rows_argmin(t, "e2m1", group_size, False).to(t.dtype))


def FP6Quant_activation(x: torch.Tensor, n_bits: int = 6, group_size: int = 128) -> torch.Tensor:
    """..._mat_qkv_fp6.py:210-243: per-group E2M3, fp16 result."""
    assert n_bits == 6
    return _STE.apply(x, lambda t: ops.quant_rows(t, "e2m3", group_size, torch.float16))


def FP6Quant_activation_per_token(x: torch.Tensor, n_bits: int = 6) -> torch.Tensor:
    """..._mat_qkv_fp6.py:245-270: one scale per token, E2M3, fp16 result."""
    assert n_bits == 6
    return _STE.apply(x, lambda t: ops.quant_rows(t, "e2m3", t.shape[-1], torch.float16))


def FP6Quant_weight(x: torch.Tensor, n_bits: int = 6) -> torch.Tensor:
    """..._mat_qkv_fp6.py:273-296: one scale per output channel, E2M3, fp16 result."""
    assert n_bits == 6
    return _STE.apply(x, lambda t: ops.quant_rows(t, "e2m3", t.shape[-1], torch.float16))


def compute_quant_error(x, w, learnable_s, Q, fmt: str = "fp4",
                        act_quant: Optional[Callable] = None, weight_quant: Optional[Callable] = None):
    """compute_quant_error_v1 of the reference (fp4: ..._mat_qkv_fp4.py:122-138, fp6: ..._fp6.py:316-333).
    act_quant / weight_quant override the quantizers (CPU tests of the loop logic)."""
    if act_quant is None:
        act_quant = FPQuant if fmt == "fp4" else FP6Quant_activation_per_token
    if weight_quant is None:
        weight_quant = FPQuant if fmt == "fp4" else FP6Quant_weight
    fp_result = torch.matmul(x, w.T)
    x_2_quant = act_quant(torch.matmul(x * learnable_s, Q))
    w_2_quant = weight_quant(torch.matmul(w / learnable_s, Q))
    quant_result = torch.matmul(x_2_quant, w_2_quant.T)
    return torch.mean((fp_result - quant_result) ** 2)


def learn_s(activations: Sequence[torch.Tensor], weight: torch.Tensor, Q: torch.Tensor, epochs: int = 50,
            lr: float = 0.01, fmt: str = "fp4", snapshot_best: bool = False, log: Optional[List[float]] = None,
            **quantizers) -> torch.Tensor:
    """The per-block loop of the reference ("v2", ..._mat_qkv_fp4.py:267-304): s starts at ones, one AdamW
    step per calibration tensor, `epochs` passes.  The reference keeps `best_s = learnable_s` - an alias
    of the live parameter, so what it saves is the LAST iterate; that is the default here too.
    snapshot_best=True returns the iterate at the end of the best epoch instead."""
    s = torch.nn.Parameter(torch.ones(weight.shape[1], device=weight.device, dtype=weight.dtype))
    opt = torch.optim.AdamW([s], lr=lr)
    best_loss, best_s = float("inf"), s
    for _ in range(epochs):
        epoch_loss = 0.0
        for x in activations:
            loss = compute_quant_error(x, weight, s, Q, fmt, **quantizers)
            loss.backward()
            opt.step()
            opt.zero_grad()
            epoch_loss += loss.item()
        avg = epoch_loss / len(activations)
        if log is not None:
            log.append(avg)
        if avg < best_loss:
            best_loss = avg
            best_s = s.detach().clone() if snapshot_best else s
    return best_s.detach()


def learn_blocks_sharded(n_blocks: int, learn_block: Callable[[int], torch.Tensor], channels: int,
                         group=None) -> List[torch.Tensor]:
    """Block b runs on rank b % world (`learn_block(b)` -> s[channels], float32); one all-gather
    (n_blocks x channels floats, 230 KB for d30) leaves every rank with the full list, in block order -
    the list the reference torch.save()s as `*_best_s_fp{4,6}.pt`."""
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    per_rank = (n_blocks + world - 1) // world
    buf = torch.zeros((per_rank, channels), dtype=torch.float32)
    for slot, b in enumerate(range(rank, n_blocks, world)):
        buf[slot] = learn_block(b).detach().to(torch.float32).cpu()
    if world == 1:
        gathered = [buf]
    else:
        dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else buf.device
        send = buf.to(dev)
        gathered = [torch.empty_like(send) for _ in range(world)]
        dist.all_gather(gathered, send, group=group)
        gathered = [g.cpu() for g in gathered]
    out: List[Optional[torch.Tensor]] = [None] * n_blocks
    for r in range(world):
        for slot, b in enumerate(range(r, n_blocks, world)):
            out[b] = gathered[r][slot].clone()
    return out  # type: ignore[return-value]
