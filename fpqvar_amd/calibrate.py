"""Weight calibration sharded over GPUs (BASELINE.json config 4).

The reference quantizes the weight of every Linear once, on one GPU, inside
``QuantizedLinear.from_float`` (tr/quant_utils.py:771-860, driven by
``quantize_VAR`` :1095-1167), and then casts the model to fp16
(evaluate_fp_quant_transform_rotate.py:131).  Every layer - every 128-group, in
fact - is independent, so here the layers are partitioned over the ranks of a
``torch.distributed`` group (one process per GPU, RCCL over xGMI), each rank
quantizes its share with the fused HIP kernels, and ONE all-gather hands every rank
the complete quantized model.  There is no other collective on this path.

Exchange formats:
  * ``"fp16"``  - the de-quantized fp16 weights themselves (what the reference keeps
                  after ``.half()``): 2 B/element on the wire.
  * ``"codes"`` - 4-bit codes (two per byte) + one fp32 scale per 128-group =
                  0.53 B/element; decoded locally with ``fpq_dequant_rows_codes``.
                  Per-group FP4 tables only.  Bit-identical to ``"fp16"``.

The quantizer is injected (``quantize=``) so that the partition / packing /
collective logic is testable on CPU with gloo; the default is the HIP path and
raises without a GPU.
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Callable, Dict, List, Mapping, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def var_linear_shapes(depth: int) -> "OrderedDict[str, Tuple[int, int]]":
    """[out, in] of the four quantized Linears of every block of VAR-d<depth>
    (C = 64*depth, mlp_ratio 4; tr/basic_var.py:107-136,262)."""
    c = 64 * depth
    shapes: "OrderedDict[str, Tuple[int, int]]" = OrderedDict()
    for b in range(depth):
        shapes[f"blocks.{b}.attn.mat_qkv"] = (3 * c, c)
        shapes[f"blocks.{b}.attn.proj"] = (c, c)
        shapes[f"blocks.{b}.ffn.fc1"] = (4 * c, c)
        shapes[f"blocks.{b}.ffn.fc2"] = (c, 4 * c)
    return shapes


def partition(sizes: Sequence[Tuple[str, int]], world: int) -> List[List[str]]:
    """Longest-processing-time-first assignment of (name, numel) to `world` ranks.
    Deterministic: every rank computes the same answer without communicating."""
    order = sorted(range(len(sizes)), key=lambda i: (-sizes[i][1], sizes[i][0]))
    load = [0] * world
    out: List[List[str]] = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda k: (load[k], k))
        out[r].append(sizes[i][0])
        load[r] += sizes[i][1]
    for names in out:
        names.sort(key=lambda n: [s[0] for s in sizes].index(n))
    return out


def default_weight_quantizer(weight_quant: str = "per_group", weight_fp_type: str = "fp_e2", w_bit: int = 4,
                             out_dtype: torch.dtype = torch.float16) -> Callable[[str, torch.Tensor], torch.Tensor]:
    """The from_float dispatch for FP formats (tr/quant_utils.py:794-855), fused with
    the driver's later ``.half()``: out = fp16(fp32 quantized weight)."""
    from . import ops
    table = {"fp_e1": "e1m2", "fp_e2": "e2m1", "fp_e3": "e3m0", "fp6_e2m3": "e2m3", "fp6_e3m2": "e3m2"}[weight_fp_type]
    assert (w_bit == 6) == weight_fp_type.startswith("fp6")

    def quantize(name: str, w: torch.Tensor) -> torch.Tensor:
        if weight_quant == "per_group":
            return ops.quant_rows(w, table, 128, out_dtype)
        if weight_quant == "per_channel" and weight_fp_type.startswith("fp6"):
            return ops.quant_rows(w, table, w.shape[-1], out_dtype)
        raise NotImplementedError(f"weight_quant={weight_quant} with {weight_fp_type}")

    return quantize


def _world(group) -> Tuple[int, int]:
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


def calibrate_sharded(weights: Mapping[str, torch.Tensor],
                      quantize: Optional[Callable[[str, torch.Tensor], torch.Tensor]] = None,
                      group=None, exchange: str = "fp16", gather: bool = True
                      ) -> Dict[str, torch.Tensor]:
    """Quantize this rank's share of `weights` and (gather=True) all-gather the rest.

    `weights` must hold the same names/shapes on every rank (the values of layers a
    rank does not own are never read).  Returns name -> quantized tensor for all
    layers (gather=True) or for the local share only.
    """
    rank, world = _world(group)
    names = list(weights.keys())
    sizes = [(n, int(weights[n].numel())) for n in names]
    plan = partition(sizes, world)
    mine = plan[rank]
    if quantize is None:
        quantize = default_weight_quantizer()

    if exchange == "codes":
        return _calibrate_codes(weights, plan, rank, world, group, gather)
    if exchange != "fp16":
        raise ValueError(f"unknown exchange format {exchange!r}")

    local = OrderedDict((n, quantize(n, weights[n])) for n in mine)
    if not gather or world == 1:
        return dict(local)

    ref = next(iter(weights.values()))
    dtype = next(iter(local.values())).dtype if local else torch.float16
    numel = {n: s for n, s in sizes}
    shard = [sum(numel[n] for n in plan[r]) for r in range(world)]
    width = max(shard)
    send = torch.zeros(width, dtype=dtype, device=ref.device)
    off = 0
    for n, q in local.items():
        send[off:off + q.numel()] = q.reshape(-1)
        off += q.numel()
    recv = [torch.empty(width, dtype=dtype, device=ref.device) for _ in range(world)]
    dist.all_gather(recv, send, group=group)          # the one collective of this path
    out: Dict[str, torch.Tensor] = {}
    for r in range(world):
        off = 0
        for n in plan[r]:
            out[n] = recv[r][off:off + numel[n]].view(weights[n].shape)
            off += numel[n]
    return {n: out[n] for n in names}


def _calibrate_codes(weights, plan, rank, world, group, gather):
    from . import ops
    names = list(weights.keys())
    mine = plan[rank]
    ref = next(iter(weights.values()))
    table = "e2m1"
    numel = {n: int(weights[n].numel()) for n in names}
    for n in names:
        assert numel[n] % 128 == 0
    code_bytes = {n: numel[n] // 2 for n in names}
    n_scales = {n: numel[n] // 128 for n in names}
    local_codes, local_scales = OrderedDict(), OrderedDict()
    for n in mine:
        c, s = ops.quant_rows_codes(weights[n].float(), table, 128, pack_nibbles=True)
        local_codes[n], local_scales[n] = c.reshape(-1), s.reshape(-1)
    if not gather or world == 1:
        return {n: ops.dequant_rows_codes(local_codes[n].view(-1, 64), local_scales[n], table, 128,
                                          torch.float16, True).view(weights[n].shape) for n in mine}
    wc = max(sum(code_bytes[n] for n in plan[r]) for r in range(world))
    ws = max(sum(n_scales[n] for n in plan[r]) for r in range(world))
    send = torch.zeros(wc + 4 * ws, dtype=torch.uint8, device=ref.device)
    off = 0
    for n in mine:
        send[off:off + code_bytes[n]] = local_codes[n]
        off += code_bytes[n]
    soff = wc
    for n in mine:
        b = local_scales[n].contiguous().view(torch.uint8)
        send[soff:soff + b.numel()] = b
        soff += b.numel()
    recv = [torch.empty_like(send) for _ in range(world)]
    dist.all_gather(recv, send, group=group)
    out = {}
    for r in range(world):
        off, soff = 0, wc
        for n in plan[r]:
            codes = recv[r][off:off + code_bytes[n]].view(-1, 64)
            scales = recv[r][soff:soff + 4 * n_scales[n]].view(torch.float32)
            out[n] = ops.dequant_rows_codes(codes, scales, table, 128, torch.float16, True).view(weights[n].shape)
            off += code_bytes[n]
            soff += 4 * n_scales[n]
    return {n: out[n] for n in names}
