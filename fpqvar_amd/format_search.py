"""Per-layer format search inner loop (reference: search/search_fp6_format.py:584-608,
827-846 and its FP4 twin search/search_fp4_format.py): for every (weight format,
activation format) pair accumulate  mean((x_j W^T - q_a(x_j) q_w(W)^T)^2)  over the
calibration activations x_j and keep the argmin.  The quantizers are this package's
fused kernels; the two GEMMs are plain library calls.  Blocks are independent, so the
search shards by block over the ranks of a process group and ends with one tiny
all-gather of (loss, w_fmt, a_fmt) per block.
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist

from . import quant_utils as qu

# the reference's candidate sets
FP6_FORMATS = ("fp6_e2m3", "fp6_e3m2")                 # per-token quantizers (search_fp6_format.py:513-554)
FP4_FORMATS = ("fp_e1", "fp_e2", "fp_e3")              # per-group-128 quantizers (search_fp4_format.py:484-553)


def quantizer(fmt: str) -> Callable[[torch.Tensor], torch.Tensor]:
    table = {
        "fp6_e2m3": lambda t: qu.fp6_quant_e2m3_per_token_cuda(t, 6),
        "fp6_e3m2": lambda t: qu.fp6_quant_e3m2_per_token_cuda(t, 6),
        "fp_e1": lambda t: qu.fp_quant_e1_per_group_cuda(t, 4, 128),
        "fp_e2": lambda t: qu.fp_quant_e2_per_group_cuda(t, 4, 128),
        "fp_e3": lambda t: qu.fp_quant_e3_per_group_cuda(t, 4, 128),
    }
    return table[fmt]


@torch.no_grad()
def search_layer(xs: Sequence[torch.Tensor], w: torch.Tensor, formats: Sequence[str] = FP6_FORMATS,
                 quant: Callable[[str], Callable] = quantizer, batched: Optional[bool] = None
                 ) -> Tuple[str, str, Dict[Tuple[str, str], float]]:
    """(best weight format, best activation format, {(w_fmt, a_fmt): summed MSE}).

    The reference walks the calibration samples one by one for every (w_fmt, a_fmt) pair: per sample one quantizer call
    (~11 torch ops + the scan kernel), two GEMMs, an MSE and a host sync (`.item()`), search_fp6_format.py:589-608.
    batched=True (default) does the same arithmetic on ALL samples of the layer at once: every quantizer here works row
    by row (per token) or group by group, so the samples are concatenated along their rows - ONE quantizer launch per
    activation format for the whole calibration set (and one per weight format), ONE GEMM per pair, the per-sample
    means as a weighted row reduction, all `len(formats)^2` losses kept on the device and read back with ONE copy.
    batched=False is the sample-by-sample loop (same quantizers), kept for comparison and for tests.

    Precondition of the batched form: the quantizer must be ROW-LOCAL (per token, or per group inside a row) - a
    per-tensor quantizer sees a different tensor once the samples are concatenated.  The built-in `quantizer` table is;
    an injected `quant` is not assumed to be: batched=None (default) means "batched for the built-in quantizers, the
    loop for anything injected", and a caller who knows his quantizer to be row-local passes batched=True.
    Both forms quantize x in ITS dtype (the reference calls the quantizer on the dumped activation as it is), cast to
    the weight's dtype for the GEMM, and subtract in float32."""
    nf = len(formats)
    if batched is None:
        batched = quant is quantizer
    if not batched:
        losses: Dict[Tuple[str, str], float] = {}
        refs = [x.to(w.dtype) @ w.t() for x in xs]
        for wf in formats:
            wq = quant(wf)(w).to(w.dtype)
            for af in formats:
                qa = quant(af)
                total = torch.zeros((), dtype=torch.float32, device=w.device)
                for x, ref in zip(xs, refs):
                    y = qa(x).to(w.dtype) @ wq.t()
                    total += torch.mean((ref.float() - y.float()) ** 2)
                losses[(wf, af)] = float(total)
    else:
        c = w.shape[-1]
        rows = [x.numel() // c for x in xs]
        x_all = torch.cat([x.reshape(-1, c) for x in xs])                             # [sum rows, C], the samples' dtype
        # sum_j mean_j((y - y_q)^2) = sum over rows of (row's squared error) / (rows_j * out): one weight per row
        w_row = torch.repeat_interleave(torch.tensor([1.0 / (r * w.shape[0]) for r in rows], dtype=torch.float32, device=w.device),
                                        torch.tensor(rows, device=w.device))
        ref = (x_all.to(w.dtype) @ w.t()).float()
        xq = {af: quant(af)(x_all).to(w.dtype) for af in formats}                     # one launch per activation format
        out = torch.empty(nf, nf, dtype=torch.float32, device=w.device)
        for i, wf in enumerate(formats):
            wq = quant(wf)(w).to(w.dtype)
            for j, af in enumerate(formats):
                d = ref - (xq[af] @ wq.t()).float()
                out[i, j] = torch.dot((d * d).sum(dim=1), w_row)
        host = out.cpu()                                                               # the layer's one synchronisation
        losses = {(wf, af): float(host[i, j]) for i, wf in enumerate(formats) for j, af in enumerate(formats)}
    best = min(losses, key=lambda k: (losses[k], formats.index(k[0]), formats.index(k[1])))
    return best[0], best[1], losses


def search_blocks_sharded(n_blocks: int, evaluate: Callable[[int], Tuple[str, str, float]],
                          formats: Sequence[str] = FP6_FORMATS, group=None) -> List[Tuple[str, str, float]]:
    """Block b is evaluated on rank b % world; every rank gets all results.
    `evaluate(b)` returns (w_fmt, a_fmt, loss).  The collective is one all-gather of
    n_blocks x (loss, w index, a index) float32 triples."""
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    per_rank = (n_blocks + world - 1) // world
    buf = torch.full((per_rank, 3), -1.0, dtype=torch.float32)
    for slot, b in enumerate(range(rank, n_blocks, world)):
        wf, af, loss = evaluate(b)
        buf[slot] = torch.tensor([loss, formats.index(wf), formats.index(af)], dtype=torch.float32)
    if world == 1:
        gathered = [buf]
    else:
        dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else buf.device
        send = buf.to(dev)
        gathered = [torch.empty_like(send) for _ in range(world)]
        dist.all_gather(gathered, send, group=group)
        gathered = [g.cpu() for g in gathered]
    out: List[Optional[Tuple[str, str, float]]] = [None] * n_blocks
    for r in range(world):
        for slot, b in enumerate(range(r, n_blocks, world)):
            loss, wi, ai = gathered[r][slot].tolist()
            out[b] = (formats[int(wi)], formats[int(ai)], loss)
    return out  # type: ignore[return-value]
