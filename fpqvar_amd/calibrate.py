"""Weight calibration sharded over GPUs (BASELINE.json config 4).

The reference quantizes the weight of every Linear once, on one GPU, inside
``QuantizedLinear.from_float`` (tr/quant_utils.py:771-860, driven by
``quantize_VAR`` :1095-1167), and then casts the model to fp16
(evaluate_fp_quant_transform_rotate.py:131).  Every layer - every 128-group, in
fact - is independent, so here the layers are partitioned over the ranks of a
``torch.distributed`` group (one process per GPU, RCCL over xGMI), each rank
quantizes its share, and ONE all-gather hands every rank the complete quantized
model.  There is no other collective on this path.

Data layout: ONE slab ``[world, width]`` of the exchange dtype per rank, ``width`` = the largest
shard.  A rank's layers are quantized by ONE launch over a device-resident segment table
(``fpq_quant_rows_segments``: fp32 in -> per-group fake-quant -> fp16 out, i.e. the reference's fp32
quantization followed by ``.half()``) that writes straight into the rank's own slot of the slab;
``all_gather_into_tensor`` (in place: the input is that slot) fills the other slots; the result
tensors are views of the slab.  No staging copy, no per-layer launch, no list-form gather.

Exchange formats:
  * ``"fp16"``  - the de-quantized fp16 weights themselves (what the reference keeps
                  after ``.half()``): 2 B/element on the wire.
  * ``"codes"`` - 4-bit codes (two per byte) + one fp32 scale per 128-group =
                  0.53 B/element; decoded locally with ``fpq_dequant_rows_codes``.
                  Per-group FP4 E2M1 only.  Bit-identical to ``"fp16"``.

A quantizer can be injected (``quantize=``) so that the partition / slab / collective logic is
testable on CPU with gloo; the default is the HIP path and raises without a GPU.
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Callable, Dict, List, Mapping, Optional, Sequence, Tuple

import torch
import torch.distributed as dist

_GROUP = 128
_FP_TABLES = {"fp_e1": "e1m2", "fp_e2": "e2m1", "fp_e3": "e3m0", "fp6_e2m3": "e2m3", "fp6_e3m2": "e3m2"}


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
    pos = {s[0]: i for i, s in enumerate(sizes)}
    for names in out:
        names.sort(key=pos.__getitem__)
    return out


def default_weight_quantizer(weight_quant: str = "per_group", weight_fp_type: str = "fp_e2", w_bit: int = 4,
                             out_dtype: torch.dtype = torch.float16) -> Callable[[str, torch.Tensor], torch.Tensor]:
    """The from_float dispatch for FP formats (tr/quant_utils.py:794-855), fused with
    the driver's later ``.half()``: out = fp16(fp32 quantized weight).  One layer per call."""
    from . import ops
    table = _FP_TABLES[weight_fp_type]
    assert (w_bit == 6) == weight_fp_type.startswith("fp6")

    def quantize(name: str, w: torch.Tensor) -> torch.Tensor:
        if weight_quant == "per_group":
            return ops.quant_rows(w, table, _GROUP, out_dtype)
        if weight_quant == "per_channel" and weight_fp_type.startswith("fp6"):
            return ops.quant_rows(w, table, w.shape[-1], out_dtype)
        raise NotImplementedError(f"weight_quant={weight_quant} with {weight_fp_type}")

    return quantize


def gather_slab(slab: torch.Tensor, rank: int, group=None) -> None:
    """THE collective of this path: every rank's slot of the [world, width] slab to every rank, in place.

    The input of all_gather_into_tensor is the rank's own slot of the output (no staging copy).  That aliasing is
    what RCCL's in-place all-gather is defined for (sendbuff == recvbuff + rank * count), but through torch.distributed
    it has only run on one-rank groups and on gloo so far (no multi-GPU box was available to this build): setting
    FPQ_GATHER_NO_ALIAS=1 sends from a separate copy of the slot instead (one extra slot-sized copy per call)."""
    import os
    send = slab[rank]
    if os.environ.get("FPQ_GATHER_NO_ALIAS"):
        send = send.clone()
    dist.all_gather_into_tensor(slab.view(-1), send, group=group)


def _world(group) -> Tuple[int, int]:
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


class LocalShard:
    """A set of fp32 layers (this rank's share) quantized per group of 128 by ONE launch.

    `weights`: name -> fp32 GPU tensor (numel % 128 == 0).  `out`: optional 1-D tensor of `out_dtype` with room for
    all of them back to back (e.g. this rank's slot of the all-gather slab); allocated when omitted.  The segment
    table (input pointer, output pointer, group count per layer) is uploaded once here; `quantize()` is then a single
    C-ABI call with no host work besides the launch."""

    def __init__(self, weights: Mapping[str, torch.Tensor], shapes: Optional[Mapping[str, Tuple[int, ...]]] = None,
                 weight_fp_type: str = "fp_e2", out: Optional[torch.Tensor] = None,
                 out_dtype: torch.dtype = torch.float16):
        from . import _lib
        self._lib = _lib
        self.table_id = _lib.TABLE_IDS[_FP_TABLES[weight_fp_type]]
        self.names = list(weights.keys())
        self.shapes = {n: tuple(shapes[n]) if shapes is not None else tuple(weights[n].shape) for n in self.names}
        self.out_dtype = out_dtype
        self._inputs = []
        self.offsets: Dict[str, int] = {}
        total = 0
        for n in self.names:
            w = weights[n]
            _lib.require_gpu(w, f"LocalShard({n})")
            if w.dtype != torch.float32:
                raise RuntimeError(f"LocalShard: {n} must be float32 (the reference quantizes the fp32 weight), got {w.dtype}")
            if w.numel() % _GROUP != 0:
                raise RuntimeError(f"LocalShard: {n} has {w.numel()} elements, not a multiple of the group size {_GROUP}")
            w = w if w.is_contiguous() else w.contiguous()
            if w.data_ptr() % 16 != 0:
                w = w.clone()
            self._inputs.append(w)
            self.offsets[n] = total
            total += w.numel()
        self.total = total
        dev = self._inputs[0].device if self._inputs else (out.device if out is not None else torch.device("cuda"))
        self.device = dev
        if out is None:
            out = torch.empty(total, dtype=out_dtype, device=dev)
        if out.dtype != out_dtype or out.dim() != 1 or out.numel() < total or not out.is_contiguous() or \
                out.data_ptr() % 16 != 0:
            raise RuntimeError("LocalShard: `out` must be a contiguous, 16-byte aligned 1-D tensor of the output dtype "
                               "with room for every layer")
        self.slab = out
        esz = out.element_size()
        rows = [w.numel() // _GROUP for w in self._inputs]
        self.max_rows = max(rows) if rows else 0
        desc = [[w.data_ptr(), out.data_ptr() + self.offsets[n] * esz, r] for n, w, r in zip(self.names, self._inputs, rows)]
        self._table = torch.tensor(desc, dtype=torch.int64).reshape(-1, 3).to(dev) if desc else None

    def quantize(self) -> torch.Tensor:
        """Enqueue the launch on the current stream; returns the slab."""
        if self._table is None:
            return self.slab
        lib, _lib = self._lib.lib(), self._lib
        with _lib.device_guard(self.device):
            _lib.check(lib.fpq_quant_rows_segments(self._table.data_ptr(), len(self.names), self.max_rows, _GROUP,
                                                   self.table_id, _lib.F32, _lib.dtype_id(self.out_dtype),
                                                   _lib.stream_ptr(self.device)), "fpq_quant_rows_segments")
        return self.slab

    def views(self) -> Dict[str, torch.Tensor]:
        return {n: self.slab[self.offsets[n]:self.offsets[n] + _numel(self.shapes[n])].view(self.shapes[n])
                for n in self.names}


def _numel(shape) -> int:
    n = 1
    for s in shape:
        n *= int(s)
    return n


class _Plan:
    """Who owns what and where it lives in the [world, width] slab (identical on every rank)."""

    def __init__(self, shapes: Mapping[str, Tuple[int, ...]], world: int):
        self.names = list(shapes.keys())
        self.shapes = {n: tuple(shapes[n]) for n in self.names}
        self.numel = {n: _numel(self.shapes[n]) for n in self.names}
        self.owners = partition([(n, self.numel[n]) for n in self.names], world)
        self.width = max(sum(self.numel[n] for n in names) for names in self.owners) if self.names else 0
        self.width = (self.width + 7) // 8 * 8           # keep every slot 16-byte aligned in fp16
        self.where: Dict[str, Tuple[int, int]] = {}
        for r, names in enumerate(self.owners):
            off = 0
            for n in names:
                self.where[n] = (r, off)
                off += self.numel[n]


class ShardedCalibration:
    """calibrate_sharded's HIP path as a reusable object: slab and segment table are built once, `run()` is one
    launch + (world > 1) one in-place all_gather_into_tensor.  `weights` needs entries for the layers THIS rank owns
    only (``plan_owners(shapes, world)[rank]``)."""

    def __init__(self, shapes: Mapping[str, Tuple[int, ...]], weights: Mapping[str, torch.Tensor], group=None,
                 weight_fp_type: str = "fp_e2", device: Optional[torch.device] = None):
        self.group = group
        self.rank, self.world = _world(group)
        self.plan = _Plan(shapes, self.world)
        mine = self.plan.owners[self.rank]
        # every check that can fail happens HERE, on every rank alike, before any collective: a rank that raised later
        # would leave the others blocked in the all-gather
        missing = [n for n in mine if n not in weights]
        if missing:
            raise RuntimeError(f"ShardedCalibration: rank {self.rank} owns {missing[:3]}... but was not given them")
        for n in mine:
            if weights[n].numel() != self.plan.numel[n]:
                raise RuntimeError(f"ShardedCalibration: {n} has {weights[n].numel()} elements, `shapes` says "
                                   f"{self.plan.shapes[n]} = {self.plan.numel[n]}")
        if device is None:   # a rank may own no layer (more ranks than layers) and hold no weights at all
            ref = next((weights[n] for n in mine), None)
            if ref is None:
                ref = next(iter(weights.values()), None)
            if ref is not None:
                device = ref.device
            elif torch.cuda.is_available():
                device = torch.device("cuda", torch.cuda.current_device())
            else:
                raise RuntimeError("ShardedCalibration: this rank holds no weights; pass `device=`")
        self.device = torch.device(device)
        # zeros, once: ranks with a short share gather the padding behind it, and a gathered slab should be a
        # deterministic function of the weights (the launches only ever write the layers)
        self.slab = torch.zeros((self.world, self.plan.width), dtype=torch.float16, device=self.device)
        self.local = LocalShard(OrderedDict((n, weights[n]) for n in mine), self.plan.shapes, weight_fp_type,
                                out=self.slab[self.rank])

    def run(self) -> Dict[str, torch.Tensor]:
        self.local.quantize()
        if self.world > 1:
            gather_slab(self.slab, self.rank, self.group)   # THE collective
        return self.views()

    def views(self) -> Dict[str, torch.Tensor]:
        p = self.plan
        return {n: self.slab[p.where[n][0], p.where[n][1]:p.where[n][1] + p.numel[n]].view(p.shapes[n]) for n in p.names}


def plan_owners(shapes: Mapping[str, Tuple[int, ...]], world: int) -> List[List[str]]:
    return _Plan(shapes, world).owners


def calibrate_sharded(weights: Mapping[str, torch.Tensor],
                      quantize: Optional[Callable[[str, torch.Tensor], torch.Tensor]] = None,
                      group=None, exchange: str = "fp16", gather: bool = True
                      ) -> Dict[str, torch.Tensor]:
    """Quantize this rank's share of `weights` and (gather=True) all-gather the rest.

    `weights` must hold the same names/shapes on every rank (the values of layers a
    rank does not own are never read).  Returns name -> quantized tensor for all
    layers (gather=True) or for the local share only.

    quantize=None: the HIP path (per-group(128) E2M1, fp32 -> fp16), one launch for the whole share.
    quantize=f(name, w): any per-layer quantizer (tests inject a CPU one); its results are copied into the slab.
    """
    rank, world = _world(group)
    if exchange == "codes":
        if quantize is not None:
            raise ValueError("exchange='codes' is the built-in per-group E2M1 path: it cannot take a custom `quantize`")
        return _calibrate_codes(weights, rank, world, group, gather)
    if exchange != "fp16":
        raise ValueError(f"unknown exchange format {exchange!r}")
    shapes = OrderedDict((n, tuple(w.shape)) for n, w in weights.items())
    if quantize is None:
        sc = ShardedCalibration(shapes, weights, group)
        if gather:
            return sc.run()
        sc.local.quantize()
        return sc.local.views()

    plan = _Plan(shapes, world)
    mine = plan.owners[rank]
    local = OrderedDict((n, quantize(n, weights[n])) for n in mine)
    if not gather or world == 1:
        return {n: local[n] for n in (mine if not gather else plan.names)}
    ref = next(iter(weights.values()))
    dtype = next(iter(local.values())).dtype if local else torch.float16
    slab = torch.empty((world, plan.width), dtype=dtype, device=ref.device)
    for n, q in local.items():
        off = plan.where[n][1]
        slab[rank, off:off + plan.numel[n]] = q.reshape(-1)
    gather_slab(slab, rank, group)                                                # the one collective of this path
    return {n: slab[plan.where[n][0], plan.where[n][1]:plan.where[n][1] + plan.numel[n]].view(plan.shapes[n])
            for n in plan.names}


class ShardedCodesCalibration:
    """The packed exchange: nibble codes + one fp32 scale per group of 128 (0.53 B per element on the wire).

    Built once (slabs, the two device-resident segment tables); `run()` is then ONE launch that quantizes this rank's
    layers straight into its slot of the codes slab (fpq_quant_rows_codes_segments: no per-layer launches, no copies),
    one all-gather, and ONE launch that decodes every layer of every rank into an fp16 slab
    (fpq_dequant_rows_codes_segments) whose views are the result - bit-equal to the fp16 exchange.  `weights`: every
    name; only this rank's layers are read (others may be zero-stride placeholders of the right shape)."""

    def __init__(self, weights, group=None, gather: bool = True, rank: Optional[int] = None, world: Optional[int] = None):
        """A reusable plan: slabs, segment tables and the output buffer are built HERE, once.  Two consequences for a caller
        that keeps the object: (1) the device pointers of the owned weights are baked into the segment table - a weight that
        is float32, contiguous and 16-byte aligned is read in place by every run() (updates are seen), anything else is
        converted ONCE into a private copy (`snapshot_names`; later updates of the original are NOT seen - rebuild the plan);
        (2) every run() returns views of the same `out` buffer, so the result of an earlier run() is overwritten."""
        from . import _lib
        self._lib = _lib
        self.snapshot_names = []
        r0, w0 = _world(group)
        self.rank = r0 if rank is None else rank
        self.world = w0 if world is None else world
        self.group, self.gather = group, bool(gather) and self.world > 1
        rank, world = self.rank, self.world
        self.names = list(weights.keys())
        self.shapes = {n: tuple(weights[n].shape) for n in self.names}
        numel = {n: int(weights[n].numel()) for n in self.names}
        for n in self.names:
            if numel[n] % _GROUP != 0:
                raise RuntimeError(f"calibrate_sharded(exchange='codes'): {n} is not a multiple of {_GROUP} elements")
        self.plan = partition([(n, numel[n]) for n in self.names], world)
        self.mine = self.plan[rank]
        dev = next(iter(weights.values())).device
        self.device = dev
        self.table_id = _lib.TABLE_IDS["e2m1"]
        code_bytes = {n: numel[n] // 2 for n in self.names}               # multiples of 64: every layer's codes stay 16-byte aligned
        n_scales = {n: numel[n] // _GROUP for n in self.names}
        wc = max(sum(code_bytes[n] for n in self.plan[r]) for r in range(world))
        wc = (wc + 15) // 16 * 16
        ws = max(sum(n_scales[n] for n in self.plan[r]) for r in range(world))
        width = (wc + 4 * ws + 15) // 16 * 16
        self.slab = torch.zeros((world, width), dtype=torch.uint8, device=dev)   # padding zeroed once (see ShardedCalibration)
        if self.slab.data_ptr() % 16 != 0:
            raise RuntimeError("calibrate_sharded(exchange='codes'): the slab is not 16-byte aligned")

        def layout(r):                                                    # (layer, codes offset, scales offset) inside slab[r]
            off, soff, rows = 0, wc, []
            for n in self.plan[r]:
                rows.append((n, off, soff))
                off += code_bytes[n]
                soff += 4 * n_scales[n]
            return rows
        self._inputs = []                                                 # contiguous fp32 inputs, kept alive with the table
        q_desc = []
        base = self.slab.data_ptr() + rank * width
        for n, off, soff in layout(rank):
            w = weights[n]
            _lib.require_gpu(w, f"calibrate_sharded({n})")
            w0 = w
            w = w.float() if w.dtype != torch.float32 else w
            w = w if w.is_contiguous() else w.contiguous()
            if w.data_ptr() % 16 != 0:
                w = w.clone()
            if w.data_ptr() != w0.data_ptr():
                self.snapshot_names.append(n)
            self._inputs.append(w)
            q_desc.append([w.data_ptr(), base + off, base + soff, n_scales[n]])
        self._q_rows = max((d[3] for d in q_desc), default=0)
        self._q_tab = torch.tensor(q_desc, dtype=torch.int64).to(dev) if q_desc else None
        ranks = range(world) if self.gather else (rank,)
        self.out_names = [n for r in ranks for n in self.plan[r]]
        self.offsets, total = {}, 0
        for n in self.out_names:
            self.offsets[n] = total
            total += numel[n]
        self.numel = numel
        self.out = torch.empty(total, dtype=torch.float16, device=dev)
        d_desc = []
        for r in ranks:
            rb = self.slab.data_ptr() + r * width
            for n, off, soff in layout(r):
                d_desc.append([rb + off, rb + soff, self.out.data_ptr() + 2 * self.offsets[n], n_scales[n]])
        self._d_rows = max((d[3] for d in d_desc), default=0)
        self._d_tab = torch.tensor(d_desc, dtype=torch.int64).to(dev) if d_desc else None
        self.gathered_bytes_per_rank = (world - 1) * width if self.gather else 0

    def run(self) -> Dict[str, torch.Tensor]:
        _lib, lib = self._lib, self._lib.lib()
        with _lib.device_guard(self.device):
            if self._q_tab is not None:
                _lib.check(lib.fpq_quant_rows_codes_segments(self._q_tab.data_ptr(), self._q_tab.shape[0], self._q_rows, _GROUP,
                                                             self.table_id, _lib.F32, 1, _lib.stream_ptr(self.device)),
                           "fpq_quant_rows_codes_segments")
            if self.gather:
                gather_slab(self.slab, self.rank, self.group)
            if self._d_tab is not None:
                _lib.check(lib.fpq_dequant_rows_codes_segments(self._d_tab.data_ptr(), self._d_tab.shape[0], self._d_rows, _GROUP,
                                                               self.table_id, _lib.F32, _lib.F16, 1, _lib.stream_ptr(self.device)),
                           "fpq_dequant_rows_codes_segments")
        res = {n: self.out[self.offsets[n]:self.offsets[n] + self.numel[n]].view(self.shapes[n]) for n in self.out_names}
        return {n: res[n] for n in self.names if n in res}


def _calibrate_codes(weights, rank, world, group, gather):
    return ShardedCodesCalibration(weights, group=group, gather=gather, rank=rank, world=world).run()
