"""BASELINE.json config 5, the part that shards: the image-generation loop of the reference drivers
(evaluate_fp_quant_transform_rotate_512x512.py:191-222, and the 256x256 twin) is independent per class and per seed:

    for i in range(1000):                 # class
        for k in range(50 // imgs):       # seed = k + 10, a batch of `imgs` images of class i
            var.autoregressive_infer_cfg(B=imgs, label_B=[i]*imgs, g_seed=k + 10, ...)
            -> class{i}_img{j + k*imgs}.png

There is no collective in the model, so N GPUs run N replicas over disjoint (class, seed) work items:
"replicas only" (DESIGN.md section 6).  This module is the host logic for that: the work list in the
reference's order, its partition over ranks, and the aggregate sample-throughput counter (one all-reduce).
The VAR model itself is the reference's vendored code and stays there.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Iterator, List, Optional, Tuple

import torch
import torch.distributed as dist


@dataclass(frozen=True)
class Batch:
    class_idx: int
    iteration: int          # k
    seed: int               # k + 10, as in the reference
    image_indices: Tuple[int, ...]   # j + k * imgs_per_batch: the numbers in the reference's file names

    @property
    def labels(self) -> List[int]:
        return [self.class_idx] * len(self.image_indices)

    def file_names(self) -> List[str]:
        return [f"class{self.class_idx}_img{j}.png" for j in self.image_indices]


def work_items(num_class: int = 1000, imgs_per_batch: int = 10, imgs_per_class: int = 50) -> Iterator[Batch]:
    """Every batch of the reference loop, in its order (imgs_per_batch = 10 at 512x512, 50 // num_iter at 256x256)."""
    num_iter = imgs_per_class // imgs_per_batch
    for i in range(num_class):
        for k in range(num_iter):
            yield Batch(i, k, k + 10, tuple(j + k * imgs_per_batch for j in range(imgs_per_batch)))


def shard(items: List[Batch], rank: int, world: int) -> List[Batch]:
    """Item n goes to rank n % world: every rank gets the same number of batches (+-1) and, because the cost of a
    batch does not depend on the class, the same amount of work; deterministic, no communication."""
    return items[rank::world]


def my_work(num_class: int = 1000, imgs_per_batch: int = 10, imgs_per_class: int = 50, group=None) -> List[Batch]:
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    return shard(list(work_items(num_class, imgs_per_batch, imgs_per_class)), rank, world)


def aggregate_throughput(images_done: int, seconds: float, group=None, device: Optional[torch.device] = None
                         ) -> Tuple[int, float]:
    """(total images over all ranks, images per second of the whole job = total / slowest rank's time)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return images_done, images_done / seconds if seconds > 0 else 0.0
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
    n = torch.tensor([float(images_done)], dtype=torch.float64, device=device)
    t = torch.tensor([float(seconds)], dtype=torch.float64, device=device)
    dist.all_reduce(n, op=dist.ReduceOp.SUM, group=group)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    total, slowest = int(n.item()), float(t.item())
    return total, total / slowest if slowest > 0 else 0.0
