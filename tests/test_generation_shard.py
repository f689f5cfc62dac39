"""Replica sharding of the image-generation loop (fpqvar_amd/generation.py): the work list is the reference's
(evaluate_fp_quant_transform_rotate_512x512.py:191-222), every item lands on exactly one rank, and the
throughput counter aggregates over a world-size-2 gloo group."""
import os
import socket

import torch.distributed as dist
import torch.multiprocessing as mp

from fpqvar_amd import generation as gen


def test_work_list_is_the_reference_loop():
    items = list(gen.work_items())
    assert len(items) == 1000 * 5
    assert items[0] == gen.Batch(0, 0, 10, tuple(range(10)))
    assert items[7].class_idx == 1 and items[7].iteration == 2 and items[7].seed == 12
    assert items[7].file_names()[0] == "class1_img20.png" and items[7].file_names()[-1] == "class1_img29.png"
    assert items[7].labels == [1] * 10
    names = [n for it in items for n in it.file_names()]
    assert len(names) == len(set(names)) == 50000                       # the 50 000 FID samples, each once
    small = list(gen.work_items(num_class=3, imgs_per_batch=25, imgs_per_class=50))   # the 256x256 driver's shape
    assert [(b.class_idx, b.seed) for b in small] == [(0, 10), (0, 11), (1, 10), (1, 11), (2, 10), (2, 11)]


def test_shards_partition_the_work():
    items = list(gen.work_items(num_class=37))
    for world in (1, 2, 3, 8):
        parts = [gen.shard(items, r, world) for r in range(world)]
        assert sorted((b.class_idx, b.iteration) for p in parts for b in p) == sorted((b.class_idx, b.iteration) for b in items)
        assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        mine = gen.my_work(num_class=5)
        done = sum(len(b.image_indices) for b in mine)
        total, rate = gen.aggregate_throughput(done, 2.0 + rank)        # rank 1 is the slow one: 3 s
        q.put((rank, [(b.class_idx, b.iteration) for b in mine], total, rate))
    finally:
        dist.destroy_process_group()


def test_replicas_gloo_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert sorted(res[0][1] + res[1][1]) == [(c, k) for c in range(5) for k in range(5)]
    assert not set(res[0][1]) & set(res[1][1])
    for r in res:
        assert r[2] == 250 and abs(r[3] - 250 / 3.0) < 1e-9
