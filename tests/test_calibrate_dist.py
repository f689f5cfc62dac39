"""Sharded weight calibration: partition logic and the world-size-2 all-gather on
CPU with gloo.  The quantizer is injected (the oracle plays the HIP kernel's
part here; the GPU test runs the real thing)."""
import os
import socket
import subprocess
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from fpqvar_amd import calibrate as cal
from oracle import fpq_oracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_var_shapes_and_partition_balance():
    shapes = cal.var_linear_shapes(30)
    assert len(shapes) == 120
    per_block = sum(o * i for (o, i) in list(shapes.values())[:4])
    assert per_block == 44_236_800                       # 44.24 M / block
    assert sum(o * i for o, i in shapes.values()) == 1_327_104_000
    sizes = [(n, o * i) for n, (o, i) in shapes.items()]
    for world in (1, 2, 4, 8):
        plan = cal.partition(sizes, world)
        flat = [n for p in plan for n in p]
        assert sorted(flat) == sorted(shapes)            # every layer exactly once
        loads = [sum(dict(sizes)[n] for n in p) for p in plan]
        assert max(loads) / (sum(loads) / world) < 1.03   # within 3 % of perfect balance
    assert sum(o * i for o, i in cal.var_linear_shapes(36).values()) == 2_293_235_712


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _toy_weights():
    g = torch.Generator().manual_seed(1234)
    shapes = {"b0.qkv": (384, 128), "b0.proj": (128, 128), "b0.fc1": (512, 128), "b0.fc2": (128, 512),
              "b1.qkv": (384, 128), "b1.proj": (128, 128), "b1.fc1": (512, 128)}
    return {n: torch.randn(*s, generator=g) * 0.02 for n, s in shapes.items()}


def _oracle_quant(name, w):
    return orc.per_group_kernel_sem(w, "e2m1", 128).to(torch.float16)


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        weights = _toy_weights()
        # poison the layers this rank does not own: they must never be read
        plan = cal.partition([(n, w.numel()) for n, w in weights.items()], world)
        for n in weights:
            if n not in plan[rank]:
                weights[n] = torch.full_like(weights[n], float("nan"))
        got = cal.calibrate_sharded(weights, quantize=_oracle_quant, exchange="fp16")
        want = {n: _oracle_quant(n, w) for n, w in _toy_weights().items()}
        ok = list(got) == list(want) and all(
            torch.equal(got[n].view(torch.int16), want[n].view(torch.int16)) and got[n].shape == want[n].shape
            for n in want)
        local = cal.calibrate_sharded(weights, quantize=_oracle_quant, gather=False)
        ok = ok and sorted(local) == sorted(plan[rank])
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", (2, 3, 8))   # 8: the node the driver scales to (one process per GPU there, gloo on the CPU here)
def test_calibrate_all_gather_gloo(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(results) == [(r, True) for r in range(world)]


def test_single_process_no_dist():
    weights = _toy_weights()
    got = cal.calibrate_sharded(weights, quantize=_oracle_quant)
    for n, w in weights.items():
        assert torch.equal(got[n], _oracle_quant(n, w))


# ---- format search sharded by block: tiny all-gather of (loss, w_fmt, a_fmt) -----------------
def _search_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from fpqvar_amd import format_search as fs
        seen = []

        def evaluate(b):
            seen.append(b)
            return fs.FP6_FORMATS[b % 2], fs.FP6_FORMATS[(b // 2) % 2], 0.5 + b

        res = fs.search_blocks_sharded(7, evaluate)
        want = [(fs.FP6_FORMATS[b % 2], fs.FP6_FORMATS[(b // 2) % 2], 0.5 + b) for b in range(7)]
        q.put((rank, res == want and seen == list(range(rank, 7, world))))
    finally:
        dist.destroy_process_group()


def test_format_search_sharded_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_search_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(results) == [(0, True), (1, True)]


def test_sharded_calibration_validates_before_any_collective():
    """A shape table that disagrees with the weights, or an owned layer that was not handed over, raises in the
    constructor - on every rank alike, before the all-gather a late failure would leave the other ranks blocked in.
    (No process group and no GPU needed: the checks come first.)"""
    from fpqvar_amd import calibrate as cal
    shapes = {"a": (4, 128), "b": (2, 128)}
    w = {"a": torch.zeros(4, 128), "b": torch.zeros(3, 128)}
    with pytest.raises(RuntimeError, match="`shapes` says"):
        cal.ShardedCalibration(shapes, w)
    with pytest.raises(RuntimeError, match="was not given"):
        cal.ShardedCalibration(shapes, {"a": torch.zeros(4, 128)})


def test_gather_slab_fallback_matches_in_place(tmp_path):
    """FPQ_GATHER_NO_ALIAS=1 (a separate send buffer) and the in-place form fill the slab identically (gloo, world 2)."""
    script = tmp_path / "g.py"
    script.write_text('''
import os, sys
sys.path.insert(0, %r)
import torch, torch.distributed as dist
from fpqvar_amd import calibrate as cal
dist.init_process_group("gloo")
r, w = dist.get_rank(), dist.get_world_size()
res = []
for mode in ("", "1"):
    if mode:
        os.environ["FPQ_GATHER_NO_ALIAS"] = mode
    slab = torch.full((w, 16), -1.0)
    slab[r] = torch.arange(16.0) + 100 * r
    cal.gather_slab(slab, r)
    res.append(slab.clone())
want = torch.stack([torch.arange(16.0) + 100 * k for k in range(w)])
assert torch.equal(res[0], want) and torch.equal(res[1], want), (res, want)
dist.destroy_process_group()
''' % ROOT)
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "FPQ_GATHER_NO_ALIAS")}
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                          "--master-port", "29731", str(script)], capture_output=True, text=True, timeout=240, env=env)
    assert out.returncode == 0, out.stderr[-2000:]


@pytest.mark.gpu
@pytest.mark.parametrize("world", (2, 3))
def test_sharded_calibration_real_ranks_share_the_gpu(world):
    """tools/rccl_world2_check.py with FPQ_CHECK_BACKEND=gloo: `world` processes on the one GPU of a test box (gloo takes
    device tensors; RCCL refuses two ranks on a device), the in-place all_gather_into_tensor on device memory, and
    calibrate_sharded with the fp16 and the codes exchange - every rank's gathered model bit-equal to per-layer launches -
    and the block-sharded format search against the single-process loop."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, FPQ_CHECK_BACKEND="gloo")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
                          "--master-port", str(29540 + world), os.path.join(root, "tools", "rccl_world2_check.py")],
                         capture_output=True, text=True, timeout=600, cwd=root, env=env)
    text = out.stdout + out.stderr
    assert out.returncode == 0, text[-2000:]
    assert text.count("in-place all_gather_into_tensor ok") == world, text[-2000:]
    assert text.count("fp16 exchange bit-equal True, codes exchange bit-equal True") == world, text[-2000:]
    assert text.count("equals the single-process result: True") == world, text[-2000:]
