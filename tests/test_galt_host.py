"""GALT loop logic on CPU (fpqvar_amd/galt.py): the objective and its straight-through gradient against
the reference's own numbers (tests/golden, section 5 of make_golden.py), the AdamW loop, and the
block-sharded all-gather under gloo.  The oracle plays the HIP quantizers' part here."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from fpqvar_amd import galt
from oracle import fpq_oracle as orc
from tests.conftest import assert_bits_equal, from_bits


def _fp4(x):
    return galt._STE.apply(x, lambda t: orc.per_group_argmin_sem(t, "e2m1", 128))


def _fp6_tok(x):
    return galt._STE.apply(x, lambda t: orc.per_token_kernel_sem(t, "e2m3"))


def _load(golden):
    return tuple(from_bits(golden[f"galt/{k}_f32"]) for k in ("x", "w", "s", "q"))


def test_objective_and_ste_gradient_match_the_reference(golden):
    x, w, s, q = _load(golden)
    for tag, aq, wq in (("fp4", _fp4, _fp4), ("fp6", _fp6_tok, _fp6_tok)):
        sp = torch.nn.Parameter(s.clone())
        loss = galt.compute_quant_error(x, w, sp, q, tag, act_quant=aq, weight_quant=wq)
        loss.backward()
        want_loss = from_bits(golden[f"galt/{tag}/loss"])
        want_grad = from_bits(golden[f"galt/{tag}/grad_s"])
        assert torch.allclose(loss.detach().float().reshape(1), want_loss, rtol=1e-6, atol=0), tag
        assert torch.allclose(sp.grad.float(), want_grad, rtol=1e-4, atol=1e-9), tag
    # the oracle's quantizers on the reference's transformed operands, bit for bit
    assert_bits_equal(orc.per_group_argmin_sem(from_bits(golden["galt/fp4/x2_f32"]), "e2m1", 128),
                      from_bits(golden["galt/fp4/x2_quant"]), "FPQuant")
    assert_bits_equal(orc.per_token_kernel_sem(from_bits(golden["galt/fp6/w2_f32"]), "e2m3"),
                      from_bits(golden["galt/fp6/w2_quant"]), "FP6Quant_weight")
    assert_bits_equal(orc.per_group_kernel_sem(from_bits(golden["galt/fp6/x2_f32"]), "e2m3", 128, out_dtype=torch.float16),
                      from_bits(golden["galt/fp6/x2_quant_group"]), "FP6Quant_activation")


def test_learn_s_reduces_the_objective_and_keeps_the_reference_aliasing(golden):
    x, w, _, q = _load(golden)
    acts = [x[:32], x[32:]]
    log = []
    s_last = galt.learn_s(acts, w, q, epochs=6, lr=0.01, fmt="fp4", log=log, act_quant=_fp4, weight_quant=_fp4)
    assert s_last.shape == (256,) and not s_last.requires_grad
    assert min(log[1:]) < log[0]
    # default = the reference's `best_s = learnable_s` alias: the LAST iterate, whatever the best epoch was
    log2 = []
    s_best = galt.learn_s(acts, w, q, epochs=6, lr=0.01, fmt="fp4", snapshot_best=True, log=log2,
                          act_quant=_fp4, weight_quant=_fp4)
    assert log2 == log
    if log.index(min(log)) == len(log) - 1:
        assert torch.equal(s_best, s_last)
    # 6 epochs x 2 steps of AdamW at lr 0.01 move s by at most ~0.12 from ones
    assert float((s_last - 1).abs().max()) < 0.13 and float((s_last - 1).abs().max()) > 0


def _free_port():
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    p = sk.getsockname()[1]
    sk.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        seen = []

        def learn_block(b):
            seen.append(b)
            return torch.full((16,), float(b)) + torch.arange(16) / 100

        got = galt.learn_blocks_sharded(5, learn_block, 16)
        q.put((rank, seen, [g.tolist() for g in got]))
    finally:
        dist.destroy_process_group()


def test_blocks_sharded_gloo_world2():
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
    assert res[0][1] == [0, 2, 4] and res[1][1] == [1, 3]          # block b on rank b % world
    want = [(torch.full((16,), float(b)) + torch.arange(16) / 100).tolist() for b in range(5)]
    assert res[0][2] == want and res[1][2] == want


def test_blocks_single_process():
    got = galt.learn_blocks_sharded(3, lambda b: torch.full((4,), float(b)), 4)
    assert [g.tolist() for g in got] == [[0.0] * 4, [1.0] * 4, [2.0] * 4]
