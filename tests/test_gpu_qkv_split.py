"""The FP4 GEMM with a split output (fpq_gemm_fp4_mx_split; include/fpq.h): mat_qkv writing q to its own tensor and k, v
straight into the KV cache's slots - bit-identical to the plain GEMM followed by the cache's copy-in
(tr/basic_var.py:173-209; kv_cache.IncrementalKVCache.append)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _dev():
    return torch.device("cuda:0")


def _operands(tokens, c, kmajor, seed):
    from fpqvar_amd import gemm
    torch.manual_seed(seed)
    x = torch.randn(tokens, c, device=_dev()).half()
    w = torch.randn(3 * c, c, device=_dev()) * 0.05
    bias = (torch.randn(3 * c, device=_dev()) * 0.1).half()
    a, wq = gemm.quantize_mx(x), gemm.quantize_mx(w)
    if kmajor:
        a = gemm.quantize_mx(x, kmajor=True)
        wk = (gemm.to_kmajor(wq[0], 4, dealt=True), gemm.to_kmajor_scales(wq[1], weight_side=True))
        return a, wk, bias, gemm.linear_fp4(*gemm.quantize_mx(x), *wq, bias)
    return a, wq, bias, gemm.linear_fp4(*a, *wq, bias)


@pytest.mark.parametrize("bsz,seq,heads", [(2, 1, 2), (3, 9, 2), (2, 25, 4), (5, 64, 2), (2, 169, 4), (3, 256, 2), (100, 1, 30), (7, 100, 30), (2, 2116, 4)])
@pytest.mark.parametrize("kmajor", [False, True])
@pytest.mark.parametrize("cfg", [None, 10, 20, 30])
def test_qkv_to_cache_equals_plain_gemm_plus_copy(bsz, seq, heads, kmajor, cfg, lib_options):
    from fpqvar_amd import gemm
    if cfg is not None:
        lib_options("FPQ_GEMM_CFG", cfg)
    c, max_len, pos = heads * 64, seq + 37, 11
    a, w, bias, qkv = _operands(bsz * seq, c, kmajor, bsz * seq + heads)
    want_q, want_k, want_v = qkv.view(bsz, seq, 3, heads, 64).unbind(2)
    cache = torch.full((2, bsz, max_len, heads, 64), 7.5, dtype=torch.float16, device=_dev())
    q = gemm.linear_fp4_qkv_to_cache(*a, *w, bias, cache, pos, seq)
    assert q.shape == (bsz, seq, c) and torch.equal(q.view(bsz, seq, heads, 64), want_q)
    assert torch.equal(cache[0, :, pos:pos + seq], want_k) and torch.equal(cache[1, :, pos:pos + seq], want_v)
    untouched = torch.ones(max_len, dtype=torch.bool, device=_dev())
    untouched[pos:pos + seq] = False
    assert bool((cache[:, :, untouched] == 7.5).all()), "the GEMM wrote outside its slots"


def test_qkv_to_cache_rejects_what_does_not_fit():
    from fpqvar_amd import gemm
    a, w, bias, _ = _operands(2 * 9, 128, True, 1)
    cache = torch.zeros(2, 2, 20, 2, 64, dtype=torch.float16, device=_dev())
    with pytest.raises(RuntimeError):
        gemm.linear_fp4_qkv_to_cache(*a, *w, bias, cache, 12, 9)        # 12 + 9 > max_len
    with pytest.raises(RuntimeError):
        gemm.linear_fp4_qkv_to_cache(*a, *w, bias, cache[:, :1], 0, 9)  # not contiguous / wrong batch
    with pytest.raises(RuntimeError):
        gemm.linear_fp4_qkv_to_cache(*a, *w, bias, cache.float(), 0, 9)


def test_incremental_cache_commit_written_equals_append():
    """five steps of a generation on two caches: append(k, v) against the split GEMM + commit_written - same cache contents, same
    views, the previous step's entries quantized exactly once in both"""
    from fpqvar_amd import gemm, kv_cache
    bsz, heads, c = 3, 4, 256
    steps = (1, 4, 9, 16, 25)
    ca = kv_cache.IncrementalKVCache(bsz, sum(steps), heads, 64, 6, device=_dev())
    cb = kv_cache.IncrementalKVCache(bsz, sum(steps), heads, 64, 6, device=_dev())
    ca.kv.zero_()
    cb.kv.zero_()
    for i, seq in enumerate(steps):
        a, w, bias, qkv = _operands(bsz * seq, c, True, 100 + i)
        _, k, v = qkv.view(bsz, seq, 3, heads, 64).unbind(2)
        ka, va = ca.append(k, v)
        gemm.linear_fp4_qkv_to_cache(*a, *w, bias, cb.kv, cb.len, seq)
        kb, vb = cb.commit_written(seq)
        assert torch.equal(ka, kb) and torch.equal(va, vb), i
    assert torch.equal(ca.kv, cb.kv) and ca.len == cb.len == sum(steps)


def test_generation_batch_with_and_without_the_split_output():
    from fpqvar_amd import var_block
    outs = []
    for flag in (True, False):
        gb = var_block.GenerationBatch("d30-256", "w4a4", depth=2, batch_rows=4, device="cuda:0", seed=3, qkv_to_cache=flag)
        assert gb.qkv_to_cache == flag
        caches = gb.new_caches("Q")
        ys = [gb.step("Q", caches, gb.new_input(pn)) for pn in gb.patch_nums[:6]]
        outs.append((ys, [c.kv[:, :, :c.len].clone() for c in caches]))
    for y0, y1 in zip(outs[0][0], outs[1][0]):
        assert torch.equal(y0, y1)
    for c0, c1 in zip(outs[0][1], outs[1][1]):
        assert torch.equal(c0, c1)
