"""The compiled host binding (fpqvar_amd/_native, csrc/quant_cuda_ext.cpp) against the ctypes path of ops.py: same C entry
points, same results, same error behaviour as the reference's module (quant/quant.cpp:17-29: RuntimeError from torch's
checks; the quant_utils.py functions: AssertionError for a wrong n_bits)."""
import os

import pytest
import torch

from tests.conftest import assert_bits_equal

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

NAMES = ("quant", "quant_rows", "quant_rows_dual", "fp_quant_e1_per_group_cuda", "fp_quant_e2_per_group_cuda",
         "fp_quant_e3_per_group_cuda", "fp_quant_e1m2_neg_e2m1_pos_per_group_cuda", "fp4_afpq_per_group_cuda",
         "fp6_quant_e2m3_per_group_cuda", "fp6_quant_e3m2_per_group_cuda", "fp6_quant_int_neg_e2m3_pos_per_group_cuda",
         "fp6_quant_per_token_contig", "fp6_quant_int_neg_e2m3_pos_per_token_contig")


def test_native_module_is_built_and_bound():
    """__graft_entry__.build() produces the module; quant_cuda.quant and the hot quant_utils names ARE its functions."""
    from fpqvar_amd import _lib, _native, quant_utils as qu
    import quant_cuda
    assert _native.fpq_version() == _lib.lib().fpq_version()
    for n in NAMES:
        assert callable(getattr(_native, n)), n
    assert quant_cuda.quant is _native.quant
    # BASELINE north_star / SURVEY 8b L1: the reference-named fused functions hang off the module `quant_cuda` itself
    assert quant_cuda.fp_quant_e2_per_group_cuda is _native.fp_quant_e2_per_group_cuda
    assert quant_cuda.fp_quant_e1m2_neg_e2m1_pos_per_group_cuda is _native.fp_quant_e1m2_neg_e2m1_pos_per_group_cuda
    assert quant_cuda.fp6_quant_int_neg_e2m3_pos_per_group_cuda is _native.fp6_quant_int_neg_e2m3_pos_per_group_cuda
    for n in quant_cuda.FP_QUANT_NAMES:
        assert getattr(quant_cuda, n) is getattr(qu, n), n
    for n in ("rotate_quant_mx", "adaln_rotate_quant_mx", "adaln_rotate_quant_token", "adaln_rotate_quant_token_codes",
              "kv_cache_step", "linear_fp4"):
        assert callable(getattr(_native, n)), n
    assert qu.fp_quant_e2_per_group_cuda is _native.fp_quant_e2_per_group_cuda
    assert qu.fp_quant_e1m2_neg_e2m1_pos_per_group_cuda is _native.fp_quant_e1m2_neg_e2m1_pos_per_group_cuda
    # no CPU path in the compiled module either
    x = torch.randn(4, 128)
    with pytest.raises(RuntimeError, match="GPU"):
        _native.fp_quant_e2_per_group_cuda(x.half(), 4)
    with pytest.raises(RuntimeError, match="GPU"):
        _native.quant(x.view(-1), qu.fp4_e2m1_grid)
    with pytest.raises(AssertionError):
        _native.fp_quant_e2_per_group_cuda(x.half(), 8)
    with pytest.raises(AssertionError):
        _native.fp6_quant_e2m3_per_group_cuda(x.half(), 4)


@pytest.mark.gpu
def test_native_equals_ctypes_path():
    from fpqvar_amd import _native, ops, quant_utils as qu
    import quant_cuda
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(5)
    x16 = (torch.randn(3, 37, 1920, generator=g) * 1.7).half().to(dev)
    x32 = (torch.randn(64, 1024, generator=g) * 0.02).to(dev)
    h = torch.nn.functional.gelu(torch.randn(50, 7680, generator=g), approximate="tanh").half().to(dev)
    h[7, 300] = float("nan")
    for x in (x16, x32):
        for name, tab in (("fp_quant_e1_per_group_cuda", "e1m2"), ("fp_quant_e2_per_group_cuda", "e2m1"), ("fp_quant_e3_per_group_cuda", "e3m0")):
            got = getattr(_native, name)(x, 4, 128)
            assert_bits_equal(got, ops.quant_rows(x, tab, 128), name)
            assert_bits_equal(getattr(qu, name)(x, 4), got, name + " via quant_utils, default group")
        for name, tab in (("fp6_quant_e2m3_per_group_cuda", "e2m3"), ("fp6_quant_e3m2_per_group_cuda", "e3m2")):
            assert_bits_equal(getattr(_native, name)(x, 6), ops.quant_rows(x, tab, 128, torch.float16), name)
        assert_bits_equal(qu.fp6_quant_e2m3_per_token_cuda(x, 6), ops.quant_rows(x, "e2m3", x.shape[-1], torch.float16), "per token")
        assert_bits_equal(qu.fp6_quant_e3m2_per_token_cuda(x, 6), ops.quant_rows(x, "e3m2", x.shape[-1], torch.float16), "per token e3m2")
        assert_bits_equal(_native.quant_rows(x, 0, 128, torch.float16), ops.quant_rows(x, "e2m1", 128, torch.float16), "out dtype")
    hc = h.clone()
    hc[7, 300] = 0.5
    for t in (h, hc):     # with and without a NaN: the default clipping strength's all-zero rule and the self-cleaning scratch
        for _ in range(2):
            assert_bits_equal(_native.fp_quant_e1m2_neg_e2m1_pos_per_group_cuda(t, 4), ops.quant_rows_dual(t, "e1m2_neg", "e2m1_pos", 128, 1.0), "dual fp4")
        assert_bits_equal(_native.fp_quant_e1m2_neg_e2m1_pos_per_group_cuda(t, 4, 128, 0.75), ops.quant_rows_dual(t, "e1m2_neg", "e2m1_pos", 128, 0.75), "dual clip 0.75")
        assert_bits_equal(_native.fp4_afpq_per_group_cuda(t, 4), ops.quant_rows_dual(t, "e2m1_neg", "e2m1_pos", 128, 1.0), "afpq")
        assert_bits_equal(_native.fp6_quant_int_neg_e2m3_pos_per_group_cuda(t, 6), ops.quant_rows_dual(t, "int_neg", "e2m3_pos", 128, None), "dual fp6")
        assert_bits_equal(qu.fp6_quant_int_neg_e2m3_pos_per_token_cuda(t, 6), ops.quant_rows_dual(t, "int_neg", "e2m3_pos", t.shape[-1], None), "dual fp6 per token")
    # non-contiguous input: reshape semantics for per-group (copy), torch's .view(-1) rule for per-token (quant_utils decides)
    xt = torch.randn(256, 64, generator=g).half().to(dev).t()
    assert_bits_equal(qu.fp_quant_e2_per_group_cuda(xt, 4, 128), ops.quant_rows(xt.contiguous(), "e2m1", 128), "transposed")
    with pytest.raises(RuntimeError):
        qu.fp6_quant_e2m3_per_token_cuda(torch.randn(2, 3, 4, 64, device=dev).half().permute(0, 2, 1, 3), 6)
    # quant_cuda.quant through both bindings
    tab = qu.fp4_e2m1_grid.to(dev)
    xs = torch.randn(1000, generator=g).to(dev) * 3
    z, idx = quant_cuda.quant(xs, tab)
    z2, idx2 = quant_cuda._quant_ctypes(xs, tab)
    assert_bits_equal(z, z2, "quant")
    assert idx.shape == idx2.shape == xs.shape and not idx.any() and idx.dtype == xs.dtype
    with pytest.raises(RuntimeError):
        quant_cuda.quant(xs.half(), tab)
    with pytest.raises(RuntimeError):
        quant_cuda.quant(xs, torch.zeros(257, device=dev))


@pytest.mark.gpu
def test_native_dual_under_graph_capture():
    """The default fc2 quantizer inside a captured graph: the NaN scratch of a capturing stream is the capture's own, and a
    replay behaves like the eager call, with and without a NaN in the input."""
    from fpqvar_amd import _native, ops
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(9)
    static = torch.nn.functional.gelu(torch.randn(40, 7680, generator=g)).half().to(dev)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        _native.fp_quant_e1m2_neg_e2m1_pos_per_group_cuda(static, 4)
    torch.cuda.current_stream().wait_stream(s)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        y = _native.fp_quant_e1m2_neg_e2m1_pos_per_group_cuda(static, 4)
    for trial, poison in enumerate((False, True, False)):
        t = torch.nn.functional.gelu(torch.randn(40, 7680, generator=g)).half()
        if poison:
            t[3, 77] = float("nan")
        static.copy_(t.to(dev))
        graph.replay()
        eager = ops.quant_rows_dual(static, "e1m2_neg", "e2m1_pos", 128, 1.0)     # eager call beside the replays, same stream
        torch.cuda.synchronize()
        assert_bits_equal(y, eager, f"replay {trial}")
        assert bool((y == 0).all()) == poison


@pytest.mark.gpu
def test_native_q_path_equals_ctypes_path(monkeypatch):
    """The operand-emitting producers, the per-token producer, the KV-cache step and the FP4 GEMM through the compiled
    binding against the same calls through ctypes (the wrappers with `_native` taken away): same C entry points."""
    from fpqvar_amd import _native, gemm, ops, rotation as rot
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(11)
    B, L, C = 4, 37, 1920
    x16 = torch.randn(B, L, C, generator=g).half().to(dev)
    x32 = torch.randn(B, L, C, generator=g).to(dev)
    sc = (torch.randn(B, 1, C, generator=g) * 0.3).half().to(dev)
    sh = (torch.randn(B, 1, C, generator=g) * 0.3).half().to(dev)
    sm = (torch.rand(C, generator=g) + 0.5).to(dev)

    def both(fn):
        assert rot._native is _native and ops._native is _native and gemm._native is _native
        a = fn()
        with monkeypatch.context() as m:
            for mod in (rot, ops, gemm):
                m.setattr(mod, "_native", None)
            b = fn()
        return a, b

    def same(a, b, what):
        a, b = (a, b) if isinstance(a, tuple) else ((a,), (b,))
        for i, (p, q) in enumerate(zip(a, b)):
            assert p.dtype == q.dtype and p.shape == q.shape and torch.equal(p.view(torch.uint8), q.view(torch.uint8)), f"{what}[{i}]"

    for x in (x16, x32):
        for smooth in (None, sm):
            same(*both(lambda: rot.rotate_quant_mx(x, smooth=smooth)), "rotate_quant_mx")
            same(*both(lambda: rot.adaln_rotate_quant_mx(x, sc, sh, smooth=smooth)), "adaln_rotate_quant_mx")
            for emit in ("values", "fp8", "fp6"):
                same(*both(lambda: rot.adaln_rotate_quant_token(x, sc, sh, "e2m3", smooth=smooth, emit=emit)), "token " + emit)
    with pytest.raises(RuntimeError):
        rot.adaln_rotate_quant_token(x16, sc, sh, "e3m2", emit="fp6")
    with pytest.raises(RuntimeError):
        rot.adaln_rotate_quant_token(x16, sc, sh, "e2m3", emit="nibbles")
    # the FP4 GEMM with and without the gated residual of the AdaLN block
    a = gemm.quantize_mx(x16.reshape(-1, C))
    w = gemm.quantize_mx((torch.randn(256, C, generator=g) * 0.02).to(dev))
    bias = torch.randn(256, generator=g).half().to(dev)
    gate = torch.randn(B, 1, 256, generator=g).half().to(dev)
    res = torch.randn(B, L, 256, generator=g).half().to(dev)
    same(*both(lambda: gemm.linear_fp4(*a, *w)), "linear_fp4")
    same(*both(lambda: gemm.linear_fp4(*a, *w, bias, gate, res)), "linear_fp4 + epilogue")
    with pytest.raises(RuntimeError):
        gemm.linear_fp4(a[0][:, :-64], a[1], *w)
    # the fused fc1 tail and its stand-alone twin (round 5)
    same(*both(lambda: gemm.linear_fp4_gelu_dual(*a, *w, bias, return_gelu=True)), "linear_fp4_gelu_dual")
    same(*both(lambda: gemm.linear_fp4_gelu_dual(*a, *w)), "linear_fp4_gelu_dual without bias / GELU output")
    yq = (torch.randn(B, L, 256, generator=g) * 1.5).half().to(dev)
    same(*both(lambda: ops.gelu_quant_rows_dual(yq, return_gelu=True)), "gelu_quant_rows_dual")
    same(*both(lambda: ops.gelu_quant_rows_dual(yq)), "gelu_quant_rows_dual without the GELU output")
    # edge cases of the compiled binding: bias / gate / residual views at an odd storage offset (2 bytes past a
    # 16-byte boundary) give the aligned result bit for bit; zero tokens / zero outputs are valid and enqueue nothing
    want = _native.linear_fp4(*a, *w, bias, gate, res)
    off = lambda t: torch.cat([t.reshape(-1)[:1], t.reshape(-1)])[1:].view(t.shape)     # same values, data_ptr + 2
    assert off(bias).data_ptr() % 16 == 2
    same(_native.linear_fp4(*a, *w, off(bias), off(gate), off(res)), want, "linear_fp4 with misaligned bias / gate / residual")
    same(_native.linear_fp4(*a, *w, off(bias)), _native.linear_fp4(*a, *w, bias), "linear_fp4 with a misaligned bias alone")
    empty_a = (a[0][:0], a[1][:0])
    assert _native.linear_fp4(*empty_a, *w, bias, gate[:0], res[:0]).shape == (0, 256)
    empty_w = (w[0][:0], w[1][:0])
    assert _native.linear_fp4(*a, *empty_w).shape == (a[0].shape[0], 0)
    # the KV-cache step
    from fpqvar_amd import kv_cache as kvc
    outs = []
    for native in (True, False):
        with monkeypatch.context() as m:
            if not native:
                m.setattr(ops, "_native", None)
            cache = kvc.IncrementalKVCache(2, 16, 30, 64, 6, device=dev)
            gg = torch.Generator().manual_seed(3)
            for n in (1, 4, 9):
                k = torch.randn(2, n, 30, 64, generator=gg).half().to(dev)
                v = torch.randn(2, n, 30, 64, generator=gg).half().to(dev)
                kk, vv = cache.append(k, v)
            outs.append((kk.clone(), vv.clone()))
    same(outs[0], outs[1], "kv_cache_step")
