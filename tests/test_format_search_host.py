"""format_search.search_layer on the CPU with injected quantizers (the oracle's): the batched form equals the
sample-by-sample loop of the reference (search/search_fp6_format.py:589-608) for row-local quantizers, is NOT the default for an
injected quantizer (it may be per-tensor), and both forms quantize the samples in their own dtype."""
import torch

from fpqvar_amd import format_search as fs
from oracle import fpq_oracle as orc


def _oracle_quant(fmt):
    tab = {"fp6_e2m3": "e2m3", "fp6_e3m2": "e3m2"}[fmt]
    return lambda t: orc.per_token_kernel_sem(t, tab)


def test_batched_equals_loop_for_row_local_quantizers():
    g = torch.Generator().manual_seed(1)
    w = (torch.randn(96, 256, generator=g) * 0.05)
    xs = [torch.randn(2, n, 256, generator=g) * (1 + j) for j, n in enumerate((1, 4, 9, 16))]
    wf, af, lb = fs.search_layer(xs, w, fs.FP6_FORMATS, quant=_oracle_quant, batched=True)
    wl, al, ll = fs.search_layer(xs, w, fs.FP6_FORMATS, quant=_oracle_quant, batched=False)
    assert (wf, af) == (wl, al) and set(lb) == set(ll)
    for k in ll:
        assert abs(lb[k] - ll[k]) <= 1e-5 * ll[k], (k, lb[k], ll[k])


def test_injected_quantizer_defaults_to_the_loop():
    g = torch.Generator().manual_seed(2)
    w = torch.randn(32, 128, generator=g) * 0.05
    xs = [torch.randn(2, 3, 128, generator=g) * s for s in (1.0, 8.0, 64.0)]

    def per_tensor(fmt):   # one scale for whatever tensor it is handed: not row-local
        def q(t):
            s = t.abs().max() / 6.0
            return orc.nearest_kernel((t / s).reshape(-1), orc.TABLES["e2m1"]).view(t.shape) * s
        return q
    _, _, default = fs.search_layer(xs, w, ("a", "b"), quant=per_tensor)
    _, _, loop = fs.search_layer(xs, w, ("a", "b"), quant=per_tensor, batched=False)
    _, _, forced = fs.search_layer(xs, w, ("a", "b"), quant=per_tensor, batched=True)
    assert default == loop
    assert abs(forced[("a", "a")] - loop[("a", "a")]) > 1e-3 * loop[("a", "a")]   # concatenation changes a per-tensor scale


def test_samples_are_quantized_in_their_own_dtype():
    """fp32 samples against an fp16 weight: the quantizer sees fp32 tensors in both forms (the reference hands the dumped
    activation to the quantizer as it is), the GEMM runs in the weight's dtype."""
    g = torch.Generator().manual_seed(3)
    w = (torch.randn(48, 128, generator=g) * 0.05).to(torch.bfloat16)   # a CPU-friendly low-precision weight dtype
    xs = [torch.randn(2, n, 128, generator=g) for n in (2, 5)]
    seen = []

    def spy(fmt):
        inner = _oracle_quant(fmt)

        def q(t):
            seen.append(t.dtype)
            return inner(t.float()).to(t.dtype)
        return q
    for batched in (True, False):
        seen.clear()
        fs.search_layer(xs, w, fs.FP6_FORMATS, quant=spy, batched=batched)
        assert torch.float32 in seen and all(d in (torch.float32, torch.bfloat16) for d in seen)
        # the activations (float32) are never cast before quantization; the weight is quantized in its own dtype
        assert seen.count(torch.bfloat16) == len(fs.FP6_FORMATS)
