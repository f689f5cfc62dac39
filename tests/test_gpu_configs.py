"""BASELINE.json's configurations at their exact shapes, HIP path (through the C ABI) against the CPU oracle.
Needs a real MI355X: run with ``-m gpu``.  Bit-exact (all NaNs equal, +0 / -0 distinguished)."""
import pytest
import torch

from oracle import fpq_oracle as orc
from tests.conftest import assert_bits_equal, from_bits

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need the GPU"
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def qu():
    import fpqvar_amd.quant_utils as m
    return m


# ---- config 1: single [4096 x 1024] fp32 tensor, per_tensor fp_e2 (E2M1) via the pure-torch path ----------------
def test_config1_per_tensor_e2m1_4096x1024(dev, qu):
    """search/baseline/plot_weight_distribution_for_motivation.py:285-294 at BASELINE.json's shape (SURVEY.md 8d:
    randn(4096, 1024) fp32, seed 0)."""
    g = torch.Generator().manual_seed(0)
    x = torch.randn(4096, 1024, generator=g)
    got, scale = qu.fp_quant_e2_per_tensor(x.to(dev))
    want = orc.per_tensor_argmin_sem(x, "e2m1")
    assert_bits_equal(got, want, "config 1 output")
    want_scale = x.abs().max() / orc.TABLES["e2m1"].abs().max()
    assert scale.shape == () and scale.dtype == torch.float32 and scale.device.type == "cuda"
    assert_bits_equal(scale, want_scale, "config 1 scale")
    # every output is a level of the one grid; the extreme element lands on +-6 * scale exactly
    lv = (orc.TABLES["e2m1"] * want_scale)
    assert bool((got.cpu().reshape(-1, 1) == lv.reshape(1, -1)).any(dim=-1).all())
    k = int(x.abs().argmax())
    assert got.reshape(-1)[k].item() == (6.0 * want_scale * torch.sign(x.reshape(-1)[k])).item()


def test_config1_per_tensor_golden(dev, qu, golden):
    """The fixture the reference's own quantize_to_nearest_grid produced (tests/golden/make_golden.py section 3)."""
    x = from_bits(golden["in/per_tensor_f32"])
    got, _ = qu.fp_quant_e2_per_tensor(x.to(dev))
    assert_bits_equal(got, from_bits(golden["out/per_tensor_argmin/e2m1"]), "per-tensor golden")


@pytest.mark.parametrize("table,fn", (("e2m1", "fp_quant_e2_per_tensor"), ("e1m2", "fp_quant_e1_per_tensor"),
                                      ("e3m0", "fp_quant_e3_per_tensor")))
@pytest.mark.parametrize("dtype", (torch.float32, torch.float16))
def test_per_tensor_dtypes_shapes_edges(dev, qu, table, fn, dtype):
    f = getattr(qu, fn)
    g = torch.Generator().manual_seed(3)
    cases = {
        "gauss": torch.randn(257, 129, generator=g),
        "heavy": torch.randn(64, 1000, generator=g) * torch.exp(0.5 * torch.randn(64, 1000, generator=g)),
        "ties": torch.tensor([0.25, -0.25, 0.75, 1.25, 1.75, 2.5, 3.5, 5, -5, -2.5, 6.0, -6.0, 0.0, -0.0]),
        "one": torch.tensor([[-3.5]]),
        "five": torch.randn(5, generator=g),
        "ragged": torch.randn(1027, generator=g),
        "zeros": torch.zeros(300),
        "nan": torch.tensor([1.0, float("nan"), -2.0, 0.5, 3.0]),
        "inf": torch.tensor([1.0, float("inf"), -2.0, 0.5, 3.0]),
        "tiny": torch.randn(70, generator=g) * 1e-7,
    }
    for name, x in cases.items():
        x = x.to(dtype)
        got, scale = f(x.to(dev))
        assert got.dtype == torch.float32 and got.shape == x.shape
        assert_bits_equal(got, orc.per_tensor_argmin_sem(x, table), f"{fn} {dtype} {name}")
        assert_bits_equal(scale, x.abs().max() / orc.TABLES[table].abs().max(), f"{fn} {dtype} {name} scale")
    # views: a transposed tensor and an unaligned base pointer take the same values
    x = torch.randn(96, 200, generator=g).to(dtype)
    got, _ = f(x.to(dev).t())
    assert_bits_equal(got, orc.per_tensor_argmin_sem(x.t(), table), f"{fn} transposed")
    flat = torch.randn(4099, generator=g).to(dtype)
    got, _ = f(flat.to(dev)[3:])
    assert_bits_equal(got, orc.per_tensor_argmin_sem(flat[3:], table), f"{fn} unaligned")
    with pytest.raises(RuntimeError):
        f(torch.empty(0, device=dev, dtype=dtype))
    with pytest.raises(RuntimeError, match="GPU"):
        f(x)


# ---- config 2: VAR-d16 mat_qkv weights, per_group g=128 fp_e2 W4 --------------------------------------------------
@pytest.mark.parametrize("block", (0, 7, 15))
def test_config2_d16_mat_qkv_weights(dev, qu, block):
    """[3072 x 1024] fp32 (randn * 0.02, seed 16 + block) through fp_quant_e2_per_group_cuda
    (tr/quant_utils.py:828-837 calls it on the fp32 weight), then the driver's .half()."""
    g = torch.Generator().manual_seed(16 + block)
    w = torch.randn(3072, 1024, generator=g) * 0.02
    got = qu.fp_quant_e2_per_group_cuda(w.to(dev), 4, 128)
    want = orc.per_group_kernel_sem(w, "e2m1", 128)
    assert_bits_equal(got, want, f"config 2 block {block}")
    from fpqvar_amd import calibrate as cal
    got16 = cal.default_weight_quantizer()("blocks.%d.attn.mat_qkv" % block, w.to(dev))
    assert_bits_equal(got16, want.half(), f"config 2 block {block}, fused .half()")
