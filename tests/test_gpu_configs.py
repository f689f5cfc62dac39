"""BASELINE.json's configurations at their exact shapes, HIP path (through the C ABI) against the CPU oracle.
Needs a real MI355X: run with ``-m gpu``.  Bit-exact (all NaNs equal, +0 / -0 distinguished)."""
import pytest
import torch

from fpqvar_amd import _lib
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


# ---- config 4's kernel: fp32 weights, groups of 128, the fast path (fpq_fast32.h) and the one-launch segment form ----
SYM = ("e2m1", "e1m2", "e3m0", "e2m3", "e3m2")


def _boundary_groups(table):
    """Groups whose scale is exactly 1 (max |x| = max|table|) and whose elements sit on, and +-1..4 ulp around, every
    level and every rounding boundary, both signs: the approximate division of the fast path must hand exactly
    these to the IEEE path."""
    tab = torch.unique(orc.TABLES[table])
    gmax = float(tab.abs().max())
    pts = torch.cat([tab, (tab[:-1] + tab[1:]) / 2])
    pts = pts[pts.abs() < gmax]
    vals = []
    for k in range(-4, 5):
        vals.append((pts.view(torch.int32) + k).view(torch.float32))
    vals = torch.cat(vals)
    vals = vals[torch.isfinite(vals) & (vals.abs() < gmax)]
    n = (vals.numel() + 126) // 127
    out = torch.zeros(n, 128)
    out[:, 0] = gmax
    flat = out[:, 1:].reshape(-1)
    flat[:vals.numel()] = vals
    out[:, 1:] = flat.view(n, 127)
    return out


@pytest.mark.parametrize("out_dtype", (torch.float16, torch.float32))
@pytest.mark.parametrize("table", SYM)
def test_fast32_groups_vs_oracle(dev, table, out_dtype):
    from fpqvar_amd import ops
    g = torch.Generator().manual_seed(41)
    parts = [torch.randn(300, 128, generator=g) * 0.02,
             torch.randn(200, 128, generator=g) * torch.exp(torch.randn(200, 128, generator=g)),
             _boundary_groups(table), _boundary_groups(table) * 0.0371, _boundary_groups(table) * 3.0e-20,
             _boundary_groups(table) * 1.7e19]
    edge = torch.randn(16, 128, generator=g)
    edge[0] = 0.0                                  # all-zero group: 0/0 -> +0
    edge[1, 5] = float("nan")
    edge[2, 7] = float("inf")
    edge[3, 9] = float("-inf")
    edge[4] *= 1e-41                               # subnormal group: scale underflows
    edge[5] *= 1e-38
    edge[6] *= 3e38 / 6                            # scale near the top of the range
    edge[7, ::2] = -0.0
    edge[8] = 1e-45                                # smallest subnormal everywhere
    edge[9, 1:] = 0.0                              # a single non-zero element
    edge[10] = -edge[10].abs()                     # single-sign groups
    edge[11] = edge[11].abs()
    edge[12] *= 1e-30                              # below 2^-90: IEEE path by the range check
    edge[13] *= 1e30
    parts.append(edge)
    x = torch.cat(parts)
    want = orc.per_group_kernel_sem(x, table, 128).to(out_dtype)
    got = ops.quant_rows(x.to(dev), table, 128, out_dtype)
    assert_bits_equal(got, want, f"fast32 {table} -> {out_dtype}")
    # ragged tail of the grid (not a multiple of the 32-group tile) and a one-group tensor
    for rows in (1, 31, 33):
        assert_bits_equal(ops.quant_rows(x[:rows].to(dev), table, 128, out_dtype), want[:rows], f"fast32 {table} rows={rows}")


@pytest.mark.parametrize("pack", (False, True))
@pytest.mark.parametrize("table", SYM)
def test_fast32_codes_vs_generic_emitter_and_oracle(dev, table, pack):
    """fp32 groups -> codes + fp32 scales through the approximate-then-verify path (groups32_codes_kernel, round 4) against
    the generic emitter (library switch FPQ_NO_FAST32: IEEE division + closed form) on the boundary-saturated groups and
    the edge rows of test_fast32_groups_vs_oracle, and decoded against the oracle's values."""
    import os
    from fpqvar_amd import ops
    if pack and table in ("e2m3", "e3m2"):
        pytest.skip("FP6 codes do not fit a nibble")
    g = torch.Generator().manual_seed(43)
    parts = [torch.randn(300, 128, generator=g) * 0.02, _boundary_groups(table), _boundary_groups(table) * 0.0371,
             _boundary_groups(table) * 3.0e-20, _boundary_groups(table) * 1.7e19]
    edge = torch.randn(14, 128, generator=g)
    edge[0] = 0.0
    edge[1, 5] = float("nan")
    edge[2, 7] = float("inf")
    edge[3, 9] = float("-inf")
    edge[4] *= 1e-41
    edge[5] *= 1e-38
    edge[6] *= 3e38 / 6
    edge[7, ::2] = -0.0
    edge[8] = 1e-45
    edge[9, 1:] = 0.0
    edge[10] = -edge[10].abs()
    edge[11] = edge[11].abs()
    edge[12] *= 1e-30
    edge[13] *= 1e30
    parts.append(edge)
    x = torch.cat(parts).to(dev)
    for rows in (x.shape[0], 1, 31, 33):
        xr = x[:rows].contiguous()
        codes, scales = ops.quant_rows_codes(xr, table, 128, pack)
        with _lib.option("FPQ_NO_FAST32", 1):
            codes_g, scales_g = ops.quant_rows_codes(xr, table, 128, pack)
        assert torch.equal(codes, codes_g), f"{table} pack={pack} rows={rows}: codes differ from the generic emitter"
        assert_bits_equal(scales, scales_g, f"{table} pack={pack} rows={rows}: scales")
        deq = ops.dequant_rows_codes(codes, scales, table, 128, torch.float32, pack)
        assert_bits_equal(deq, orc.per_group_kernel_sem(xr.cpu(), table, 128), f"{table} pack={pack} rows={rows}: decoded values vs oracle")


def test_fast32_codes_equal_generic_on_64m_weights(dev):
    import os
    from fpqvar_amd import ops
    g = torch.Generator(device=dev).manual_seed(6)
    for kind in ("weights", "uniform"):
        x = torch.randn(1 << 26, device=dev, generator=g) * 0.02 if kind == "weights" else (torch.rand(1 << 26, device=dev, generator=g) * 2 - 1)
        codes, scales = ops.quant_rows_codes(x.view(-1, 128), "e2m1", 128, True)
        with _lib.option("FPQ_NO_FAST32", 1):
            codes_g, scales_g = ops.quant_rows_codes(x.view(-1, 128), "e2m1", 128, True)
        assert torch.equal(codes, codes_g) and torch.equal(scales.view(torch.int32), scales_g.view(torch.int32)), kind


def test_fast32_equals_ieee_path_on_64m_weights(dev):
    """Fast path (approximate division + verification) against the generic kernel (IEEE division) on 2^26 random
    weights per distribution, on the GPU: bit-equal, i.e. every value the approximation cannot decide was caught."""
    import os
    from fpqvar_amd import ops
    g = torch.Generator(device=dev).manual_seed(5)
    for table in ("e2m1", "e2m3"):
        for kind in ("weights", "uniform"):
            x = torch.randn(1 << 26, device=dev, generator=g) * 0.02 if kind == "weights" else \
                (torch.rand(1 << 26, device=dev, generator=g) * 2 - 1)
            fast = ops.quant_rows(x, table, 128, torch.float16)
            with _lib.option("FPQ_NO_FAST32", 1):
                slow = ops.quant_rows(x, table, 128, torch.float16)
            assert bool((fast.view(torch.int16) == slow.view(torch.int16)).all()), f"{table} {kind}"
            rows = slice(0, 64 * 128)
            assert_bits_equal(fast[rows], orc.per_group_kernel_sem(x[rows].cpu(), table, 128).half(), f"{table} {kind} vs oracle")


def test_one_launch_over_a_segment_table(dev):
    """LocalShard: layers of different sizes, one launch, results are views of one slab, bit-equal to the oracle."""
    from fpqvar_amd import calibrate as cal
    g = torch.Generator().manual_seed(9)
    shapes = {"a.qkv": (384, 128), "a.proj": (128, 128), "a.fc1": (512, 256), "a.fc2": (128, 512), "tiny": (1, 128),
              "odd": (3, 5, 128)}
    w = {n: torch.randn(*s, generator=g) * 0.02 for n, s in shapes.items()}
    w["a.proj"][0, :128] = 0.0
    for table, fp_type in (("e2m1", "fp_e2"), ("e2m3", "fp6_e2m3")):
        for out_dtype in (torch.float16, torch.float32):
            shard = cal.LocalShard({n: t.to(dev) for n, t in w.items()}, weight_fp_type=fp_type, out_dtype=out_dtype)
            shard.quantize()
            views = shard.views()
            assert list(views) == list(shapes)
            for n in shapes:
                assert views[n].shape == shapes[n] and views[n].data_ptr() >= shard.slab.data_ptr()
                assert_bits_equal(views[n], orc.per_group_kernel_sem(w[n], table, 128).to(out_dtype), f"segment {n} {table}")
    # a non-contiguous input is made contiguous (as the reference's reshape would), an empty share is a no-op
    t = torch.randn(128, 256, generator=g)
    shard = cal.LocalShard({"t": t.to(dev).t()})
    shard.quantize()
    assert_bits_equal(shard.views()["t"], orc.per_group_kernel_sem(t.t().contiguous(), "e2m1", 128).half(), "transposed")
    assert cal.LocalShard({}, out=torch.empty(8, dtype=torch.float16, device=dev)).quantize().numel() == 8
    with pytest.raises(RuntimeError):
        cal.LocalShard({"h": torch.zeros(128, device=dev, dtype=torch.float16)})
    with pytest.raises(RuntimeError):
        cal.LocalShard({"r": torch.zeros(100, device=dev)})


def test_sharded_calibration_object_world1(dev):
    from fpqvar_amd import calibrate as cal
    shapes = cal.var_linear_shapes(2)
    shapes = {n: (o // 16, 128) for n, (o, i) in shapes.items()}          # small layers, groups of 128
    g = torch.Generator().manual_seed(3)
    w = {n: torch.randn(*s, generator=g) * 0.02 for n, s in shapes.items()}
    sc = cal.ShardedCalibration(shapes, {n: t.to(dev) for n, t in w.items()})
    got = sc.run()
    for n in shapes:
        assert_bits_equal(got[n], orc.per_group_kernel_sem(w[n], "e2m1", 128).half(), n)
    again = sc.run()                                 # reusable: same slab, same views
    assert all(again[n].data_ptr() == got[n].data_ptr() for n in shapes)


def _config4_weights(depth, dev):
    """Synthetic weights of every quantized Linear of VAR-d<depth> (randn * 0.02, seed 30 k + block; no real
    checkpoint exists offline), generated on the device - 1.33 G (d30) / 2.29 G (d36) fp32 elements."""
    from fpqvar_amd import calibrate as cal
    shapes = cal.var_linear_shapes(depth)
    kinds = {"attn.mat_qkv": 0, "attn.proj": 1, "ffn.fc1": 2, "ffn.fc2": 3}
    w = {}
    for n, s in shapes.items():
        _, b, a, k = n.split(".")
        g = torch.Generator(device=dev).manual_seed(depth * kinds[a + "." + k] + int(b))
        w[n] = torch.randn(*s, device=dev, generator=g) * 0.02
    return shapes, w


def _oracle_slices(name, w, got, n_groups=64):
    """first / last `n_groups` groups of a layer against the CPU oracle (the reference's fp32 quantization + .half())"""
    flat, gflat = w.reshape(-1, 128), got.reshape(-1, 128)
    for sl in (slice(0, n_groups), slice(flat.shape[0] - n_groups, flat.shape[0])):
        want = orc.per_group_kernel_sem(flat[sl].cpu(), "e2m1", 128).half()
        assert_bits_equal(gflat[sl], want, f"{name} groups {sl.start}..{sl.stop} vs oracle")


@pytest.mark.parametrize("depth", (30, 36))
def test_config4_one_launch_full_size(dev, depth):
    """BASELINE config 4 at its real size (tr/quant_utils.py:828-837 via quantize_VAR :1095-1167): all 120 (d30) / 144
    (d36) Linears in ONE fpq_quant_rows_segments launch - the launch shape bench.py times - bit for bit against one
    fpq_quant_rows launch per layer on EVERY layer, and against the oracle on the first / last 64 groups of the first,
    a middle and the last segment."""
    from fpqvar_amd import calibrate as cal, ops
    shapes, w = _config4_weights(depth, dev)
    assert len(w) == 4 * depth
    sc = cal.ShardedCalibration(shapes, w)
    assert len(sc.local.names) == 4 * depth and sc.local.total == sum(t.numel() for t in w.values())
    got = sc.run()
    torch.cuda.synchronize()
    names = list(shapes)
    for n in names:
        per_layer = ops.quant_rows(w[n], "e2m1", 128, torch.float16)
        assert got[n].shape == w[n].shape and got[n].dtype == torch.float16
        same = bool((got[n].view(torch.int16) == per_layer.view(torch.int16)).all())
        assert same, f"d{depth} {n}: one-launch segment output differs from the per-layer launch"
    for n in (names[0], names[len(names) // 2], names[len(names) // 2 + 2], names[-1]):
        _oracle_slices(f"d{depth} {n}", w[n], got[n])
    # the slab is exactly the layers back to back (world 1): nothing outside the segments was written or skipped
    assert sc.slab.numel() >= sc.local.total
    again = sc.run()
    assert all(again[n].data_ptr() == got[n].data_ptr() for n in names)


def test_config4_d30_codes_exchange_full_size(dev):
    """The packed exchange format of the sharded calibration (nibble codes + one fp32 scale per group, 0.53 B per
    element) at d30 size: decoded result bit-equal to the one-launch fp16 form on every layer."""
    from fpqvar_amd import calibrate as cal
    shapes, w = _config4_weights(30, dev)
    ref = cal.ShardedCalibration(shapes, w).run()
    got = cal.calibrate_sharded(w, exchange="codes")
    assert list(got) == list(shapes)
    for n in shapes:
        assert got[n].dtype == torch.float16 and got[n].shape == w[n].shape
        assert bool((got[n].view(torch.int16) == ref[n].view(torch.int16)).all()), f"codes exchange differs on {n}"
    _oracle_slices("codes " + list(shapes)[-1], w[list(shapes)[-1]], got[list(shapes)[-1]])


def test_fp16_exchange_three_ranks_on_one_gpu(dev, monkeypatch):
    """The fp16 slab exchange (ShardedCalibration: one launch into the rank's slot, one in-place all-gather, views of the
    slab) as three ranks would run it, replayed on one GPU with the collective replaced by a recorder / a replayer:
    every rank's slot offsets and the views every rank takes agree; bit-equal to per-layer fpq_quant_rows."""
    from fpqvar_amd import calibrate as cal, ops
    shapes = {f"l{i}": s for i, s in enumerate(((384, 128), (128, 512), (640, 256), (256, 128), (1024, 384), (128, 128), (896, 640)))}
    g = torch.Generator().manual_seed(10)
    w = {n: (torch.randn(*s, generator=g) * 0.02).to(dev) for n, s in shapes.items()}
    want = {n: ops.quant_rows(w[n], "e2m1", 128, torch.float16) for n in shapes}
    world, slots = 3, {}
    owners = cal.plan_owners(shapes, world)
    assert sorted(n for o in owners for n in o) == sorted(shapes)
    for r in range(world):
        monkeypatch.setattr(cal, "_world", lambda group, r=r: (r, world))
        monkeypatch.setattr(cal, "gather_slab", lambda slab, rank, group=None: slots.__setitem__(rank, slab[rank].clone()))
        cal.ShardedCalibration(shapes, {n: w[n] for n in owners[r]}).run()
    assert sorted(slots) == [0, 1, 2]

    def replay(slab, rank, group=None):
        for r in range(world):
            slab[r].copy_(slots[r])
    monkeypatch.setattr(cal, "gather_slab", replay)
    for r in range(world):
        monkeypatch.setattr(cal, "_world", lambda group, r=r: (r, world))
        got = cal.ShardedCalibration(shapes, {n: w[n] for n in owners[r]}).run()
        for n in shapes:
            assert_bits_equal(got[n], want[n], f"rank {r}: {n}")


def test_codes_exchange_three_ranks_on_one_gpu(dev, monkeypatch):
    """The packed exchange as THREE ranks would run it, replayed on one GPU: every rank quantizes its share into its slot
    (the collective replaced by a recorder), then each rank's decode runs on a slab filled with all recorded slots -
    offsets, slot widths and the decode table must agree between ranks that never talk about them.  Bit-equal to the
    one-launch fp16 form on every layer, on every rank."""
    from fpqvar_amd import calibrate as cal
    shapes = {f"l{i}": s for i, s in enumerate(((384, 128), (128, 512), (640, 256), (256, 128), (1024, 384), (128, 128), (896, 640)))}
    g = torch.Generator().manual_seed(9)
    w = {n: (torch.randn(*s, generator=g) * 0.02).to(dev) for n, s in shapes.items()}
    ref = cal.ShardedCalibration(shapes, w).run()
    world = 3
    slots = {}

    def record(slab, rank, group=None):
        slots[rank] = slab[rank].clone()
    monkeypatch.setattr(cal, "gather_slab", record)
    plan = cal.partition([(n, w[n].numel()) for n in w], world)
    for r in range(world):
        own = {n: (w[n] if n in plan[r] else torch.zeros((), device=dev).expand(shapes[n])) for n in shapes}
        cal._calibrate_codes(own, r, world, None, True)
    assert sorted(slots) == [0, 1, 2] and len({v.numel() for v in slots.values()}) == 1

    def replay(slab, rank, group=None):
        for r in range(world):
            slab[r].copy_(slots[r])
    monkeypatch.setattr(cal, "gather_slab", replay)
    for r in range(world):
        own = {n: (w[n] if n in plan[r] else torch.zeros((), device=dev).expand(shapes[n])) for n in shapes}
        got = cal._calibrate_codes(own, r, world, None, True)
        assert list(got) == list(shapes)
        for n in shapes:
            assert got[n].shape == w[n].shape and got[n].dtype == torch.float16
            assert bool((got[n].view(torch.int16) == ref[n].view(torch.int16)).all()), f"rank {r}: {n}"


@pytest.mark.parametrize("in_dtype", (torch.float32, torch.float16))
@pytest.mark.parametrize("pack", (True, False))
def test_codes_segments_equal_single_tensor_calls(dev, in_dtype, pack):
    """fpq_quant_rows_codes_segments / fpq_dequant_rows_codes_segments (the packed calibration exchange: many layers, one
    launch each way) against fpq_quant_rows_codes / fpq_dequant_rows_codes per layer, on layers of very different sizes
    (1 group ... 40000 groups, so that most workgroups of the small segments fall through), both dtypes, nibble-packed
    and byte codes; and the decoded values against the oracle's fake quantization."""
    from fpqvar_amd import _lib, ops
    g = torch.Generator().manual_seed(5)
    groups = (1, 3, 40000, 257, 16, 1024)
    xs = [(torch.randn(r, 128, generator=g) * (0.02 + 0.3 * i)).to(in_dtype).to(dev) for i, r in enumerate(groups)]
    xs[1][0, :] = 0
    cb = 64 if pack else 128
    codes = [torch.full((r, cb), 0xAB, dtype=torch.uint8, device=dev) for r in groups]
    scales = [torch.full((r,), float("nan"), dtype=in_dtype, device=dev) for r in groups]
    tab = torch.tensor([[x.data_ptr(), c.data_ptr(), s.data_ptr(), r] for x, c, s, r in zip(xs, codes, scales, groups)],
                       dtype=torch.int64).to(dev)
    lib = _lib.lib()
    rc = lib.fpq_quant_rows_codes_segments(tab.data_ptr(), len(groups), max(groups), 128, _lib.TABLE_IDS["e2m1"],
                                           _lib.dtype_id(in_dtype), int(pack), _lib.stream_ptr(dev))
    assert rc == 0
    outs = [torch.full((r, 128), float("nan"), dtype=torch.float16, device=dev) for r in groups]
    dtab = torch.tensor([[c.data_ptr(), s.data_ptr(), o.data_ptr(), r] for c, s, o, r in zip(codes, scales, outs, groups)],
                        dtype=torch.int64).to(dev)
    rc = lib.fpq_dequant_rows_codes_segments(dtab.data_ptr(), len(groups), max(groups), 128, _lib.TABLE_IDS["e2m1"],
                                             _lib.dtype_id(in_dtype), _lib.F16, int(pack), _lib.stream_ptr(dev))
    assert rc == 0
    for i, (x, c, s, o) in enumerate(zip(xs, codes, scales, outs)):
        c1, s1 = ops.quant_rows_codes(x, "e2m1", 128, pack_nibbles=pack)
        assert torch.equal(c.view(-1), c1.view(-1)), f"segment {i}: codes"
        assert_bits_equal(s, s1.view(-1), f"segment {i}: scales")
        o1 = ops.dequant_rows_codes(c1, s1, "e2m1", 128, torch.float16, pack)
        assert_bits_equal(o, o1.view_as(o), f"segment {i}: decoded")
        assert_bits_equal(o, orc.per_group_kernel_sem(x.cpu(), "e2m1", 128).half(), f"segment {i}: decoded vs oracle")
    # argument checks: nothing is enqueued for a bad call
    assert lib.fpq_quant_rows_codes_segments(tab.data_ptr(), len(groups), max(groups), 64, _lib.TABLE_IDS["e2m1"],
                                             _lib.dtype_id(in_dtype), int(pack), _lib.stream_ptr(dev)) != 0
    assert lib.fpq_dequant_rows_codes_segments(None, 1, 1, 128, _lib.TABLE_IDS["e2m1"], _lib.F32, _lib.F16, 1, _lib.stream_ptr(dev)) != 0
    assert lib.fpq_quant_rows_codes_segments(None, 0, 0, 128, _lib.TABLE_IDS["e2m1"], _lib.F32, 1, _lib.stream_ptr(dev)) == 0


@pytest.mark.parametrize("out_dtype", (torch.float16, torch.float32))
@pytest.mark.parametrize("cols", (1920, 2304, 7680, 9216, 512, 1000, 2048, 2056, 5000 * 2))
def test_fast32_long_rows_vs_oracle(dev, cols, out_dtype):
    """Per-channel fp32 weights (fp6_quant_e2m3_per_token_cuda on [out, in], tr/quant_utils.py:808-811; fp16 result)
    and the same rows with fp32 output: the wavefront- / workgroup-per-row fast path against the oracle, incl. rows
    whose scale is outside the fast path's range, boundary values and ragged row counts."""
    from fpqvar_amd import ops
    g = torch.Generator().manual_seed(cols)
    rows = 37
    x = torch.randn(rows, cols, generator=g) * 0.02
    x[3] *= torch.exp(torch.randn(cols, generator=g))
    x[4] = 0.0
    x[5, 17] = float("nan")
    x[6, 5] = float("inf")
    x[7] *= 1e-41
    x[8] *= 1e30
    x[9, ::2] = -0.0
    x[10, 1:] = 0.0
    tab = torch.unique(orc.TABLES["e2m3"])
    pts = torch.cat([tab, (tab[:-1] + tab[1:]) / 2])
    for k, r in zip(range(-3, 4), range(11, 18)):        # rows with scale 1 and elements on / around every boundary
        vals = (pts.view(torch.int32) + k).view(torch.float32)
        vals = vals[torch.isfinite(vals) & (vals.abs() < 7.5)]
        x[r] = 0.0
        x[r, 0] = 7.5
        x[r, 1:1 + vals.numel()] = vals[:cols - 1]
    for table in ("e2m3", "e3m2", "e2m1"):
        want = orc.per_token_kernel_sem(x, table, out_dtype)
        got = ops.quant_rows(x.to(dev), table, cols, out_dtype)
        assert_bits_equal(got, want, f"rows32 {table} cols={cols} -> {out_dtype}")
    assert_bits_equal(ops.quant_rows(x[:1].to(dev), "e2m3", cols, out_dtype), orc.per_token_kernel_sem(x[:1], "e2m3", out_dtype), "one row")


# ---- BASELINE config 4, the format search at its real size (search/search_fp6_format.py:576-608, search_fp4_format.py:782-821) ----
def _config4_search_layer(dev, block=0, n=100):
    """One d30 mat_qkv layer and its calibration dump: w [5760 x 1920], 100 samples x_j [2, pn^2, 1920] over the ten scale
    steps (models/basic_var.py:55-61 writes one tensor per (label pair, step)), 13600 rows in all; heavy-tailed like
    pre-quantization activations."""
    g = torch.Generator(device=dev).manual_seed(400 + block)
    pns = (1, 2, 3, 4, 5, 6, 8, 10, 13, 16)
    xs = []
    for j in range(n):
        shape = (2, pns[j % 10] ** 2, 1920)
        xs.append((torch.randn(shape, device=dev, generator=g) * torch.exp(0.5 * torch.randn(shape, device=dev, generator=g))).half())
    w = (torch.randn(5760, 1920, device=dev, generator=g) * 0.02).half()
    return xs, w


def test_config4_format_search_d30_layer(dev, qu):
    from fpqvar_amd import format_search as fs
    xs, w = _config4_search_layer(dev)
    assert sum(x.numel() // 1920 for x in xs) == 13600
    for formats in (fs.FP6_FORMATS, fs.FP4_FORMATS):
        wf, af, lb = fs.search_layer(xs, w, formats)                      # batched: one quantizer launch per format
        wl, al, ll = fs.search_layer(xs, w, formats, batched=False)       # the reference's sample-by-sample order
        assert set(lb) == set(ll) and len(lb) == len(formats) ** 2
        for key in ll:
            assert abs(lb[key] - ll[key]) <= 2e-3 * ll[key], (key, lb[key], ll[key])
        assert (wf, af) == (wl, al)
        assert (wf, af) == min(ll, key=ll.get)
    # what the losses are made of: the quantized operands of two samples and of a slice of the weight, against the oracle
    tab6, tab4 = {"fp6_e2m3": "e2m3", "fp6_e3m2": "e3m2"}, {"fp_e1": "e1m2", "fp_e2": "e2m1", "fp_e3": "e3m0"}
    for j in (3, 77):
        xc = xs[j].cpu()
        for f, t in tab6.items():
            assert_bits_equal(fs.quantizer(f)(xs[j]), orc.per_token_kernel_sem(xc, t), f"sample {j} {f}")
        for f, t in tab4.items():
            assert_bits_equal(fs.quantizer(f)(xs[j]), orc.per_group_kernel_sem(xc, t, 128), f"sample {j} {f}")
    wc = w[:256].cpu()
    for f, t in tab6.items():
        assert_bits_equal(fs.quantizer(f)(w)[:256], orc.per_token_kernel_sem(wc, t), f"weight {f}")
    for f, t in tab4.items():
        assert_bits_equal(fs.quantizer(f)(w)[:256], orc.per_group_kernel_sem(wc, t, 128), f"weight {f}")
    # the batched form's precondition: a sample's rows come out of the concatenated launch as out of its own
    x_all = torch.cat([x.reshape(-1, 1920) for x in xs])
    for f in list(tab6) + list(tab4):
        q_all = fs.quantizer(f)(x_all)
        o = sum(x.numel() // 1920 for x in xs[:77])
        assert torch.equal(q_all[o:o + xs[77].numel() // 1920].view(torch.int16), fs.quantizer(f)(xs[77]).reshape(-1, 1920).view(torch.int16))


def test_format_search_batched_default_only_for_row_local_quantizers(dev, qu):
    """An injected quantizer is not assumed to be row-local: the default then is the sample-by-sample loop (a per-tensor
    quantizer sees another tensor once the samples are concatenated); fp32 samples against an fp16 weight are quantized
    in THEIR dtype in both forms."""
    from fpqvar_amd import format_search as fs
    g = torch.Generator(device=dev).manual_seed(5)
    xs = [torch.randn(2, n, 512, device=dev, generator=g) * (1 + j) for j, n in enumerate((4, 9, 16))]     # fp32 samples
    w = (torch.randn(256, 512, device=dev, generator=g) * 0.05).half()
    _, _, lb = fs.search_layer(xs, w, fs.FP4_FORMATS)
    _, _, ll = fs.search_layer(xs, w, fs.FP4_FORMATS, batched=False)
    for key in ll:
        assert abs(lb[key] - ll[key]) <= 1e-4 * ll[key], (key, lb[key], ll[key])

    def per_tensor(fmt):
        return lambda t: qu.fp_quant_e2_per_tensor(t)[0].to(t.dtype)
    _, _, ld = fs.search_layer(xs, w, ("a", "b"), quant=per_tensor)                    # default: the loop
    _, _, lf = fs.search_layer(xs, w, ("a", "b"), quant=per_tensor, batched=False)
    assert ld == lf
    _, _, lt = fs.search_layer(xs, w, ("a", "b"), quant=per_tensor, batched=True)      # forced: another (wrong) number
    assert abs(lt[("a", "a")] - lf[("a", "a")]) > 1e-3 * lf[("a", "a")]


def test_config4_format_search_sharded_two_ranks_share_the_gpu():
    """search_blocks_sharded at config-4 size: 4 blocks of [5760 x 1920] x 100 samples on 2 real ranks sharing the GPU (gloo)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, FPQ_CHECK_BACKEND="gloo", FPQ_CHECK_ONLY="search", FPQ_CHECK_SEARCH="config4")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", "29547", os.path.join(root, "tools", "rccl_world2_check.py")],
                         capture_output=True, text=True, timeout=300, cwd=root, env=env)
    text = out.stdout + out.stderr
    assert out.returncode == 0, text[-2000:]
    assert text.count("equals the single-process result: True") == 2, text[-2000:]
