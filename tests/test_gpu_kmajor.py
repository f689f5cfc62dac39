"""K-major operand images (include/fpq.h, "K-MAJOR OPERAND IMAGES"): the converter against the definition restated in torch,
every *_km producer against the converter applied to its row-major sibling (bit for bit, scales included), and the *_km GEMMs
against the row-major GEMMs on the same codes (bit for bit: same arithmetic, only the operand's address pattern differs)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _dev():
    return torch.device("cuda:0")


def image_by_definition(codes: torch.Tensor, seg: int, dealt: bool) -> torch.Tensor:
    """include/fpq.h: image[s, j, p * 16 + b] = codes[row(j), s * seg + c(j, p) * 16 + b]"""
    rows, row_bytes = codes.shape
    steps, cps = row_bytes // seg, seg // 16
    image_rows = (rows + 63) // 64 * 64 if dealt else rows
    j = torch.arange(image_rows, device=codes.device)
    row = (j & ~63) + 4 * (j & 15) + ((j >> 4) & 3) if dealt else j
    p = torch.arange(cps, device=codes.device)
    if seg == 64:
        pi = torch.tensor([0, 2, 3, 1], device=codes.device)[(j & 15) >> 2]
        c = p[None, :] ^ pi[:, None]                                        # [j, p]
    else:
        c = (p[None, :] - ((j >> 3) & 1)[:, None]) % 6
    src = torch.cat([codes, codes.new_zeros(1, row_bytes)])                 # row `rows` = the zero row of the padding
    row = torch.where(row < rows, row, torch.full_like(row, rows))
    chunks = src.view(rows + 1, steps, cps, 16)[row]                        # [j, s, c, 16]
    out = torch.take_along_dim(chunks, c[:, None, :, None].expand(image_rows, steps, cps, 16), dim=2)
    return out.permute(1, 0, 2, 3).reshape(steps, image_rows, seg).contiguous()


@pytest.mark.parametrize("rows", [1, 16, 33, 301, 392, 4096])
@pytest.mark.parametrize("bits,seg", [(4, 64), (6, 96)])
@pytest.mark.parametrize("dealt", [False, True])
def test_converter_matches_the_definition(rows, bits, seg, dealt):
    from fpqvar_amd import gemm
    torch.manual_seed(rows)
    codes = torch.randint(0, 256, (rows, 15 * seg), dtype=torch.uint8, device=_dev())
    got = gemm.to_kmajor(codes, bits, dealt=dealt)
    want = image_by_definition(codes, seg, dealt)
    assert got.shape == want.shape and torch.equal(got, want)


@pytest.mark.parametrize("rows", [1, 7, 64, 301, 392])
@pytest.mark.parametrize("dtype", [torch.float16, torch.float32])
@pytest.mark.parametrize("weight_side", [False, True])
def test_scale_converter_matches_the_definition(rows, dtype, weight_side):
    """include/fpq.h: scale_image[g, r] = (float) scales[r, g], rows rounded up to 4 (64 on the weight side), zero padding"""
    from fpqvar_amd import gemm
    torch.manual_seed(rows)
    sc = (torch.rand(rows, 15, device=_dev()) + 0.01).to(dtype)
    got = gemm.to_kmajor_scales(sc, weight_side=weight_side)
    pad = (rows + 63) // 64 * 64 if weight_side else (rows + 3) // 4 * 4
    want = torch.zeros(15, pad, dtype=torch.float32, device=_dev())
    want[:, :rows] = sc.float().t()
    assert got.dtype == torch.float32 and torch.equal(got, want)


def test_converter_rejects_bad_arguments():
    from fpqvar_amd import gemm
    with pytest.raises(RuntimeError):
        gemm.to_kmajor(torch.zeros(4, 100, dtype=torch.uint8, device=_dev()), 4)
    with pytest.raises(RuntimeError):
        gemm.to_kmajor(torch.zeros(4, 128, dtype=torch.uint8, device=_dev()), 5)
    assert gemm.to_kmajor(torch.zeros(0, 128, dtype=torch.uint8, device=_dev()), 4).shape == (2, 0, 64)


# ---- producers -------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("rows,k", [(1, 128), (37, 1920), (300, 1024), (5000, 1920), (65, 2304)])
def test_quantize_mx_km(rows, k):
    from fpqvar_amd import gemm
    torch.manual_seed(rows + k)
    x = (torch.randn(rows, k, device=_dev()) * 3).half()
    codes, scales = gemm.quantize_mx(x)
    image, scales_km = gemm.quantize_mx(x, kmajor=True)
    assert image.shape == (k // 128, rows, 64)
    assert torch.equal(gemm.to_kmajor_scales(scales), scales_km)
    assert torch.equal(image, gemm.to_kmajor(codes, 4))


@pytest.mark.parametrize("rows,k,dtype", [(1, 128, torch.float16), (37, 1920, torch.float16), (4000, 1920, torch.float16), (65, 2304, torch.float16),
                                          (3, 8192, torch.float16), (50, 1920, torch.float32), (9, 16384, torch.float16)])
def test_quantize_fp6_km(rows, k, dtype):
    from fpqvar_amd import gemm
    torch.manual_seed(rows + k)
    x = (torch.randn(rows, k, device=_dev()) * 3).to(dtype)
    codes, scales = gemm.quantize_fp6(x)
    image, scales_km = gemm.quantize_fp6(x, kmajor=True)
    assert image.shape == (k // 128, rows, 96)
    assert torch.equal(scales, scales_km)
    assert torch.equal(image, gemm.to_kmajor(codes, 6))


@pytest.mark.parametrize("rows,c,dtype,smooth", [(1, 128, torch.float16, False), (300, 1920, torch.float16, False), (4099, 1920, torch.float16, True),
                                                 (77, 1024, torch.float32, False), (513, 2304, torch.float32, True)])
def test_rotate_quant_mx_km(rows, c, dtype, smooth):
    from fpqvar_amd import gemm, rotation
    torch.manual_seed(rows + c)
    x = (torch.randn(rows, c, device=_dev()) * 2).to(dtype)
    sm = (torch.rand(c, device=_dev()) + 0.5) if smooth else None
    codes, scales = rotation.rotate_quant_mx(x, smooth=sm)
    image, scales_km = rotation.rotate_quant_mx(x, smooth=sm, kmajor=True)
    assert torch.equal(gemm.to_kmajor_scales(scales), scales_km)
    assert torch.equal(image, gemm.to_kmajor(codes, 4))


@pytest.mark.parametrize("b,l,c,dtype", [(2, 1, 1920, torch.float16), (3, 100, 1920, torch.float16), (2, 2116, 1920, torch.float32), (5, 9, 1024, torch.float16),
                                         (4, 64, 1024, torch.float16), (3, 25, 2304, torch.float16), (2, 49, 2560, torch.float32), (3, 16, 128, torch.float16)])
def test_adaln_rotate_quant_mx_km(b, l, c, dtype):
    from fpqvar_amd import gemm, rotation
    torch.manual_seed(b * l + c)
    x = torch.randn(b, l, c, device=_dev()).to(dtype)
    scale = (torch.randn(b, 1, c, device=_dev()) * 0.2).half()
    shift = (torch.randn(b, 1, c, device=_dev()) * 0.2).half()
    sm = torch.rand(c, device=_dev()) + 0.5
    codes, scales = rotation.adaln_rotate_quant_mx(x, scale, shift, smooth=sm)
    image, scales_km = rotation.adaln_rotate_quant_mx(x, scale, shift, smooth=sm, kmajor=True)
    assert image.shape == (c // 128, b * l, 64)
    assert torch.equal(gemm.to_kmajor_scales(scales), scales_km)
    assert torch.equal(image, gemm.to_kmajor(codes, 4))


@pytest.mark.parametrize("b,l,c,dtype", [(2, 1, 1920, torch.float16), (3, 100, 1920, torch.float16), (2, 1024, 1920, torch.float32), (3, 25, 2304, torch.float16),
                                         (2, 49, 2560, torch.float32), (5, 9, 1024, torch.float16), (3, 16, 128, torch.float16)])
def test_adaln_rotate_quant_token_fp6_km(b, l, c, dtype):
    from fpqvar_amd import gemm, rotation
    torch.manual_seed(b * l + c)
    x = torch.randn(b, l, c, device=_dev()).to(dtype)
    scale = (torch.randn(b, 1, c, device=_dev()) * 0.2).half()
    shift = (torch.randn(b, 1, c, device=_dev()) * 0.2).half()
    codes, scales = rotation.adaln_rotate_quant_token(x, scale, shift, emit="fp6")
    image, scales_km = rotation.adaln_rotate_quant_token(x, scale, shift, emit="fp6", kmajor=True)
    assert image.shape == (c // 128, b * l, 96)
    assert torch.equal(scales, scales_km)
    assert torch.equal(image, gemm.to_kmajor(codes, 6))


def test_producers_refuse_what_they_cannot_write():
    from fpqvar_amd import rotation
    x = torch.randn(2, 8, 3072, device=_dev()).half()     # beyond one wavefront per row: the k-major form does not exist
    sc = torch.zeros(2, 1, 3072, device=_dev()).half()
    rotation.adaln_rotate_quant_mx(x, sc, sc)
    with pytest.raises(RuntimeError):
        rotation.adaln_rotate_quant_mx(x, sc, sc, kmajor=True)
    with pytest.raises(RuntimeError):
        rotation.adaln_rotate_quant_token(x[..., :1920].contiguous(), sc[..., :1920].contiguous(), sc[..., :1920].contiguous(), emit="fp8", kmajor=True)


# ---- GEMMs -----------------------------------------------------------------------------------------------------------
SHAPES = [(1, 128, 8), (33, 256, 128), (301, 1920, 392), (700, 1920, 1920), (4356, 1920, 5760), (2500, 1024, 3072), (513, 2304, 2304),
          (100, 8192, 256), (4100, 3840, 640)]   # (long K: 64 / 30 groups of scale planes, the bigger tiles' LDS images no longer fit twice)


@pytest.mark.parametrize("tokens,k,outs", SHAPES)
@pytest.mark.parametrize("cfg", [None, 10, 20, 30])
def test_linear_fp4_km_equals_row_major(tokens, k, outs, cfg, lib_options):
    from fpqvar_amd import gemm
    if cfg is not None:
        lib_options("FPQ_GEMM_CFG", cfg)
    torch.manual_seed(tokens + outs)
    x = torch.randn(tokens, k, device=_dev()).half()
    w = torch.randn(outs, k, device=_dev()) * 0.05
    bias = (torch.randn(outs, device=_dev()) * 0.1).half()
    (ac, asc), (wc, wsc) = gemm.quantize_mx(x), gemm.quantize_mx(w)
    ai, wi = gemm.to_kmajor(ac, 4), gemm.to_kmajor(wc, 4, dealt=True)
    asi, wsi = gemm.to_kmajor_scales(asc), gemm.to_kmajor_scales(wsc, weight_side=True)
    assert torch.equal(gemm.linear_fp4(ac, asc, wc, wsc, bias), gemm.linear_fp4(ai, asi, wi, wsi, bias))
    assert torch.equal(gemm.linear_fp4(ac, asc, wc, wsc), gemm.linear_fp4(ai, asi, wi, wsi, outs=outs))
    if tokens % 3 == 0 or tokens == 33:
        bsz = 3 if tokens % 3 == 0 else 1
        gate = torch.randn(bsz, 1, outs, device=_dev()).half()
        res = torch.randn(tokens, outs, device=_dev()).half()
        assert torch.equal(gemm.linear_fp4(ac, asc, wc, wsc, bias, gate, res), gemm.linear_fp4(ai, asi, wi, wsi, bias, gate, res))


@pytest.mark.parametrize("tokens,k,outs", [(1, 128, 128), (33, 256, 128), (301, 1920, 384), (4356, 1920, 7680), (700, 1920, 1920)])
@pytest.mark.parametrize("cfg", [None, 10, 20, 30])
def test_linear_fp4_gelu_dual_km_equals_row_major(tokens, k, outs, cfg, lib_options):
    from fpqvar_amd import gemm
    if cfg is not None:
        lib_options("FPQ_GEMM_CFG", cfg)
    torch.manual_seed(tokens + outs)
    x = torch.randn(tokens, k, device=_dev()).half()
    w = torch.randn(outs, k, device=_dev()) * 0.05
    bias = (torch.randn(outs, device=_dev()) * 0.1).half()
    (ac, asc), (wc, wsc) = gemm.quantize_mx(x), gemm.quantize_mx(w)
    ai, wi = gemm.to_kmajor(ac, 4), gemm.to_kmajor(wc, 4, dealt=True)
    asi, wsi = gemm.to_kmajor_scales(asc), gemm.to_kmajor_scales(wsc, weight_side=True)
    q0, h0 = gemm.linear_fp4_gelu_dual(ac, asc, wc, wsc, bias, return_gelu=True)
    q1, h1 = gemm.linear_fp4_gelu_dual(ai, asi, wi, wsi, bias, return_gelu=True)
    assert torch.equal(q0, q1) and torch.equal(h0, h1)


@pytest.mark.parametrize("tokens,k,outs", SHAPES)
@pytest.mark.parametrize("cfg", [None, 0, 1])
def test_linear_fp6_km_equals_row_major(tokens, k, outs, cfg, lib_options):
    from fpqvar_amd import gemm
    if cfg is not None:
        lib_options("FPQ_GEMM6_CFG", cfg)
    torch.manual_seed(tokens + outs)
    x = torch.randn(tokens, k, device=_dev()).half()
    w = torch.randn(outs, k, device=_dev()) * 0.05
    bias = (torch.randn(outs, device=_dev()) * 0.1).half()
    (ac, asc), (wc, wsc) = gemm.quantize_fp6(x), gemm.quantize_fp6(w)
    ai, wi = gemm.to_kmajor(ac, 6), gemm.to_kmajor(wc, 6, dealt=True)
    assert torch.equal(gemm.linear_fp6(ac, asc, wc, wsc, bias), gemm.linear_fp6(ai, asc, wi, wsc, bias))
    if tokens % 3 == 0:
        gate = torch.randn(3, 1, outs, device=_dev()).half()
        res = torch.randn(tokens, outs, device=_dev()).half()
        assert torch.equal(gemm.linear_fp6(ac, asc, wc, wsc, bias, gate, res), gemm.linear_fp6(ai, asc, wi, wsc, bias, gate, res))


def test_mixed_layouts_are_an_error():
    from fpqvar_amd import gemm
    x = torch.randn(64, 256, device=_dev()).half()
    w = torch.randn(128, 256, device=_dev())
    (ac, asc), (wc, wsc) = gemm.quantize_mx(x), gemm.quantize_mx(w)
    asi, wsi = gemm.to_kmajor_scales(asc), gemm.to_kmajor_scales(wsc, weight_side=True)
    with pytest.raises(RuntimeError):
        gemm.linear_fp4(gemm.to_kmajor(ac, 4), asi, wc, wsc)
    with pytest.raises(RuntimeError):
        gemm.linear_fp4(ac, asc, gemm.to_kmajor(wc, 4, dealt=True), wsi)
    with pytest.raises(RuntimeError):   # an activation-side (undealt, unpadded) image is not a weight image
        gemm.linear_fp4(gemm.to_kmajor(ac, 4), asi, gemm.to_kmajor(wc[:100], 4), gemm.to_kmajor_scales(wsc[:100]))
    with pytest.raises(RuntimeError):   # images need scale IMAGES (fp32 [K/128, rows]), not the row-major fp16 scales
        gemm.linear_fp4(gemm.to_kmajor(ac, 4), asc, gemm.to_kmajor(wc, 4, dealt=True), wsi)
    with pytest.raises(RuntimeError):   # a width the weight image cannot have
        gemm.linear_fp4(gemm.to_kmajor(ac, 4), asi, gemm.to_kmajor(wc, 4, dealt=True), wsi, outs=32)


@pytest.mark.parametrize("cls_name", ["FP4Linear", "FP4LinearGeluDual", "FP6Linear"])
def test_modules_hold_kmajor_weights(cls_name):
    from fpqvar_amd import gemm
    torch.manual_seed(3)
    lin = torch.nn.Linear(1920, 1920 if cls_name != "FP4LinearGeluDual" else 7680).to(_dev())
    cls = getattr(gemm, cls_name)
    plain, km = cls.from_float(lin), cls.from_float(lin, kmajor=True)
    assert not plain.kmajor and km.kmajor and km.w_codes.dim() == 3
    x = torch.randn(2, 150, 1920, device=_dev()).half()
    assert torch.equal(plain(x), km(x))
    km = km.half()                       # the driver's var.half(): scales stay fp32, the image stays uint8
    assert km.w_scales.dtype == plain.w_scales.dtype and km.w_codes.dtype == torch.uint8
    assert torch.equal(plain(x), km(x))


def test_kmajor_path_under_graph_replay():
    """producer -> k-major GEMM captured in a hipGraph and replayed on new data (the generation loop's form)"""
    from fpqvar_amd import gemm, rotation
    torch.manual_seed(5)
    lin = torch.nn.Linear(1920, 5760).to(_dev())
    mod = gemm.FP4Linear.from_float(lin, kmajor=True)
    x = torch.randn(2, 324, 1920, device=_dev()).half()
    sc = (torch.randn(2, 1, 1920, device=_dev()) * 0.1).half()

    def step():
        a, s = rotation.adaln_rotate_quant_mx(x, sc, sc, kmajor=True)
        return mod.forward_operands(a, s)

    step()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        y = step()
    x.copy_(torch.randn_like(x))
    g.replay()
    torch.cuda.synchronize()
    a, s = rotation.adaln_rotate_quant_mx(x, sc, sc)
    want = gemm.linear_fp4(a, s, *gemm.quantize_mx(lin.weight.detach().float()), lin.bias.detach().half())
    assert mod.w_scales.shape == (15, 5760) and mod.w_scales.dtype == torch.float32
    assert torch.equal(y, want)


def test_quantize_var_uses_kmajor_operands_and_survives_a_state_dict_round_trip():
    """quantize_VAR(real_fp4 / real_fp6): k-major weights by default, bit-identical outputs to kmajor_operands=False, and the
    3-D image buffers travel through state_dict / load_state_dict like any other buffer."""
    import copy
    from fpqvar_amd import gemm, quant_linear as ql

    class FFN(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.fc1, self.act, self.fc2 = torch.nn.Linear(256, 1024), torch.nn.GELU(approximate="tanh"), torch.nn.Linear(1024, 256)

        def forward(self, x):
            return self.fc2(self.act(self.fc1(x)))

    class Attn(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.mat_qkv, self.proj = torch.nn.Linear(256, 768), torch.nn.Linear(256, 256)

        def forward(self, x):
            return self.proj(self.mat_qkv(x)[..., :256])

    class Toy(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.attn, self.ffn = Attn(), FFN()

        def forward(self, x):
            return self.ffn(self.attn(x))

    cfg4 = dict(weight_quant="per_group", act_quant="per_group", w_bit=4, a_bit=4, act_quant_sym=True, activation_fp_quant=True,
                weight_fp_quant=True, act_fp_type="fp_e2", weight_fp_type="fp_e2", fc2_fp_type="fp_e1m2_neg_e2m1_pos", real_fp4=True)
    cfg6 = dict(weight_quant="per_channel", act_quant="per_token", w_bit=6, a_bit=6, act_quant_sym=True, activation_fp_quant=True,
                weight_fp_quant=True, act_fp_type="fp6_e2m3", weight_fp_type="fp6_e2m3", fc2_fp_type="fp6_int_neg_e2m3_pos", real_fp6=True)
    torch.manual_seed(11)
    x = torch.randn(3, 40, 256, device=_dev()).half()
    for cfg in (cfg4, dict(cfg4, fuse_ffn=True), cfg6):
        base = Toy().to(_dev())
        km = ql.quantize_VAR(copy.deepcopy(base), **cfg).half()
        rm = ql.quantize_VAR(copy.deepcopy(base), kmajor_operands=False, **cfg).half()
        n_images = sum(1 for m in km.modules() if isinstance(m, (gemm.FP4Linear, gemm.FP6Linear)) and m.kmajor)
        assert n_images == 3 and not any(getattr(m, "kmajor", False) for m in rm.modules())
        y = km(x)
        assert torch.equal(y, rm(x))
        again = ql.quantize_VAR(copy.deepcopy(base), **cfg).half()
        for m in again.modules():          # wipe the weights, then restore them from the first model's state_dict
            if isinstance(m, (gemm.FP4Linear, gemm.FP6Linear)):
                m.w_codes.zero_()
        again.load_state_dict(km.state_dict())
        assert torch.equal(again(x), y)


def test_random_shapes_agree_with_the_row_major_path():
    """forty seeded random (tokens, K, outs) triples: producer images == converted row-major outputs, k-major GEMMs == row-major
    GEMMs (FP4 with bias and an odd width, FP6), every tiling the dispatcher may pick"""
    import random
    from fpqvar_amd import gemm
    rnd = random.Random(2026)
    for case in range(40):
        tokens = rnd.choice([1, 2, 3, 5, 17, 63, 64, 65, 127, 129, 255, 257, 300, 511, 777, 1025, 2049])
        k = 128 * rnd.randint(1, 20)
        outs = 8 * rnd.randint(1, 120)
        torch.manual_seed(case)
        x = (torch.randn(tokens, k, device=_dev()) * rnd.choice([0.01, 1.0, 30.0])).half()
        w = torch.randn(outs, k, device=_dev()) * 0.05
        bias = (torch.randn(outs, device=_dev()) * 0.1).half() if rnd.random() < 0.7 else None
        # FP4
        (ac, asc), (wc, wsc) = gemm.quantize_mx(x), gemm.quantize_mx(w)
        ai, asi = gemm.quantize_mx(x, kmajor=True)
        assert torch.equal(ai, gemm.to_kmajor(ac, 4)) and torch.equal(asi, gemm.to_kmajor_scales(asc)), (case, tokens, k)
        wi, wsi = gemm.to_kmajor(wc, 4, dealt=True), gemm.to_kmajor_scales(wsc, weight_side=True)
        assert torch.equal(gemm.linear_fp4(ac, asc, wc, wsc, bias), gemm.linear_fp4(ai, asi, wi, wsi, bias, outs=outs)), (case, tokens, k, outs)
        # FP6
        (ac6, as6), (wc6, ws6) = gemm.quantize_fp6(x), gemm.quantize_fp6(w)
        ai6, as6k = gemm.quantize_fp6(x, kmajor=True)
        assert torch.equal(ai6, gemm.to_kmajor(ac6, 6)) and torch.equal(as6, as6k), (case, tokens, k)
        assert torch.equal(gemm.linear_fp6(ac6, as6, wc6, ws6, bias), gemm.linear_fp6(ai6, as6k, gemm.to_kmajor(wc6, 6, dealt=True), ws6, bias)), (case, tokens, k, outs)
