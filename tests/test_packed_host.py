"""Host logic of the packed weight format (fpqvar_amd/packed.py): 6-bit packing and the safetensors
container.  No GPU: the PackedWeight objects are built by hand."""
import pytest
import torch

from fpqvar_amd import packed


def test_pack6_roundtrip_and_layout():
    g = torch.Generator().manual_seed(0)
    codes = torch.randint(0, 63, (37, 128), generator=g, dtype=torch.uint8)
    p = packed.pack6(codes)
    assert p.shape == (37, 96) and p.dtype == torch.uint8
    assert torch.equal(packed.unpack6(p), codes)
    # bit layout: code i occupies bits [6i, 6i+6) of the 24-bit little-endian word
    one = torch.tensor([[1, 2, 3, 62]], dtype=torch.uint8)
    word = 1 | (2 << 6) | (3 << 12) | (62 << 18)
    assert packed.pack6(one).tolist() == [[word & 255, (word >> 8) & 255, (word >> 16) & 255]]
    with pytest.raises(RuntimeError):
        packed.pack6(torch.zeros(2, 6, dtype=torch.uint8))
    with pytest.raises(RuntimeError):
        packed.unpack6(torch.zeros(2, 4, dtype=torch.uint8))


def test_container_roundtrip(tmp_path):
    g = torch.Generator().manual_seed(1)
    a = packed.PackedWeight(torch.randint(0, 255, (24, 64), generator=g, dtype=torch.uint8), torch.rand(24, generator=g),
                            "e2m1", 128, (8, 384), "float16", 128, 42, torch.rand(384, generator=g) + 0.5,
                            torch.randn(8, generator=g))
    b = packed.PackedWeight(torch.randint(0, 255, (8, 288), generator=g, dtype=torch.uint8),
                            torch.rand(8, generator=g).half(), "e2m3", 384, (8, 384))
    path = str(tmp_path / "w.safetensors")
    n = packed.save_packed(path, {"blocks.0.attn.mat_qkv": a, "blocks.0.ffn.fc2": b}, extra={"depth": 30})
    assert n == a.nbytes() + b.nbytes()
    got = packed.load_packed(path)
    assert list(got) == ["blocks.0.attn.mat_qkv", "blocks.0.ffn.fc2"]
    for name, want in (("blocks.0.attn.mat_qkv", a), ("blocks.0.ffn.fc2", b)):
        p = got[name]
        assert (p.table, p.cols, p.shape, p.out_dtype, p.rotate_block, p.rotate_seed) == \
               (want.table, want.cols, want.shape, want.out_dtype, want.rotate_block, want.rotate_seed)
        assert torch.equal(p.codes, want.codes) and torch.equal(p.scales, want.scales)
        assert (p.smooth is None) == (want.smooth is None) and (p.bias is None) == (want.bias is None)
    assert torch.equal(got["blocks.0.attn.mat_qkv"].smooth, a.smooth)
    assert got["blocks.0.attn.mat_qkv"].bits == 4 and got["blocks.0.ffn.fc2"].bits == 6


def test_container_rejects_foreign_files(tmp_path):
    from safetensors.torch import save_file
    path = str(tmp_path / "x.safetensors")
    save_file({"t": torch.zeros(2)}, path)
    with pytest.raises(RuntimeError, match="fpqvar-packed"):
        packed.load_packed(path)


def test_dequantize_needs_the_gpu_library():
    p = packed.PackedWeight(torch.zeros(2, 64, dtype=torch.uint8), torch.ones(2), "e2m1", 128, (2, 128))
    with pytest.raises(RuntimeError):
        p.dequantize()          # CPU tensors: the product path has no CPU fallback
    with pytest.raises(RuntimeError, match="E2M1"):
        packed.PackedWeight(torch.zeros(2, 96, dtype=torch.uint8), torch.ones(2), "e2m3", 128, (2, 128)).fp4_operands()


def test_container_rejects_truncated_or_inconsistent_layers(tmp_path):
    """load_packed checks the header against the tensor sizes: a file whose codes / scales do not cover the shape it
    claims is refused at load time, not handed to a kernel that would read past the allocation."""
    import json
    from safetensors.torch import save_file
    g = torch.Generator().manual_seed(2)
    good = {"table": "e2m1", "cols": 128, "shape": [8, 384], "out_dtype": "float16", "rotate_block": 0, "rotate_seed": 0}
    codes, scales = torch.randint(0, 255, (24, 64), generator=g, dtype=torch.uint8), torch.rand(24, generator=g)

    def write(name, header, tensors):
        path = str(tmp_path / name)
        save_file(tensors, path, metadata={"format": packed.FORMAT, "layers": json.dumps({"l": header})})
        return path

    assert list(packed.load_packed(write("ok.safetensors", good, {"l.codes": codes, "l.scales": scales}))) == ["l"]
    cases = {
        "short_codes": (good, {"l.codes": codes[:20], "l.scales": scales}),
        "short_scales": (good, {"l.codes": codes, "l.scales": scales[:5]}),
        "bigger_shape": (dict(good, shape=[16, 384]), {"l.codes": codes, "l.scales": scales}),
        "wrong_cols": (dict(good, cols=100), {"l.codes": codes, "l.scales": scales}),
        "codes_dtype": (good, {"l.codes": codes.to(torch.int16), "l.scales": scales}),
        "scales_dtype": (good, {"l.codes": codes, "l.scales": scales.double()}),
        "unknown_table": (dict(good, table="e5m2"), {"l.codes": codes, "l.scales": scales}),
        "no_scales": (good, {"l.codes": codes}),
        "six_bit_as_nibbles": (dict(good, table="e2m3"), {"l.codes": codes, "l.scales": scales}),
    }
    for name, (header, tensors) in cases.items():
        with pytest.raises(RuntimeError):
            packed.load_packed(write(name + ".safetensors", header, tensors))
    missing = dict(good)
    del missing["cols"]
    with pytest.raises(RuntimeError, match="cols"):
        packed.load_packed(write("missing.safetensors", missing, {"l.codes": codes, "l.scales": scales}))
