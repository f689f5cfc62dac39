"""CPU-only: the rotation pieces of fpqvar_amd.rotation against the vectors produced by the
reference's hadamard_utils / rotation_utils (tests/golden/make_golden.py)."""
import torch

from fpqvar_amd import rotation as rot
from oracle import fpq_oracle as orc


def test_q128_equals_reference(golden):
    q_ref = torch.from_numpy(golden["rot/q128_f64"])
    assert torch.equal(rot.random_hadamard_matrix(128, "cpu", 42), q_ref)
    qb = rot.block_random_hadamard_matrix(1920, 128, "cpu", 42)
    assert qb.shape == (1920, 1920)
    for i in (0, 7, 14):
        assert torch.equal(qb[i * 128:(i + 1) * 128, i * 128:(i + 1) * 128], q_ref)
    assert float(qb[:128, 128:].abs().max()) == 0.0
    assert bool(golden["rot/blocks_identical_1920"])


def test_sign_mask_roundtrip():
    d = rot.sign_vector(128, 42)
    assert torch.equal(d, orc.sign_vector(128, 42))
    before = torch.get_rng_state()
    rot.sign_vector(128, 42)
    assert torch.equal(before, torch.get_rng_state())        # global RNG untouched
    m = rot.sign_mask(d)
    back = torch.tensor([-1.0 if (m[j // 32] >> (j % 32)) & 1 else 1.0 for j in range(128)], dtype=torch.float64)
    assert torch.equal(back, d)
    assert d[:16].tolist() == [-1, 1, -1, -1, -1, 1, -1, -1, -1, 1, -1, -1, -1, -1, 1, -1]


def test_weight_side_helpers():
    g = torch.Generator().manual_seed(0)
    w = torch.randn(64, 256, generator=g)
    q = rot.block_random_hadamard_matrix(256, 128, "cpu", 42)
    wr = rot.rotate_weight(w, q)
    assert wr.dtype == w.dtype
    # Q is orthogonal up to the float32 sqrt: (W Q) Q^T ~ W
    torch.testing.assert_close(rot.rotate_weight(wr, q.t()), w, rtol=1e-5, atol=1e-5)
    s = torch.rand(256, generator=g) + 0.5
    assert torch.equal(rot.transform_weight(w, s), w / s)


def test_model_level_preprocessing_matches_the_reference(golden):
    """transform_model then rotate_model (block mode) on a toy model: weights bit-equal to what the reference's
    learnable_transformation/transform_model_utils.py and rotate_utils/rotation_utils.py produce (tests/golden, section 6)."""
    import torch
    from fpqvar_amd import rotation as rot
    from tests.conftest import assert_bits_equal, from_bits

    class Blk(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.attn, self.ffn = torch.nn.Module(), torch.nn.Module()
            self.attn.mat_qkv = torch.nn.Linear(128, 384, bias=False)
            self.ffn.fc1 = torch.nn.Linear(128, 64)

    class Toy(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.C = 128
            self.blocks = torch.nn.ModuleList([Blk() for _ in range(2)])

    toy = Toy()
    for i in range(2):
        toy.blocks[i].attn.mat_qkv.weight.data = from_bits(golden[f"prep/w0/qkv{i}"])
        toy.blocks[i].ffn.fc1.weight.data = from_bits(golden[f"prep/w0/fc1{i}"])
    s_qkv = [from_bits(golden[f"prep/s/qkv{i}"]) for i in range(2)]
    s_fc1 = [from_bits(golden[f"prep/s/fc1{i}"]) for i in range(2)]
    rot.transform_model(toy, s_qkv, s_fc1)
    rot.rotate_model(toy, "cpu", True)
    for i in range(2):
        assert_bits_equal(toy.blocks[i].attn.mat_qkv.weight.data, from_bits(golden[f"prep/w1/qkv{i}"]), f"mat_qkv {i}")
        assert_bits_equal(toy.blocks[i].ffn.fc1.weight.data, from_bits(golden[f"prep/w1/fc1{i}"]), f"fc1 {i}")


def test_full_width_hadamard_matches_the_reference(golden):
    """The non-block rotation (rotate_utils/rotation_utils.py:211-222 with hadamard_utils.get_hadK: had60 x 2^5 for
    1920, had36 x 2^6 for 2304, ...): the generated Paley and Williamson tables equal the reference's literal ones, Q for VAR's
    widths is bit-equal (SHA-256 of the float64 bytes + sampled rows), rotate_model(block_rotate=False) reproduces
    the reference's rotated weights."""
    import hashlib
    import pytest
    import torch
    from fpqvar_amd import rotation as rot
    from tests.conftest import assert_bits_equal, from_bits

    for k in (12, 20, 28, 36, 40, 60, 108, 140, 52, 156, 172):   # Paley I / II; 52, 156, 172: Williamson arrays
        had, kk = rot.get_hadK(k)
        assert kk == k and torch.equal(had, torch.from_numpy(golden[f"had/table/{k}"]).double()), k
        assert torch.equal(had @ had.T, k * torch.eye(k, dtype=torch.float64))
        assert torch.equal(rot.get_hadK(k, transpose=True)[0], had.T)
    assert rot.get_hadK(1024) == (None, 1)
    assert rot.get_hadK(1920)[1] == 60 and rot.get_hadK(2304)[1] == 36 and rot.get_hadK(1280)[1] == 40 and rot.get_hadK(1536)[1] == 12
    for n, k in ((2 * 52, 52), (4 * 172, 172), (8 * 156, 156), (11008, 172)):   # 11008 = 172 * 64
        assert rot.get_hadK(n)[1] == k
        h = rot.hadamard_matrix(n) if n < 2000 else None
        assert h is None or torch.equal(h @ h.T, n * torch.eye(n, dtype=torch.float64))
    for n in (1280, 1536, 1920, 2304):
        q = rot.random_hadamard_matrix(n, "cpu", 42)
        assert q.dtype == torch.float64 and q.shape == (n, n)
        digest = hashlib.sha256(q.numpy().tobytes()).digest()
        assert digest == golden[f"had/q_sha256/{n}"].tobytes(), n
        assert torch.equal(q[:3, :64], torch.from_numpy(golden[f"had/q_corner/{n}"]))
        assert torch.equal(q[-1], torch.from_numpy(golden[f"had/q_lastrow/{n}"]))
        err = (q @ q.T - torch.eye(n, dtype=torch.float64)).abs().max()
        assert float(err) < 1e-6          # orthogonal up to float32(sqrt(n)), as the reference's Q is
    assert torch.equal(rot.get_orthogonal_matrix(1920, "hadamard", "cpu"), rot.random_hadamard_matrix(1920, "cpu", 42))
    with pytest.raises(ValueError):
        rot.get_orthogonal_matrix(128, "random", "cpu")

    class Blk(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.attn, self.ffn = torch.nn.Module(), torch.nn.Module()
            self.attn.mat_qkv = torch.nn.Linear(240, 72, bias=False)
            self.ffn.fc1 = torch.nn.Linear(240, 40)

    class Toy(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.C = 240
            self.blocks = torch.nn.ModuleList([Blk() for _ in range(2)])

    toy = Toy()
    for i in range(2):
        toy.blocks[i].attn.mat_qkv.weight.data = from_bits(golden[f"prep_full/w0/qkv{i}"])
        toy.blocks[i].ffn.fc1.weight.data = from_bits(golden[f"prep_full/w0/fc1{i}"])
    rot.rotate_model(toy, "cpu", False)
    for i in range(2):       # W . Q is a float64 GEMM over 240 terms: its summation order is the BLAS's, compare to fp32 rounding
        for got, key in ((toy.blocks[i].attn.mat_qkv.weight.data, f"prep_full/w1/qkv{i}"), (toy.blocks[i].ffn.fc1.weight.data, f"prep_full/w1/fc1{i}")):
            want = from_bits(golden[key])
            assert got.dtype == want.dtype and got.shape == want.shape
            ulp = (got.view(torch.int32) - want.view(torch.int32)).abs().max()
            assert int(ulp) <= 1, (key, int(ulp))
