"""Randomized-Hadamard rotation pieces (reference: rotate_utils/hadamard_utils.py:63-99,
rotate_utils/rotation_utils.py:69-104) and the fused online rotate + quant op.

The reference builds a dense block-diagonal Q (15 or 18 identical 128x128 blocks
``diag(D) . H128 / sqrt(128)``, D = +-1 drawn with ``torch.manual_seed(42)``) and
multiplies activations by it with a dense fp16 GEMM on every forward
(tr/basic_var.py:263,266).  Here Q exists only for the offline weight side and for
tests; online, ``rotate_quant`` runs the 128-point butterfly inside the quant kernel.
"""
from __future__ import annotations

import ctypes
from typing import Optional, Tuple

import torch

from ._lib import TABLE_IDS, check, dtype_id, lib, require_gpu, stream_ptr, device_guard


def sign_vector(size: int = 128, seed: int = 42) -> torch.Tensor:
    """D of random_hadamard_matrix(size, device, seed): ``torch.manual_seed(seed);
    randint(0, 2, (size,)) * 2 - 1`` on the CPU generator (hadamard_utils.py:92-99),
    drawn from a private generator so the global RNG state is left alone."""
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    return torch.randint(low=0, high=2, size=(size,), generator=g).to(torch.float64) * 2 - 1


def sylvester(n: int) -> torch.Tensor:
    assert n & (n - 1) == 0, "Sylvester's construction gives power-of-two sizes only"
    h = torch.ones(1, 1, dtype=torch.float64)
    while h.shape[0] < n:
        h = torch.cat([torch.cat([h, h], 1), torch.cat([h, -h], 1)], 0)
    return h


# ---- Hadamard matrices of the non-power-of-two orders the full-width rotation needs ---------------------------
# The reference keeps literal 12 .. 172-row tables (rotate_utils/hadamard_utils.py:164-4204, taken from Sloane's
# library via quip-sharp).  Eight of them ARE Paley's constructions entry for entry (checked against the reference's
# tables; fixtures in tests/golden): orders 12, 20, 60, 108, 140 = Paley I over GF(11), GF(19), GF(59), GF(107),
# GF(139); orders 36 and 28 = Paley II over GF(17) and GF(13); order 40 = [[H20, H20], [H20, -H20]].  They are
# generated here from the quadratic-residue character instead of being stored.  VAR's widths need exactly these:
# 1280 = 40 * 32 (d20), 1536 = 12 * 128 (d24), 1920 = 60 * 32 (d30), 2304 = 36 * 64 (d36).  Orders 172, 156 and 52
# (other constructions) are not provided.
def _chi(q: int):
    """Quadratic-residue character of GF(q), q an odd prime: chi[x] = +1 (non-zero square), -1 (non-square), 0."""
    squares = {(x * x) % q for x in range(1, q)}
    return [0] + [1 if x in squares else -1 for x in range(1, q)]


def paley_hadamard_i(q: int) -> torch.Tensor:
    """Order q + 1, q = 3 (mod 4) prime, normalised as the reference's tables are: first column +1, first row
    (+1, -1, ..., -1), the q x q core is the right-circulant I - S with S[i][j] = chi(j - i)."""
    assert q % 4 == 3
    chi = torch.tensor(_chi(q), dtype=torch.float64)
    idx = (torch.arange(q)[None, :] - torch.arange(q)[:, None]) % q
    h = torch.ones(q + 1, q + 1, dtype=torch.float64)
    h[0, 1:] = -1.0
    h[1:, 1:] = torch.eye(q, dtype=torch.float64) - chi[idx]
    return h


def paley_hadamard_ii(q: int) -> torch.Tensor:
    """Order 2 (q + 1), q = 1 (mod 4) prime: with C the symmetric conference matrix [[0, 1], [1, chi(j - i)]],
    H = [[C + I, C - I], [C - I, -C - I]] (the block layout of the reference's tables)."""
    assert q % 4 == 1
    chi = torch.tensor(_chi(q), dtype=torch.float64)
    idx = (torch.arange(q)[None, :] - torch.arange(q)[:, None]) % q
    c = torch.ones(q + 1, q + 1, dtype=torch.float64)
    c[0, 0] = 0.0
    c[1:, 1:] = chi[idx]
    eye = torch.eye(q + 1, dtype=torch.float64)
    return torch.cat([torch.cat([c + eye, c - eye], 1), torch.cat([c - eye, -c - eye], 1)], 0)


def williamson_hadamard(n: int, minus) -> torch.Tensor:
    """Williamson's array  [[A, B, C, D], [-B, A, -D, C], [-C, D, A, -B], [-D, -C, B, A]]  of four symmetric circulant
    +-1 matrices of order n (row i of a block = its first row rotated right by i); `minus[t]` is the bit mask of the -1
    entries of block t's first row.  AA' + BB' + CC' + DD' = 4n I makes it a Hadamard matrix of order 4n.  The reference's
    literal had52 / had156 / had172 (hadamard_utils.py:666, 2057, 2998 - Sloane's had.52.will, had.156.will,
    had.172.will) are exactly this array for n = 13, 39, 43 with the first rows below."""
    idx = (torch.arange(n).view(1, n) - torch.arange(n).view(n, 1)) % n          # [i, j] -> (j - i) mod n
    blocks = []
    for m in minus:
        row = torch.tensor([-1.0 if (m >> i) & 1 else 1.0 for i in range(n)], dtype=torch.float64)
        assert torch.equal(row[1:], row[1:].flip(0)), "a Williamson block is symmetric"
        blocks.append(row[idx])
    a, b, c, d = blocks
    return torch.cat([torch.cat([a, b, c, d], 1), torch.cat([-b, a, -d, c], 1),
                      torch.cat([-c, d, a, -b], 1), torch.cat([-d, -c, b, a], 1)], 0)


_WILLIAMSON = {52: (13, (0x161a, 0x1c0e, 0xb34, 0x1ede)),
               156: (39, (0x1afb3cdf58, 0xecf5af370, 0x1975bdae98, 0x72be247d4e)),
               172: (43, (0x730a26450ce, 0x207ac935e04, 0x14d42f42b28, 0x385b3fcda1c))}

_PALEY = {172: lambda: williamson_hadamard(*_WILLIAMSON[172]), 156: lambda: williamson_hadamard(*_WILLIAMSON[156]),
          52: lambda: williamson_hadamard(*_WILLIAMSON[52]),
          140: lambda: paley_hadamard_i(139), 108: lambda: paley_hadamard_i(107), 60: lambda: paley_hadamard_i(59),
          36: lambda: paley_hadamard_ii(17), 28: lambda: paley_hadamard_ii(13),
          40: lambda: torch.kron(sylvester(2), paley_hadamard_i(19)),          # the reference's had40 is the doubled had20
          20: lambda: paley_hadamard_i(19), 12: lambda: paley_hadamard_i(11)}


def get_hadK(n: int, transpose: bool = False):
    """(hadK, K) with the reference's decision ladder (hadamard_utils.py:7-60): the first K of 172, 156, 140, 108,
    60, 52, 36, 28, 40, 20, 12 that divides n (n / K must then be a power of two), else (None, 1) for a power of two."""
    for k in (172, 156, 140, 108, 60, 52, 36, 28, 40, 20, 12):
        if n % k == 0:
            assert (n // k) & (n // k - 1) == 0
            h = _PALEY[k]()           # every order of the reference's ladder is generated (Paley I / II, Williamson)
            return (h.T.contiguous() if transpose else h), k
    assert n & (n - 1) == 0
    return None, 1


def hadamard_matrix(n: int) -> torch.Tensor:
    """The +-1 matrix matmul_hadU applies (hadamard_utils.py:63-85): its radix-2 passes run over the LOW index bits in
    natural order and the K x K table over the rest, i.e. hadK (x) Sylvester(n / K)."""
    had_k, k = get_hadK(n)
    low = sylvester(n // k)
    return low if had_k is None else torch.kron(had_k, low)


def random_hadamard_matrix(size: int, device, seed: int) -> torch.Tensor:
    """matmul_hadU(diag(D)) = diag(D) . H_size^T / float32(sqrt(size)) in float64 (hadamard_utils.py:63-99), any size
    get_hadK accepts (every entry is +-1 over the float32 square root: no rounding depends on summation order)."""
    d = sign_vector(size, seed)
    return ((d[:, None] * hadamard_matrix(size).T) / torch.tensor(size).sqrt()).to(device)


def block_random_hadamard_matrix(total_size: int = 1920, block_size: int = 128, device="cuda", seed: int = 42
                                 ) -> torch.Tensor:
    """Block-diagonal Q with IDENTICAL blocks (every block re-seeds with `seed`,
    rotation_utils.py:69-104)."""
    assert total_size % block_size == 0
    q = random_hadamard_matrix(block_size, device, seed)
    return torch.block_diag(*([q] * (total_size // block_size)))


def sign_mask(d: torch.Tensor) -> Tuple[int, int, int, int]:
    """128 signs -> 4 x uint32, bit j set <=> d[j] == -1."""
    assert d.numel() == 128
    bits = (d.reshape(-1) < 0).to(torch.int64).tolist()
    return tuple(sum(b << k for k, b in enumerate(bits[32 * w:32 * w + 32])) for w in range(4))


def rotate_weight(w: torch.Tensor, q: torch.Tensor) -> torch.Tensor:
    """Offline weight side: W <- W . Q in float64, back to W's dtype (rotation_utils.py:129-154)."""
    return (w.to(torch.float64) @ q.to(w.device, torch.float64)).to(w.dtype)


def transform_weight(w: torch.Tensor, s: torch.Tensor) -> torch.Tensor:
    """GALT smoothing, weight side: W <- W / s per input channel (transform_model_utils.py:8-28)."""
    return w / s.to(w.device)


# ---- model-level preprocessing, same names as the reference (rotate_utils/rotation_utils.py:129-154,211-243;
# learnable_transformation/transform_model_utils.py:8-28).  `layer` is one AdaLN block with .attn.mat_qkv / .ffn.fc1.
def rotate_mat_qkv(layer, Q) -> None:
    """W_q, W_k, W_v <- W . Q in float64 (row-wise, so rotating the stacked weight at once is the same arithmetic)."""
    w = layer.attn.mat_qkv.weight.data
    layer.attn.mat_qkv.weight.data = rotate_weight(w, Q)


def rotate_fc1(layer, Q) -> None:
    w = layer.ffn.fc1.weight.data
    layer.ffn.fc1.weight.data = rotate_weight(w, Q)


def get_orthogonal_matrix(size: int, mode: str = "hadamard", device=None, seed: int = 42) -> torch.Tensor:
    """rotation_utils.py:57-63 ('random' draws a QR factor from the global RNG there; only 'hadamard' is reproducible
    and used by the reference's drivers)."""
    if mode != "hadamard":
        raise ValueError(f"Unknown mode {mode}" if mode != "random" else "mode 'random' is not provided (unseeded QR)")
    return random_hadamard_matrix(size, device, seed)


def rotate_model(model, device, block_rotate: bool = True) -> None:
    """The reference's offline weight rotation for every block, mat_qkv and fc1 (rotation_utils.py:211-240):
    block_rotate=True - the run scripts' mode - with the block-diagonal Q (128-wide blocks, seed 42);
    block_rotate=False with the full-width randomized Hadamard matrix of size model.C (had60 x 2^5 for 1920, had36 x 2^6
    for 2304).  The ONLINE side of the full-width mode is a dense x . Q (no 128-block structure for the fused
    kernel to exploit), exactly as in the reference."""
    if block_rotate:
        q = block_random_hadamard_matrix(total_size=model.C, block_size=128, device=device, seed=42)
    else:
        q = get_orthogonal_matrix(model.C, mode="hadamard", device=device)
    for layer in model.blocks:
        rotate_mat_qkv(layer, q)
        rotate_fc1(layer, q)


def transform_mat_qkv(layer, mat_qkv_best_s) -> None:
    w = layer.attn.mat_qkv.weight.data
    layer.attn.mat_qkv.weight.data = transform_weight(w, mat_qkv_best_s).to(w.dtype)


def transform_fc1(layer, fc1_best_s) -> None:
    w = layer.ffn.fc1.weight.data
    layer.ffn.fc1.weight.data = transform_weight(w, fc1_best_s).to(w.dtype)


def transform_model(model, mat_qkv_best_s, fc1_best_s) -> None:
    """GALT smoothing, weight side: W <- W / s per block with the learned per-channel vectors."""
    for idx, layer in enumerate(model.blocks):
        transform_mat_qkv(layer, mat_qkv_best_s[idx])
        transform_fc1(layer, fc1_best_s[idx])


_DEFAULT_MASK = None
_DEFAULT_MASK_TUPLE = None

try:   # the compiled binding (csrc/quant_cuda_ext.cpp) for the hot, plain calls of the producers
    from . import _native
except ImportError:   # pragma: no cover - build() always produces it
    _native = None
if __import__("os").environ.get("FPQ_NO_NATIVE") == "1":   # the A/B tools time variant builds of the library through ctypes (_lib.use_variant)
    _native = None


def _native_ok(x, d, smooth, c) -> bool:
    """The compiled fast path takes the usual arguments only: default sign vector, contiguous x, float32 [C] smooth."""
    return _native is not None and d is None and x.is_cuda and x.is_contiguous() and (
        smooth is None or (smooth.dtype is torch.float32 and smooth.dim() == 1 and smooth.shape[0] == c and
                           smooth.device == x.device and smooth.is_contiguous()))


def _default_mask_tuple():
    global _DEFAULT_MASK_TUPLE
    if _DEFAULT_MASK_TUPLE is None:
        _DEFAULT_MASK_TUPLE = tuple(int(v) for v in sign_mask(sign_vector(128, 42)))
    return _DEFAULT_MASK_TUPLE


def _mask_arg(d: Optional[torch.Tensor]):
    """The 128 sign bits as the C ABI takes them; the default D (seed 42) is computed once - building it costs more
    host time than the kernel runs."""
    global _DEFAULT_MASK
    if d is None:
        if _DEFAULT_MASK is None:
            _DEFAULT_MASK = (ctypes.c_uint32 * 4)(*sign_mask(sign_vector(128, 42)))   # read-only to the library: shared
        return _DEFAULT_MASK
    return (ctypes.c_uint32 * 4)(*sign_mask(d.cpu()))


def rotate_quant(x: torch.Tensor, table: str = "e2m1", d: Optional[torch.Tensor] = None,
                 smooth: Optional[torch.Tensor] = None, return_rotated: bool = False):
    """out = fp_quant_*_per_group_cuda( half(x*smooth) @ half(Q_block) , 128 ) in one launch.

    x: [..., C] fp16 or fp32 on the GPU, C % 128 == 0.  Returns fp16 (and the rotated
    fp16 tensor when return_rotated)."""
    require_gpu(x, "rotate_quant")
    if x.dtype not in (torch.float16, torch.float32):
        raise RuntimeError(f"rotate_quant: x must be float16 or float32, got {x.dtype}")
    c = x.shape[-1]
    if c % 128 != 0:
        raise RuntimeError("rotate_quant: the last dimension must be a multiple of 128")
    if not return_rotated and _native_ok(x, d, smooth, c):
        return _native.rotate_quant(x, TABLE_IDS[table], _default_mask_tuple(), smooth)
    mask = _mask_arg(d)
    xc = x if x.is_contiguous() else x.contiguous()
    sm, sm_ptr = _smooth_ptr(smooth, c, x.device)
    out = torch.empty(x.shape, dtype=torch.float16, device=x.device)
    rot = torch.empty_like(out) if return_rotated else None
    with device_guard(x.device):
        check(lib().fpq_rotate_quant_rows(xc.data_ptr(), out.data_ptr(), rot.data_ptr() if rot is not None else None,
                                          x.numel() // c, c, dtype_id(x.dtype), sm_ptr, mask, TABLE_IDS[table],
                                          stream_ptr(x.device)), "fpq_rotate_quant_rows")
    return (out, rot) if return_rotated else out


def _mod_rows(t: torch.Tensor, bsz: int, c: int) -> torch.Tensor:
    """[B, 1, C] / [B, C] modulation tensor as B contiguous rows (no tensor op when it already is)."""
    if t.is_contiguous() and t.numel() == bsz * c:
        return t
    return t.reshape(bsz, c).contiguous()


def _smooth_ptr(smooth, c, device):
    if smooth is None:
        return None, None
    if smooth.dtype is torch.float32 and smooth.dim() == 1 and smooth.shape[0] == c and smooth.device == device \
            and smooth.is_contiguous():
        return smooth, smooth.data_ptr()          # the usual case: no tensor ops on the host path
    sm = smooth.detach().to(device=device, dtype=torch.float32).reshape(-1).contiguous()
    if sm.numel() == 1:
        sm = sm.expand(c).contiguous()
    if sm.numel() != c:
        raise RuntimeError("smooth must have one entry per channel")
    return sm, sm.data_ptr()


def _kmajor_mx_scales(rows: int, c: int, device) -> torch.Tensor:
    from .gemm import kmajor_mx_scales
    return kmajor_mx_scales(rows, c, device)


def rotate_quant_mx(x: torch.Tensor, d: Optional[torch.Tensor] = None, smooth: Optional[torch.Tensor] = None,
                    kmajor: bool = False) -> Tuple[torch.Tensor, torch.Tensor]:
    """rotate_quant(x, "e2m1") emitting the FP4 GEMM's operands instead of values:
    (codes uint8 [rows, C/2], scales fp16 [rows, C/128]); level(code) * scale == rotate_quant(x) bit for bit.
    kmajor: the activation side's k-major images (include/fpq.h) - codes [C/128, rows, 64], scales fp32 [C/128, rows rounded up to 4]."""
    require_gpu(x, "rotate_quant_mx")
    if x.dtype not in (torch.float16, torch.float32) or x.shape[-1] % 128 != 0:
        raise RuntimeError("rotate_quant_mx: x must be float16/float32 with a last dimension that is a multiple of 128")
    c = x.shape[-1]
    rows = x.numel() // c
    if _native_ok(x, d, smooth, c):
        return _native.rotate_quant_mx(x, _default_mask_tuple(), smooth, kmajor)
    mask = _mask_arg(d)
    xc = x if x.is_contiguous() else x.contiguous()
    sm, sm_ptr = _smooth_ptr(smooth, c, x.device)
    codes = torch.empty((c // 128, rows, 64) if kmajor else (rows, c // 2), dtype=torch.uint8, device=x.device)
    scales = _kmajor_mx_scales(rows, c, x.device) if kmajor else torch.empty((rows, c // 128), dtype=torch.float16, device=x.device)
    fn = lib().fpq_rotate_quant_rows_codes_mx_km if kmajor else lib().fpq_rotate_quant_rows_codes_mx
    with device_guard(x.device):
        check(fn(xc.data_ptr(), codes.data_ptr(), scales.data_ptr(), rows, c, dtype_id(x.dtype), sm_ptr, mask, stream_ptr(x.device)),
              "fpq_rotate_quant_rows_codes_mx_km" if kmajor else "fpq_rotate_quant_rows_codes_mx")
    return codes, scales


def adaln_rotate_quant_mx(x: torch.Tensor, scale: torch.Tensor, shift: torch.Tensor, d: Optional[torch.Tensor] = None,
                          smooth: Optional[torch.Tensor] = None, eps: float = 1e-6, kmajor: bool = False
                          ) -> Tuple[torch.Tensor, torch.Tensor]:
    """adaln_rotate_quant(x, scale, shift, "e2m1") emitting (codes uint8 [B*L, C/2], scales fp16 [B*L, C/128]).
    kmajor (C <= 2560): the activation side's k-major images - codes [C/128, B*L, 64], scales fp32 [C/128, B*L rounded up to 4]."""
    require_gpu(x, "adaln_rotate_quant_mx")
    if x.dim() != 3:
        raise RuntimeError("adaln_rotate_quant_mx: x must be [B, L, C]")
    bsz, seq, c = x.shape
    if c % 128 != 0 or c > 4096:
        raise RuntimeError("adaln_rotate_quant_mx: C must be a multiple of 128 and at most 4096")
    if scale.dtype != shift.dtype or scale.dtype not in (torch.float16, torch.float32):
        raise RuntimeError("adaln_rotate_quant_mx: scale and shift must both be float16 or both float32")
    sc = _mod_rows(scale, bsz, c)
    sh = _mod_rows(shift, bsz, c)
    if sc.device == x.device and sh.device == x.device and _native_ok(x, d, smooth, c):
        return _native.adaln_rotate_quant_mx(x, sc, sh, _default_mask_tuple(), smooth, float(eps), kmajor)
    mask = _mask_arg(d)
    xc = x if x.is_contiguous() else x.contiguous()
    sm, sm_ptr = _smooth_ptr(smooth, c, x.device)
    codes = torch.empty((c // 128, bsz * seq, 64) if kmajor else (bsz * seq, c // 2), dtype=torch.uint8, device=x.device)
    scales = (_kmajor_mx_scales(bsz * seq, c, x.device) if kmajor
              else torch.empty((bsz * seq, c // 128), dtype=torch.float16, device=x.device))
    fn = lib().fpq_adaln_rotate_quant_rows_codes_mx_km if kmajor else lib().fpq_adaln_rotate_quant_rows_codes_mx
    with device_guard(x.device):
        check(fn(xc.data_ptr(), codes.data_ptr(), scales.data_ptr(), bsz * seq, c, dtype_id(x.dtype), sc.data_ptr(),
                 sh.data_ptr(), dtype_id(sc.dtype), seq, float(eps), sm_ptr, mask, stream_ptr(x.device)),
              "fpq_adaln_rotate_quant_rows_codes_mx_km" if kmajor else "fpq_adaln_rotate_quant_rows_codes_mx")
    return codes, scales


def adaln_rotate_quant_token(x: torch.Tensor, scale: torch.Tensor, shift: torch.Tensor, table: str = "e2m3",
                             d: Optional[torch.Tensor] = None, smooth: Optional[torch.Tensor] = None, eps: float = 1e-6,
                             emit: str = "values", kmajor: bool = False):
    """The fused producer for the per-token configurations (W6A6): LayerNorm, modulate, smooth, rotate, then
    fp6_quant_*_per_token_cuda with one scale per token row.  emit="values": fp16 [B, L, C];
    emit="fp8": (codes uint8 [B*L, C], scales fp16 [B*L]) for gemm.linear_fp8; emit="fp6" (table e2m3 only):
    (dense 6-bit codes uint8 [B*L, C * 3 / 4], scales) for gemm.linear_fp6.  C <= 2560.
    kmajor (emit="fp6" only): the codes as the activation side's k-major image [C/128, B*L, 96]."""
    require_gpu(x, "adaln_rotate_quant_token")
    if x.dim() != 3:
        raise RuntimeError("adaln_rotate_quant_token: x must be [B, L, C]")
    bsz, seq, c = x.shape
    if c % 128 != 0 or c > 2560:
        raise RuntimeError("adaln_rotate_quant_token: C must be a multiple of 128 and at most 2560")
    if scale.dtype != shift.dtype or scale.dtype not in (torch.float16, torch.float32):
        raise RuntimeError("adaln_rotate_quant_token: scale and shift must both be float16 or both float32")
    sc = _mod_rows(scale, bsz, c)
    sh = _mod_rows(shift, bsz, c)
    if emit not in ("values", "fp8", "fp6"):
        raise RuntimeError(f"adaln_rotate_quant_token: unknown emit {emit!r}")
    if emit == "fp6" and table != "e2m3":
        raise RuntimeError("adaln_rotate_quant_token: emit='fp6' is the E2M3 operand format")
    if kmajor and emit != "fp6":
        raise RuntimeError("adaln_rotate_quant_token: kmajor is a layout of the FP6 operand codes (emit='fp6')")
    if sc.device == x.device and sh.device == x.device and _native_ok(x, d, smooth, c):
        if emit == "values":
            return _native.adaln_rotate_quant_token(x, sc, sh, TABLE_IDS[table], _default_mask_tuple(), smooth, float(eps))
        return _native.adaln_rotate_quant_token_codes(x, sc, sh, TABLE_IDS[table], 8 if emit == "fp8" else 6, _default_mask_tuple(),
                                                      smooth, float(eps), kmajor)
    mask = _mask_arg(d)
    xc = x if x.is_contiguous() else x.contiguous()
    sm, sm_ptr = _smooth_ptr(smooth, c, x.device)
    rows = bsz * seq
    with device_guard(x.device):
        if emit in ("fp8", "fp6"):
            if emit == "fp6" and table != "e2m3":
                raise RuntimeError("adaln_rotate_quant_token: emit='fp6' is the E2M3 operand format")
            codes = torch.empty((c // 128, rows, 96) if kmajor else (rows, c if emit == "fp8" else c * 3 // 4), dtype=torch.uint8,
                                device=x.device)
            scales = torch.empty((rows,), dtype=torch.float16, device=x.device)
            fn = (lib().fpq_adaln_rotate_quant_token_rows_codes_fp8 if emit == "fp8"
                  else lib().fpq_adaln_rotate_quant_token_rows_codes_fp6_km if kmajor
                  else lib().fpq_adaln_rotate_quant_token_rows_codes_fp6)
            check(fn(xc.data_ptr(), codes.data_ptr(), scales.data_ptr(), rows, c, dtype_id(x.dtype), sc.data_ptr(), sh.data_ptr(),
                     dtype_id(sc.dtype), seq, float(eps), sm_ptr, mask, TABLE_IDS[table], stream_ptr(x.device)),
                  "fpq_adaln_rotate_quant_token_rows_codes_" + emit)
            return codes, scales
        if emit != "values":
            raise RuntimeError(f"adaln_rotate_quant_token: unknown emit {emit!r}")
        out = torch.empty(x.shape, dtype=torch.float16, device=x.device)
        check(lib().fpq_adaln_rotate_quant_token_rows(
            xc.data_ptr(), out.data_ptr(), None, None, None, rows, c, dtype_id(x.dtype), sc.data_ptr(), sh.data_ptr(),
            dtype_id(sc.dtype), seq, float(eps), sm_ptr, mask, TABLE_IDS[table], stream_ptr(x.device)),
            "fpq_adaln_rotate_quant_token_rows")
    return out


def adaln_rotate_quant(x: torch.Tensor, scale: torch.Tensor, shift: torch.Tensor, table: str = "e2m1",
                       d: Optional[torch.Tensor] = None, smooth: Optional[torch.Tensor] = None, eps: float = 1e-6,
                       return_intermediates: bool = False):
    """act_quant( matmul( LN(x).mul(scale.add(1)).add_(shift).mul(smooth), Q_block ) ) in one launch
    (tr/basic_var.py:263,266 + tr/quant_utils.py:765 under the driver's fp16 autocast).

    x: [B, L, C] fp16/fp32 on the GPU; scale, shift: [B, 1, C] (or [B, C]), both fp16 or both fp32.
    Returns fp16 [B, L, C] (plus h and the rotated tensor when return_intermediates)."""
    require_gpu(x, "adaln_rotate_quant")
    if x.dim() != 3:
        raise RuntimeError("adaln_rotate_quant: x must be [B, L, C]")
    bsz, seq, c = x.shape
    if c % 128 != 0 or c > 4096:
        raise RuntimeError("adaln_rotate_quant: C must be a multiple of 128 and at most 4096")
    if scale.dtype != shift.dtype or scale.dtype not in (torch.float16, torch.float32):
        raise RuntimeError("adaln_rotate_quant: scale and shift must both be float16 or both float32")
    sc = _mod_rows(scale, bsz, c)
    sh = _mod_rows(shift, bsz, c)
    if not return_intermediates and sc.device == x.device and sh.device == x.device and _native_ok(x, d, smooth, c):
        return _native.adaln_rotate_quant(x, sc, sh, TABLE_IDS[table], _default_mask_tuple(), smooth, float(eps))
    mask = _mask_arg(d)
    xc = x if x.is_contiguous() else x.contiguous()
    sm, sm_ptr = _smooth_ptr(smooth, c, x.device)
    out = torch.empty(x.shape, dtype=torch.float16, device=x.device)
    h = torch.empty_like(out) if return_intermediates else None
    y = torch.empty_like(out) if return_intermediates else None
    with device_guard(x.device):
        check(lib().fpq_adaln_rotate_quant_rows(
            xc.data_ptr(), out.data_ptr(), h.data_ptr() if h is not None else None,
            y.data_ptr() if y is not None else None, bsz * seq, c, dtype_id(x.dtype), sc.data_ptr(), sh.data_ptr(),
            dtype_id(sc.dtype), seq, float(eps), sm_ptr, mask, TABLE_IDS[table], stream_ptr(x.device)),
            "fpq_adaln_rotate_quant_rows")
    return (out, h, y) if return_intermediates else out
