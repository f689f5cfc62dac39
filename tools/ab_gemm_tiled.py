#!/usr/bin/env python3
"""EXPERIMENT: the FP4 / FP6 GEMMs fed from PRE-TILED operand images (every LDS-DMA piece 1 KiB contiguous) against the shipped
row-major operands.  Needs tools/ab/libtiled.so = tools/build_variant.sh --gemm tiled -DFPQ_GEMM_TILED_OPERANDS=1, whose two
LDS-DMA kernels read the images this script builds with torch from the row-major codes; results must be bit-equal.
usage: ab_gemm_tiled.py tools/ab/libtiled.so"""
import os
import sys

os.environ["FPQ_NO_NATIVE"] = "1"   # the compiled binding is linked to the stock library
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from fpqvar_amd import _lib, gemm  # noqa: E402

stock = _lib.lib()
variant = _lib.use_variant(sys.argv[1])
dev = torch.device("cuda:0")
torch.manual_seed(0)


def use(l):
    _lib._lib = l


def pad_rows(codes, mult):
    r = codes.shape[0]
    p = (-r) % mult
    return codes if p == 0 else torch.cat([codes, codes.new_zeros(p, codes.shape[1])])


def deal(codes):
    """weight rows in the order the kernels deal them over a wavefront's four 16-row tiles: new row (grp * 4 + n) * 16 + q
    holds output 64 * grp + 4 * q + n"""
    r, b = codes.shape
    return codes.view(r // 64, 16, 4, b).permute(0, 2, 1, 3).reshape(r, b)


def tile_fp4(codes, dealt):
    """[R, G * 64] row-major nibble codes -> [R / 16][G][16 rows][4 physical chunks][16 bytes]; physical chunk pc of row q holds
    logical chunk pc ^ perm(q) (glds_chunk_perm, fpq_gemm_fp4.h)"""
    x = pad_rows(codes, 64 if dealt else 16)
    if dealt:
        x = deal(x)
    r, b = x.shape
    g = b // 64
    x = x.view(r // 16, 16, g, 4, 16)
    q = torch.arange(16, device=x.device)
    perm = (0x78 >> ((q >> 2) << 1)) & 3
    src = torch.arange(4, device=x.device)[None, :] ^ perm[:, None]          # [q, pc] -> logical chunk
    x = torch.take_along_dim(x, src[None, :, None, :, None].expand(r // 16, 16, g, 4, 16), dim=3)
    return x.permute(0, 2, 1, 3, 4).contiguous().view(-1)


def tile_fp6(codes, dealt):
    """[R, S * 96] row-major 6-bit codes -> [R / 32][S][32 rows][6 physical chunks][16 bytes]; physical chunk pc of row r holds
    logical chunk (pc - rot(r)) mod 6, rot(r) = (r >> 3) & 1 (fp6_rot, fpq_gemm_fp6.h)"""
    x = pad_rows(codes, 64 if dealt else 32)
    if dealt:
        x = deal(x)
    r, b = x.shape
    s = b // 96
    x = x.view(r // 32, 32, s, 6, 16)
    rr = torch.arange(32, device=x.device)
    rot = (rr >> 3) & 1
    src = (torch.arange(6, device=x.device)[None, :] - rot[:, None]) % 6
    x = torch.take_along_dim(x, src[None, :, None, :, None].expand(r // 32, 32, s, 6, 16), dim=3)
    return x.permute(0, 2, 1, 3, 4).contiguous().view(-1)


def burst(fn, n=20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def raw_fp6(l, a, sa, w, sw, bias, T, O, K):
    out = torch.empty(T, O, dtype=torch.float16, device=dev)
    rc = l.fpq_gemm_fp6_rows(a.data_ptr(), sa.data_ptr(), _lib.dtype_id(sa.dtype), w.data_ptr(), sw.data_ptr(), _lib.dtype_id(sw.dtype),
                             bias.data_ptr(), out.data_ptr(), T, O, K, _lib.stream_ptr(dev))
    assert rc == 0, rc
    return out


def raw_fp4(l, a, sa, w, sw, bias, T, O, K):
    out = torch.empty(T, O, dtype=torch.float16, device=dev)
    rc = l.fpq_gemm_fp4_mx(a.data_ptr(), sa.data_ptr(), w.data_ptr(), sw.data_ptr(), _lib.dtype_id(sw.dtype), bias.data_ptr(), out.data_ptr(),
                           T, O, K, _lib.stream_ptr(dev))
    assert rc == 0, rc
    return out


def raw_fc1(l, a, sa, w, sw, bias, T, O, K):
    out = torch.empty(T, O, dtype=torch.float16, device=dev)
    rc = l.fpq_gemm_fp4_gelu_dual(a.data_ptr(), sa.data_ptr(), w.data_ptr(), sw.data_ptr(), _lib.dtype_id(sw.dtype), bias.data_ptr(),
                                  out.data_ptr(), None, T, O, K, None, _lib.stream_ptr(dev))
    assert rc == 0, rc
    return out


print(f"# stock {os.path.basename(stock._name)} (row-major operands) against {os.path.basename(sys.argv[1])} (pre-tiled operand images); ms = best of 5 alternating bursts of 20")
for T, K, O in ((65536, 1920, 5760), (65536, 1920, 7680), (16900, 1920, 1920), (4356, 1920, 5760), (301, 1920, 392), (33, 1920, 128)):
    x = torch.randn(T, K, device=dev).half()
    w = torch.randn(O, K, device=dev) * 0.02
    b = (torch.randn(O, device=dev) * 0.1).half()
    a6, w6 = gemm.quantize_fp6(x), gemm.quantize_fp6(w)
    a4, w4 = gemm.quantize_mx(x), gemm.quantize_mx(w)
    jobs = [("fp6", raw_fp6, a6, w6, tile_fp6), ("fp4", raw_fp4, a4, w4, tile_fp4)]
    if O % 128 == 0:
        jobs.append(("fp4_fc1", raw_fc1, a4, w4, tile_fp4))
    for name, fn, (ac, asc), (wc, wsc), tiler in jobs:
        at, wt = tiler(ac, False), tiler(wc, True)
        run = {"stock": lambda: fn(stock, ac, asc, wc, wsc, b, T, O, K), "variant": lambda: fn(variant, at, asc, wt, wsc, b, T, O, K)}
        same = torch.equal(run["stock"](), run["variant"]())
        if T >= 4000:
            best = {"stock": 1e9, "variant": 1e9}
            for _ in range(5):
                for tag in best:
                    run[tag]()
                    best[tag] = min(best[tag], burst(run[tag]))
            print(f"{name:8s} [{T} x {K}] -> {O}: row-major {best['stock']:.4f} ms, tiled {best['variant']:.4f} ms ({best['stock'] / best['variant']:.3f} x), bit-equal {same}")
        else:
            print(f"{name:8s} [{T} x {K}] -> {O}: bit-equal {same}")
        assert same, name
