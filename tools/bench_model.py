#!/usr/bin/env python3
"""Model-shaped end to end (BASELINE.json config 3): the transformer part of one VAR-d30
256x256 generation batch - 10 scale steps (patch_nums 1..16, 680 tokens), 30 AdaLN blocks per step, B = 100 rows per
token (50 images x CFG), W4A4 per-group fp_e2 + fc2 dual format + block rotation + GALT smoothing + KV cache in FP6
(run.sh line 4) - with random weights (no checkpoints exist offline), every block sharing one set of weight tensors.
Word embedding, class conditioning, the VQVAE decoder and sampling are not part of the quantized path and are left out.

Three ways to run everything around the attention core:
  R  Level 0 of INTEGRATION.md: the reference's own op sequence (its ~11 torch ops per quantizer around
     quant_cuda.quant, dense fp16 GEMM with the block-diagonal Q, fp16 Linears on de-quantized tensors, the whole KV
     cache re-quantized at every step)
  F  Level 1 + 1b: one launch per quantizer, fused LayerNorm/modulate/smooth/rotate/quant producer, incremental KV
  Q  F with mat_qkv / proj / fc1 on the FP4 matrix cores (producers emit the GEMM operands directly, proj applies
     the block's gate and residual in its epilogue) and attention by fpq_attention_blhc straight off the cache views
Prints the time per batch for each and the speed-ups.
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.nn.functional as Fn  # noqa: E402

import quant_cuda  # noqa: E402  (the drop-in module)
from fpqvar_amd import gemm, kv_cache, ops, quant_utils as qu, rotation as rot  # noqa: E402

MODELS = {   # name: (depth, patch_nums, rows per token = images x CFG); SURVEY.md section 8 header, configs C3 / C5
    "d30-256": (30, (1, 2, 3, 4, 5, 6, 8, 10, 13, 16), 100),
    "d36-512": (36, (1, 2, 3, 4, 6, 9, 13, 18, 24, 32), 20),
}


def ref_sym(x, grid, group=None, out_dtype=None):
    """fp_quant_e2_per_group_cuda / fp6_quant_e2m3_per_token_cuda as the reference spells them."""
    shape = x.shape
    xs = x.reshape(-1, group) if group else x
    scale = xs.abs().max(dim=-1, keepdim=True)[0] / grid.abs().max()
    q, _ = quant_cuda.quant((xs / scale).view(-1).to(torch.float32), grid)
    return (q.view(xs.shape) * scale).view(shape).to(out_dtype or x.dtype)


def ref_dual(x, gneg, gpos, group=128):
    clip = 1.0 * x.abs().max()
    x = torch.clamp(x, -clip, clip)
    shape = x.shape
    xs = x.reshape(-1, group)
    zeros = torch.zeros_like(xs)
    xn_, xp_ = torch.where(xs <= 0, xs, zeros), torch.where(xs > 0, xs, zeros)
    sn = xn_.abs().max(dim=-1, keepdim=True)[0] / gneg.abs().max()
    sp = xp_.abs().max(dim=-1, keepdim=True)[0] / gpos.abs().max()
    qa, _ = quant_cuda.quant((xn_ / sn).view(-1).to(torch.float32), gneg)
    qb, _ = quant_cuda.quant((xp_ / sp).view(-1).to(torch.float32), gpos)
    return (qa.view(xs.shape) * sn + qb.view(xs.shape) * sp).view(shape).to(x.dtype)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="d30-256", choices=tuple(MODELS),
                    help="d30-256: VAR-d30 256x256 (C = 1920); d36-512: VAR-d36 512x512 (C = 2304, 2240 tokens)")
    ap.add_argument("--depth", type=int, default=None, help="run fewer blocks than the model has")
    ap.add_argument("--batch", type=int, default=None)
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--paths", default="F,Q,R")
    ap.add_argument("--config", default="w4a4", choices=("w4a4", "w6a6"),
                    help="w4a4: run.sh line 4 (per-group fp_e2, fc2 dual FP4); w6a6: run.sh line 10 (per-token / per-channel fp6_e2m3, fc2 dual FP6)")
    ap.add_argument("--no-graphs", action="store_true")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    heads, PATCH_NUMS, rows = MODELS[args.model]
    C, H, HID, B, depth = 64 * heads, heads, 4 * 64 * heads, args.batch or rows, args.depth or heads
    hd = C // H
    s_qkv, s_fc1 = torch.rand(C, device=dev) + 0.5, torch.rand(C, device=dev) + 0.5
    q64 = rot.block_random_hadamard_matrix(C, 128, dev, 42)
    q32 = q64.float()

    def lin_w(o, i, smooth=None, rotate=False):
        w = torch.randn(o, i, device=dev) * 0.02
        if smooth is not None:
            w = rot.transform_weight(w, smooth)
        return rot.rotate_weight(w, q64) if rotate else w

    w32 = {"qkv": lin_w(3 * C, C, s_qkv, True), "proj": lin_w(C, C), "fc1": lin_w(HID, C, s_fc1, True), "fc2": lin_w(C, HID)}
    W6 = args.config == "w6a6"
    if W6:
        wq = {n: qu.fp6_quant_e2m3_per_token_cuda(w, 6) for n, w in w32.items()}
        fp4 = {n: gemm.quantize_fp6(w32[n]) for n in ("qkv", "proj", "fc1")}        # operands of the row-scaled GEMMs
    else:
        wq = {n: qu.fp_quant_e2_per_group_cuda(w, 4, 128).half() for n, w in w32.items()}
        fp4 = {n: gemm.quantize_mx(w32[n]) for n in ("qkv", "proj", "fc1")}
    mods = [[(torch.randn(B, 1, C, device=dev) * 0.2).half() for _ in range(6)] for _ in range(depth)]
    e2m1 = qu.fp4_e2m1_grid.to(dev)
    e2m3 = qu.fp6_e2m3_grid.to(dev)
    gneg = torch.tensor([-1.75, -1.5, -1.25, -1.0, -0.75, -0.5, -0.25, 0.0], device=dev)
    gpos = torch.tensor([0.0, 0.5, 1.0, 1.5, 2.0, 3.0, 4.0, 6.0], device=dev)
    ineg, e2m3p = qu.int_neg_grid.to(dev), qu.e2m3_pos_grid.to(dev)

    # the reference's op sequences for the two configurations
    def r_act(t):      # activation quantizer of mat_qkv / proj / fc1
        return ref_sym(t, e2m3, None, torch.float16) if W6 else ref_sym(t, e2m1, 128)

    def r_fc2(t):      # fc2's dual-format input quantizer
        if not W6:
            return ref_dual(t, gneg, gpos)
        zeros = torch.zeros_like(t)
        xn_, xp_ = torch.where(t <= 0, t, zeros), torch.where(t > 0, t, zeros)
        sn = xn_.abs().max(dim=-1, keepdim=True)[0] / ineg.abs().max()
        sp = xp_.abs().max(dim=-1, keepdim=True)[0] / e2m3p.abs().max()
        qa, _ = quant_cuda.quant((xn_ / sn).view(-1).to(torch.float32), ineg)
        qb, _ = quant_cuda.quant((xp_ / sp).view(-1).to(torch.float32), e2m3p)
        return (qa.view(t.shape) * sn + qb.view(t.shape) * sp).to(t.dtype)

    # this library's single launches
    def f_act(t):
        return qu.fp6_quant_e2m3_per_token_cuda(t, 6) if W6 else qu.fp_quant_e2_per_group_cuda(t, 4, 128)

    def f_fc2(t):
        return qu.fp6_quant_int_neg_e2m3_pos_per_token_cuda(t, 6) if W6 else qu.fp_quant_e1m2_neg_e2m1_pos_per_group_cuda(t, 4, 128)

    def f_producer(t, sc, sh, sm):
        return rot.adaln_rotate_quant_token(t, sc, sh, "e2m3", smooth=sm) if W6 else rot.adaln_rotate_quant(t, sc, sh, "e2m1", smooth=sm)

    def q_producer_linear(t, sc, sh, sm, name):
        if W6:
            return gemm.linear_fp6(*rot.adaln_rotate_quant_token(t, sc, sh, "e2m3", smooth=sm, emit="fp6"), *fp4[name])
        return gemm.linear_fp4(*rot.adaln_rotate_quant_mx(t, sc, sh, smooth=sm), *fp4[name])

    def q_proj(t2d, gate, resid):       # x + proj(a).mul(gamma1), gate and residual applied in the GEMM epilogue
        if W6:
            return gemm.linear_fp6(*gemm.quantize_fp6(t2d), *fp4["proj"], None, gate, resid)
        return gemm.linear_fp4(*gemm.quantize_mx(t2d), *fp4["proj"], None, gate, resid)
    max_len = sum(p * p for p in PATCH_NUMS)

    def attend(q, kc, vc):                      # q [B,L,H,c]; kc, vc [B,Ltot,H,c] (flash layout, as the KV runs use)
        o = Fn.scaled_dot_product_attention(q.transpose(1, 2), kc.transpose(1, 2), vc.transpose(1, 2))
        return o.transpose(1, 2).reshape(q.shape[0], q.shape[1], C)

    def new_caches(path):
        if path == "R":
            return [None] * depth
        return [kv_cache.IncrementalKVCache(B, max_len, H, hd, 6, device=dev) for _ in range(depth)]

    def run(path):
        caches = new_caches(path)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for pn in PATCH_NUMS:
            x = torch.randn(B, pn * pn, C, device=dev).half()
            step(path, caches, x)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) * 1e3

    def run_graphed(path):
        """One hipGraph per scale step (static shapes), captured in step order so that the KV-cache bookkeeping on
        the host advances exactly as in an eager run; a batch is then 10 graph launches."""
        caches = new_caches(path)
        pool = torch.cuda.graph_pool_handle()
        graphs, inputs = [], []
        for pn in PATCH_NUMS:
            x = torch.randn(B, pn * pn, C, device=dev).half()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, pool=pool):
                step(path, caches, x)
            graphs.append(g)
            inputs.append(x)
        best = float("inf")
        for _ in range(args.reps + 1):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for g in graphs:
                g.replay()
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) * 1e3)
        del graphs
        return best

    def step(path, caches, x):
        if True:
            L = x.shape[1]
            for b in range(depth):
                g1, g2, sc1, sc2, sh1, sh2 = mods[b]
                if path == "R":
                    with torch.autocast("cuda", dtype=torch.float16):
                        x1 = torch.matmul(Fn.layer_norm(x, (C,), eps=1e-6).mul(sc1.add(1)).add_(sh1).mul(s_qkv), q32)
                        qkv = Fn.linear(r_act(x1), wq["qkv"]).view(B, L, 3, H, hd)
                        q, k, v = qkv.unbind(2)
                        if caches[b] is None:
                            kc, vc = k, v
                        else:                                        # tr/basic_var.py:186-209: whole cache, every step
                            ck, cv = caches[b]
                            ck = ref_sym(ck.contiguous(), e2m3, None, torch.float16)
                            cv = ref_sym(cv.contiguous(), e2m3, None, torch.float16)
                            kc, vc = torch.cat((ck, k), dim=1), torch.cat((cv, v), dim=1)
                        caches[b] = (kc, vc)
                        a = Fn.linear(r_act(attend(q, kc, vc)), wq["proj"])
                        x = x + a.mul(g1)
                        x2 = torch.matmul(Fn.layer_norm(x, (C,), eps=1e-6).mul(sc2.add(1)).add_(sh2).mul(s_fc1), q32)
                        h = Fn.gelu(Fn.linear(r_act(x2), wq["fc1"]), approximate="tanh")
                        x = x + Fn.linear(r_fc2(h), wq["fc2"]).mul(g2)
                    continue
                if path == "F":
                    qkv = Fn.linear(f_producer(x, sc1, sh1, s_qkv), wq["qkv"])
                else:
                    qkv = q_producer_linear(x, sc1, sh1, s_qkv, "qkv")
                q, k, v = qkv.view(B, L, 3, H, hd).unbind(2)
                kc, vc = caches[b].append(k, v)
                a = attend(q, kc, vc) if path == "F" else ops.attention_blhc(q, kc, vc, hd ** -0.5).view(B, L, C)
                if path == "F":
                    x = ops.gate_residual(Fn.linear(f_act(a), wq["proj"]), g1, x)
                else:
                    x = q_proj(a.view(B * L, C), g1, x).view(B, L, C)
                if path == "F":
                    h = Fn.linear(f_producer(x, sc2, sh2, s_fc1), wq["fc1"])
                else:
                    h = q_producer_linear(x, sc2, sh2, s_fc1, "fc1").view(B, L, HID)
                h = Fn.gelu(h, approximate="tanh")
                x = ops.gate_residual(Fn.linear(f_fc2(h), wq["fc2"]), g2, x)
        return x

    res = {"workload": f"VAR-{args.model} transformer part, {depth} blocks x {len(PATCH_NUMS)} steps ({max_len} tokens), B={B} (CFG), {args.config.upper()} + FP6 KV cache, random weights",
           "depth": depth, "batch_rows": B}
    paths = args.paths.split(",")
    for path in paths:
        run(path)                                   # warm-up (allocator, kernel load)
        res[f"{path}_ms_per_batch"] = round(min(run(path) for _ in range(args.reps)), 1)
        torch.cuda.empty_cache()
        if args.no_graphs:
            continue
        try:
            res[f"{path}_ms_per_batch_hipgraph"] = round(run_graphed(path), 1)
        except Exception as e:      # extra information only
            res[f"{path}_ms_per_batch_hipgraph"] = f"error: {str(e)[:120]}"
        torch.cuda.empty_cache()
    if "R" in paths:
        for pth in ("F", "Q"):
            if pth in paths:
                res[f"speedup_{pth}_vs_R"] = round(res["R_ms_per_batch"] / res[f"{pth}_ms_per_batch"], 2)
    for pth in paths:
        res[f"images_per_s_{pth}"] = round((B // 2) / (res[f"{pth}_ms_per_batch"] / 1e3), 1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
