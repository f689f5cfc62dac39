"""The transformer part of one generation batch of the reference's W4A4 / W6A6 runs, at the shapes of BASELINE.json
configs 3 and 5: ten scale steps (tr/var.py:175), `depth` AdaLN blocks per step
(tr/basic_var.py:253-269), B = images x CFG rows per token, random weights (no checkpoints exist offline; every block
shares one set of weight tensors), KV cache in FP6 (run.sh:4).  Word embedding, class conditioning, sampling and the
VQVAE decoder are not part of the quantized path and are left out: this is the host logic that strings the path's
kernels together the way the model does, for the model-level figures of bench.py (`generation`) and
tools/bench_model.py - the VAR model itself stays the reference's vendored code.

Three ways to run everything around the attention core:
  R  the reference's own op sequence on this GPU (Level 0 of INTEGRATION.md): its ~11 torch ops per quantizer around
     quant_cuda.quant (tr/quant_utils.py:313-330,415-452,503-517), the dense fp16 GEMM with the block-diagonal Q
     (tr/basic_var.py:263,266), fp16 Linears on de-quantized tensors (tr/quant_utils.py:767), the whole KV cache
     re-quantized at every step (tr/basic_var.py:186-209)
  F  one launch per quantizer, the fused LayerNorm / modulate / smooth / rotate / quant producer, incremental KV cache, GELU
     fused in front of fc2's input quantizer, attention by fpq_attention_blhc off the cache views (the reference calls
     flash_attn_func there, tr/basic_var.py:211; rounds 1 - 4 timed this path with torch's SDPA: sdpa_in_f=True); the Linears
     stay fp16 GEMMs on fake-quantized values (the reference's numerics)
  Q  F with mat_qkv / proj / fc1 on the FP4 (W6A6: FP6) matrix cores - the producers emit the GEMM operands, proj applies
     the block's gate and residual in its epilogue, fc1 applies GELU and fc2's dual-format input quantizer in its epilogue
     (gemm.linear_fp4_gelu_dual; W4A4 only) - and attention by fpq_attention_blhc straight off the cache views
"""
from __future__ import annotations

import os
import time
from typing import Dict, List, Optional, Sequence

import torch
import torch.nn.functional as Fn

from . import gemm, kv_cache, ops, quant_utils as qu, rotation as rot

MODELS = {   # name: (heads = depth, patch_nums, rows per token = images x CFG); SURVEY.md section 8 header, configs C3 / C5
    "d30-256": (30, (1, 2, 3, 4, 5, 6, 8, 10, 13, 16), 100),
    "d36-512": (36, (1, 2, 3, 4, 6, 9, 13, 18, 24, 32), 20),
}
WHAT = {
    "d30-256": "VAR-d30 256x256, 50 images with CFG (evaluate_fp_quant_transform_rotate.py:187-199)",
    "d36-512": "VAR-d36 512x512, 10 images with CFG (evaluate_fp_quant_transform_rotate_512x512.py:54,62,192-214)",
}
PATHS = ("R", "F", "Q")
TUNED_GEMMS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tunableop_gfx950.csv")


class tuned_torch_gemms:
    """`with tuned_torch_gemms():` - torch's OWN GEMMs (fc2 on every path; every Linear and the rotation of paths F / R) pick
    their kernel from selections recorded once on an MI355X for the shapes of the two models (torch's TunableOp: the fastest of
    the hipBLASLt / rocBLAS solutions per shape; `fpqvar_amd/tunableop_gfx950.csv`, made by running tools/bench_model.py under
    PYTORCH_TUNABLEOP_TUNING=1).  Nothing is tuned at run time; a file whose validators (torch / hipBLASLt / rocBLAS versions,
    gfx950) do not match the box is ignored by torch, which then selects as it does by default."""

    def __init__(self, path: str = TUNED_GEMMS):
        self.path = path

    def __enter__(self):
        import torch.cuda.tunable as tn
        self.before = (tn.is_enabled(), tn.tuning_is_enabled())
        self.active = os.path.exists(self.path)
        if self.active:
            # torch works on a private copy: it may rewrite its results file when the process ends, and the recorded file is source
            import shutil
            import tempfile
            self.copy = os.path.join(tempfile.gettempdir(), f"fpq_tunableop_{os.getpid()}.csv")
            shutil.copyfile(self.path, self.copy)
            tn.enable(True)
            tn.tuning_enable(False)
            tn.set_filename(self.copy, False)
            self.active = bool(tn.read_file(self.copy))
            if not self.active:
                tn.enable(self.before[0])
        return self

    def __exit__(self, *exc):
        import torch.cuda.tunable as tn
        tn.enable(self.before[0])
        tn.tuning_enable(self.before[1])
        return False


def _ref_sym(x, grid, group=None, out_dtype=None):
    """fp_quant_e2_per_group_cuda / fp6_quant_e2m3_per_token_cuda as the reference spells them (tr/quant_utils.py:313-330,503-517)."""
    import quant_cuda
    shape = x.shape
    xs = x.reshape(-1, group) if group else x
    scale = xs.abs().max(dim=-1, keepdim=True)[0] / grid.abs().max()
    q, _ = quant_cuda.quant((xs / scale).view(-1).to(torch.float32), grid)
    return (q.view(xs.shape) * scale).view(shape).to(out_dtype or x.dtype)


def _ref_dual(x, gneg, gpos, group=128, clip=True):
    """fp_quant_e1m2_neg_e2m1_pos_per_group_cuda (tr/quant_utils.py:415-452; the FP6 twin :577-646 has no global clip)."""
    import quant_cuda
    if clip:
        c = 1.0 * x.abs().max()
        x = torch.clamp(x, -c, c)
    shape = x.shape
    xs = x.reshape(-1, group) if group else x
    zeros = torch.zeros_like(xs)
    xn_, xp_ = torch.where(xs <= 0, xs, zeros), torch.where(xs > 0, xs, zeros)
    sn = xn_.abs().max(dim=-1, keepdim=True)[0] / gneg.abs().max()
    sp = xp_.abs().max(dim=-1, keepdim=True)[0] / gpos.abs().max()
    qa, _ = quant_cuda.quant((xn_ / sn).view(-1).to(torch.float32), gneg)
    qb, _ = quant_cuda.quant((xp_ / sp).view(-1).to(torch.float32), gpos)
    return (qa.view(xs.shape) * sn + qb.view(xs.shape) * sp).view(shape).to(x.dtype)


class GenerationBatch:
    """Weights, modulation vectors and the step function of one model-shaped batch on `device`."""

    def __init__(self, model: str = "d30-256", config: str = "w4a4", depth: Optional[int] = None,
                 batch_rows: Optional[int] = None, device=None, seed: int = 0, fused_fc1: bool = True, sdpa_in_f: bool = False,
                 kmajor: bool = True, qkv_to_cache: bool = True):
        assert model in MODELS and config in ("w4a4", "w6a6")
        self.model, self.config = model, config
        heads, self.patch_nums, rows = MODELS[model]
        dev = self.dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.C, self.H = 64 * heads, heads
        self.HID, self.B, self.depth = 4 * self.C, batch_rows or rows, depth or heads
        self.hd = self.C // self.H
        self.max_len = sum(p * p for p in self.patch_nums)
        self.W6 = config == "w6a6"
        self.fused_fc1 = fused_fc1 and not self.W6 and hasattr(gemm, "linear_fp4_gelu_dual")
        self.sdpa_in_f = sdpa_in_f                              # path F with torch's SDPA instead of fpq_attention_blhc (rounds 1 - 4 timed it that way)
        self.fused_gelu_quant = fused_fc1                       # path F: GELU + fc2's input quantizer in one pass over the fc1 output
        self.kmajor = kmajor                                    # path Q: operands as k-major images (include/fpq.h): contiguous LDS-DMA pieces
        # path Q (W4A4): mat_qkv writes k and v straight into the KV cache's slots (fpq_gemm_fp4_mx_split): no copy-in pass
        self.qkv_to_cache = qkv_to_cache and not self.W6 and hasattr(gemm, "linear_fp4_qkv_to_cache")
        C, HID, B = self.C, self.HID, self.B
        g = torch.Generator(device=dev).manual_seed(seed)
        self.gen = g
        self.s_qkv = torch.rand(C, device=dev, generator=g) + 0.5
        self.s_fc1 = torch.rand(C, device=dev, generator=g) + 0.5
        q64 = rot.block_random_hadamard_matrix(C, 128, dev, 42)
        self.q32 = q64.float()

        def lin_w(o, i, smooth=None, rotate=False):
            w = torch.randn(o, i, device=dev, generator=g) * 0.02
            if smooth is not None:
                w = rot.transform_weight(w, smooth)
            return rot.rotate_weight(w, q64) if rotate else w

        w32 = {"qkv": lin_w(3 * C, C, self.s_qkv, True), "proj": lin_w(C, C), "fc1": lin_w(HID, C, self.s_fc1, True),
               "fc2": lin_w(C, HID)}
        if self.W6:
            self.wq = {n: qu.fp6_quant_e2m3_per_token_cuda(w, 6) for n, w in w32.items()}
            self.wop = {n: gemm.quantize_fp6(w32[n]) for n in ("qkv", "proj", "fc1")}   # operands of the row-scaled GEMMs
            if kmajor:
                self.wop = {n: (gemm.to_kmajor(c, 6, dealt=True), sc) for n, (c, sc) in self.wop.items()}
        else:
            self.wq = {n: qu.fp_quant_e2_per_group_cuda(w, 4, 128).half() for n, w in w32.items()}
            self.wop = {n: gemm.quantize_mx(w32[n]) for n in ("qkv", "proj", "fc1")}
            if kmajor:
                self.wop = {n: (gemm.to_kmajor(c, 4, dealt=True), gemm.to_kmajor_scales(sc, weight_side=True)) for n, (c, sc) in self.wop.items()}
        del w32
        self.mods = [[(torch.randn(B, 1, C, device=dev, generator=g) * 0.2).half() for _ in range(6)] for _ in range(self.depth)]
        self.e2m1 = qu.fp4_e2m1_grid.to(dev)
        self.e2m3 = qu.fp6_e2m3_grid.to(dev)
        self.gneg = torch.tensor([-1.75, -1.5, -1.25, -1.0, -0.75, -0.5, -0.25, 0.0], device=dev)
        self.gpos = torch.tensor([0.0, 0.5, 1.0, 1.5, 2.0, 3.0, 4.0, 6.0], device=dev)
        self.ineg, self.e2m3p = qu.int_neg_grid.to(dev), qu.e2m3_pos_grid.to(dev)

    # ---- the quantizers of the three paths -------------------------------------------------------------------------
    def r_act(self, t):       # activation quantizer of mat_qkv / proj / fc1, the reference's op sequence
        return _ref_sym(t, self.e2m3, None, torch.float16) if self.W6 else _ref_sym(t, self.e2m1, 128)

    def r_fc2(self, t):       # fc2's dual-format input quantizer, the reference's op sequence
        if self.W6:
            return _ref_dual(t, self.ineg, self.e2m3p, None, clip=False)
        return _ref_dual(t, self.gneg, self.gpos)

    def f_act(self, t):
        return qu.fp6_quant_e2m3_per_token_cuda(t, 6) if self.W6 else qu.fp_quant_e2_per_group_cuda(t, 4, 128)

    def f_fc2(self, t):
        return qu.fp6_quant_int_neg_e2m3_pos_per_token_cuda(t, 6) if self.W6 else qu.fp_quant_e1m2_neg_e2m1_pos_per_group_cuda(t, 4, 128)

    def act_then_fc2_quant(self, y):
        """`fc2.act_quant(act(y))` for an fc1 output y: one pass (GELU in front of the dual quantizer) or GELU, then the quantizer."""
        if not self.fused_gelu_quant:
            return self.f_fc2(Fn.gelu(y, approximate="tanh"))
        if self.W6:
            return qu.gelu_fp6_quant_int_neg_e2m3_pos_per_token_cuda(y, 6)
        return qu.gelu_fp_quant_e1m2_neg_e2m1_pos_per_group_cuda(y, 4, 128)

    def f_producer(self, t, sc, sh, sm):
        if self.W6:
            return rot.adaln_rotate_quant_token(t, sc, sh, "e2m3", smooth=sm)
        return rot.adaln_rotate_quant(t, sc, sh, "e2m1", smooth=sm)

    def q_producer_linear(self, t, sc, sh, sm, name):
        if self.W6:
            return gemm.linear_fp6(*rot.adaln_rotate_quant_token(t, sc, sh, "e2m3", smooth=sm, emit="fp6", kmajor=self.kmajor), *self.wop[name])
        return gemm.linear_fp4(*rot.adaln_rotate_quant_mx(t, sc, sh, smooth=sm, kmajor=self.kmajor), *self.wop[name])

    def q_proj(self, t2d, gate, resid):       # x + proj(a).mul(gamma1), gate and residual applied in the GEMM epilogue
        if self.W6:
            return gemm.linear_fp6(*gemm.quantize_fp6(t2d, kmajor=self.kmajor), *self.wop["proj"], None, gate, resid)
        return gemm.linear_fp4(*gemm.quantize_mx(t2d, kmajor=self.kmajor), *self.wop["proj"], None, gate, resid)

    def q_fc1_gelu_dual(self, t, sc, sh):
        """fc2's quantized input straight out of the fc1 GEMM: GELU(tanh) and the dual E1M2-/E2M1+ quantizer in its epilogue."""
        return gemm.linear_fp4_gelu_dual(*rot.adaln_rotate_quant_mx(t, sc, sh, smooth=self.s_fc1, kmajor=self.kmajor), *self.wop["fc1"])

    def attend(self, q, kc, vc):               # q [B,L,H,c]; kc, vc [B,Ltot,H,c] (flash layout, as the KV runs use)
        o = Fn.scaled_dot_product_attention(q.transpose(1, 2), kc.transpose(1, 2), vc.transpose(1, 2))
        return o.transpose(1, 2).reshape(q.shape[0], q.shape[1], self.C)

    def new_caches(self, path):
        if path == "R":
            return [None] * self.depth
        return [kv_cache.IncrementalKVCache(self.B, self.max_len, self.H, self.hd, 6, device=self.dev) for _ in range(self.depth)]

    def new_input(self, pn):
        return torch.randn(self.B, pn * pn, self.C, device=self.dev, generator=self.gen).half()

    # ---- one scale step: `depth` blocks over x [B, pn^2, C] ----------------------------------------------------------
    def step(self, path, caches, x):
        B, C, H, hd, HID = self.B, self.C, self.H, self.hd, self.HID
        L = x.shape[1]
        for b in range(self.depth):
            g1, g2, sc1, sc2, sh1, sh2 = self.mods[b]
            if path == "R":
                with torch.autocast("cuda", dtype=torch.float16):
                    x1 = torch.matmul(Fn.layer_norm(x, (C,), eps=1e-6).mul(sc1.add(1)).add_(sh1).mul(self.s_qkv), self.q32)
                    qkv = Fn.linear(self.r_act(x1), self.wq["qkv"]).view(B, L, 3, H, hd)
                    q, k, v = qkv.unbind(2)
                    if caches[b] is None:
                        kc, vc = k, v
                    else:                                        # tr/basic_var.py:186-209: whole cache, every step
                        ck, cv = caches[b]
                        ck = _ref_sym(ck.contiguous(), self.e2m3, None, torch.float16)
                        cv = _ref_sym(cv.contiguous(), self.e2m3, None, torch.float16)
                        kc, vc = torch.cat((ck, k), dim=1), torch.cat((cv, v), dim=1)
                    caches[b] = (kc, vc)
                    a = Fn.linear(self.r_act(self.attend(q, kc, vc)), self.wq["proj"])
                    x = x + a.mul(g1)
                    x2 = torch.matmul(Fn.layer_norm(x, (C,), eps=1e-6).mul(sc2.add(1)).add_(sh2).mul(self.s_fc1), self.q32)
                    h = Fn.gelu(Fn.linear(self.r_act(x2), self.wq["fc1"]), approximate="tanh")
                    x = x + Fn.linear(self.r_fc2(h), self.wq["fc2"]).mul(g2)
                continue
            if path == "Q" and self.qkv_to_cache:
                q = gemm.linear_fp4_qkv_to_cache(*rot.adaln_rotate_quant_mx(x, sc1, sh1, smooth=self.s_qkv, kmajor=self.kmajor), *self.wop["qkv"],
                                                 None, caches[b].kv, caches[b].len, L).view(B, L, H, hd)
                kc, vc = caches[b].commit_written(L)
            else:
                if path == "F":
                    qkv = Fn.linear(self.f_producer(x, sc1, sh1, self.s_qkv), self.wq["qkv"])
                else:
                    qkv = self.q_producer_linear(x, sc1, sh1, self.s_qkv, "qkv")
                q, k, v = qkv.view(B, L, 3, H, hd).unbind(2)
                kc, vc = caches[b].append(k, v)
            a = self.attend(q, kc, vc) if (path == "F" and self.sdpa_in_f) else ops.attention_blhc(q, kc, vc, hd ** -0.5).view(B, L, C)
            if path == "F":
                x = ops.gate_residual(Fn.linear(self.f_act(a), self.wq["proj"]), g1, x)
            else:
                x = self.q_proj(a.view(B * L, C), g1, x).view(B, L, C)
            if path == "F":
                hq = self.act_then_fc2_quant(Fn.linear(self.f_producer(x, sc2, sh2, self.s_fc1), self.wq["fc1"]))
            elif self.fused_fc1:
                hq = self.q_fc1_gelu_dual(x, sc2, sh2).view(B, L, HID)
            else:
                hq = self.act_then_fc2_quant(self.q_producer_linear(x, sc2, sh2, self.s_fc1, "fc1").view(B, L, HID))
            x = ops.gate_residual(Fn.linear(hq, self.wq["fc2"]), g2, x)
        return x

    # ---- timing ----------------------------------------------------------------------------------------------------
    def run_eager(self, path) -> float:
        """ms for the ten steps launched eagerly (host launch costs included)."""
        caches = self.new_caches(path)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for pn in self.patch_nums:
            self.step(path, caches, self.new_input(pn))
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) * 1e3

    def capture(self, path):
        """One hipGraph per scale step (static shapes), captured in step order so that the KV-cache bookkeeping on the
        host advances exactly as in an eager run; a batch is then ten graph launches."""
        caches = self.new_caches(path)
        pool = torch.cuda.graph_pool_handle()
        graphs, keep = [], [caches]
        for pn in self.patch_nums:
            x = self.new_input(pn)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, pool=pool):
                y = self.step(path, caches, x)
            graphs.append(g)
            keep.append((x, y))
        return graphs, keep

    @staticmethod
    def replay(graphs) -> float:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for g in graphs:
            g.replay()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) * 1e3

    def time_path(self, path, reps: int = 3, graphs: bool = True) -> Dict[str, float]:
        """One eager warm-up batch (allocator, kernel load), then - graphs: capture + best of `reps` replays; eager: best of
        `reps` eager batches."""
        out = {"warmup_eager_ms": round(self.run_eager(path), 1)}
        torch.cuda.empty_cache()
        if graphs:
            gr, keep = self.capture(path)
            self.replay(gr)                                                  # first replay: graph upload
            out["ms_per_batch"] = round(min(self.replay(gr) for _ in range(reps)), 2)
            out["clock"] = "ten hipGraph replays (one per scale step), host wall clock around them, best of %d" % reps
            del gr, keep
        else:
            out["ms_per_batch"] = round(min(self.run_eager(path) for _ in range(reps)), 2)
            out["clock"] = "eager launches, host wall clock, best of %d" % reps
        torch.cuda.empty_cache()
        out["images_per_s"] = round((self.B // 2) / (out["ms_per_batch"] / 1e3), 1)
        return out

    def describe(self) -> str:
        return (f"VAR-{self.model} transformer part, {self.depth} blocks x {len(self.patch_nums)} steps ({self.max_len} tokens), "
                f"B={self.B} rows per token (CFG), {self.config.upper()} + FP6 KV cache, random weights")


def generation_record(models: Sequence[str] = ("d30-256", "d36-512"), paths: Sequence[str] = PATHS, config: str = "w4a4",
                      reps: int = 3, device=None, seed: int = 0, depth: Optional[int] = None, tuned_gemms: bool = True) -> List[dict]:
    """bench.py's `generation` entries (this rank's replica): one record per (model, path).  tuned_gemms: torch's own GEMMs
    - the same ones on every path - run with the recorded TunableOp selections (tuned_torch_gemms), and the record says so."""
    out = []
    for model in models:
        gb = GenerationBatch(model, config, depth=depth, device=device, seed=seed)
        for path in paths:
            rec = {"model": model, "path": path, "config": config, "images_per_batch": gb.B // 2, "what": WHAT[model]}
            if path == "Q":
                rec["operands"] = "k-major images (include/fpq.h)" if gb.kmajor else "row-major codes"
                rec["kv_cache"] = ("k, v written into the cache's slots by mat_qkv's GEMM (fpq_gemm_fp4_mx_split), one quantization pass per step"
                                   if gb.qkv_to_cache else "one launch per step: quantization pass + copy-in of k, v")
            try:
                if tuned_gemms:
                    with tuned_torch_gemms() as tg:
                        rec.update(gb.time_path(path, reps))
                    rec["torch_gemms"] = ("TunableOp selections recorded for these shapes (fpqvar_amd/tunableop_gfx950.csv), no tuning at run time"
                                          if tg.active else "torch's default selection (the recorded TunableOp file did not validate on this box)")
                else:
                    rec.update(gb.time_path(path, reps))
                    rec["torch_gemms"] = "torch's default selection"
            except Exception as e:   # one failing path must not hide the others
                rec["error"] = repr(e)[:200]
                torch.cuda.empty_cache()
            out.append(rec)
        del gb
        torch.cuda.empty_cache()
    return out
