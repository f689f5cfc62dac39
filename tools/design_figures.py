#!/usr/bin/env python3
"""DESIGN.md section 4's figures table, written from a bench.py line so that every number in it can be traced:
   python tools/design_figures.py profiles/r05_bench_n1.json          -> prints the block
   python tools/design_figures.py profiles/r05_bench_n1.json write    -> splices it between DESIGN.md's figures markers"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ROWS = [   # (label, bytes per element, other_kernels key)
    ("dual E1M2⁻/E2M1⁺ g=128 `[65536×7680]` (two launches; fc2's input, A5)", "4", "dual_fc2_e1m2neg_e2m1pos_fp16_65536x7680"),
    ("GELU(tanh) + the same quantizer in ONE pass over the fc1 output (round 5; two launches)", "4", "gelu_dual_fc2_one_pass_fp16_65536x7680"),
    ("dual INT⁻/E2M3⁺ g=128 `[65536×7680]` (A8)", "4", "dual_fc2_intneg_e2m3pos_per_group_fp16_65536x7680"),
    ("dual INT⁻/E2M3⁺ per token `[65536×7680]` (A8)", "4", "dual_fc2_intneg_e2m3pos_per_token_fp16_65536x7680"),
    ("E2M3 per token `[65536×1920]`, FP6 conversion hardware (A6)", "4", "fp6_e2m3_per_token_fp16_65536x1920"),
    ("rotate + quant, values `[65536×1920]` (F1)", "4", "fused_rotate_quant_e2m1_fp16_65536x1920"),
    ("rotate + quant, FP4 operands (F1 → F2)", "2.52", "rotate_quant_codes_mx_fp16_65536x1920"),
    ("adaLN producer, fp16 rows `[65500×1920]`, values (A12 / F1)", "4", "adaln_rotate_quant_e2m1_fp16_65500x1920"),
    ("adaLN producer, fp32 rows (the model's residual stream), values", "6", "adaln_rotate_quant_e2m1_fp32rows_65500x1920"),
    ("adaLN producer, fp32 rows, FP4 operands", "4.52", "adaln_rotate_quant_codes_mx_fp32rows_65500x1920"),
    ("adaLN producer, fp32 rows, per-token E4M3 bytes / dense 6-bit codes", "5 / 4.75", ("adaln_rotate_quant_token_codes_fp8_fp32rows_65500x1920", "adaln_rotate_quant_token_codes_fp6_fp32rows_65500x1920")),
    ("adaLN producer, fp16 rows, FP4 operands / E4M3 bytes / 6-bit codes (vector-issue bound; the reference never feeds fp16 rows here)", "2.52 / 3 / 2.75", ("adaln_rotate_quant_codes_mx_fp16_65500x1920", "adaln_rotate_quant_token_codes_fp8_fp16_65500x1920", "adaln_rotate_quant_token_codes_fp6_fp16_65500x1920")),
    ("config 5, C = 2304 `[44800×2304]`: adaLN fp32 rows values / operands; fp16 rows values", "6 / 4.52 / 4", ("config5_adaln_rotate_quant_e2m1_fp32rows_44800x2304", "config5_adaln_rotate_quant_codes_mx_fp32rows_44800x2304", "config5_adaln_rotate_quant_e2m1_fp16rows_44800x2304")),
    ("config 5: E2M1 g=128 `[44800×2304]`; dual `[44800×9216]`", "4", ("config5_act_quant_e2m1_g128_fp16_44800x2304", "config5_dual_fc2_e1m2neg_e2m1pos_fp16_44800x9216")),
    ("fp32 weights g=128 `[32768×1920]` → fp32 / → fp16; per channel E2M3 → fp16 (A3 / A6 on weights)", "8 / 6 / 6", ("weights_e2m1_per_group_fp32_to_fp32_32768x1920", "weights_e2m1_per_group_fp32_to_fp16_32768x1920", "weights_e2m3_per_channel_fp32_to_fp16_32768x1920")),
    ("config 1 per-tensor E2M1 `[4096×1024]` fp32 (two launches, launch-bound: 16 MB)", "12", "config1_fp_quant_e2_per_tensor_fp32_4096x1024"),
    ("config 2 d16 `mat_qkv` ×16 `[3072×1024]` fp32: 16 eager calls / ONE segment launch", "8", ("config2_d16_mat_qkv_16_calls_fp32_to_fp32", "config2_d16_mat_qkv_one_segment_launch_fp32_to_fp32")),
]


def cell(o, key):
    keys = key if isinstance(key, tuple) else (key,)
    parts = []
    for k in keys:
        v = o.get(k)
        parts.append("-" if not v or "error" in v else f"{v['ms'] * 1e3:.1f} µs, **{v['frac_of_8TBps']:.3f}**")
    return " / ".join(parts)


def block(d, src):
    o = d.get("other_kernels", {})
    rf = d["roofline"]
    out = [f"Figures of `{src}` (the driver-style `bench.py` line; `ms` = mean over every timed launch, fraction of 8 TB/s from it):", "",
           "| kernel at its BASELINE shape (SURVEY §8a row) | algorithmic B / element | µs per launch, fraction of 8 TB/s |", "|---|---|---|",
           f"| **headline** E2M1 g=128 fp16 `[65536×1920]` (A3; `rows16_lut_subwave_kernel`, levels from the FP4 conversion hardware) | 4 | "
           f"{rf['kernel_ms'] * 1e3:.1f} µs by HIP events, **{rf['frac']:.4f}**; value {d['value']:.0f} Gelem/s; PMC traffic "
           f"{(rf['traffic'] or 0) / 1e6:.2f} MB vs {rf['algorithmic_bytes'] / 1e6:.2f} MB algorithmic |"]
    for label, bpe, key in ROWS:
        out.append(f"| {label} | {bpe} | {cell(o, key)} |")
    wc = d.get("weight_calibration") or {}
    if "ms" in wc:
        out.append(f"| config 4: all 120 Linears of VAR-d30, fp32 → fp16, ONE launch ({wc['elements'] / 1e9:.3f} G elements) | 6 | "
                   f"{wc['ms']:.3f} ms, **{wc['frac_of_8TBps_per_gpu']:.3f}** |")
    for k, name in (("config4_format_search_d30_mat_qkv_fp6_2x2_100_samples", "FP6 2×2"), ("config4_format_search_d30_mat_qkv_fp4_3x3_100_samples", "FP4 3×3")):
        if k in o and "ms_per_layer" in o[k]:
            out.append(f"| config 4: format search of one d30 `mat_qkv` layer × 100 samples, batched, {name} (A14) | - | {o[k]['ms_per_layer']:.2f} ms per layer incl. its read-back |")
    fs = d.get("format_search_sharded") or {}
    if "ms" in fs:
        out.append(f"| config 4: the 30-layer search, sharded form at N = {fs['n_gpus']} | - | {fs['ms']:.1f} ms ({fs['ms_local']:.1f} local + gather), {fs['layers_per_s']:.0f} layers/s |")
    for key, name in (("config3_steps", "config 3 (d30)"), ("config5_steps", "config 5 (d36-512)")):
        c = d.get(key) or {}
        if "time_weighted_frac_of_8TBps" in c:
            out.append(f"| {name}: the four quantizer calls of a block over the ten scale steps, fp32 residual stream, cold (§4c) | 6 / 4 / 4 | "
                       f"{c['block_us_over_the_ten_steps']:.0f} µs per block, time-weighted **{c['time_weighted_frac_of_8TBps']:.3f}** "
                       f"(bound of the five-launch sequence {c.get('bound_frac')}, launch floor {c.get('launch_floor_us')} µs; "
                       f"adaLN / act / dual {c['by_kernel']['adaln']:.3f} / {c['by_kernel']['act']:.3f} / {c['by_kernel']['dual']:.3f}) |")
    gen = d.get("generation") or []
    by = {(g["model"], g["path"]): g for g in gen if "ms_per_batch" in g}
    for model in ("d30-256", "d36-512"):
        if all((model, p) in by for p in "RFQ"):
            r, f, q = (by[(model, p)] for p in "RFQ")
            out.append(f"| `generation` {model} W4A4, transformer part of one batch: reference op sequence / fused fake-quant / matrix cores | - | "
                       f"{r['ms_per_batch']:.0f} / {f['ms_per_batch']:.0f} / **{q['ms_per_batch']:.0f} ms** = {r['images_per_s']} / {f['images_per_s']} / "
                       f"{q['images_per_s']} images/s ({r['ms_per_batch'] / q['ms_per_batch']:.1f} × the reference's sequence) |")
    cb = d.get("cpu_baseline") or {}
    if cb:
        out.append(f"| CPU baseline (`kind: {cb['kind']}`: the reference's pure-torch path restated, {cb['cores']} threads, `[8192×1920]` sample) | - | {cb['value']} Gelem/s |")
    return "\n".join(out)


def main():
    src = sys.argv[1]
    with open(os.path.join(ROOT, src)) as f:
        d = json.load(f)
    text = block(d, src)
    if len(sys.argv) > 2 and sys.argv[2] == "write":
        p = os.path.join(ROOT, "DESIGN.md")
        s = open(p).read()
        a, b = s.index("<!-- figures:begin"), s.index("<!-- figures:end -->")
        a = s.index("\n", a) + 1
        open(p, "w").write(s[:a] + text + "\n" + s[b:])
        print("DESIGN.md: figures block rewritten from", src)
    else:
        print(text)


if __name__ == "__main__":
    main()
