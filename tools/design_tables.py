#!/usr/bin/env python3
"""The round-4 tables of DESIGN.md, regenerated from the committed profiles (so that every figure in them can be traced):
   python tools/design_tables.py figures   -> the "kernel at its BASELINE shape" table (profiles/r04_bench_n1.json + r04_pmc_*.txt)
   python tools/design_tables.py steps     -> section 4c's per-step table and the time-weighted summary (profiles/r04_steps_*.json)
   python tools/design_tables.py write     -> both, spliced into DESIGN.md"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")


def line(path):
    return json.loads(open(os.path.join(P, path)).read().strip().split("\n")[-1])


def pmc(case):
    t = open(os.path.join(P, f"r04_pmc_{case}.txt")).read()
    us = float(re.search(r"kernel-trace average\) ([0-9.]+) us", t).group(1))
    fr = float(re.search(r"= ([0-9.]+) of 8 TB/s", t).group(1))
    valu = float(re.search(r"= ([0-9.]+) vector instructions per element", t).group(1))
    busy = float(re.search(r"cycles = ([0-9.]+) of them", t).group(1))
    return f"{us:.1f}, {fr:.3f}, {valu:.2f}, {100 * busy:.0f} %"


def figures():
    b, e = line("r04_bench_n1.json"), line("r04_bench_n1_commit_9065354.json")
    o, oe = b["other_kernels"], e["other_kernels"]

    def k(name, d=o):
        v = d[name]
        return f"{1e3 * v['ms']:.1f}, {v['frac_of_8TBps']:.3f}"

    def both(name):
        return f"{k(name)} ({k(name, oe)})" if name in oe else k(name)
    stats = open(os.path.join(P, "r04_bench_kernel_stats.csv")).read()
    m = re.search(r'"rows16_lut_subwave_kernel<16, false, 1[^"]*",(\d+),\d+,([0-9.]+)', stats)
    tr = json.load(open(os.path.join(P, "r04_pmc_traffic.json")))
    r, re_ = b["roofline"], e["roofline"]
    rows = [
        ("headline E2M1 g=128, fp16 `[65536×1920]`", "4",
         f"{1e3 * r['kernel_ms']:.1f} by HIP events over {b['steps']} steps, **{r['frac']:.4f}**; value {b['value']:.0f} Gelem/s "
         f"({1e3 * re_['kernel_ms']:.1f}, {re_['frac']:.4f}; {e['value']:.0f})",
         f"{float(m.group(2)) / 1e3:.2f} over {m.group(1)} launches = **{503316480 / (float(m.group(2)) * 1e-9) / 8e12:.3f}** (`r04_bench_kernel_stats.csv`); "
         f"{pmc('sym')} (`r04_pmc_sym.txt`); traffic {tr['traffic_bytes_per_launch'] / 1e6:.2f} MB = {tr['ratio']:.5f} × algorithmic (`r04_pmc_traffic.json`)"),
        ("dual E1M2⁻/E2M1⁺ g=128 `[65536×7680]` (two launches)", "4", both("dual_fc2_e1m2neg_e2m1pos_fp16_65536x7680"), "-"),
        ("dual INT⁻/E2M3⁺ g=128 / per token `[65536×7680]`", "4",
         f"{k('dual_fc2_intneg_e2m3pos_per_group_fp16_65536x7680')} / {k('dual_fc2_intneg_e2m3pos_per_token_fp16_65536x7680')} "
         f"({k('dual_fc2_intneg_e2m3pos_per_group_fp16_65536x7680', oe)} / {k('dual_fc2_intneg_e2m3pos_per_token_fp16_65536x7680', oe)})",
         f"per group: {pmc('dual6')} - the one value-emitting quantizer bound by vector issue (`r04_pmc_dual6.txt`)"),
        ("E2M3 per token `[65536×1920]`, levels from the FP6 conversion hardware (§4c)", "4", both("fp6_e2m3_per_token_fp16_65536x1920"),
         f"{pmc('token6')} (`r04_pmc_token6.txt`; table form: 84.5, 0.745, 10.57, 39 %, `r04_pmc_token6_table.txt`)"),
        ("rotate + quant, values / FP4 operands", "4 / 2.52",
         f"{k('fused_rotate_quant_e2m1_fp16_65536x1920')} / {k('rotate_quant_codes_mx_fp16_65536x1920')} "
         f"({k('fused_rotate_quant_e2m1_fp16_65536x1920', oe)} / {k('rotate_quant_codes_mx_fp16_65536x1920', oe)})",
         f"{pmc('rotate')} / {pmc('rotate_codes')}"),
        ("adaLN producer, fp16 rows `[65500×1920]`, values / FP4 operands", "4 / 2.52",
         f"{k('adaln_rotate_quant_e2m1_fp16_65500x1920')} / {k('adaln_rotate_quant_codes_mx_fp16_65500x1920')} "
         f"({k('adaln_rotate_quant_e2m1_fp16_65500x1920', oe)} / {k('adaln_rotate_quant_codes_mx_fp16_65500x1920', oe)})",
         f"{pmc('adaln')} / {pmc('adaln_codes')}"),
        ("adaLN producer, fp32 rows (the model's residual stream), values / FP4 operands / E4M3 per token", "6 / 4.52 / 5",
         " / ".join(k(n) for n in ("adaln_rotate_quant_e2m1_fp32rows_65500x1920", "adaln_rotate_quant_codes_mx_fp32rows_65500x1920",
                                   "adaln_rotate_quant_token_codes_fp8_fp32rows_65500x1920")) + " (" +
         " / ".join(k(n, oe) for n in ("adaln_rotate_quant_e2m1_fp32rows_65500x1920", "adaln_rotate_quant_codes_mx_fp32rows_65500x1920",
                                       "adaln_rotate_quant_token_codes_fp8_fp32rows_65500x1920")) + ")",
         f"values: {pmc('adaln32')}"),
        ("adaLN producer at C = 2304 `[44800×2304]` (config 5): fp32 rows values / operands; fp16 rows values", "6 / 4.52 / 4",
         " / ".join(k(n) for n in ("config5_adaln_rotate_quant_e2m1_fp32rows_44800x2304", "config5_adaln_rotate_quant_codes_mx_fp32rows_44800x2304",
                                   "config5_adaln_rotate_quant_e2m1_fp16rows_44800x2304")) + " (" +
         " / ".join(k(n, oe) for n in ("config5_adaln_rotate_quant_e2m1_fp32rows_44800x2304", "config5_adaln_rotate_quant_codes_mx_fp32rows_44800x2304",
                                       "config5_adaln_rotate_quant_e2m1_fp16rows_44800x2304")) + ")", "-"),
        ("config 5's other two calls: E2M1 g=128 `[44800×2304]`; dual `[44800×9216]`", "4",
         f"{k('config5_act_quant_e2m1_g128_fp16_44800x2304')}; {k('config5_dual_fc2_e1m2neg_e2m1pos_fp16_44800x9216')} "
         f"({k('config5_act_quant_e2m1_g128_fp16_44800x2304', oe)}; {k('config5_dual_fc2_e1m2neg_e2m1pos_fp16_44800x9216', oe)})", "-"),
    ]
    lines = ["| kernel at its BASELINE shape | B / element | bench r04: µs, fraction of 8 TB/s (the same line at commit 9065354 on another box) | rocprofv3 r04: avg µs, fraction, VALU per element, vector pipe busy |",
             "|---|---|---|---|"]
    for row in rows:
        lines.append("| " + " | ".join(row) + " |")
    wc, wce = b["weight_calibration"], e["weight_calibration"]
    m2 = re.search(r'groups32_lut_kernel[^,]*",(\d+),\d+,([0-9.]+)', stats)

    def g(name, d=o):
        return f"{d[name]['ms']:.4f} ms = {d[name]['TFLOPs']:.0f} TFLOP/s" if name in d else "-"
    c1, c2a, c2b = o["config1_fp_quant_e2_per_tensor_fp32_4096x1024"], o["config2_d16_mat_qkv_16_calls_fp32_to_fp32"], o["config2_d16_mat_qkv_one_segment_launch_fp32_to_fp32"]
    lines += [
        f"| fp32 weights g=128 → fp16, d30 all-Linear, ONE launch (config 4) | 6 | {wc['ms']:.3f} ms, **{wc['frac_of_8TBps_per_gpu']:.3f}** ({wce['ms']:.3f} ms, {wce['frac_of_8TBps_per_gpu']:.3f}) | "
        f"{float(m2.group(2)) / 1e6:.3f} ms over {m2.group(1)} launches (`r04_bench_kernel_stats.csv`); traffic {tr['calibration_kernel']['ratio']:.5f} × algorithmic |",
        "| fp32 weights → nibble codes + fp32 scales and back (the packed exchange, world 1) | 4.53 + 2.53 | 1.66 ms for both launches = 0.70 (r3: 2.20 ms; `r04_calib_codes_n1.txt`) | - |",
        f"| config 1 per-tensor E2M1 `[4096×1024]` fp32 (two launches) | 12 | {1e3 * c1['ms']:.1f}, {c1['frac_of_8TBps']:.3f} (launch-bound: 16 MB) | - |",
        f"| config 2 d16 `mat_qkv` ×16 `[3072×1024]` fp32 → fp32: 16 calls / one segment launch | 8 | {1e3 * c2a['ms']:.1f}, {c2a['frac_of_8TBps']:.3f} (eager, host-bound) / {1e3 * c2b['ms']:.1f}, {c2b['frac_of_8TBps']:.3f} | - |",
        f"| config 4 format search, one d30 `mat_qkv` layer × 100 samples, batched: FP6 2×2 / FP4 3×3 | - | {o['config4_format_search_d30_mat_qkv_fp6_2x2_100_samples']['ms_per_layer']:.3f} / "
        f"{o['config4_format_search_d30_mat_qkv_fp4_3x3_100_samples']['ms_per_layer']:.3f} ms per layer | - |",
        f"| FP4 GEMM (per-group scales) `[65536×1920]·[1920→5760]` | - | {g('gemm_fp4_w4a4_mat_qkv_65536x1920x5760')} ({g('gemm_fp4_w4a4_mat_qkv_65536x1920x5760', oe)}; start of the round: 0.7075 ms = 2049) | "
        "`r04_pmc_gemm_fp4.txt`: 10.48 vector instructions per MFMA, matrix pipe 26 % busy; §8 item 6 |",
        f"| FP6 / FP8 GEMM (per-token × per-channel scales), same shape | - | {g('gemm_fp6_w6a6_mat_qkv_65536x1920x5760')} / {g('gemm_fp8_rows_mat_qkv_65536x1920x5760')} "
        "(start of the round, same-process A/B: 0.786 / 0.925 ms) | §8 item 6 |",
    ]
    return lines


def steps():
    out = []
    names = {"adaln": "adaLN producer (× 2 per block)", "act": "E2M1 g=128 (proj input)", "dual": "dual E1M2⁻/E2M1⁺ (fc2 input, 2 launches)"}
    heads = {"d30": "VAR-d30, fp32 rows, cold (commit {c}; µs per call, `profiles/r04_steps_d30_fp32.json`; printed by `python tools/design_tables.py steps`):",
             "d36": "VAR-d36 512², fp32 rows, cold (`profiles/r04_steps_d36_fp32.json`):"}
    for model in ("d30", "d36"):
        d = json.load(open(os.path.join(P, f"r04_steps_{model}_fp32.json")))
        st = d["steps"]
        out += [heads[model].format(c=d["git_head"]), "",
                "| rows | " + " | ".join(str(s["rows"]) for s in st) + " | Σ | of 8 TB/s | fixed + slope |", "|" + "---|" * (len(st) + 4)]
        for key, label in names.items():
            bk = d["by_kernel"][key]
            f = bk["fit"]
            out.append(f"| {label} | " + " | ".join(f"{s[key]['us']:.1f}" for s in st) + f" | {bk['sum_us']:.1f} | {bk['frac_of_8TBps']:.3f} | "
                       f"{f['fixed_us_per_call']:.2f} µs + {f['ns_per_row']:.3f} ns/row ({f['frac_of_8TBps_of_the_slope']:.3f}) |")
        out += ["| block (2 + 1 + 1 calls) | " + " | ".join(f"{s['block_us']:.0f}" for s in st) +
                f" | **{d['block_us_over_the_ten_steps']:.1f}** | **{d['time_weighted_frac_of_8TBps']:.3f}** | |", ""]

    def tw(tag, m):
        return json.load(open(os.path.join(P, f"r04_steps_{m}_{tag}.json")))["time_weighted_frac_of_8TBps"]
    b = line("r04_bench_n1.json")
    out += ["| time-weighted fraction of 8 TB/s | d30 | d36-512 |", "|---|---|---|",
            f"| fp32 rows, cold (the bench line of the same commit and box: {b['config3_steps']['time_weighted_frac_of_8TBps']:.3f} / {b['config5_steps']['time_weighted_frac_of_8TBps']:.3f}) | "
            f"**{tw('fp32', 'd30'):.3f}** (six boxes this round: 0.618 - 0.624; r3 kernels, same tool: 0.611) | **{tw('fp32', 'd36'):.3f}** (0.569 - 0.575; r3 kernels: 0.560) |",
            f"| fp16 rows, cold | {tw('fp16', 'd30'):.3f} | {tw('fp16', 'd36'):.3f} |",
            f"| fp32 rows, resident | {tw('fp32_resident', 'd30'):.3f} | {tw('fp32_resident', 'd36'):.3f} |",
            "| round 3's figure (fp16 rows, resident, best of 5) | 0.63 → 0.641 with this tool's medians | - |", ""]
    return out


def write():
    """Splice both blocks into DESIGN.md (between their first line and the paragraph that follows them)."""
    path = os.path.join(ROOT, "profiles", "NOTES_r04.md")   # the round-4 notebook (DESIGN.md until round 5)
    s = open(path).read()
    a = s.index("| kernel at its BASELINE shape |")
    b = s.index("| kernel | used for (SURVEY §8a row)")
    s = s[:a] + "\n".join(figures()) + "\n\n" + s[b:]
    a = s.index("VAR-d30, fp32 rows, cold (commit")
    b = s.index("**Why 0.70 is not there")
    s = s[:a] + "\n".join(steps()) + "\n" + s[b:]
    open(path, "w").write(s)
    print("profiles/NOTES_r04.md: round-4 figures table and section 4c step tables rewritten from profiles/")


if __name__ == "__main__":
    if sys.argv[1:] == ["write"]:
        write()
    else:
        print("\n".join(steps() if sys.argv[1:] == ["steps"] else figures()))
