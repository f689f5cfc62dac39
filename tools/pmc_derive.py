#!/usr/bin/env python3
"""The derived lines under a tools/pmc_run.sh summary: time, fraction of 8 TB/s at the case's algorithmic bytes, vector
instructions per element, vector-pipe busy share, HBM traffic = (FETCH_SIZE x 2 + WRITE_SIZE) KiB - the gfx950 correction
of the guide's HBM section, exactly as profiles/r03_pmc_*.txt computed it by hand.
usage: pmc_derive.py <prof_one.py case> <gpurun_out/pmc_<case> directory>"""
import csv
import re
import sys

CASES = {   # elements per launch, algorithmic bytes per element, what
    "adaln": (100 * 655 * 1920, 4.0, "fp16 rows -> E2M1 values (fp16)"),
    "adaln32": (100 * 655 * 1920, 6.0, "fp32 rows -> E2M1 values (fp16)"),
    "adaln_codes": (100 * 655 * 1920, 2.0 + 0.5 + 2.0 / 128, "fp16 rows -> E2M1 codes + fp16 group scales"),
    "rotate": (65536 * 1920, 4.0, "fp16 -> rotated, E2M1 values"),
    "rotate_smooth": (65536 * 1920, 4.0, "fp16 -> smoothed, rotated, E2M1 values"),
    "rotate_codes": (65536 * 1920, 2.0 + 0.5 + 2.0 / 128, "fp16 -> rotated, E2M1 codes + fp16 group scales"),
    "sym": (65536 * 1920, 4.0, "fp16 -> E2M1 g=128 values (the headline kernel)"),
    "token6": (65536 * 1920, 4.0, "fp16 -> E2M3 per token (rows of 1920)"),
    "group6": (65536 * 1920, 4.0, "fp16 -> E2M3 g=128"),
    "dual": (65536 * 1920, 4.0, "fp16 -> dual E1M2- / E2M1+ g=128 (no fix-up launch: clipping strength None)"),
    "dual6": (65536 * 7680, 4.0, "fp16 -> dual INT- / E2M3+ g=128"),
}


def counters(path):
    out = {}
    try:
        for line in open(path):
            p = line.split()
            if len(p) >= 2 and re.match(r"^[A-Z_0-9]+$", p[0]):
                out[p[0]] = float(p[1])
    except OSError:
        pass
    return out


def main():
    case, d = sys.argv[1], sys.argv[2]
    if case not in CASES:
        return
    n, bpe, what = CASES[case]
    c = {}
    for k in ("p1", "p2", "p3", "p4"):
        c.update(counters(f"{d}/{k}.summary.txt"))
    t_ns, best_total = None, -1.0
    try:
        for r in csv.reader(open(f"{d}/kernel_stats.csv")):   # Name, Calls, TotalDurationNs, AverageNs, ...
            ours = r and ("_GLOBAL__N_" in r[0] or r[0].startswith("void (anonymous namespace)::")) and "at::native" not in r[0]
            if ours and "zero_if_flag" not in r[0] and float(r[2]) > best_total:   # this library's kernel with the most time
                best_total, t_ns = float(r[2]), float(r[3])
    except (OSError, ValueError, IndexError):
        pass
    print(f"# derived ({what}; {n} elements, {bpe:.3f} algorithmic B/element = {n * bpe / 1e6:.1f} MB per launch):")
    if t_ns:
        print(f"#   kernel time (kernel-trace average) {t_ns / 1e3:.1f} us -> {n * bpe / t_ns:.0f} GB/s = {n * bpe / t_ns / 8000:.3f} of 8 TB/s")
    if "SQ_INSTS_VALU" in c:
        print(f"#   SQ_INSTS_VALU x 64 lanes / elements = {c['SQ_INSTS_VALU'] * 64 / n:.2f} vector instructions per element")
    if "GRBM_GUI_ACTIVE" in c and "SQ_ACTIVE_INST_VALU" in c:
        cyc = c["GRBM_GUI_ACTIVE"] / 8
        busy = c["SQ_ACTIVE_INST_VALU"] / 1024 * 4
        print(f"#   GRBM_GUI_ACTIVE / 8 XCDs = {cyc:.0f} cycles per launch; SQ_ACTIVE_INST_VALU / 1024 SIMDs x 4 = {busy:.0f} cycles = {busy / cyc:.2f} of them")
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        tr = (c["FETCH_SIZE"] * 2 + c["WRITE_SIZE"]) * 1024 / 1e6
        print(f"#   HBM traffic: FETCH_SIZE x 2 (gfx950 correction) + WRITE_SIZE = {tr:.1f} MB per launch = {tr / (n * bpe / 1e6):.3f} x algorithmic")


if __name__ == "__main__":
    main()
