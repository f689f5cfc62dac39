#!/usr/bin/env python3
"""gpurun_out/prof_<round>/* (written by tools/collect_profiles.sh on the GPU box; FPQ_ROUND, default r05) -> profiles/<round>_*: the PMC summaries as they
are, the step profiles with one line per step, the bench line and kernel stats, and <round>_pmc_traffic.json recomputed from the
FETCH_SIZE / WRITE_SIZE passes of the bench command.      python tools/install_profiles.py [pmc] [steps] [bench]"""
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RND = os.environ.get("FPQ_ROUND", "r05")
SRC = os.path.join(ROOT, "gpurun_out", "prof_" + RND)
DST = os.path.join(ROOT, "profiles")
what = sys.argv[1:] or ["pmc", "steps", "bench"]

if "pmc" in what:
    for f in sorted(os.listdir(SRC)):
        if f.startswith("pmc_") and f.endswith(".txt"):
            shutil.copy(os.path.join(SRC, f), os.path.join(DST, RND + "_" + f))
            print("profiles/" + RND + "_" + f)

if "steps" in what:
    for f in sorted(os.listdir(SRC)):
        m = re.match(r"steps_(.*)\.json$", f)
        if not m:
            continue
        d = json.load(open(os.path.join(SRC, f)))
        steps, eager = d.pop("steps"), d.pop("eager", None)
        out = json.dumps(d, indent=1)[:-2] + ',\n "steps": [\n' + ",\n".join("  " + json.dumps(s) for s in steps) + "\n ]"
        if eager:
            out += ',\n "eager": [\n' + ",\n".join("  " + json.dumps(s) for s in eager) + "\n ]"
        name = RND + "_steps_" + m.group(1).replace("d36-512", "d36") + ".json"
        open(os.path.join(DST, name), "w").write(out + "\n}\n")
        json.load(open(os.path.join(DST, name)))
        print("profiles/" + name, d["git_head"], d["time_weighted_frac_of_8TBps"])

if "bench" in what:
    shutil.copy(os.path.join(SRC, "bench_n1.json"), os.path.join(DST, RND + "_bench_n1.json"))
    shutil.copy(os.path.join(SRC, "bench_under_rocprof.json"), os.path.join(DST, RND + "_bench_n1_under_rocprofv3.json"))
    shutil.copy(os.path.join(SRC, "bench_kernel_stats.csv"), os.path.join(DST, RND + "_bench_kernel_stats.csv"))

    def counters(path):
        head, vals = open(path).readline().strip(), []
        for line in open(path):
            p = line.split()
            if len(p) >= 2 and p[0] in ("FETCH_SIZE", "WRITE_SIZE"):
                vals.append(float(p[1]))
        return head, vals
    head, (f, cf) = counters(os.path.join(SRC, "bench_fetch.txt"))
    _, (w, cw) = counters(os.path.join(SRC, "bench_write.txt"))
    t = json.load(open(os.path.join(DST, (RND if os.path.exists(os.path.join(DST, RND + "_pmc_traffic.json")) else "r04") + "_pmc_traffic.json")))   # the previous record is the template
    t["measured"] = head.lstrip("# ").replace("measured: ", "") + " (tools/collect_profiles.sh bench)"
    t["FETCH_SIZE_KB_per_launch"], t["WRITE_SIZE_KB_per_launch"] = f, w
    t["traffic_bytes_per_launch"] = (2 * f + w) * 1024
    t["ratio"] = t["traffic_bytes_per_launch"] / t["algorithmic_bytes_per_launch"]
    c = t["calibration_kernel"]
    c["FETCH_SIZE_KB_per_launch"], c["WRITE_SIZE_KB_per_launch"] = cf, cw
    c["traffic_bytes_per_launch"] = (2 * cf + cw) * 1024
    c["ratio"] = c["traffic_bytes_per_launch"] / c["algorithmic_bytes_per_launch"]
    json.dump(t, open(os.path.join(DST, RND + "_pmc_traffic.json"), "w"), indent=1)
    line = json.loads(open(os.path.join(SRC, "bench_n1.json")).read().strip().splitlines()[-1])
    print("profiles/" + RND + "_bench_n1.json", line["value"], line["roofline"]["frac"], "traffic ratio", round(t["ratio"], 6))
