#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (kernel-trace stats + separate --pmc passes) into the
small files committed under profiles/.

    python tools/summarize_prof.py <gpurun_out/prof dir> <tag>   ->  profiles/<tag>_*.{csv,json}
"""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name, n=110):
    return name if len(name) <= n else name[:n] + "..."


def main():
    src, tag = sys.argv[1], sys.argv[2]
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    out = {}
    stats = glob.glob(os.path.join(src, "kt", "**", "*kernel_stats.csv"), recursive=True)
    if stats:
        rows = list(csv.DictReader(open(stats[0])))
        with open(os.path.join(ROOT, "profiles", f"{tag}_kernel_stats.csv"), "w") as f:
            w = csv.writer(f)
            w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
            for r in rows:
                w.writerow([short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"],
                            r["MinNs"], r["MaxNs"], r["StdDev"]])
        hot = [r for r in rows if "rows16_lut_subwave_kernel" in r["Name"]]
        if hot:
            out["kernel"] = short(hot[0]["Name"])
            out["calls"] = int(hot[0]["Calls"])
            out["avg_ns"] = float(hot[0]["AverageNs"])
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        files = glob.glob(os.path.join(src, counter.lower(), "**", "*counter_collection.csv"), recursive=True)
        vals = []
        for fn in files:
            for r in csv.DictReader(open(fn)):
                if r.get("Counter_Name") == counter and "rows16_lut_subwave_kernel" in r.get("Kernel_Name", ""):
                    vals.append(float(r["Counter_Value"]))
        if vals:
            out[counter + "_avg_per_launch_raw"] = sum(vals) / len(vals)
            out[counter + "_launches"] = len(vals)
    if "FETCH_SIZE_avg_per_launch_raw" in out and "WRITE_SIZE_avg_per_launch_raw" in out:
        # MI355X_MICROARCH.md, HBM: counters are in KiB; on gfx950 FETCH_SIZE reports exactly half
        # of the bytes of a wide (16 B/lane) coalesced streaming read -> double it; WRITE_SIZE is exact.
        fetch = out["FETCH_SIZE_avg_per_launch_raw"] * 1024 * 2
        write = out["WRITE_SIZE_avg_per_launch_raw"] * 1024
        out["fetch_bytes_corrected"] = fetch
        out["write_bytes"] = write
        out["traffic_bytes_per_launch"] = fetch + write
        out["algorithmic_bytes_per_launch"] = 65536 * 1920 * 4
        out["note"] = ("FETCH_SIZE*1024*2 (gfx950 half-count correction for 16 B/lane streaming reads) + "
                       "WRITE_SIZE*1024; separate --pmc passes")
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json"), "w") as f:
            json.dump(out, f, indent=1)
    with open(os.path.join(ROOT, "profiles", f"{tag}_summary.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
