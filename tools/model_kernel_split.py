#!/usr/bin/env python3
"""rocprofv3 --kernel-trace --stats output of `tools/bench_model.py --paths Q --no-graphs --reps 1` (two batches:
warm-up + one timed) -> per-batch kernel split, written as profiles/<tag>.csv.

    python tools/model_kernel_split.py <rocprof output dir> <tag> [batches]
"""
import csv
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    src, tag = sys.argv[1], sys.argv[2]
    batches = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    stats = glob.glob(os.path.join(src, "**", "*kernel_stats.csv"), recursive=True)
    rows = list(csv.DictReader(open(stats[0])))
    total = sum(float(r["TotalDurationNs"]) for r in rows)
    with open(os.path.join(ROOT, "profiles", f"{tag}.csv"), "w") as f:
        w = csv.writer(f)
        w.writerow(["Name", "CallsPerBatch", "MsPerBatch", "AverageUs", "Percent"])
        for r in rows:
            w.writerow([r["Name"][:120], int(r["Calls"]) // batches, round(float(r["TotalDurationNs"]) / batches / 1e6, 3),
                        round(float(r["AverageNs"]) / 1e3, 1), round(100.0 * float(r["TotalDurationNs"]) / total, 2)])
    print("kernel ms per batch:", round(total / batches / 1e6, 1))


if __name__ == "__main__":
    main()
