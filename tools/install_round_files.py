#!/usr/bin/env python3
"""After tools/collect_profiles.sh {pmc,steps,bench} + the model / GEMM runs of a round's last library have been merged back
into gpurun_out/: install them under profiles/ (tools/install_profiles.py), refresh the model-generation record, the matrix-core
path's kernel split, the FP4 GEMM counter file's measured block and the K sweeps, and rewrite DESIGN.md's round-4 tables.
    python tools/install_round_files.py <commit> [previous bench line to keep as profiles/r04_bench_n1_commit_<c>.json]"""
import csv
import json
import os
import subprocess
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(R, "gpurun_out")
P = os.path.join(R, "profiles")
H = sys.argv[1]
sha = open(os.path.join(G, "model", "lib.sha")).read().split()[0]
subprocess.run([sys.executable, os.path.join(R, "tools", "install_profiles.py")], check=True)

# model-shaped batches: the new runs in front, everything older kept under a key that says what it preceded
old = json.load(open(os.path.join(P, "r04_model_generation.json")))
runs = {f: json.loads(open(os.path.join(G, "model", f + ".json")).read().strip().split("\n")[-1])
        for f in ("d30-256_w4a4", "d30-256_w6a6", "d36-512_w4a4", "d36-512_w6a6")}
if old["measured"].split(",")[0] != f"commit {H} (the round's last library)":
    hist = {k: v for k, v in old.items() if k not in ("measured", "runs")}
    old = {"measured": f"commit {H} (the round's last library), libfpq_hip.so sha256 {sha}, tools/bench_model.py --model <m> --config <c>, one MI355X box",
           "runs": runs, "previous_state_of_the_round": {"measured": old["measured"], "runs": old["runs"]}, **hist}
else:
    old["runs"] = runs
json.dump(old, open(os.path.join(P, "r04_model_generation.json"), "w"), indent=1)

# kernel split of the matrix-core path
subprocess.run([sys.executable, os.path.join(R, "tools", "model_kernel_split.py"), os.path.join(G, "model"), "r04_tmp_Q", "2"], check=True)
src = open(os.path.join(P, "r04_tmp_Q.csv")).read()
rows = list(csv.DictReader(src.split("\n")))
tot = sum(float(r["MsPerBatch"]) for r in rows)
fp4 = sum(float(r["MsPerBatch"]) for r in rows if "gemm_fp4" in r["Name"])
open(os.path.join(P, "r04_model_d30_Q_path_kernel_stats.csv"), "w").write(
    f"# measured: commit {H}, libfpq_hip.so sha256 {sha}; rocprofv3 --kernel-trace --stats -- python3 tools/bench_model.py --model d30-256 "
    f"--paths Q --no-graphs --reps 1 (two batches: warm-up + one timed; figures per batch; {tot:.1f} ms of kernels per batch; the FP4 GEMM "
    f"{fp4:.1f} ms - before this round's GEMM work, commit 37e2df9: 76.9 ms as 900 launches of the 128 x 128 tiling)\n" + src)
os.remove(os.path.join(P, "r04_tmp_Q.csv"))

# the FP4 GEMM counter file: stamp + measured block replaced, the derived text kept
std = open(os.path.join(P, "r04_pmc_gemm.txt")).read().split("\n")[0]
a = open(os.path.join(G, "pmc_gemm_a.txt")).read()
gem = a[a.index("_ZN12_GLOBAL__N_120gemm_fp4_glds_kernel"):].rstrip().split("\n")
path = os.path.join(P, "r04_pmc_gemm_fp4.txt")
lines = open(path).read().split("\n")
i0 = [i for i, l in enumerate(lines) if l.startswith("_ZN12_GLOBAL__N_120gemm_fp4_glds_kernelIfLi8ELi4")][0]
i1 = [i for i, l in enumerate(lines) if l.startswith("# derived")][0]
open(path, "w").write("\n".join([std] + lines[1:i0] + gem + lines[i1:]))

# the FP6 GEMM counters: the MFMA pass appended to the standard file
a6 = open(os.path.join(G, "pmc_gemm6_a.txt")).read()
p6 = os.path.join(P, "r04_pmc_gemm6.txt")
t6 = open(p6).read()
if "SQ_INSTS_MFMA" not in t6:
    open(p6, "w").write(t6.rstrip("\n") + "\n# tools/pmc_pass.sh gemm6_a gemm6 SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES ... (one more --pmc pass of the same launch):\n" +
                        a6[a6.index("_ZN12_GLOBAL__N_120gemm_fp6_rows_kernel"):])

# K sweeps of the three GEMMs (one box)
with open(os.path.join(P, "r04_gemm_k_sweep.txt"), "w") as f:
    f.write(f"# FP4 / FP6 / FP8 GEMM [65536 x K] . [5760 x K]^T, time against K / 128 (tools/gemm_k_sweep.py [fp4|fp6|fp8]); commit {H}, "
            f"libfpq_hip.so sha256 {sha}, one box\n")
    for k in ("fp4", "fp6", "fp8"):
        f.write("".join(l for l in open(os.path.join(G, f"gemm_k_{k}.log")) if "amdgpu.ids" not in l))
subprocess.run([sys.executable, os.path.join(R, "tools", "design_tables.py"), "write"], check=True)
print("installed; now check the prose of DESIGN.md / README.md against profiles/r04_bench_n1.json")
