#!/bin/bash
# The adaLN producer over the ten scale steps under FPQ_ADALN_ROWS = rows per workgroup (cold inputs, hipGraph replay):
#   tools/sweep_steps.sh <outdir> "<rows values>" "<models>" "<dtypes>"
out=${1:-gpurun_out/sweep}; mkdir -p $out
for m in ${3:-d30 d36-512}; do for d in ${4:-fp32 fp16}; do for r in ${2:-4 8 12 16}; do
  FPQ_ADALN_ROWS=$r python tools/bench_small_steps.py --model $m --rows $d --ops adaln > $out/adaln_${m}_${d}_rows$r.json 2>> $out/err.log || exit 1
  echo "$m $d rows/wg $r ok"
done; done; done
python - "$out" <<'PY'
import glob, json, os, sys
out = sys.argv[1]
tab = {}
for f in sorted(glob.glob(os.path.join(out, "adaln_*_rows*.json"))):
    d = json.load(open(f))
    key = (d["model"], d["rows_dtype_of_the_residual_stream"])
    r = int(f.rsplit("rows", 1)[1].split(".")[0])
    tab.setdefault(key, {})[r] = [s["adaln"]["us"] for s in d["steps"]]
    tab[key]["rows"] = [s["rows"] for s in d["steps"]]
with open(os.path.join(out, "summary.txt"), "w") as fh:
    for key, v in tab.items():
        print(key, file=fh)
        print("   rows      " + "".join(f"{x:>9d}" for x in v["rows"]), file=fh)
        for r in sorted(k for k in v if k != "rows"):
            print(f"   rows/wg {r:<3d}" + "".join(f"{x:9.2f}" for x in v[r]) + f"   sum {sum(v[r]):8.1f}", file=fh)
print(open(os.path.join(out, "summary.txt")).read())
PY
