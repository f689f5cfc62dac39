#!/bin/bash
# usage: tools/pmc_p1.sh <tag> <prof_one.py case>   (environment selects the kernel variant) - one rocprofv3 --pmc pass with the
# wavefront-time counters; prints the summary and the average resident wavefronts per SIMD
export TMPDIR=/tmp
out=$PWD/gpurun_out/pmc1_$1
mkdir -p $out
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $out/p1 -- python3 tools/prof_one.py $2 > $out/p1.log 2>&1
python3 tools/pmc_summary.py $out/p1 > $out/p1.summary.txt
rm -rf $out/p1
echo "== $1"; cat $out/p1.summary.txt
python3 - $out/p1.summary.txt <<'PY'
import sys
v = {}
for l in open(sys.argv[1]):
    p = l.split()
    if len(p) >= 2 and p[0].startswith(("SQ_", "GRBM")):
        v[p[0]] = float(p[1])
cyc = v["GRBM_GUI_ACTIVE"] / 8
print("   avg resident wavefronts per SIMD: %.2f   VALU-issuing share of wavefront time: %.2f   at s_waitcnt: %.2f   issue-stalled: %.2f" % (
    v["SQ_WAVE_CYCLES"] * 4 / 1024 / cyc, v["SQ_ACTIVE_INST_VALU"] / v["SQ_WAVE_CYCLES"], v["SQ_WAIT_ANY"] / v["SQ_WAVE_CYCLES"],
    v["SQ_WAIT_INST_ANY"] / v["SQ_WAVE_CYCLES"]))
PY
