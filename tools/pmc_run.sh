#!/bin/bash
# usage: tools/pmc_run.sh <prof_one.py case> -- four separate rocprofv3 --pmc passes (never mixed with tracing) +
# one --kernel-trace --stats pass; summaries land in gpurun_out/pmc_<case>/
set -e
which=$1
out=$PWD/gpurun_out/pmc_$which
mkdir -p $out
export TMPDIR=/tmp
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE"
P2="SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM"
rocprofv3 --pmc $P1 --output-format csv -d $out/p1 -- python3 tools/prof_one.py $which > $out/p1.log 2>&1
rocprofv3 --pmc $P2 --output-format csv -d $out/p2 -- python3 tools/prof_one.py $which > $out/p2.log 2>&1 || echo "p2 failed" >> $out/p2.log
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/p3 -- python3 tools/prof_one.py $which > $out/p3.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/p4 -- python3 tools/prof_one.py $which > $out/p4.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -- python3 tools/prof_one.py $which > $out/kt.log 2>&1
for p in p1 p2 p3 p4; do python3 tools/pmc_summary.py $out/$p > $out/$p.summary.txt 2>&1 || true; done
find $out/kt -name "*kernel_stats.csv" -exec cp {} $out/kernel_stats.csv \;
# keep only the summaries (the raw per-dispatch CSVs are large)
rm -rf $out/p1 $out/p2 $out/p3 $out/p4 $out/kt
echo "pmc $which done"
