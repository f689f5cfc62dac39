#!/bin/bash
# usage: tools/pmc_pass.sh <tag> <prof_one.py case> <counter> [counter ...]   - ONE rocprofv3 --pmc pass (never mixed with
# tracing); per-kernel averages to gpurun_out/pmc_<tag>.txt and stdout.  FPQ_PROF_LIB=<lib.so> profiles a variant build.
export TMPDIR=/tmp
tag=$1; which=$2; shift 2
out=$PWD/gpurun_out/pmcpass_$tag
mkdir -p $out
rocprofv3 --pmc "$@" --output-format csv -d $out/p -- python3 tools/prof_one.py $which > $out/log 2>&1
python3 tools/pmc_summary.py $out/p > $PWD/gpurun_out/pmc_$tag.txt
rm -rf $out
echo "== $tag"; cat $PWD/gpurun_out/pmc_$tag.txt
