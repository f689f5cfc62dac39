#!/bin/bash
# Round-2 evidence, one GPU call: counters (separate --pmc passes, never mixed with tracing) for the kernels DESIGN.md
# quotes, and the rocprofv3 --kernel-trace --stats summary + FETCH/WRITE passes of the bench.py command itself.
# Output: gpurun_out/prof_r02/*.txt|csv|json (copied to profiles/ by hand after reading them).
set -e
out=$PWD/gpurun_out/prof_r02
mkdir -p $out
export TMPDIR=/tmp
for k in sym rotate dual dual6 calib channel adaln adaln32; do
  tools/pmc_run.sh $k > /dev/null 2>&1
  { echo "# tools/pmc_run.sh $k  (tools/prof_one.py $k; averages per launch over 5 launches)"; cat gpurun_out/pmc_$k/p1.summary.txt gpurun_out/pmc_$k/p2.summary.txt gpurun_out/pmc_$k/p3.summary.txt gpurun_out/pmc_$k/p4.summary.txt; echo "# rocprofv3 --kernel-trace --stats of the same script:"; grep -E "^\"_ZN12_GLOBAL__N_|^\"void \(anonymous namespace\)::" gpurun_out/pmc_$k/kernel_stats.csv | sed -E 's/^"(_ZN12_GLOBAL__N_[0-9]*[a-z_0-9]*)[^"]*"/\1/; s/^"void \(anonymous namespace\)::([a-z_0-9]*<[^>]*>)[^"]*"/\1/'; } > $out/pmc_$k.txt
  echo "pmc $k ok"
done
FPQ_ADALN_V1=1 tools/pmc_run.sh adaln > /dev/null 2>&1
{ echo "# FPQ_ADALN_V1=1 tools/pmc_run.sh adaln  (the round-1 kernel, adaln_rotate_quant16_kernel)"; cat gpurun_out/pmc_adaln/p1.summary.txt gpurun_out/pmc_adaln/p2.summary.txt gpurun_out/pmc_adaln/p3.summary.txt gpurun_out/pmc_adaln/p4.summary.txt; grep -E "^\"_ZN12_GLOBAL__N_|^\"void \(anonymous namespace\)::" gpurun_out/pmc_adaln/kernel_stats.csv | sed -E 's/^"(_ZN12_GLOBAL__N_[0-9]*[a-z_0-9]*)[^"]*"/\1/; s/^"void \(anonymous namespace\)::([a-z_0-9]*<[^>]*>)[^"]*"/\1/'; } > $out/pmc_adaln_round1_kernel.txt
echo "pmc adaln v1 ok"
# the butterfly forms of the rotation (before the matrix-core transform), same counters
for k in rotate adaln; do
  FPQ_ROT_BUTTERFLY=1 tools/pmc_run.sh $k > /dev/null 2>&1
  { echo "# FPQ_ROT_BUTTERFLY=1 tools/pmc_run.sh $k  (the butterfly form of the 128-point transform, DPP / permlane exchanges)"; cat gpurun_out/pmc_$k/p1.summary.txt gpurun_out/pmc_$k/p2.summary.txt gpurun_out/pmc_$k/p3.summary.txt gpurun_out/pmc_$k/p4.summary.txt; echo "# rocprofv3 --kernel-trace --stats of the same script:"; grep -E "^\"_ZN12_GLOBAL__N_|^\"void \(anonymous namespace\)::" gpurun_out/pmc_$k/kernel_stats.csv | sed -E 's/^"(_ZN12_GLOBAL__N_[0-9]*[a-z_0-9]*)[^"]*"/\1/; s/^"void \(anonymous namespace\)::([a-z_0-9]*<[^>]*>)[^"]*"/\1/'; } > $out/pmc_${k}_butterfly.txt
  echo "pmc $k butterfly ok"
done
# the bench command itself
rocprofv3 --kernel-trace --stats --output-format csv -d $out/bench_kt -- python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline > $out/bench_under_rocprof.json 2> $out/bench_kt.log
find $out/bench_kt -name "*kernel_stats.csv" -exec cp {} $out/bench_kernel_stats_full.csv \;
grep -E "^\"Name\"|^\"_ZN12_GLOBAL__N_|^\"void \(anonymous namespace\)::" $out/bench_kernel_stats_full.csv | sed -E 's/^"(_ZN12_GLOBAL__N_[0-9]*[a-z_0-9]*)[^"]*"/"\1"/; s/^"void \(anonymous namespace\)::([a-z_0-9]*<[^>]*>)[^"]*"/"\1"/' > $out/bench_kernel_stats.csv
rm -rf $out/bench_kt $out/bench_kernel_stats_full.csv
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/bench_f -- python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline > /dev/null 2> $out/bench_f.log
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/bench_w -- python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline > /dev/null 2> $out/bench_w.log
python3 tools/pmc_summary.py $out/bench_f rows16_lut_subwave > $out/bench_fetch.txt
python3 tools/pmc_summary.py $out/bench_w rows16_lut_subwave > $out/bench_write.txt
python3 tools/pmc_summary.py $out/bench_f groups32 >> $out/bench_fetch.txt
python3 tools/pmc_summary.py $out/bench_w groups32 >> $out/bench_write.txt
rm -rf $out/bench_f $out/bench_w
echo "bench profiles ok"
