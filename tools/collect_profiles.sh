#!/bin/bash
# A round's evidence (FPQ_ROUND, default r05), every file headed by WHAT was measured: the commit (FPQ_GIT_HEAD, set by the caller: the GPU box has no
# .git) and the SHA-256 of the library that ran.  Counters in separate --pmc passes (never mixed with tracing) for the kernels
# DESIGN.md quotes, and the rocprofv3 --kernel-trace --stats summary + FETCH / WRITE passes of the bench.py command itself.
#   tools/collect_profiles.sh pmc [cases]  -> gpurun_out/prof_<round>/pmc_<case>.txt          (one GPU call)
#   tools/collect_profiles.sh bench        -> gpurun_out/prof_<round>/bench_*.{json,csv,txt}  (another)
#   tools/collect_profiles.sh steps        -> gpurun_out/prof_<round>/steps_*.json            (the ten scale steps of d30 / d36-512)
# python tools/install_profiles.py copies them to profiles/<round>_* .
set -e
round=${FPQ_ROUND:-r05}
out=$PWD/gpurun_out/prof_$round
mkdir -p $out
export TMPDIR=/tmp
sha=$(sha256sum fpqvar_amd/libfpq_hip.so | cut -c1-64)
stampline="# measured: commit ${FPQ_GIT_HEAD:-unknown}, libfpq_hip.so sha256 $sha, $(date -u +%Y-%m-%dT%H:%MZ)"
strip='s/^"(_ZN12_GLOBAL__N_[0-9]*[a-z_0-9]*)[^"]*"/\1/; s/^"void \(anonymous namespace\)::([a-z_0-9]*<[^>]*>)[^"]*"/\1/'
if [ "$1" = pmc ]; then
  for k in ${2:-adaln adaln32 adaln_codes rotate rotate_codes sym token6 dual6}; do
    tools/pmc_run.sh $k > /dev/null 2>&1
    { echo "$stampline"
      echo "# tools/pmc_run.sh $k  (tools/prof_one.py $k; averages per launch over 6 launches on three inputs in turn)"
      cat gpurun_out/pmc_$k/p1.summary.txt gpurun_out/pmc_$k/p2.summary.txt gpurun_out/pmc_$k/p3.summary.txt gpurun_out/pmc_$k/p4.summary.txt
      echo "# rocprofv3 --kernel-trace --stats of the same script:"
      grep -E "^\"_ZN12_GLOBAL__N_|^\"void \(anonymous namespace\)::" gpurun_out/pmc_$k/kernel_stats.csv | sed -E "$strip"
      python3 tools/pmc_derive.py $k gpurun_out/pmc_$k; } > $out/pmc_$k.txt
    echo "pmc $k ok"
  done
elif [ "$1" = steps ]; then
  for m in d30 d36-512; do
    python3 tools/bench_small_steps.py --model $m --rows fp32 --mode rotating --eager > $out/steps_${m}_fp32.json 2>> $out/steps.log
    python3 tools/bench_small_steps.py --model $m --rows fp16 --mode rotating > $out/steps_${m}_fp16.json 2>> $out/steps.log
    python3 tools/bench_small_steps.py --model $m --rows fp32 --mode resident > $out/steps_${m}_fp32_resident.json 2>> $out/steps.log
    echo "steps $m ok"
  done
else
  # the bench command itself: the line, the same command under --kernel-trace --stats, and the two traffic passes
  python3 bench.py > $out/bench_n1.json 2> $out/bench_n1.log
  echo "bench ok"
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/bench_kt -- python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --skip generation,format_search,steps > $out/bench_under_rocprof.json 2> $out/bench_kt.log
  find $out/bench_kt -name "*kernel_stats.csv" -exec cp {} $out/bench_kernel_stats_full.csv \;
  { echo "$stampline"; echo "# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --skip generation,format_search,steps (this library's kernels only)"
    grep -E "^\"Name\"|^\"_ZN12_GLOBAL__N_|^\"void \(anonymous namespace\)::" $out/bench_kernel_stats_full.csv | sed -E 's/^"(_ZN12_GLOBAL__N_[0-9]*[a-z_0-9]*)[^"]*"/"\1"/; s/^"void \(anonymous namespace\)::([a-z_0-9]*<[^>]*>)[^"]*"/"\1"/'; } > $out/bench_kernel_stats.csv
  rm -rf $out/bench_kt $out/bench_kernel_stats_full.csv
  echo "bench kernel trace ok"
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/bench_f -- python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline --skip generation,format_search,steps,other_kernels > /dev/null 2> $out/bench_f.log
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/bench_w -- python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline --skip generation,format_search,steps,other_kernels > /dev/null 2> $out/bench_w.log
  { echo "$stampline"; python3 tools/pmc_summary.py $out/bench_f rows16_lut_subwave; python3 tools/pmc_summary.py $out/bench_f groups32; } > $out/bench_fetch.txt
  { echo "$stampline"; python3 tools/pmc_summary.py $out/bench_w rows16_lut_subwave; python3 tools/pmc_summary.py $out/bench_w groups32; } > $out/bench_write.txt
  rm -rf $out/bench_f $out/bench_w
  echo "bench profiles ok"
fi
