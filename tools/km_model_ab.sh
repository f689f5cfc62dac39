set -e
mkdir -p gpurun_out/r05r
for m in d30-256 d36-512; do
  for f in "" "--row-major-operands"; do
    timeout -k 10 300 python tools/bench_model.py --model $m --paths Q --tuned-gemms --reps 3 $f >> gpurun_out/r05r/model_km_ab.txt 2>&1
  done
done
timeout -k 10 300 python tools/bench_model.py --model d30-256 --config w6a6 --paths Q --tuned-gemms --reps 3 >> gpurun_out/r05r/model_km_ab.txt 2>&1
timeout -k 10 300 python tools/bench_model.py --model d30-256 --config w6a6 --paths Q --tuned-gemms --reps 3 --row-major-operands >> gpurun_out/r05r/model_km_ab.txt 2>&1
grep -v amdgpu.ids gpurun_out/r05r/model_km_ab.txt
