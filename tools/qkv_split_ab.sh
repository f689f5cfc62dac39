set -e
mkdir -p gpurun_out/r05ak
for m in d30-256 d36-512; do
  for f in "" "--qkv-copy-in"; do
    timeout -k 10 300 python tools/bench_model.py --model $m --paths Q --tuned-gemms --reps 3 $f >> gpurun_out/r05ak/qkv_split_ab.txt 2>&1
  done
done
grep -v amdgpu gpurun_out/r05ak/qkv_split_ab.txt | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l)
    print(d['workload'].split(',')[0], 'k, v into the cache by the GEMM' if d['qkv_to_cache'] else 'copy-in pass', 'Q', d['Q_ms_per_batch_hipgraph'], 'ms per batch (hipGraph)')
"
