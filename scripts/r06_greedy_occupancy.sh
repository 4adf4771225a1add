#!/bin/bash
# Round 6: is GreedyCD's sweep bound by latency (waves per SIMD) or by instruction issue?  Resident workgroups per CU capped by dynamic LDS.
export NMFX_DEV=1
R="$(cd "$(dirname "$0")/.." && pwd)"; O="$R/gpurun_out/r06w"; mkdir -p "$O"; cd "$R"
B="python bench.py --no-cpu-baseline --alg greedycd --steps 10 --warmup 5 --all-events"
for n in 8 6 4 3 2 1; do
  NMFX_GREEDY_WGS_PER_CU=$n $B > "$O/greedy_wgs_$n.json" 2>> "$O/err.log"
done
python - <<'PY'
import json
for n in (8,6,4,3,2,1):
    d=json.load(open('gpurun_out/r06w/greedy_wgs_%d.json'%n))
    k={x['name']:x['avg_us'] for x in d['kernels']}
    print(n, 'workgroups per CU =', n, 'waves per SIMD:', d['ms_per_step'], 'greedy_W', round(k['greedy_W'],1), 'greedy_H', round(k['greedy_H'],1))
PY
