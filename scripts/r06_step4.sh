#!/bin/bash
export NMFX_DEV=1
R="$(cd "$(dirname "$0")/.." && pwd)"; O="$R/gpurun_out/r06g"; mkdir -p "$O"; cd "$R"
python bench.py --no-cpu-baseline --alg alspgrad --dtype f64 --p 32768 --n 4096 --k 512 --steps 2 --warmup 1 --all-events > "$O/bench_alspgrad_c5_shard_all_events.json" 2>> "$O/err.log"
python - <<'PY'
import json
d=json.load(open('gpurun_out/r06g/bench_alspgrad_c5_shard_all_events.json')); print(d['ms_per_step'], d['ms_per_step_no_events'])
for k in d['kernels']: print(k)
PY
