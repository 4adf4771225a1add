#!/bin/bash
# Round 6: C4 and C5 at FULL size on one GPU (the per-config lines are the 8-GPU problems' shard shapes; the verdict asked for these too).
export NMFX_DEV=1
R="$(cd "$(dirname "$0")/.." && pwd)"; O="$R/gpurun_out/r06z"; mkdir -p "$O"; cd "$R"
: > "$O/fullsize.jsonl"
timeout 900 python bench.py --no-cpu-baseline --alg projals --p 16384 --n 131072 --k 256 --steps 10 --warmup 3 >> "$O/fullsize.jsonl" 2>> "$O/err.log"
timeout 1500 python bench.py --no-cpu-baseline --alg alspgrad --dtype f64 --p 32768 --n 32768 --k 512 --steps 2 --warmup 1 >> "$O/fullsize.jsonl" 2>> "$O/err.log"
python - <<'PY'
import json
for l in open('gpurun_out/r06z/fullsize.jsonl'):
    d=json.loads(l); print(d['config']['workload'][:60], d['ms_per_step'], d.get('frac_of_mfma_peak'))
PY
tail -3 "$O/err.log" | grep -v amdgpu
