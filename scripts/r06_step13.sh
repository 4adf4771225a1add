#!/bin/bash
# Round 6: the LDS-panel potrf (k > 256, Float64 k > 128) with the shared elimination, two trailing tiles in flight: tests, k = 512 lines.
export NMFX_DEV=1
R="$(cd "$(dirname "$0")/.." && pwd)"; O="$R/gpurun_out/r06y"; mkdir -p "$O"; cd "$R"
timeout 1200 python -m pytest tests/test_gpu_utils.py tests/test_gpu_projals_alspgrad.py tests/test_golden.py tests/test_frontend.py tests/test_gpu_c4_c5.py -x -q -m gpu -k "projals or pdsolve or pdrsolve or adddiag or posdef or rsvd or nndsvd or c4" > "$O/pytest_a.log" 2>&1
tail -3 "$O/pytest_a.log"
B="python bench.py --no-cpu-baseline --alg projals --steps 30 --warmup 10 --no-events"
: > "$O/lines.jsonl"
$B --p 8192 --n 16384 --k 512 >> "$O/lines.jsonl" 2>> "$O/err.log"
$B --p 8192 --n 16384 --k 512 >> "$O/lines.jsonl" 2>> "$O/err.log"
$B --p 4096 --n 4096 --k 512 >> "$O/lines.jsonl" 2>> "$O/err.log"
$B --dtype f64 --p 8192 --n 8192 --k 256 >> "$O/lines.jsonl" 2>> "$O/err.log"
python - <<'PY'
import json
for l in open('gpurun_out/r06y/lines.jsonl'):
    d=json.loads(l); print(d['config']['workload'][:50], d['ms_per_step'])
PY
