#!/bin/bash
export NMFX_DEV=1
R="$(cd "$(dirname "$0")/.." && pwd)"; O="$R/gpurun_out/s3"; mkdir -p "$O"; cd "$R"
export GPU_MAX_HW_QUEUES=24
B="python bench.py --no-cpu-baseline --alg projals --steps 30 --warmup 5"
$B --no-events > "$O/projals_after_reduce.json" 2>/dev/null
$B --all-events > "$O/projals_all_events2.json" 2>/dev/null
timeout 900 python -m pytest tests/test_gpu_projals_alspgrad.py tests/test_gpu_c4_c5.py -q -m gpu -k "projals or c4" 2>&1 | tail -4
python - <<'PY'
import json
d=json.load(open("gpurun_out/s3/projals_after_reduce.json")); print("projals ms/step", d["ms_per_step"])
d=json.load(open("gpurun_out/s3/projals_all_events2.json"))
for k in d["kernels"]: print(f"{k['name']:26s} {k['avg_us']:8.2f}")
PY
