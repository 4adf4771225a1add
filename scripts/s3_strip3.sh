#!/bin/bash
export NMFX_DEV=1
R="$(cd "$(dirname "$0")/.." && pwd)"; O="$R/gpurun_out/s3"; mkdir -p "$O"; cd "$R"
export GPU_MAX_HW_QUEUES=24
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I nmf.jl_amd/csrc scripts/kbench/potrs_bench.hip -o /tmp/potrs_bench 2> "$O/potrs_bench_build.err"
timeout 300 /tmp/potrs_bench 6 1 > "$O/potrs_strip_bench2.log" 2>&1
timeout 900 python -m pytest tests/test_gpu_projals_alspgrad.py tests/test_gpu_utils.py -q -m gpu -k "projals or pdsolve" 2>&1 | tail -6 > "$O/tests_potrs4.log"
: > "$O/projals_k512.jsonl"
for st in 1 0; do NMFX_POTRS_STRIP=$st python bench.py --no-cpu-baseline --alg projals --p 8192 --n 16384 --k 512 --steps 20 --warmup 5 --no-events >> "$O/projals_k512.jsonl" 2>/dev/null; done
cat "$O/potrs_strip_bench2.log"; tail -4 "$O/tests_potrs4.log"
python - <<'PY'
import json
for l in open("gpurun_out/s3/projals_k512.jsonl"):
    try: d=json.loads(l)
    except Exception: continue
    print(d.get("ms_per_step"), d.get("config",{}).get("workload"))
PY
