#!/bin/bash
# strip-kernel potrs: stand-alone bench with its host check, the tests that exercise the route, A/B of the projals iteration; and a kernel
# trace of the side-stream tail (which launches really overlap)
export NMFX_DEV=1
R="$(cd "$(dirname "$0")/.." && pwd)"; O="$R/gpurun_out/s3"; mkdir -p "$O"; cd "$R"
export GPU_MAX_HW_QUEUES=24
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I nmf.jl_amd/csrc scripts/kbench/potrs_bench.hip -o /tmp/potrs_bench 2> "$O/potrs_bench_build.err"
timeout 300 /tmp/potrs_bench 6 1 > "$O/potrs_strip_bench.log" 2>&1
timeout 300 /tmp/potrs_bench 6 > "$O/potrs_panel_bench.log" 2>&1
timeout 900 python -m pytest tests/test_gpu_utils.py tests/test_gpu_projals_alspgrad.py tests/test_golden.py -x -q -m gpu 2>&1 | tail -15 > "$O/tests_potrs.log"
B="python bench.py --no-cpu-baseline --alg projals --steps 30 --warmup 5 --no-events"
: > "$O/projals_ab.jsonl"
for st in 1 0; do NMFX_POTRS_STRIP=$st $B >> "$O/projals_ab.jsonl" 2>/dev/null; done
NMFX_POTRS=1 NMFX_POTRS_STRIP=0 $B >> "$O/projals_ab.jsonl" 2>/dev/null
python bench.py --no-cpu-baseline --alg projals --steps 30 --warmup 5 --all-events > "$O/projals_all_events.json" 2>/dev/null
cd /tmp && export TMPDIR=/tmp
NMFX_SIDE_TAIL=1 rocprofv3 --kernel-trace -d "$O/trace_side" -o t -- python "$R/bench.py" --no-cpu-baseline --sim-ranks 8 --steps 12 --warmup 3 --no-events --transport rccl > "$O/trace_side.log" 2>&1
python "$R/scripts/trace_timeline.py" "$(find "$O/trace_side" -name '*.db' | head -1)" gemm 60 > "$O/trace_side_timeline.txt" 2>&1
rm -rf "$O/trace_side"
cd "$R"
cat "$O/potrs_strip_bench.log"; tail -4 "$O/tests_potrs.log"
python - <<'PY'
import json
for l in open("gpurun_out/s3/projals_ab.jsonl"):
    try: d=json.loads(l)
    except Exception: continue
    print(d.get("ms_per_step"), d.get("config",{}).get("workload"))
PY
head -70 "$O/trace_side_timeline.txt"
