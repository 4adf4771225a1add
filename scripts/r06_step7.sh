#!/bin/bash
# Round 6: register-resident potrf (chol.hpp: potrf_reg_kernel) -- standalone timing / correctness, projals tests, ProjectedALS A/B lines.
export NMFX_DEV=1
R="$(cd "$(dirname "$0")/.." && pwd)"; O="$R/gpurun_out/r06i"; mkdir -p "$O"; cd "$R"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I nmf.jl_amd/csrc -I include scripts/kbench/potrf_bench.hip -o /tmp/potrf_bench 2> "$O/potrf_bench_build.log"
timeout 120 /tmp/potrf_bench > "$O/potrf_bench.log" 2>&1
cat "$O/potrf_bench.log"
timeout 900 python -m pytest tests/test_gpu_utils.py tests/test_gpu_projals_alspgrad.py tests/test_golden.py -x -q -m gpu -k "projals or pdsolve or pdrsolve or adddiag or posdef" > "$O/pytest_a.log" 2>&1
tail -5 "$O/pytest_a.log"
B="python bench.py --no-cpu-baseline --alg projals --steps 30 --warmup 10"
: > "$O/projals.jsonl"
for rep in 1 2; do
  $B --no-events >> "$O/projals.jsonl" 2>> "$O/err.log"
  NMFX_CHOL_UNROLLED=0 $B --no-events >> "$O/projals.jsonl" 2>> "$O/err.log"
  NMFX_POTRF_REG=0 $B --no-events >> "$O/projals.jsonl" 2>> "$O/err.log"
done
for s in 2 4; do NMFX_CHOL_SLOTS=$s $B --no-events >> "$O/projals.jsonl" 2>> "$O/err.log"; done
$B --p 4096 --n 4096 --no-events >> "$O/projals.jsonl" 2>> "$O/err.log"
NMFX_POTRF_REG=0 $B --p 4096 --n 4096 --no-events >> "$O/projals.jsonl" 2>> "$O/err.log"
NMFX_CHOL_SLOTS=0 $B --p 4096 --n 4096 --no-events >> "$O/projals.jsonl" 2>> "$O/err.log"
$B --all-events > "$O/projals_all_events.json" 2>> "$O/err.log"
python - <<'PY'
import json
for l in open('gpurun_out/r06i/projals.jsonl'):
    d=json.loads(l); print(d['config'].get('workload'), d['ms_per_step'])
d=json.load(open('gpurun_out/r06i/projals_all_events.json'))
print(d['ms_per_step'], [(k['name'],round(k['avg_us'],1)) for k in d['kernels']])
PY
tail -5 "$O/err.log"
