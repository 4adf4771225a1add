#!/bin/bash
R="$(cd "$(dirname "$0")/.." && pwd)"; O="$R/gpurun_out/s3"; mkdir -p "$O"; cd "$R"
hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/kbench/mfma_f64_probe.hip -o /tmp/mfma_f64_probe 2>/dev/null
hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/kbench/group_barrier_probe.hip -o /tmp/group_barrier_probe 2>/dev/null
timeout 120 /tmp/mfma_f64_probe > "$O/mfma_f64_probe.log" 2>&1
timeout 120 /tmp/group_barrier_probe 200 > "$O/group_barrier_probe.log" 2>&1
cat "$O/mfma_f64_probe.log" "$O/group_barrier_probe.log"
timeout 600 python -m pytest tests/test_gpu_projals_alspgrad.py -q -m gpu -k "projals" 2>&1 | tail -5
