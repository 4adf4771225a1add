#!/bin/bash
# Round 6: potrf_reg_kernel alone (scripts/kbench/potrf_bench.hip: timing, phase stamps, correctness)
R="$(cd "$(dirname "$0")/.." && pwd)"; O="$R/gpurun_out/r06i"; mkdir -p "$O"; cd "$R"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I nmf.jl_amd/csrc -I include scripts/kbench/potrf_bench.hip -o /tmp/potrf_bench 2> "$O/potrf_bench_build.log"
timeout 120 /tmp/potrf_bench > "$O/potrf_bench.log" 2>&1
grep -v "^  step [0-7]: stage" "$O/potrf_bench.log"
