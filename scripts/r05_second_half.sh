#!/bin/bash
# Round 5, second half -- how the measurements behind profiles/r05_potrs_*, r05_projals_*, r05_mfma_f64_*, r05_gemm_bench_f64_*,
# r05_group_barrier_* and r05_side_stream_* were taken (GPU box; everything lands under gpurun_out/r05s/).  The per-configuration bench
# lines and the rocprofv3 view of the ProjectedALS line are scripts/r05_final_refresh.sh.
export NMFX_DEV=1   # development switches (csrc/comm.hpp: dev_env)
R="$(cd "$(dirname "$0")/.." && pwd)"; O="$R/gpurun_out/r05s"; mkdir -p "$O"; cd "$R"
export GPU_MAX_HW_QUEUES=24
CC="hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I nmf.jl_amd/csrc"
# potrs! by strips: stand-alone bench with its host check (argument 2 = 1), round 4's panel kernel for comparison (no argument)
$CC scripts/kbench/potrs_bench.hip -o /tmp/potrs_bench 2> "$O/potrs_bench_build.err"
timeout 300 /tmp/potrs_bench 6 1 > "$O/potrs_strip_bench.log" 2>&1
timeout 300 /tmp/potrs_bench 6 > "$O/potrs_panel_bench.log" 2>&1
# its SQ counters (own pass, kernel trace only)
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 \
    --kernel-trace --output-format csv -d "$O/pmc" -o p -- /tmp/potrs_bench 3 1 > "$O/pmc_run.log" 2>&1 )
python scripts/pmc_summary.py "$O/pmc" potrs_strip_kernel > "$O/potrs_strip_sq_counters.txt"; rm -rf "$O/pmc"
# ProjectedALS: the three H-solve kernels against each other, the f32 error study on both routes, k = 512
B="python bench.py --no-cpu-baseline --alg projals --steps 30 --warmup 5 --no-events"
: > "$O/projals_strip_vs_product_vs_panel.jsonl"
NMFX_POTRS_STRIP=1 $B >> "$O/projals_strip_vs_product_vs_panel.jsonl" 2>/dev/null
NMFX_POTRS=0 $B >> "$O/projals_strip_vs_product_vs_panel.jsonl" 2>/dev/null
NMFX_POTRS=1 NMFX_POTRS_STRIP=0 $B >> "$O/projals_strip_vs_product_vs_panel.jsonl" 2>/dev/null
: > "$O/projals_k512.jsonl"
for st in 1 0; do NMFX_POTRS_STRIP=$st python bench.py --no-cpu-baseline --alg projals --p 8192 --n 16384 --k 512 --steps 20 --warmup 5 --no-events >> "$O/projals_k512.jsonl" 2>/dev/null; done
python scripts/projals_f32_error.py > "$O/projals_f32_error.log" 2>&1
# probes: the Float64 matrix-core instruction's ceiling, the Float64 product over the contraction, the synchronisation of a persistent 2-D grid
hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/kbench/mfma_f64_probe.hip -o /tmp/mfma_f64_probe 2>/dev/null && timeout 120 /tmp/mfma_f64_probe > "$O/mfma_f64_probe.log" 2>&1
hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/kbench/group_barrier_probe.hip -o /tmp/group_barrier_probe 2>/dev/null && timeout 120 /tmp/group_barrier_probe 200 > "$O/group_barrier_probe.log" 2>&1
$CC scripts/kbench/gemm_bench.hip -o /tmp/gemm_bench 2>/dev/null && timeout 300 /tmp/gemm_bench 5 8 > "$O/gemm_bench_f64_contraction_sweep.log" 2>&1
ls -la "$O"
