#!/bin/bash
R="$(cd "$(dirname "$0")/.." && pwd)"; O="$R/gpurun_out/s3pmc"; mkdir -p "$O"; cd "$R"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I nmf.jl_amd/csrc scripts/kbench/potrs_bench.hip -o /tmp/pb 2>/dev/null
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d "$O/pmc" -o p -- /tmp/pb 3 2 > "$O/run.log" 2>&1
f=$(find "$O/pmc" -name '*counter_collection.csv' | head -1)
python "$R/scripts/pmc_summary.py" "$O/pmc" potrs_strip_kernel | tee "$O/summary.txt"
rm -rf "$O/pmc"
