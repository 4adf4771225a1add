#!/bin/bash
# GPU box: build and run the greedy sweep bench on the stored state (scripts/kbench/data/greedy_state.bin, from greedy_state.py)
cd "$(dirname "$0")"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I../../nmf.jl_amd/csrc -I../../include greedy_bench.hip -o greedy_bench 2>/dev/null || { echo build failed; exit 1; }
./greedy_bench data/greedy_state.bin
