#!/bin/bash
export NMFX_DEV=1   # the NMFX_* development switches below are honoured only with it (csrc/comm.hpp)
# Round-5 evidence run (GPU box): rocprofv3 summaries of the headline command, one bench line per config (each with its cpu_baseline),
# per-launch event tables, the simulated-rank timings (RCCL stand-in and peer-window launch sequence, A/B of the round's changes),
# bench.py --gpus N launching its own ranks on one device.  Everything lands under gpurun_out/r05p/.
R="$(cd "$(dirname "$0")/.." && pwd)"
O="$R/gpurun_out/r05p"; mkdir -p "$O"
cd "$R"
export GPU_MAX_HW_QUEUES=24
bash scripts/profile_bench.sh r05p/prof_multmse > "$O/prof_multmse.log" 2>&1
python bench.py --steps 20 --warmup 5 > "$O/driver_20_steps.json" 2>/dev/null
python bench.py --steps 20 --warmup 5 --prewarm-ms 0 --no-cpu-baseline > "$O/driver_20_steps_without_prewarm.json" 2>/dev/null
bash scripts/bench_configs.sh > "$O/bench_configs.jsonl" 2> "$O/bench_configs.err"
B="python bench.py --no-cpu-baseline"
NMFX_XT=0 $B --traffic none > "$O/multmse_without_transposed_images.json" 2>/dev/null
$B --all-events --traffic none > "$O/multmse_all_events.json" 2>/dev/null
$B --p 4096 --n 4096 --k 64 --steps 500 --warmup 50 --no-events > "$O/c2_no_events.json" 2>/dev/null
: > "$O/simranks.jsonl"
for g in 2 4 8; do for tr in rccl p2p; do
  $B --sim-ranks $g --steps 50 --no-events --transport $tr >> "$O/simranks.jsonl" 2>/dev/null
done; done
NMFX_W_BLOCKED=0 $B --sim-ranks 8 --steps 50 --no-events --transport rccl > "$O/simranks8_rccl_unpacked_every_iteration.json" 2>/dev/null
NMFX_XT=0 NMFX_W_BLOCKED=0 $B --sim-ranks 8 --steps 50 --no-events --transport rccl > "$O/simranks8_rccl_round4_products_and_unpack.json" 2>/dev/null
$B --sim-ranks 8 --steps 50 --all-events --transport rccl > "$O/simranks8_rccl_all_events.json" 2>/dev/null
$B --sim-ranks 8 --steps 50 --all-events --transport p2p > "$O/simranks8_p2p_all_events.json" 2>/dev/null
: > "$O/simranks8_c4_c5.jsonl"
$B --sim-ranks 8 --alg projals --p 16384 --n 131072 --k 256 --steps 10 --warmup 3 --no-events --transport rccl >> "$O/simranks8_c4_c5.jsonl" 2>/dev/null
$B --sim-ranks 8 --alg alspgrad --dtype f64 --p 32768 --n 32768 --k 512 --steps 2 --warmup 1 --transport rccl >> "$O/simranks8_c4_c5.jsonl" 2>/dev/null
# plain `python bench.py --gpus N` (no launcher in front): N ranks on ONE device over the peer windows -- plumbing, not a scaling number
NMFX_BENCH_BACKEND=gloo-p2p python bench.py --gpus 4 --steps 20 --warmup 5 --p 8192 --n 8192 --no-cpu-baseline > "$O/bench_gpus4_self_launched_one_gpu.json" 2> "$O/bench_gpus4.err"
NMFX_BENCH_BACKEND=gloo-p2p python bench.py --gpus 8 --steps 20 --warmup 5 --p 8192 --n 8192 --no-cpu-baseline > "$O/bench_gpus8_self_launched_one_gpu.json" 2> "$O/bench_gpus8.err"
python scripts/alspgrad_gradient_modes.py > "$O/alspgrad_gradient_modes.jsonl" 2>/dev/null
(hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I nmf.jl_amd/csrc scripts/kbench/gemm_bench.hip -o /tmp/gemm_bench 2>/dev/null; /tmp/gemm_bench 8 7) > "$O/gemm_bench_shard_shapes_4_vs_8_waves.log" 2>&1
ls -la "$O"
