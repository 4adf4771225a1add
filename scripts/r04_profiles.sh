#!/bin/bash
export NMFX_DEV=1   # the NMFX_* development switches below are honoured only with it (csrc/comm.hpp)
# Round-4 evidence run (GPU box): rocprofv3 summaries of the headline and the multdiv command, one bench line per config, per-launch
# event tables, the simulated-rank timings (RCCL stand-in and peer-window launch sequence), the multi-process bench on one GPU.
# Everything lands under gpurun_out/r04p/.
R="$(cd "$(dirname "$0")/.." && pwd)"
O="$R/gpurun_out/r04p"; mkdir -p "$O"
cd "$R"
export GPU_MAX_HW_QUEUES=24
bash scripts/profile_bench.sh r04p/prof_multmse > "$O/prof_multmse.log" 2>&1
BENCH_ARGS="--alg multdiv --steps 30 --warmup 10" bash scripts/profile_bench.sh r04p/prof_multdiv > "$O/prof_multdiv.log" 2>&1
python bench.py --steps 20 --warmup 5 > "$O/driver_20_steps.json" 2>/dev/null
bash scripts/bench_configs.sh > "$O/bench_configs.jsonl" 2> "$O/bench_configs.err"
B="python bench.py --no-cpu-baseline"
$B --p 4096 --n 4096 --k 64 --steps 200 --warmup 50 --all-events > "$O/c2_all_events.json" 2>/dev/null
$B --p 4096 --n 4096 --k 64 --steps 500 --warmup 50 --no-events > "$O/c2_no_events.json" 2>/dev/null
$B --alg multdiv --steps 30 --warmup 10 --all-events > "$O/multdiv_all_events.json" 2>/dev/null
$B --alg projals --steps 30 --warmup 10 --all-events > "$O/projals_all_events.json" 2>/dev/null
$B --alg projals --steps 30 --warmup 10 --no-events > "$O/projals_no_events.json" 2>/dev/null
$B --alg greedycd --steps 20 --warmup 10 --all-events > "$O/greedycd_all_events.json" 2>/dev/null
$B --k 160 --no-events > "$O/multmse_k160.json" 2>/dev/null
NMFX_K_GRANULE=128 $B --k 160 --no-events > "$O/multmse_k160_padded_to_256.json" 2>/dev/null
: > "$O/simranks.jsonl"
for g in 2 4 8; do for tr in rccl p2p; do
  $B --sim-ranks $g --steps 50 --no-events --transport $tr >> "$O/simranks.jsonl" 2>/dev/null
done; done
$B --sim-ranks 8 --steps 50 --all-events --transport rccl > "$O/simranks8_rccl_all_events.json" 2>/dev/null
$B --sim-ranks 8 --steps 50 --all-events --transport p2p > "$O/simranks8_p2p_all_events.json" 2>/dev/null
: > "$O/simranks8_c4_c5.jsonl"
$B --sim-ranks 8 --alg projals --p 16384 --n 131072 --k 256 --steps 10 --warmup 3 --no-events --transport rccl >> "$O/simranks8_c4_c5.jsonl" 2>/dev/null
NMFX_RS_FUSED=0 $B --sim-ranks 8 --alg projals --p 16384 --n 131072 --k 256 --steps 10 --warmup 3 --no-events --transport rccl >> "$O/simranks8_c4_c5.jsonl" 2>/dev/null
$B --sim-ranks 8 --alg projals --p 16384 --n 131072 --k 256 --steps 10 --warmup 3 --no-events --transport p2p >> "$O/simranks8_c4_c5.jsonl" 2>/dev/null
$B --sim-ranks 8 --alg alspgrad --dtype f64 --p 32768 --n 32768 --k 512 --steps 2 --warmup 1 --transport rccl >> "$O/simranks8_c4_c5.jsonl" 2>/dev/null
# bench.py --gpus N as N processes on ONE device over the peer windows (plumbing + protocol check, not a scaling number)
scripts/bench_multiproc_1gpu.sh 4 --steps 20 --warmup 5 --p 8192 --n 8192 --no-cpu-baseline > "$O/bench_4proc_one_gpu.json" 2> "$O/bench_4proc_one_gpu.err"
scripts/bench_multiproc_1gpu.sh 8 --steps 20 --warmup 5 --p 8192 --n 8192 --no-cpu-baseline > "$O/bench_8proc_one_gpu.json" 2> "$O/bench_8proc_one_gpu.err"
for tiny in 1 0; do NMFX_P2P_TINY=$tiny scripts/bench_multiproc_1gpu.sh 2 --alg alspgrad --dtype f64 --p 8192 --n 8192 --k 256 --steps 2 --warmup 1 --no-cpu-baseline --no-events > "$O/alspgrad_2proc_tiny$tiny.json" 2>/dev/null; done
(cd scripts/kbench && hipcc --offload-arch=gfx950 -O3 -std=c++17 ipc_probe.hip -o ipc_probe 2>/dev/null; for n in 2 4 8; do timeout 100 ./ipc_probe $n 0; done) > "$O/ipc_probe.log" 2>&1
# round 4, second half: GreedyCD's sweep (kernel bench on a stored state when present), instruction issue rates, Float64 trial-step shapes, factorisations stand-alone
(cd scripts/kbench && hipcc --offload-arch=gfx950 -O3 valu_rate_probe.hip -o valu_rate_probe 2>/dev/null && ./valu_rate_probe) > "$O/valu_rate_probe.log" 2>&1
if [ ! -f scripts/kbench/data/greedy_state.bin ]; then mkdir -p scripts/kbench/data; python scripts/kbench/greedy_state.py scripts/kbench/data/greedy_state.bin 12 > "$O/greedy_state.log" 2>&1; fi
scripts/kbench/run_greedy_bench.sh > "$O/greedy_bench.log" 2>&1
(hipcc --offload-arch=gfx950 -O3 -std=c++17 -I nmf.jl_amd/csrc scripts/kbench/gemm_bench.hip -o scripts/kbench/gemm_bench 2>/dev/null; scripts/kbench/gemm_bench 6 5) > "$O/gemm_bench_f64_trial_step_shapes.log" 2>&1
(hipcc --offload-arch=gfx950 -O3 -std=c++17 -I nmf.jl_amd/csrc scripts/kbench/potrf_bench.hip -o /tmp/potrf_bench 2>/dev/null; /tmp/potrf_bench) > "$O/potrf_bench_standalone.log" 2>&1
python scripts/fixed_overhead_probe.py 2>&1 | grep -v amdgpu.ids > "$O/iterate_call_overhead.log"
ls -la "$O"
