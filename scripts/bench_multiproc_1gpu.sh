#!/bin/bash
# Development aid (1-GPU box): bench.py --gpus N as N PROCESSES that all use device 0, gloo rendezvous, the real peer-window exchange
# between the processes (NMFX_BENCH_BACKEND=gloo-p2p).  Timings are NOT scaling numbers (the ranks share one GPU); what this checks is
# the multi-process code path of bench.py: handle exchange, verification, timed region, consistency check, the JSON line.
#   scripts/bench_multiproc_1gpu.sh 4 --steps 20 --warmup 5 --p 4096 --n 4096
N=$1; shift
export NMFX_BENCH_BACKEND=gloo-p2p NMFX_BENCH_DEVICE=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=${MASTER_PORT:-29541} WORLD_SIZE=$N
pids=()
for ((r = 1; r < N; r++)); do
    RANK=$r LOCAL_RANK=$r python bench.py --gpus $N "$@" > /dev/null 2> gpurun_out/bench_mp_rank$r.err &
    pids+=($!)
done
RANK=0 LOCAL_RANK=0 python bench.py --gpus $N "$@"
rc=$?
for p in "${pids[@]}"; do wait $p || rc=$?; done
exit $rc
