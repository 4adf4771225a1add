#!/bin/bash
# One bench.py line per BASELINE.json config (single-GPU form: C4 / C5 as the per-GPU shard of the 8-GPU problem).
# usage (on the GPU box): bash scripts/bench_configs.sh > gpurun_out/bench_configs.jsonl
set +e
cd "$(dirname "$0")/.."
python bench.py --p 4096 --n 4096 --k 64 --steps 200 --warmup 50                       # C2
python bench.py                                                                                           # C3 multmse (headline)
python bench.py --alg multdiv --steps 30 --warmup 10                                    # C3 multdiv
python bench.py --alg projals --steps 30 --warmup 10                                    # C4 per-GPU shard (16384 x 16384)
python bench.py --alg alspgrad --dtype f64 --p 32768 --n 4096 --k 512 --steps 3 --warmup 1   # C5 per-GPU shard, reference default maxsubiter = 200
python bench.py --no-cpu-baseline --dtype f64 --p 8192 --n 8192 --k 256 --steps 30 --warmup 10            # f64 multmse
python bench.py --no-cpu-baseline --alg cd --steps 30 --warmup 10                                         # CoordinateDescent at the C3 shape
python bench.py --no-cpu-baseline --alg greedycd --steps 20 --warmup 10                                   # GreedyCD (nnmf's default algorithm) at the C3 shape
python bench.py --no-cpu-baseline --precision bf16x3                                                        # OPT-IN mixed-precision big GEMMs (not the headline)
python bench.py --no-cpu-baseline --precision bf16x3 --alg multdiv --steps 30 --warmup 10                  # OPT-IN, multdiv
