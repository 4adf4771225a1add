mkdir -p gpurun_out/r04; export GPU_MAX_HW_QUEUES=24
timeout 900 python -m pytest tests/test_gpu_utils.py tests/test_gpu_projals_alspgrad.py -x -q -m gpu > gpurun_out/r04/test_projals3.log 2>&1; tail -3 gpurun_out/r04/test_projals3.log
B="python bench.py --no-cpu-baseline"
for i in 1 2; do
$B --alg projals --n 131072 --sim-ranks 8 --no-events --steps 10 --warmup 3 --transport rccl > gpurun_out/r04/sim8_c4_fused_$i.json 2>&1
NMFX_RS_FUSED=0 $B --alg projals --n 131072 --sim-ranks 8 --no-events --steps 10 --warmup 3 --transport rccl > gpurun_out/r04/sim8_c4_unfused_$i.json 2>&1
done
$B --alg projals --n 131072 --sim-ranks 8 --no-events --steps 10 --warmup 3 --transport p2p > gpurun_out/r04/sim8_c4_fused_p2p.json 2>&1
for f in sim8_c4_fused_1 sim8_c4_unfused_1 sim8_c4_fused_2 sim8_c4_unfused_2 sim8_c4_fused_p2p; do python -c "import json,sys; d=json.loads(open(\"gpurun_out/r04/$f.json\").read().strip().splitlines()[-1]); print(\"$f\", d[\"ms_per_step\"])"; done
# ALSPGrad line-search scalars: inside the decision kernels vs as window all-reduces, 2 processes on one GPU
for tiny in 1 0; do NMFX_P2P_TINY=$tiny scripts/bench_multiproc_1gpu.sh 2 --alg alspgrad --dtype f64 --p 8192 --n 8192 --k 256 --steps 2 --warmup 1 --no-cpu-baseline --no-events > gpurun_out/r04/alspgrad_2proc_tiny$tiny.json 2> gpurun_out/r04/alspgrad_2proc_tiny$tiny.err; python -c "import json; d=json.loads(open('gpurun_out/r04/alspgrad_2proc_tiny$tiny.json').read().strip().splitlines()[-1]); print('tiny=$tiny', d['ms_per_step'], d.get('inner_iters_per_step'), d.get('backtracks_per_step'), d.get('multi_gpu_consistency'))"; done
