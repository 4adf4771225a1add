mkdir -p gpurun_out/r04; export GPU_MAX_HW_QUEUES=24
timeout 2800 python -m pytest tests/ -x -q -m gpu > gpurun_out/r04/test_all_gpu2.log 2>&1; tail -4 gpurun_out/r04/test_all_gpu2.log
B="python bench.py --no-cpu-baseline"
$B > gpurun_out/r04/bench_multmse_50.json 2>&1
$B --steps 20 --warmup 5 > gpurun_out/r04/bench_multmse_20.json 2>&1
$B --sim-ranks 8 --no-events --transport rccl > gpurun_out/r04/sim8_rccl_buf.json 2>&1
$B --sim-ranks 8 --no-events --transport p2p > gpurun_out/r04/sim8_p2p_buf.json 2>&1
$B --alg multdiv > gpurun_out/r04/bench_multdiv.json 2>&1
$B --alg projals --no-events > gpurun_out/r04/bench_projals_noev.json 2>&1
$B --alg alspgrad --dtype f64 --p 32768 --n 4096 --k 512 --steps 2 --warmup 1 > gpurun_out/r04/bench_alspgrad_c5shard.json 2>&1
$B --dtype f64 --p 8192 --n 8192 > gpurun_out/r04/bench_multmse_f64.json 2>&1
for f in bench_multmse_50 bench_multmse_20 sim8_rccl_buf sim8_p2p_buf bench_multdiv bench_projals_noev bench_alspgrad_c5shard bench_multmse_f64; do python -c "
import json
d=json.loads(open('gpurun_out/r04/$f.json').read().strip().splitlines()[-1])
print('$f', d['ms_per_step'], d.get('ms_per_step_no_events'), d['frac_of_mfma_peak'], (d.get('roofline') or {}).get('frac'), [(k['name'],k['avg_us']) for k in d['kernels'][:3]])"; done
