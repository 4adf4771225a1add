mkdir -p gpurun_out/r04; export GPU_MAX_HW_QUEUES=24
timeout 1200 python -m pytest tests/test_gpu_k_granularity.py tests/test_gpu_utils.py tests/test_gpu_projals_alspgrad.py -x -q -m gpu > gpurun_out/r04/test_kgran.log 2>&1; tail -15 gpurun_out/r04/test_kgran.log
B="python bench.py --no-cpu-baseline --no-events"
for k in 160 192 256; do
$B --k $k > gpurun_out/r04/multmse_k$k.json 2>&1
python -c "import json; d=json.loads(open('gpurun_out/r04/multmse_k$k.json').read().strip().splitlines()[-1]); print('k=$k', d['ms_per_step'])"
done
NMFX_K_GRANULE=128 $B --k 160 > gpurun_out/r04/multmse_k160_old.json 2>&1; python -c "import json; d=json.loads(open('gpurun_out/r04/multmse_k160_old.json').read().strip().splitlines()[-1]); print('k=160 (K=256 padding)', d['ms_per_step'])"
for alg in projals cd greedycd multdiv; do $B --alg $alg --k 160 --steps 10 --warmup 3 > gpurun_out/r04/${alg}_k160.json 2>&1; NMFX_K_GRANULE=128 $B --alg $alg --k 160 --steps 10 --warmup 3 > gpurun_out/r04/${alg}_k160_old.json 2>&1; python -c "
import json
a=json.loads(open('gpurun_out/r04/${alg}_k160.json').read().strip().splitlines()[-1]); b=json.loads(open('gpurun_out/r04/${alg}_k160_old.json').read().strip().splitlines()[-1])
print('$alg k=160: K=192', a['ms_per_step'], ' K=256', b['ms_per_step'])"; done
