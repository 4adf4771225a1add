mkdir -p gpurun_out/r04; export GPU_MAX_HW_QUEUES=24
timeout 900 python -m pytest tests/test_gpu_utils.py tests/test_gpu_projals_alspgrad.py -x -q -m gpu > gpurun_out/r04/test_projals2.log 2>&1; tail -3 gpurun_out/r04/test_projals2.log
B="python bench.py --no-cpu-baseline"
for i in 1 2; do
$B --alg projals --no-events > gpurun_out/r04/projals_c3_noev_$i.json 2>&1
NMFX_POTRS=0 $B --alg projals --no-events > gpurun_out/r04/projals_c3_noev_nopotrs_$i.json 2>&1
done
$B --alg projals --all-events > gpurun_out/r04/projals_c3_all.json 2>&1
$B --alg projals --dtype f64 --p 8192 --n 8192 --k 128 --all-events > gpurun_out/r04/projals_f64_all.json 2>&1
NMFX_POTRS=0 $B --alg projals --dtype f64 --p 8192 --n 8192 --k 128 --all-events > gpurun_out/r04/projals_f64_all_nopotrs.json 2>&1
for f in projals_c3_noev_1 projals_c3_noev_nopotrs_1 projals_c3_noev_2 projals_c3_noev_nopotrs_2; do python -c "import json,sys; d=json.loads(open(\"gpurun_out/r04/$f.json\").read().strip().splitlines()[-1]); print(\"$f\", d[\"ms_per_step\"], d[\"frac_of_mfma_peak\"])"; done
for f in projals_c3_all projals_f64_all projals_f64_all_nopotrs; do python -c "
import json
d=json.loads(open('gpurun_out/r04/$f.json').read().strip().splitlines()[-1])
print('$f', d['ms_per_step'], [(k['name'],k['avg_us']) for k in d['kernels'] if k['name'] in ('potrs_clampH','gemm_UinvtB','gemm_UinvY_clampH','trtri_WtW')])"; done
python scripts/projals_f32_error.py 2>&1 | tail -8
