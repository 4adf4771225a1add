mkdir -p gpurun_out/r04; export GPU_MAX_HW_QUEUES=24
(cd scripts/kbench && hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I ../../nmf.jl_amd/csrc gemm_bench.hip -o gemm_bench 2>/dev/null && ./gemm_bench 6 1 > ../../gpurun_out/r04/gemm_prio.log 2>&1)
timeout 1500 python -m pytest tests/test_gpu_utils.py tests/test_gpu_projals_alspgrad.py tests/test_gpu_localcomm.py tests/test_gpu_comm.py tests/test_gpu_peer.py tests/test_gpu_track_stop.py -x -q -m gpu > gpurun_out/r04/test_projals.log 2>&1; tail -5 gpurun_out/r04/test_projals.log
B="python bench.py --no-cpu-baseline"
$B --alg projals --no-events > gpurun_out/r04/projals_c3_noev.json 2>&1
NMFX_POTRS=0 $B --alg projals --no-events > gpurun_out/r04/projals_c3_noev_nopotrs.json 2>&1
$B --alg projals --all-events > gpurun_out/r04/projals_c3_all.json 2>&1
$B --alg projals > gpurun_out/r04/projals_c3_sampled.json 2>&1
$B --alg projals --n 131072 --sim-ranks 8 --no-events --steps 10 --warmup 3 --transport rccl > gpurun_out/r04/sim8_projals_c4_new.json 2>&1
$B --alg projals --n 131072 --sim-ranks 8 --all-events --steps 10 --warmup 3 --transport rccl > gpurun_out/r04/sim8_projals_c4_new_all.json 2>&1
for f in projals_c3_noev projals_c3_noev_nopotrs projals_c3_sampled sim8_projals_c4_new; do python -c "import json,sys; d=json.loads(open(\"gpurun_out/r04/$f.json\").read().strip().splitlines()[-1]); print(\"$f\", d[\"ms_per_step\"], d[\"frac_of_mfma_peak\"])"; done
grep -v "^---" gpurun_out/r04/gemm_prio.log | sort | head -80
