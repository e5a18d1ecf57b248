mkdir -p gpurun_out/r05i
timeout -k 10 600 python -m pytest tests/test_gpu_backbone.py -q -x -k "algebraic" 2>&1 | grep -v amdgpu.ids | tail -4 || exit 1
bash tools/ab_env.sh PRIFIT_POOL_ALG c2 2 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05i/ab_pool_alg_c2.txt
PRIFIT_SPAN_SHAPES=1 PRIFIT_BENCH_EVENTS=all python3 bench.py --workload c2 --no-cpu-baseline --no-extra --steps 20 --warmup 6 > gpurun_out/r05i/c2_shapes.json 2> /dev/null
python3 tools/fam_table.py gpurun_out/r05i/c2_shapes.json > gpurun_out/r05i/c2_table.txt; grep -E "pool_alg|ms/step" gpurun_out/r05i/c2_table.txt
