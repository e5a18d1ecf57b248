mkdir -p gpurun_out/r05j
timeout -k 10 600 python -m pytest tests/test_gpu_backbone.py -q -x -k "algebraic" 2>&1 | grep -v amdgpu.ids | tail -3 || exit 1
L=prifit_amd/lib/libprifit_hip.so
cp $L /tmp/new.so
for v in new palg1 palg3; do
  if [ $v = new ]; then cp /tmp/new.so $L; else cp prifit_amd/lib/variants/$v.so $L; fi
  echo "== lib $v"; timeout -k 10 200 python tools/pool_alg_bench.py 2>&1 | grep -E "winners|Error|error|assert" 
done 2>&1 | tee gpurun_out/r05j/palg_probe.txt
cp /tmp/new.so $L
bash tools/ab_env.sh PRIFIT_POOL_ALG c2 2 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05j/ab_pool_alg_c2.txt
PRIFIT_SPAN_SHAPES=1 PRIFIT_BENCH_EVENTS=all python3 bench.py --workload c2 --no-cpu-baseline --no-extra --steps 20 --warmup 6 > gpurun_out/r05j/c2_shapes.json 2> /dev/null
python3 tools/fam_table.py gpurun_out/r05j/c2_shapes.json > gpurun_out/r05j/c2_table.txt; grep -E "^pool_alg|^ms/step" gpurun_out/r05j/c2_table.txt
