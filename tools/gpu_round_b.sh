mkdir -p gpurun_out/r05b
L=prifit_amd/lib/libprifit_hip.so
cp $L /tmp/new.so
for v in new r04like new r04like; do
  if [ $v = new ]; then cp /tmp/new.so $L; else cp prifit_amd/lib/variants/r04like.so $L; fi
  echo "== lib $v"; python tools/gather_bwd_bench.py 2>&1 | grep -v amdgpu.ids; python tools/chord_bench.py 2>&1 | grep -v amdgpu.ids
done > gpurun_out/r05b/ab_kernels.txt 2>&1
cp /tmp/new.so $L
cat gpurun_out/r05b/ab_kernels.txt
PAD=1 python tools/gather_bwd_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05b/gather_pad.txt
timeout -k 10 900 python -m pytest tests -m gpu -q -k "dgcnn or nms_with or col_sum or gather or sa_ or chord or fit" > gpurun_out/r05b/gpu_tests.log 2>&1; tail -8 gpurun_out/r05b/gpu_tests.log
for w in c5 c2; do python bench.py --workload $w --steps 60 --no-cpu-baseline > gpurun_out/r05b/${w}_bench.json 2> gpurun_out/r05b/${w}_bench.err; python - <<PY
import json
d=json.loads(open("gpurun_out/r05b/${w}_bench.json").read().strip().splitlines()[-1])
print("$w", d["value"], d["ms_per_step"], "host", d["host_enqueue_ms_per_step"])
PY
done
for w in c2 c3; do python bench.py --workload $w --steps 60 --no-cpu-baseline --no-extra --graph > gpurun_out/r05b/${w}_graph_bench.json 2> gpurun_out/r05b/${w}_graph_bench.err; python - <<PY
import json
d=json.loads(open("gpurun_out/r05b/${w}_graph_bench.json").read().strip().splitlines()[-1])
print("$w --graph", d["value"], d["ms_per_step"], "host", d["host_enqueue_ms_per_step"])
PY
done
python bench.py --steps 60 --no-cpu-baseline --no-extra > gpurun_out/r05b/c3_bench.json 2> gpurun_out/r05b/c3_bench.err; python - <<PY
import json
d=json.loads(open("gpurun_out/r05b/c3_bench.json").read().strip().splitlines()[-1])
print("c3", d["value"], d["ms_per_step"], "host", d["host_enqueue_ms_per_step"])
PY
