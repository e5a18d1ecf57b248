mkdir -p gpurun_out/r05c
python tools/gather_bwd_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05c/gather.txt
PAD=1 python tools/gather_bwd_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05c/gather_pad.txt
timeout -k 10 900 python -m pytest tests -m gpu -q -k "dgcnn or nms_with or gather or sa_ or backbone" > gpurun_out/r05c/gpu_tests.log 2>&1; tail -5 gpurun_out/r05c/gpu_tests.log
for i in 1 2; do python bench.py --workload c2 --steps 60 --no-cpu-baseline > gpurun_out/r05c/c2_bench_$i.json 2> gpurun_out/r05c/c2_bench_$i.err; python - <<PY
import json
d=json.loads(open("gpurun_out/r05c/c2_bench_$i.json").read().strip().splitlines()[-1])
print("c2", d["value"], d["ms_per_step"], "host", d["host_enqueue_ms_per_step"])
PY
done
python tools/host_profile.py > gpurun_out/r05c/host_profile.txt 2>&1; tail -45 gpurun_out/r05c/host_profile.txt
nproc; cat /proc/loadavg
