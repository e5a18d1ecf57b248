mkdir -p gpurun_out/r05e
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > gpurun_out/r05e/smoke.log 2>&1 || { tail -30 gpurun_out/r05e/smoke.log; exit 1; }
tail -2 gpurun_out/r05e/smoke.log
timeout -k 10 1500 python -m pytest tests -m gpu -q --durations=8 > gpurun_out/r05e/gpu_tests.log 2>&1; echo "pytest rc $?"; tail -16 gpurun_out/r05e/gpu_tests.log
python bench.py --steps 60 --no-cpu-baseline > gpurun_out/r05e/c3_bench.json 2> gpurun_out/r05e/c3_bench.err
python - <<PY
import json
d=json.loads(open("gpurun_out/r05e/c3_bench.json").read().strip().splitlines()[-1])
print("c3", d["value"], d["ms_per_step"], "host", d["host_enqueue_ms_per_step"])
for k,v in d["extra_summary"].items(): print("  ", k, v)
PY
