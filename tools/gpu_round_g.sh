mkdir -p gpurun_out/r05g
for e in clustered25; do
PRIFIT_SPAN_SHAPES=1 PRIFIT_BENCH_EVENTS=all python3 bench.py --embedding $e --no-cpu-baseline --no-extra --steps 20 --warmup 6 > gpurun_out/r05g/${e}_shapes.json 2> gpurun_out/r05g/${e}.err
python3 tools/fam_table.py gpurun_out/r05g/${e}_shapes.json > gpurun_out/r05g/${e}_table.txt
python3 bench.py --embedding $e --no-cpu-baseline --no-extra --steps 60 > gpurun_out/r05g/${e}_bench.json 2>> gpurun_out/r05g/${e}.err
python3 - <<PY
import json
d=json.loads(open("gpurun_out/r05g/${e}_bench.json").read().strip().splitlines()[-1])
print("$e", d["value"], d["ms_per_step"], "host", d["host_enqueue_ms_per_step"], "fallbacks", d["speculation_fallbacks"], d["config"].get("clusters_per_shape"))
PY
head -12 gpurun_out/r05g/${e}_table.txt; grep -E "^(nms|kth|sample|ellips|sdf|membership|ms_rows)" gpurun_out/r05g/${e}_table.txt
done
