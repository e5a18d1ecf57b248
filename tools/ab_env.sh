#!/bin/bash
# usage (GPU box): tools/ab_env.sh <ENV_VAR> <workload> [reps]  -- same-box A/B of one environment switch (0 vs 1), alternating runs
var=$1; w=$2; reps=${3:-2}
root=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $root/gpurun_out/ab
for r in $(seq 1 $reps); do
  for v in 0 1; do
    env $var=$v python3 $root/bench.py --workload $w --steps 40 --warmup 10 --no-cpu-baseline --no-extra > $root/gpurun_out/ab/${var}_${w}_${v}_$r.json 2> $root/gpurun_out/ab/${var}_${w}_${v}_$r.err
    python3 - "$root/gpurun_out/ab/${var}_${w}_${v}_$r.json" "$var=$v $w" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
g = d.get("roofline_grouping") or {}
print(sys.argv[2], "shapes/s %.1f  ms/step %.3f  host %.2f  grouping ms %.3f" % (d["value"], d["ms_per_step"], d["host_enqueue_ms_per_step"], g.get("ms_per_step", 0)))
PY
  done
done
