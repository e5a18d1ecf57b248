#!/bin/bash
# per-kernel times of tools/pool_alg_bench.py (the algebraic backward's passes alone)
root=${GRAFT_REPO_ROOT:-/root/repo}; out=$root/gpurun_out/r05r; mkdir -p $out
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_pa -o pa -- python3 $root/tools/pool_alg_bench.py > $out/prof_pa.log 2>&1 )
python3 - $out/prof_pa/*kernel_stats.csv <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "pool_alg" in r["Name"] or "slab_sum" in r["Name"]:
        print("%-90s calls %5s avg %8.1f us min %8.1f max %8.1f" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
rm -rf $out/prof_pa
