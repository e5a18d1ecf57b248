#!/bin/bash
# per-kernel times of the row-sparse backward alone at R live rows: tools/gpu_rows_prof.sh R
root=${GRAFT_REPO_ROOT:-/root/repo}; out=$root/gpurun_out/r05n; mkdir -p $out
export MS_ROWS_ONE=1 MS_ROWS_R=$1
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_rows -o rows -- python3 $root/tools/ms_rows_bench.py > $out/prof_rows.log 2>&1 )
python3 - $out/prof_rows/*kernel_stats.csv <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "ms_rows" in r["Name"]:
        print("%-60s calls %5s avg %8.1f us" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
rm -rf $out/prof_rows
