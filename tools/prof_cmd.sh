#!/bin/bash
# usage (on the GPU box): tools/prof_cmd.sh <tag> <script.py> [args…] -> per-kernel stats of any python script
tag=$1; shift
root=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/prof_$tag -o $tag -- python3 "$@" > $root/gpurun_out/prof_$tag.log 2>&1
cd $root
python3 - "$tag" <<'PY'
import csv, glob, sys
f = glob.glob("gpurun_out/prof_%s/**/*kernel_stats.csv" % sys.argv[1], recursive=True)
if not f:
    print(open("gpurun_out/prof_%s.log" % sys.argv[1]).read()[-3000:]); sys.exit(1)
for r in list(csv.DictReader(open(f[0])))[:30]:
    print("%-72s %6s %12s %10s %8s %8s" % (r["Name"][:72], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["MinNs"], r["MaxNs"]))
PY
