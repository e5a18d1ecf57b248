#!/bin/bash
# usage (on the GPU box): tools/prof_stats.sh <tag> [bench args…]  -> gpurun_out/prof_<tag>/, prints top kernels
tag=$1; shift
root=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/prof_$tag -o $tag -- python3 $root/bench.py --no-cpu-baseline --steps 5 --warmup 3 "$@" > $root/gpurun_out/prof_$tag.log 2>&1
cd $root
python3 - "$tag" <<'PY'
import csv, glob, sys
f = glob.glob("gpurun_out/prof_%s/**/*kernel_stats.csv" % sys.argv[1], recursive=True)
if not f:
    print(open("gpurun_out/prof_%s.log" % sys.argv[1]).read()[-3000:]); sys.exit(1)
for r in list(csv.DictReader(open(f[0])))[:40]:
    print("%-72s %6s %12s %10s %6s" % (r["Name"][:72], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"]))
PY
