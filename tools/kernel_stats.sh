#!/bin/bash
# usage (GPU box): tools/kernel_stats.sh <tag> <workload> [pattern]  -- rocprofv3 kernel statistics of one bench run into
# gpurun_out/<tag>/<workload>_kernel_stats.csv, and the rows matching `pattern` (default: the 30 largest) per step
tag=$1; w=$2; pat=${3:-}
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/$tag
mkdir -p $out
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_$w -o $w -- python3 $root/bench.py --workload $w --no-cpu-baseline --no-extra --steps 40 --warmup 3 > $out/prof_$w.log 2>&1 )
cp $out/prof_$w/*kernel_stats.csv $out/${w}_kernel_stats.csv 2>/dev/null || cp $out/prof_$w/*/*kernel_stats.csv $out/${w}_kernel_stats.csv
rm -rf $out/prof_$w
python3 - $out/${w}_kernel_stats.csv "$pat" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
pat = sys.argv[2]
# 3 warm-up + 40 timed + 5 of the host-time pass + 2 of the launch count (round 6); for c3 / c5 bench.py sends its first warm-up
# step through the speculative runner's fall-back once (forward + backward twice): one step-equivalent more
steps = 50.0 if "c2_kernel_stats" in sys.argv[1] else 51.0
tot = sum(int(r["TotalDurationNs"]) for r in rows)
print("kernel time %.3f ms/step, %.1f launches/step" % (tot / steps / 1e6, sum(int(r["Calls"]) for r in rows) / steps))
sel = [r for r in rows if pat and any(p in r["Name"] for p in pat.split(","))] if pat else rows[:30]
for r in sel:
    print("%-84s %5.1f/step %8.1f us %7.3f ms/step" % (r["Name"][:84], int(r["Calls"]) / steps, float(r["AverageNs"]) / 1e3,
                                                       int(r["TotalDurationNs"]) / steps / 1e6))
PY
