#!/bin/bash
# usage (GPU box): tools/step_trace.sh <tag> <workload>  -- the kernel sequence of ONE timed step (between two optimizer launches),
# start offset / duration / name, into gpurun_out/<tag>/<workload>_step_trace.txt
tag=$1; w=$2
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/$tag
mkdir -p $out
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --output-format csv -d $out/trace_$w -o $w -- python3 $root/bench.py --workload $w --no-cpu-baseline --no-extra --steps 6 --warmup 3 > $out/trace_$w.log 2>&1 )
python3 - $out/trace_$w $out/${w}_step_trace.txt <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
opt = [i for i, r in enumerate(rows) if "FusedOptimizer" in r["Kernel_Name"]]
# the optimizer's launches come in a burst per step: take the span between the last two bursts
bursts = [i for k, i in enumerate(opt) if k == 0 or i - opt[k - 1] > 20]
a, b = bursts[-3], bursts[-2]
t0 = int(rows[a]["Start_Timestamp"])
with open(sys.argv[2], "w") as o:
    prev_end = t0
    for r in rows[a:b]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        o.write("%9.1f us  +%6.1f gap  %8.1f us  %s\n" % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3, r["Kernel_Name"][:150]))
        prev_end = max(prev_end, e)
    o.write("step: %.1f us, %d launches\n" % ((int(rows[b]["Start_Timestamp"]) - t0) / 1e3, b - a))
print(open(sys.argv[2]).read()[-200:])
PY
rm -rf $out/trace_$w
