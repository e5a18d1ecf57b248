#!/bin/bash
# usage (on the GPU box): tools/prof_pmc.sh <tag> [bench args…]
# Two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over a short bench run, then
# tools/pmc_traffic.py turns the per-dispatch counters into HBM bytes per launch for every GEMM family
# -> gpurun_out/pmc_<tag>/traffic.json (copy to profiles/ to have bench.py report roofline.traffic).
tag=$1; shift
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/pmc_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $out/$c -o $c -- python3 $root/bench.py --no-cpu-baseline --steps 2 --warmup 1 "$@" > $out/$c.log 2>&1
done
cd $root
python3 tools/pmc_traffic.py $out
