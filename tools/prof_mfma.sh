#!/bin/bash
# usage (on the GPU box): tools/prof_mfma.sh <tag> [bench args…]
# One rocprofv3 --pmc pass (SQ_VALU_MFMA_BUSY_CYCLES, GRBM_GUI_ACTIVE) with the kernel trace over a short bench run;
# tools/pmc_mfma.py turns it into matrix-core utilisation per kernel family -> gpurun_out/mfma_<tag>/mfma.json
tag=$1; shift
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/mfma_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out/pmc -o mfma -- python3 $root/bench.py --no-cpu-baseline --steps 2 --warmup 1 "$@" > $out/run.log 2>&1
cd $root
python3 tools/pmc_mfma.py $out
