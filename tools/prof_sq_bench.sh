#!/bin/bash
# usage (on the GPU box): tools/prof_sq_bench.sh <tag> [bench args…]   SQ counter passes over a short bench run
tag=$1; shift
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/sq_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out/p1 -o p1 -- python3 $root/bench.py --no-cpu-baseline --steps 2 --warmup 1 "$@" > $out/p1.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS --kernel-trace --output-format csv -d $out/p2 -o p2 -- python3 $root/bench.py --no-cpu-baseline --steps 2 --warmup 1 "$@" > $out/p2.log 2>&1
cd $root
python3 tools/pmc_sq.py $out
