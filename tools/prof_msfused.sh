#!/bin/bash
# usage (on the GPU box): tools/prof_msfused.sh <tag>
# SQ counter passes over the fused mean-shift micro-benchmark (tools/msfused_bench.py): where do the waves spend their
# cycles (parked on s_waitcnt / barriers, issue stalls, active) -> gpurun_out/msf_<tag>/
tag=$1
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/msf_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $out/counters.txt 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out/p1 -o p1 -- python3 $root/tools/msfused_bench.py > $out/p1.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU --kernel-trace --output-format csv -d $out/p2 -o p2 -- python3 $root/tools/msfused_bench.py > $out/p2.log 2>&1
cd $root
python3 tools/pmc_sq.py $out
