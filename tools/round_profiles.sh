#!/bin/bash
# usage (GPU box): tools/round_profiles.sh <tag>  -- everything profiles/ keeps for a round, into gpurun_out/<tag>/
# (bench lines of c3 / c2 / c5, rocprofv3 kernel statistics of each, shape tables, PMC traffic + MFMA passes of c3, the launch
# census, one B = 24 pass of the CPU baseline).  Progress lines go to stdout (a silent long run is taken to be hung).
tag=${1:-round}
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/$tag
mkdir -p $out
cd $root
say() { echo "[$(date +%H:%M:%S)] $*"; }
say "bench c3 (headline: 200 steps, extras, cpu baseline)"
python3 bench.py > $out/c3_bench.json 2> $out/c3_bench.err
say "bench c3 with the driver's arguments"
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/c3_driver_args_bench.json 2> $out/c3_driver_args_bench.err
for w in c2 c5; do
  say "bench $w"
  python3 bench.py --workload $w > $out/${w}_bench.json 2> $out/${w}_bench.err
done
for w in c3 c2 c5; do
  say "rocprofv3 kernel stats $w"
  ( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_$w -o $w -- python3 $root/bench.py --workload $w --no-cpu-baseline --no-extra --steps 40 --warmup 3 > $out/prof_$w.log 2>&1 )
  cp $out/prof_$w/*kernel_stats.csv $out/${w}_kernel_stats.csv 2>/dev/null || cp $out/prof_$w/*/*kernel_stats.csv $out/${w}_kernel_stats.csv
  rm -rf $out/prof_$w
  say "shape table $w"
  PRIFIT_SPAN_SHAPES=1 PRIFIT_BENCH_EVENTS=all python3 bench.py --workload $w --no-cpu-baseline --no-extra --steps 20 --warmup 6 > $out/${w}_shapes.json 2> /dev/null
  python3 tools/fam_table.py $out/${w}_shapes.json > $out/${w}_shapes_table.txt
done
say "PMC traffic (FETCH_SIZE / WRITE_SIZE passes) c3"
bash tools/prof_pmc.sh ${tag}_c3 --no-extra > $out/pmc_traffic.log 2>&1
cp gpurun_out/pmc_${tag}_c3/traffic.json $out/pmc_traffic.json; rm -rf gpurun_out/pmc_${tag}_c3
say "PMC traffic c5"
bash tools/prof_pmc.sh ${tag}_c5 --no-extra --workload c5 > $out/pmc_traffic_c5.log 2>&1
cp gpurun_out/pmc_${tag}_c5/traffic.json $out/pmc_traffic_c5.json; rm -rf gpurun_out/pmc_${tag}_c5
say "PMC MFMA busy c3"
bash tools/prof_mfma.sh ${tag}_c3 --no-extra > $out/pmc_mfma.log 2>&1
cp gpurun_out/mfma_${tag}_c3/mfma.json $out/pmc_mfma.json; rm -rf gpurun_out/mfma_${tag}_c3
say "SQ / TCC counter passes c3 (tools/prof_sq_bench.sh + two more passes) -> per-kernel summaries"
bash tools/prof_sq_bench.sh ${tag} --no-extra > $out/sq_summary.txt 2>&1
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INST_CYCLES_VMEM --kernel-trace --output-format csv -d $root/gpurun_out/sq_${tag}/p3 -o p3 -- python3 $root/bench.py --no-cpu-baseline --no-extra --steps 2 --warmup 1 > $root/gpurun_out/sq_${tag}/p3.log 2>&1
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCP_TCC_READ_REQ_sum --kernel-trace --output-format csv -d $root/gpurun_out/sq_${tag}/p4 -o p4 -- python3 $root/bench.py --no-cpu-baseline --no-extra --steps 2 --warmup 1 > $root/gpurun_out/sq_${tag}/p4.log 2>&1 )
python3 tools/pmc_kernels.py gpurun_out/sq_${tag} $out/sq_stream.json "gemm_stream_kernel,gemm_stream_bwd_kernel,gemm_stream_tn_kernel" > $out/sq_stream.txt 2>&1
python3 tools/pmc_kernels.py gpurun_out/sq_${tag} $out/sq_all.json "gemm,ms_fused,sa_group,pool_bwd,bn_relu,chord,kth,sample_nn" > /dev/null 2>&1
rm -rf gpurun_out/sq_${tag}
say "host time by phase (tools/host_split.py)"
for w in c3 c5 c2; do python3 tools/host_split.py $w 2>/dev/null | grep -v amdgpu.ids >> $out/host_split.txt; done
say "launch census c3 / c5"
PRIFIT_BENCH_CENSUS=$out/c3_launch_census.txt python3 bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-extra > /dev/null 2>&1
PRIFIT_BENCH_CENSUS=$out/c5_launch_census.txt python3 bench.py --workload c5 --steps 5 --warmup 3 --no-cpu-baseline --no-extra > /dev/null 2>&1
say "CPU baseline, one B = 24 sample"
python3 bench.py --steps 5 --warmup 3 --no-extra --cpu-baseline-shapes 24 > $out/c3_cpu_baseline_b24.json 2> /dev/null
say "done"
ls -la $out | tail -30
