#!/bin/bash
out=gpurun_out/r05y; mkdir -p $out
for rep in 1 2; do for v in 0 1; do for w in c2 c3; do
  PRIFIT_GATHER_BWD_CSR=$v python bench.py --workload $w --no-cpu-baseline --no-extra --steps 40 --warmup 8 > $out/g_${v}_${w}_$rep.json 2> $out/g_${v}_${w}_$rep.err
  echo "PRIFIT_GATHER_BWD_CSR=$v $w $(python tools/fam_table.py $out/g_${v}_${w}_$rep.json | head -1)"
done; done; done
