#!/bin/bash
# same-box A/B: PRIFIT_POOL_ALG=0 against the default ("auto": the algebraic backward on SA2 scale 1 only) and "1" (all layers)
out=gpurun_out/r05r; mkdir -p $out
for rep in 1 2; do
  for v in 0 auto 1; do
    for w in c2 c3; do
      PRIFIT_POOL_ALG=$v python bench.py --workload $w --no-cpu-baseline --no-extra --steps 40 --warmup 8 > $out/pa_${v}_${w}_$rep.json 2> $out/pa_${v}_${w}_$rep.err
      echo "PRIFIT_POOL_ALG=$v $w $(python tools/fam_table.py $out/pa_${v}_${w}_$rep.json | head -1)"
    done
  done
done
