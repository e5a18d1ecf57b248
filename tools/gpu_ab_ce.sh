#!/bin/bash
# same-box A/B of the segmentation loss kernel (c2): PRIFIT_CE_KERNEL=0 -> torch's F.cross_entropy
out=gpurun_out/r05u; mkdir -p $out
for rep in 1 2; do for v in 0 1; do
  PRIFIT_CE_KERNEL=$v python bench.py --workload c2 --no-cpu-baseline --no-extra --steps 40 --warmup 8 > $out/ce_${v}_$rep.json 2> $out/ce_${v}_$rep.err
  echo "PRIFIT_CE_KERNEL=$v c2 $(python tools/fam_table.py $out/ce_${v}_$rep.json | head -1)"
done; done
