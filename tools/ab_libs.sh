#!/bin/bash
# same-box A/B of two builds of the library: tools/ab_libs.sh A.so B.so [bench args]   (alternates A B A B)
A=$1; B=$2; shift 2
for i in 1 2; do
  for L in $A $B; do
    cp $L prifit_amd/lib/libprifit_hip.so
    python bench.py --no-cpu-baseline --steps 100 "$@" > gpurun_out/ab.json && echo "$L $(python tools/fam_table.py gpurun_out/ab.json | head -1)"
  done
done
