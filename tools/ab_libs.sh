#!/bin/bash
# same-box A/B of two builds of the library: tools/ab_libs.sh A.so B.so [bench args]   (alternates A B A B)
# The build under test is chosen with PRIFIT_LIB (prifit_amd/_lib.py): the product library is never overwritten.
A=$1; B=$2; shift 2
for i in 1 2; do
  for L in $A $B; do
    PRIFIT_LIB=$(realpath $L) python bench.py --no-cpu-baseline --no-extra --steps 100 "$@" > gpurun_out/ab.json && echo "$L $(python tools/fam_table.py gpurun_out/ab.json | head -1)"
  done
done
