#!/bin/bash
out=gpurun_out/r05r; mkdir -p $out
for v in 0 1; do
  for w in c3 c2; do
    PRIFIT_POOL_ALG=$v PRIFIT_SPAN_SHAPES=1 PRIFIT_BENCH_EVENTS=all python3 bench.py --workload $w --no-cpu-baseline --no-extra --steps 20 --warmup 6 > $out/tab_${v}_$w.json 2>/dev/null
    python3 tools/fam_table.py $out/tab_${v}_$w.json > $out/tab_${v}_$w.txt
    echo "$v $w $(head -1 $out/tab_${v}_$w.txt)"
  done
done
