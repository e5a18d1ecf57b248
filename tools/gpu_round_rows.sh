#!/bin/bash
# A/B of the row-sparse mean-shift backward (tables + one dX pass) against the library built before it
set -e
out=gpurun_out/r05n; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_fit.py -q -x -k "row_sparse or both_mean_shift or convex_loss_end or center or selfsup" > $out/tests.log 2>&1 || { tail -30 $out/tests.log; exit 1; }
tail -3 $out/tests.log
L=prifit_amd/lib
cp $L/libprifit_hip.so $L/new.so
for rep in 1 2; do
  for v in old new; do
    if [ $v = old ]; then cp $L/libprifit_hip_old.so $L/libprifit_hip.so; else cp $L/new.so $L/libprifit_hip.so; fi
    for w in c3 c5; do
      timeout -k 10 300 python bench.py --workload $w --no-cpu-baseline --no-extra --steps 30 --warmup 5 > $out/${v}_${w}_$rep.json 2> $out/${v}_${w}_$rep.err
      python - $out/${v}_${w}_$rep.json $v $w <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], sys.argv[3], "%.3f ms/step"%d["ms_per_step"])
PY
    done
  done
done
cp $L/new.so $L/libprifit_hip.so
python tools/fam_table.py $out/new_c3_2.json > $out/c3_table.txt; python tools/fam_table.py $out/old_c3_2.json > $out/c3_table_old.txt
grep -E "^ms/step|ms_rows" $out/c3_table.txt $out/c3_table_old.txt || true
