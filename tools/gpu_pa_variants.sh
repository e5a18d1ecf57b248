#!/bin/bash
L=prifit_amd/lib
cp $L/libprifit_hip.so $L/cur.so
for v in cur variants/pa_4_1 variants/pa_4_4 variants/pa_8_2 variants/pa_4_8; do
  cp $L/$v.so $L/libprifit_hip.so
  echo "== $v"; timeout -k 10 200 python tools/pool_alg_bench.py 2>&1 | grep "winners" | sed 's/.*own launches://' | tr '\n' ' '; echo
done
cp $L/cur.so $L/libprifit_hip.so; rm $L/cur.so
