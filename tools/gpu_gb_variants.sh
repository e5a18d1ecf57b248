#!/bin/bash
L=prifit_amd/lib
cp $L/libprifit_hip.so $L/cur.so
for v in cur variants/gb16 variants/gb4; do
  cp $L/$v.so $L/libprifit_hip.so
  echo "== $v"; bash tools/kernel_stats.sh r05y c2 gather_bwd | grep gather_bwd
done
cp $L/cur.so $L/libprifit_hip.so; rm $L/cur.so
