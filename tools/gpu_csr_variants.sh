#!/bin/bash
L=prifit_amd/lib
cp $L/libprifit_hip.so $L/cur.so
for v in cur variants/csr20 variants/csr40; do
  cp $L/$v.so $L/libprifit_hip.so
  echo "== $v"; bash tools/kernel_stats.sh r05w c5 edge_csr | grep edge_csr
done
cp $L/cur.so $L/libprifit_hip.so; rm $L/cur.so
