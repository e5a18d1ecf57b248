mkdir -p gpurun_out/r05d
L=prifit_amd/lib/libprifit_hip.so
cp $L /tmp/new.so
for v in new glbp1 glbp2 glbp3; do
  if [ $v = new ]; then cp /tmp/new.so $L; else cp prifit_amd/lib/variants/$v.so $L; fi
  echo "== lib $v"; python tools/gather_bwd_bench.py 2>&1 | grep "fused"
done 2>&1 | tee gpurun_out/r05d/gather_probe.txt
cp /tmp/new.so $L
