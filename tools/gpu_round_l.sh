mkdir -p gpurun_out/r05l
python tools/gather_bwd_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05l/gather.txt
PAD=1 python tools/gather_bwd_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05l/gather_pad.txt
timeout -k 10 600 python -m pytest tests -m gpu -q -k "gather or sa_ or backbone or train_step" 2>&1 | tail -3
