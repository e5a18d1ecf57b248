mkdir -p gpurun_out/r05h
timeout -k 10 600 python -m pytest tests/test_gpu_backbone.py -q -x -k "algebraic or fused_bn_reduce or fused_da_dw" -s 2>&1 | grep -v amdgpu.ids | tail -25
