mkdir -p gpurun_out/r05a
python tools/dbg_c5_selfsup.py > gpurun_out/r05a/dbg_c5.txt 2>&1; tail -25 gpurun_out/r05a/dbg_c5.txt
timeout -k 10 900 python -m pytest tests -m gpu -q --deselect tests/test_gpu_bench_step_parity.py -k "dgcnn or nms_with or col_sum or fit or train_step or backbone" > gpurun_out/r05a/gpu_tests2.log 2>&1
tail -15 gpurun_out/r05a/gpu_tests2.log
python tools/rocblas_ref.py > gpurun_out/r05a/rocblas_ref.txt 2>&1; cat gpurun_out/r05a/rocblas_ref.txt
python bench.py --steps 50 > gpurun_out/r05a/c3_bench.json 2> gpurun_out/r05a/c3_bench.err; tail -c 2500 gpurun_out/r05a/c3_bench.json
