mkdir -p gpurun_out/r3b
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -k "edge" > gpurun_out/r3b/pytest_edge.log 2>&1; tail -15 gpurun_out/r3b/pytest_edge.log
timeout 300 python scripts/bench_edge16.py > gpurun_out/r3b/bench_edge.txt 2>&1; cat gpurun_out/r3b/bench_edge.txt
