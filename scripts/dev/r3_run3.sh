mkdir -p gpurun_out/r3c
timeout 1700 python -m pytest tests -m gpu -q > gpurun_out/r3c/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3c/pytest.log
timeout 600 python bench.py > gpurun_out/r3c/bench.json 2> gpurun_out/r3c/bench.err; echo "bench rc=$?" >> gpurun_out/r3c/bench.err
tail -4 gpurun_out/r3c/pytest.log; tail -3 gpurun_out/r3c/bench.err; head -c 600 gpurun_out/r3c/bench.json
