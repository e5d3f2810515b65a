mkdir -p gpurun_out/r3a
./scripts/dev/edge9_probe > gpurun_out/r3a/edge9_probe.txt 2>&1
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r3a/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3a/pytest.log
timeout 600 python bench.py > gpurun_out/r3a/bench.json 2> gpurun_out/r3a/bench.err; echo "bench rc=$?" >> gpurun_out/r3a/bench.err
bash scripts/dev/run_profiles_r3.sh v0 headline > gpurun_out/r3a/prof.log 2>&1
tail -3 gpurun_out/r3a/pytest.log; cat gpurun_out/r3a/edge9_probe.txt; head -c 1500 gpurun_out/r3a/bench.json
