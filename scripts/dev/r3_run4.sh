mkdir -p gpurun_out/r3d
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -k "packing or packs" > gpurun_out/r3d/pytest_pack.log 2>&1; tail -6 gpurun_out/r3d/pytest_pack.log
timeout 900 python -m pytest tests/test_train_gpu.py tests/test_regressions_gpu.py -q -x > gpurun_out/r3d/pytest_train.log 2>&1; tail -6 gpurun_out/r3d/pytest_train.log
for m in magnet_cnn_2d magnet_gnn; do timeout 300 python scripts/dev/find_train_glue.py $m 2>&1 | grep -v Warn | head -8; done
timeout 300 python bench.py --no-cpu-baseline --no-extra-cases --no-precision-modes --steps 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('mpnn_2d train', d['training_step']['ms_per_step'], 'rollout', d['ms_per_step'])"
