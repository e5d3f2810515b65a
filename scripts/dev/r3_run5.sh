# the fused-dW2 edge backward: parity tests, microbench, the training tests, the training leg
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_train_gpu.py -x -q -m gpu -k "edge_backward" 2>&1 | tail -5
timeout 300 python scripts/bench_edge_bwd.py 2>&1 | tail -8
if [ "$1" = full ]; then
timeout 1500 python -m pytest tests/test_train_gpu.py -x -q -m gpu 2>&1 | tail -5
for v in fused_w fused; do MAGNET_EDGE_BWD=$v python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra-cases --no-precision-modes --train-steps 3 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v', json.dumps(d['training_step']))"; done
fi
