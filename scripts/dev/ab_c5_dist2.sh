for i in 1 2 3; do for v in 0 1; do
  if [ $v = 1 ]; then export MAGNET_MLP_DIST2=1; else unset MAGNET_MLP_DIST2; fi
  python bench.py --workload magnet_gnn --traj 1 --steps 10 --warmup 2 --train-steps 0 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('dist2=$v c5', round(d['ms_per_step'],3), d['roofline']['launch_ms'], d['roofline']['frac'])"
done; done
