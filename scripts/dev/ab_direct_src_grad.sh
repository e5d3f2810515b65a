# Dev: A/B of MAGNET_DIRECT_SRC_GRAD (dense-source gradients written by the data GEMM vs dX0 + column copy + torch adds) on the C5 bench line
R=${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2; do for V in 1 0; do
  echo -n "MAGNET_DIRECT_SRC_GRAD=$V: "
  MAGNET_DIRECT_SRC_GRAD=$V python3 $R/bench.py --workload magnet_gnn --traj 1 --no-ref-shapes 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); t=d['training_step']
print('rollout', round(d['ms_per_step'],2), 'training', t['ms_per_step'], 'peak GB', t['peak_mem_GB'])"
done; done
