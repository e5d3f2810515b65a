# Dev: the C5 leg of the bench line (1 trajectory) N times in fresh processes: how stable is its training_step?
R=${GRAFT_REPO_ROOT:-/root/repo}
for i in $(seq 1 ${1:-5}); do
  python3 $R/bench.py --workload magnet_gnn --traj 1 --no-ref-shapes 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); t=d['training_step']
print('rollout', round(d['ms_per_step'],2), 'training', t['ms_per_step'], 'warm-up steps', t.get('warmup_step_ms'), 'discarded', t.get('discarded_ms_per_step'), 'peak GB', t['peak_mem_GB'])"
done
