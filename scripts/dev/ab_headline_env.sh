# headline leg alone, one environment variable alternating (usage: ab_headline_env.sh NAME v1 v2 ...; ROUNDS=3): rollout ms, edge kernel, ms per launch, frac, clock
R=${GRAFT_REPO_ROOT:-.}; NAME=$1; shift; V=$@; N=${ROUNDS:-3}
for i in $(seq $N); do for v in $V; do
  env $NAME=$v python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --train-steps 0 --no-extra-cases --no-precision-modes --no-ref-shapes 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('$NAME=$v', round(d['ms_per_step'],3), r['kernel'], r['launch_ms'], r['frac'], r.get('clock_mhz'))"
done; done
