# same-box alternating A/B of environment knobs on the headline leg: bash scripts/dev/r3_env_ab.sh "X=0" "MAGNET_FUSED_STATS=1" ...
R=$GRAFT_REPO_ROOT; cd $R
HEAD="--steps 20 --warmup 2 --no-cpu-baseline --no-extra-cases --train-steps 0 --no-precision-modes"
for i in 1 2 3; do for v in "$@"; do
  env $v python bench.py $HEAD 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v  rollout ms', round(d['ms_per_step'],3), 'edge launch ms', d['roofline']['launch_ms'])"
done; done
