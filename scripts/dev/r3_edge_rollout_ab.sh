# in-rollout A/B of the edge-kernel generations (the rollout runs at the package power limit: equal kernels in isolation need not be
# equal there).  usage: bash scripts/dev/r3_edge_rollout_ab.sh
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3s; mkdir -p $O; cd $R
HEAD="--steps 20 --warmup 2 --no-cpu-baseline --no-extra-cases --train-steps 0 --no-precision-modes"
for i in 1 2; do for v in "X=0" "MAGNET_EDGE_T32=1" "MAGNET_EDGE_T32=2" "MAGNET_EDGE_SINGLE=1"; do
  env $v python bench.py $HEAD > $O/edge_ab.log 2>&1
  python - $O/edge_ab.log "$v" <<'PY'
import json, sys
for l in open(sys.argv[1]):
    if l.startswith('{"metric"'):
        d = json.loads(l); r = d["roofline"]
        print(sys.argv[2], "rollout ms", round(d["ms_per_step"], 3), "edge launch ms", r.get("launch_ms"), r.get("kernel", "")[:40], flush=True)
PY
done; done
