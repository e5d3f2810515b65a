# same-box A/B of library builds: bash scripts/dev/r3_tail_ab.sh product notail | product bl0 ...  (variants/libmagnet_hip_<name>.so)
# (round 3: the wait states behind every asm split vs a build without them; biases from LDS vs from global memory)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3s; mkdir -p $O; cd $R
HEAD="--steps 20 --warmup 2 --no-cpu-baseline --no-extra-cases --train-steps 3 --no-precision-modes"
VARS=${@:-product notail}
for i in 1 2; do for v in $VARS; do
  if [ $v = product ]; then unset MAGNET_LIB_PATH; else export MAGNET_LIB_PATH=$R/magnet_amd/csrc/variants/libmagnet_hip_$v.so; fi
  python bench.py $HEAD > $O/tail_ab.log 2>&1
  python - $O/tail_ab.log "$v headline" <<'PY'
import json, sys
for l in open(sys.argv[1]):
    if l.startswith('{"metric"'):
        d = json.loads(l); r = d["roofline"]; t = d.get("training_step") or {}
        print(sys.argv[2], "rollout ms", round(d["ms_per_step"], 3), "edge launch ms", r.get("launch_ms"), "train ms", t.get("ms_per_step"), flush=True)
PY
  for w in magnet_gnn magnet_cnn_2d; do
    python bench.py --workload $w --traj 1 --steps 10 --warmup 2 --train-steps 3 --no-cpu-baseline > $O/tail_ab.log 2>&1
    python - $O/tail_ab.log "$v $w" <<'PY'
import json, sys
for l in open(sys.argv[1]):
    if l.startswith('{"metric"'):
        d = json.loads(l); r = d["roofline"]; t = d.get("training_step") or {}
        print(sys.argv[2], "rollout ms", round(d["ms_per_step"], 3), "chain launch ms", r.get("launch_ms"), "train ms", t.get("ms_per_step"), flush=True)
PY
  done
done; done
