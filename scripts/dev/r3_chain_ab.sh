# A/B of the one-accumulator 64-wide chains (MAGNET_CHAIN_GEMM): C4b rollout + training step, both forms, twice each
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3s; mkdir -p $O; cd $R
python scripts/bench_chain_single.py 32 64 > $O/chain64_c.txt 2>&1; tail -n 2 $O/chain64_c.txt
for i in 1 2; do for m in fp16x3s fp16x3; do
  MAGNET_CHAIN_GEMM=$m python bench.py --workload magnet_cnn_2d --traj 1 --steps 10 --warmup 2 --train-steps 3 --no-cpu-baseline > $O/c4b_${m}_$i.log 2>&1
  python - $O/c4b_${m}_$i.log $m <<'PY'
import json, sys
for l in open(sys.argv[1]):
    if l.startswith('{"metric"'):
        d = json.loads(l); t = d.get("training_step") or {}
        print(sys.argv[2], "rollout ms", round(d["ms_per_step"], 3), "train ms", t.get("ms_per_step"), flush=True)
PY
done; done
