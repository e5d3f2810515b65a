#!/bin/bash
# First lease of an 8-GPU MI355X node: ONE command that yields the 1 -> 2 -> 4 -> 8 curve instead of a debug session (VERDICT r3 item 9;
# the reference scales with `trainer.gpus=N trainer.strategy=ddp`, README.md:124-125).
#
#     bash scripts/scale_first_lease.sh [out_dir]
#
# Every bench.py invocation below starts its N ranks as FRESH torchrun children before this shell's python has touched a GPU
# (bench.py: launch_ranks), one rank per GPU over RCCL / xGMI.  What is checked and printed per N:
#   * headline only (rollout, no data-path collective):  ms per rollout per rank, world size the process group saw, distinct trajectory
#     seeds / input checksums per rank, aggregate node-updates/s and its ratio to N x the N = 1 value;
#   * three training steps: ms per step, allreduce_ms (one flat 2.1 MB gradient buffer: SURVEY.md 5 estimates ~35 us at 2 GPUs,
#     ~250 us ring-bound at 8), and EQUAL parameter checksums on every rank after the last step (same averaged gradient, same Adam step);
#   * two training steps of the C5 case (magnet_gnn, 256 LR nodes + 256^2 queries): the same with a 10.3 MB gradient buffer.
# Exit code 1 when any of that fails.
set -u
cd "$(dirname "$0")/.."
OUT=${1:-gpurun_out/scale_first_lease}
mkdir -p "$OUT"
export HSA_ENABLE_IPC_MODE_LEGACY=${HSA_ENABLE_IPC_MODE_LEGACY:-0}
NG=$(python -c "import torch; print(torch.cuda.device_count())")      # counting devices does not initialise the GPU
echo "visible GPUs: $NG"
rc=0
for N in 1 2 4 8; do
  [ "$N" -gt "$NG" ] && { echo "N=$N skipped (only $NG GPUs)"; continue; }
  python bench.py --gpus $N --steps 10 --warmup 2 --no-cpu-baseline --no-extra-cases --no-precision-modes --train-steps 0 \
      > "$OUT/headline_n$N.json" 2> "$OUT/headline_n$N.err" || { echo "N=$N headline run FAILED (see $OUT/headline_n$N.err)"; rc=1; continue; }
  python bench.py --gpus $N --steps 2 --warmup 1 --no-cpu-baseline --no-extra-cases --no-precision-modes --train-steps 3 \
      > "$OUT/train_n$N.json" 2> "$OUT/train_n$N.err" || { echo "N=$N training run FAILED (see $OUT/train_n$N.err)"; rc=1; continue; }
  # the larger gradient buffer: magnet_gnn C5 (2.57 M parameters = 10.3 MB per all-reduce; SURVEY.md 5 estimates 35 us direct / 250 us ring at 8)
  python bench.py --gpus $N --workload magnet_gnn --traj 1 --steps 2 --warmup 1 --train-steps 2 \
      > "$OUT/train_c5_n$N.json" 2> "$OUT/train_c5_n$N.err" || { echo "N=$N C5 training run FAILED (see $OUT/train_c5_n$N.err)"; rc=1; continue; }
done
python - "$OUT" <<'PY' || rc=1
import glob, json, os, sys
out = sys.argv[1]
base, ok = None, True
for n in (1, 2, 4, 8):
    hp, tp = os.path.join(out, f"headline_n{n}.json"), os.path.join(out, f"train_n{n}.json")
    if not (os.path.exists(hp) and os.path.getsize(hp)):
        continue
    h = json.loads(open(hp).read().strip().splitlines()[-1])
    d = h["dist"]
    base = base or h["value"]
    seeds = d.get("trajectory_seed_by_rank", [])
    line = (f"N={n}: world_size_seen {d.get('world_size_seen')}  backend {d.get('backend')}  ms/rollout per rank {d.get('ms_per_step_by_rank')}  "
            f"value {h['value']:.4g}  scaling vs N x (N=1) {h['value'] / (n * base):.3f}  distinct seeds {len(set(seeds))}/{len(seeds)}")
    if d.get("world_size_seen") != n or len(set(seeds)) != len(seeds):
        ok, line = False, line + "  <-- WRONG world size or repeated trajectories"
    print(line)
    if os.path.exists(tp) and os.path.getsize(tp):
        t = json.loads(open(tp).read().strip().splitlines()[-1])
        tr, sums = t["training_step"], t["dist"].get("param_abs_sum_by_rank_after_training", [])
        same = len(set(sums)) <= 1
        print(f"      training: {tr['ms_per_step']} ms/step, allreduce_ms {tr.get('allreduce_ms')}, parameter checksums equal on all ranks: {same}")
        ok = ok and same
    cp = os.path.join(out, f"train_c5_n{n}.json")
    if os.path.exists(cp) and os.path.getsize(cp):
        c = json.loads(open(cp).read().strip().splitlines()[-1])["training_step"]
        hi, lo = c.get("param_abs_sum_max_over_ranks"), c.get("param_abs_sum_min_over_ranks")
        same = hi is not None and lo is not None and hi == lo          # (absent keys are a failure, not None == None)
        print(f"      C5 (magnet_gnn) training: {c['ms_per_step']} ms/step, allreduce_ms {c.get('allreduce_ms')} for {c.get('allreduce_bytes')} bytes, "
              f"parameter checksums equal on all ranks: {same}")
        ok = ok and same
print("OK" if ok else "FAILED")
sys.exit(0 if ok else 1)
PY
exit $rc
