# Dev: A/B of MAGNET_ACCUM_PARAM_GRADS (parameter gradients added to p.grad by the one-kernel Linear backward vs torch's AccumulateGrad)
# on the reference's training shapes: steady-state step time of steps 3..7 (scripts/dev/steps_timeline.py), alternating processes
R=${GRAFT_REPO_ROOT:-/root/repo}
for W in gnn_64 gnn_512 cnn_64; do for rep in 1 2; do for V in 1 0; do
  echo -n "$W MAGNET_ACCUM_PARAM_GRADS=$V: "
  MAGNET_ACCUM_PARAM_GRADS=$V python3 $R/scripts/dev/steps_timeline.py $W 8 2>/dev/null | awk '/^step [3-7]/{s+=$3; n++} END{printf "%.2f ms per step (steps 3-7)\n", s/n}'
done; done; done
