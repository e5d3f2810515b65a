# one environment knob 0 / 1 (default MAGNET_FLAT_GRADS; KNOB=MAGNET_X0_FREE ...): loss trajectories of 8 steps must be the same
# bits, the step time is printed beside them.  usage: bash scripts/dev/ab_flat_grads.sh [cases...]
R=${GRAFT_REPO_ROOT:-.}
for c in ${@:-gnn_64 cnn_64 c4a gnn_512}; do
  for v in 0 1; do
    env ${KNOB:-MAGNET_FLAT_GRADS}=$v python $R/scripts/dev/many_steps.py $c 8 2>/dev/null | awk '{printf "%s ", $2} END {print ""}' > /tmp/traj_$v.txt
    t=$(env ${KNOB:-MAGNET_FLAT_GRADS}=$v python $R/scripts/dev/one_train_step.py $c 2>/dev/null | grep "per step" | tail -1)
    echo "$c env ${KNOB:-MAGNET_FLAT_GRADS}=$v losses: $(cat /tmp/traj_$v.txt) | $t"
  done
  cmp -s /tmp/traj_0.txt /tmp/traj_1.txt && echo "$c: same trajectory" || echo "$c: TRAJECTORIES DIFFER"
done
