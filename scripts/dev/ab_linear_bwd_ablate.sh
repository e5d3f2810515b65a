# Dev: time magnet_linear_bwd128_f32 under the ablation libraries of scripts/dev/build_lb_ablations.sh
R=${GRAFT_REPO_ROOT:-/root/repo}
for M in 2105344 218752; do
  echo "== M=$M"; echo -n "shipped: "; python3 $R/scripts/dev/bench_linear_bwd.py $M 2>/dev/null | grep "W=128"
  for L in $R/magnet_amd/csrc/libmagnet_hip_abl*.so; do
    echo -n "$(basename $L .so | sed s/libmagnet_hip_//): "; MAGNET_LIB_PATH=$L python3 $R/scripts/dev/bench_linear_bwd.py $M 2>/dev/null | grep "W=128"
  done
  echo -n "shipped: "; python3 $R/scripts/dev/bench_linear_bwd.py $M 2>/dev/null | grep "W=128"
done
