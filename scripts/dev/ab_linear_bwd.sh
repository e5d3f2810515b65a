# Dev: A/B of two builds of the library on magnet_linear_bwd128/64_f32 (alternating processes on one box)
#   bash scripts/dev/ab_linear_bwd.sh <variant .so> [rows ...]
R=${GRAFT_REPO_ROOT:-/root/repo}; V=$1; shift; ROWS=${@:-"2105344 218752"}
for rep in 1 2; do for M in $ROWS; do
  echo "-- shipped  M=$M"; python3 $R/scripts/dev/bench_linear_bwd.py $M
  echo "-- variant  M=$M"; MAGNET_LIB_PATH=$V python3 $R/scripts/dev/bench_linear_bwd.py $M
done; done
