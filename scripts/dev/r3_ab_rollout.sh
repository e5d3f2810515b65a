# A/B of the edge kernel variants INSIDE the rollout (the package is power limited there, unlike in an isolated kernel loop):
# wall time of the rollout, then the in-graph kernel durations from a kernel trace of each variant
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ab_r3; mkdir -p $O
HEAD="--steps 20 --warmup 3 --no-cpu-baseline --no-extra-cases --train-steps 0 --no-precision-modes"
cd $R
for rep in 1 2 3; do
for v in 0 1 2; do
  export MAGNET_EDGE_T32=$v
  python bench.py $HEAD 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('T32=$v', 'rollout %.2f ms' % d['ms_per_step'], r['kernel'], 'launch %.4f ms' % r['launch_ms'], 'clock', r['clock_mhz'])"
done; done | tee $O/wall.txt
cd /tmp && export TMPDIR=/tmp
for v in 0 1 2; do
  export MAGNET_EDGE_T32=$v
  timeout -k 5 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t$v -- python3 $R/bench.py $HEAD > $O/t$v.log 2>&1
  cp $O/t$v/*/*_kernel_stats.csv $O/kernel_stats_t$v.csv; rm -rf $O/t$v
  echo "== T32=$v"; head -8 $O/kernel_stats_t$v.csv | cut -c1-150
done
