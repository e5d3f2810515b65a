# kernel stats of the MAgNet rollouts (no training leg): gpurun_out/c5_stats.csv, gpurun_out/c4b_stats.csv
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for w in magnet_gnn:c5 magnet_cnn_2d:c4b; do
  O=$R/gpurun_out/${w#*:}_stats; mkdir -p $O
  timeout -k 5 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 $R/bench.py --workload ${w%%:*} --traj 1 --steps 5 --warmup 2 --no-cpu-baseline --train-steps 0 > $O/log.txt 2>&1
  cp $O/*/*_kernel_stats.csv $R/gpurun_out/${w#*:}_stats.csv
done
