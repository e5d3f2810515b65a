# kernel stats of one MAgNet inference rollout leg (arg: magnet_gnn | magnet_cnn_2d): gpurun_out/<name>_fwd_stats.csv
R=$GRAFT_REPO_ROOT; W=${1:-magnet_gnn}; O=$R/gpurun_out/${W}_fwd; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 $R/bench.py --workload $W --traj 1 --steps 5 --warmup 2 --no-cpu-baseline --train-steps 0 > $O/log.txt 2>&1
cp $O/*/*_kernel_stats.csv $R/gpurun_out/${W}_fwd_stats.csv; rm -rf $O/*/; grep '^{"metric"' $O/log.txt | cut -c1-200
