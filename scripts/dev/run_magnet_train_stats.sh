# kernel stats of a MAgNet training leg (arg: magnet_gnn | magnet_cnn_2d): gpurun_out/c5_train_stats.csv
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/c5_train_stats; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 5 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 $R/bench.py --workload ${1:-magnet_gnn} --traj 1 --steps 1 --warmup 1 --no-cpu-baseline --train-steps 2 > $O/log.txt 2>&1
cp $O/*/*_kernel_stats.csv $R/gpurun_out/${1:-magnet_gnn}_train_stats.csv; rm -rf $O/*/; tail -1 $O/log.txt | cut -c1-100   # (the kernel trace itself would blow the 64 MiB merge limit)
