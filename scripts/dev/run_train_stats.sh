# kernel stats of the MPNN_2d training step alone: gpurun_out/mpnn2d_train_stats.csv  (usage: run_train_stats.sh [tag])
R=$GRAFT_REPO_ROOT; T=${1:-x}; O=$R/gpurun_out/train_stats_$T; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 5 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 $R/scripts/bench_train.py --steps 3 --warmup 1 > $O/log.txt 2>&1
cp $O/*/*_kernel_stats.csv $R/gpurun_out/mpnn2d_train_stats_$T.csv; rm -rf $O/*/; tail -3 $O/log.txt | cut -c1-600
