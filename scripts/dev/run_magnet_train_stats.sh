# kernel stats of the MAgNet[GNN] training leg: gpurun_out/c5_train_stats.csv
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/c5_train_stats; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 5 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 $R/bench.py --workload magnet_gnn --traj 1 --steps 1 --warmup 1 --no-cpu-baseline --train-steps 2 > $O/log.txt 2>&1
cp $O/*/*_kernel_stats.csv $R/gpurun_out/c5_train_stats.csv; tail -1 $O/log.txt | cut -c1-100
