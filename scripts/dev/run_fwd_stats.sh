# kernel stats of the forward rollout only (no training leg): gpurun_out/fwd_stats.csv
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/fwd_stats; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 5 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --train-steps 0 > $O/log.txt 2>&1
cp $O/*/*_kernel_stats.csv $R/gpurun_out/fwd_stats.csv; tail -1 $O/log.txt | cut -c1-200
