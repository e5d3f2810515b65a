# kernel stats of the headline inference rollout alone: gpurun_out/fwd_stats_<tag>.csv
R=$GRAFT_REPO_ROOT; T=${1:-x}; O=$R/gpurun_out/fwd_stats_$T; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --train-steps 0 --no-extra-cases > $O/log.txt 2>&1
cp $O/*/*_kernel_stats.csv $R/gpurun_out/fwd_stats_$T.csv; rm -rf $O/*/; grep '^{"metric"' $O/log.txt | cut -c1-300
