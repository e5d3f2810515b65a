# Round profile set: kernel stats + HBM traffic (two PMC passes) of the default bench command.  usage: run_profiles.sh v9
R=$GRAFT_REPO_ROOT; V=${1:-vX}; O=$R/gpurun_out/prof_$V; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 5 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/stats.log 2>&1
timeout -k 5 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --train-steps 0 > $O/fetch.log 2>&1
timeout -k 5 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --train-steps 0 > $O/write.log 2>&1
cd $R && python3 scripts/collect_traffic.py $O/fetch $O/write $O/traffic_$V.json
cp $O/stats/*/*_kernel_stats.csv $O/bench_kernel_stats_$V.csv
cp $O/fetch/*/*_counter_collection.csv $O/pmc_bench_${V}_FETCH_SIZE.csv; cp $O/write/*/*_counter_collection.csv $O/pmc_bench_${V}_WRITE_SIZE.csv
rm -rf $O/stats $O/fetch $O/write
ls -la $O
