# Round-2 profile set of the DEFAULT bench command: kernel stats + HBM traffic (two separate PMC passes).  usage: run_profiles_r2.sh v2
R=$GRAFT_REPO_ROOT; V=${1:-vX}; O=$R/gpurun_out/prof_r2_$V; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 5 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/stats.log 2>&1
timeout -k 5 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --train-steps 0 > $O/fetch.log 2>&1
timeout -k 5 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --train-steps 0 > $O/write.log 2>&1
cd $R && python3 scripts/collect_traffic.py $O/fetch $O/write $O/traffic_$V.json > /dev/null
cp $O/stats/*/*_kernel_stats.csv $O/bench_kernel_stats_$V.csv
python3 - $O $V <<'PY'
import csv, glob, sys
O, V = sys.argv[1:3]
for name in ("fetch", "write"):     # keep only the roofline kernels' rows of the raw counter files (the full files are tens of MB)
    f = glob.glob(f"{O}/{name}/*/*_counter_collection.csv")[0]
    keep = ("k_mpnn_edge", "k_segment_mean", "k_mlp_bf16<1, 5,")
    rows = list(csv.reader(open(f)))
    with open(f"{O}/pmc_bench_{V}_{name.upper()}_SIZE.csv", "w", newline="") as g:
        w = csv.writer(g)
        w.writerow(rows[0])
        ki = rows[0].index("Kernel_Name")
        w.writerows(r for r in rows[1:] if any(k in r[ki] for k in keep))
PY
grep '^{"metric"' $O/stats.log | tail -1 > $O/bench_$V.json
rm -rf $O/stats $O/fetch $O/write
ls -la $O
