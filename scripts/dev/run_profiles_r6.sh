# Round-6 profile set.  usage (on the GPU box, from the repo root): bash scripts/dev/run_profiles_r6.sh v1 [headline|c4b|c5|traffic|all]
#   headline : kernel stats of the HEADLINE LEG ALONE (no training leg, no extra cases, no CPU baseline, no precision modes), so that the
#              average duration of the edge kernel in the CSV is the headline launches' and roofline.frac can be recomputed from one row
#   c4b / c5 : the same for `--workload magnet_cnn_2d` / `magnet_gnn` (rollout only) PLUS the per-shape table (kernel x grid size:
#              scripts/dev/agg_trace.py) - the --stats row of the chain kernel mixes the LR-graph and the joint-graph launches
#   traffic  : the two separate --pmc passes (FETCH_SIZE, WRITE_SIZE) of the default line's rollouts -> traffic_$V.json
# The program itself follows `--` (python3 bench.py ...), never a shell or env wrapper.
R=$GRAFT_REPO_ROOT; V=${1:-vX}; WHAT=${2:-all}; O=$R/gpurun_out/prof_r6_$V; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
HEAD="--steps 10 --warmup 2 --no-cpu-baseline --no-extra-cases --train-steps 0 --no-precision-modes"
stats() {   # stats <name> <bench args...>
  local name=$1; shift
  timeout -k 5 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$name -- python3 $R/bench.py "$@" > $O/$name.log 2>&1
  cp $O/$name/*/*_kernel_stats.csv $O/${name}_kernel_stats_$V.csv 2>/dev/null
  python3 $R/scripts/dev/agg_trace.py $O/$name $O/${name}_by_shape_$V.txt > /dev/null 2>&1
  grep '^{"metric"' $O/$name.log | tail -1 > $O/${name}_profiled_$V.json
  rm -rf $O/$name
}
if [ $WHAT = all ] || [ $WHAT = headline ]; then stats headline $HEAD; fi
if [ $WHAT = all ] || [ $WHAT = c4b ]; then stats c4b --workload magnet_cnn_2d --traj 1 --steps 10 --warmup 2 --train-steps 0; fi
if [ $WHAT = all ] || [ $WHAT = c5 ]; then stats c5 --workload magnet_gnn --traj 1 --steps 10 --warmup 2 --train-steps 0; fi
if [ $WHAT = all ] || [ $WHAT = traffic ]; then
  P="--steps 1 --warmup 1 --no-cpu-baseline --train-steps 0 --no-precision-modes --no-ref-shapes"
  timeout -k 5 500 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -- python3 $R/bench.py $P > $O/fetch.log 2>&1
  timeout -k 5 500 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -- python3 $R/bench.py $P > $O/write.log 2>&1
  cd $R && python3 scripts/collect_traffic.py $O/fetch $O/write $O/traffic_$V.json > /dev/null
  python3 - $O $V <<'PY'
import csv, glob, sys
O, V = sys.argv[1:3]
for name in ("fetch", "write"):     # keep only the roofline kernels' rows of the raw counter files (the full files are tens of MB)
    f = glob.glob(f"{O}/{name}/*/*_counter_collection.csv")[0]
    keep = ("k_mpnn_edge16", "k_mpnn_edge_stream32", "k_mpnn_edge_rows32", "k_mpnn_edge_rowsr", "k_mpnn_edge_rowsg", "k_node_halves<5>", "k_segment_mean", "k_mlp_bf16<1, 5, 4, 4, 2, false, true, 2", "k_mlp_bf16<3, 5, 2, 1, 2, false, true, 2",
            "k_node_update2<9, true>")
    rows = list(csv.reader(open(f)))
    gi = rows[0].index("Grid_Size")
    ki = rows[0].index("Kernel_Name")
    sel = [r for r in rows[1:] if any(k in r[ki] for k in keep)]
    big = {}
    for r in sel:
        big[r[ki]] = max(big.get(r[ki], 0), int(r[gi]))
    with open(f"{O}/pmc_bench_{V}_{name.upper()}_SIZE.csv", "w", newline="") as g:
        w = csv.writer(g)
        w.writerow(rows[0])
        seen = {}
        for r in sel:                                                       # the roofline launches: the largest grid of each kernel,
            if int(r[gi]) == big[r[ki]]:                                    # first 40 rows each (the JSON holds the means over all)
                seen[r[ki]] = seen.get(r[ki], 0) + 1
                if seen[r[ki]] <= 40:
                    w.writerow(r)
PY
  rm -rf $O/fetch $O/write
fi
ls -la $O
