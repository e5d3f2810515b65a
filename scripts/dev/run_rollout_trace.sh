# per-shape kernel table of the rollout (inference) leg of one workload (arg: magnet_gnn | magnet_cnn_2d): gpurun_out/rollout_trace_<name>.txt
R=$GRAFT_REPO_ROOT; N=${1:-magnet_gnn}; O=$R/gpurun_out/rollout_trace_$N; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export MAGNET_BENCH_GRAPH=0 MAGNET_BENCH_PREWARM_S=0
timeout -k 5 600 rocprofv3 --kernel-trace --output-format csv -d $O -- python3 $R/bench.py --workload $N --traj 1 --steps 4 --warmup 1 --no-cpu-baseline --train-steps 0 > $O/log.txt 2>&1
python3 $R/scripts/dev/agg_trace.py $O $R/gpurun_out/rollout_trace_$N.txt | head -40
rm -rf $O
