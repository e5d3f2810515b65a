# per-shape kernel table of the validation rollouts of one reference shape (eager launches, so that every kernel shows): gpurun_out/val_trace_<name>.txt
R=$GRAFT_REPO_ROOT; N=${1:-gnn_64}; O=$R/gpurun_out/val_trace_$N; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 5 600 rocprofv3 --kernel-trace --output-format csv -d $O -- python3 $R/scripts/dev/one_val_rollout.py $N eager > $O/log.txt 2>&1
tail -2 $O/log.txt
python3 $R/scripts/dev/agg_trace.py $O $R/gpurun_out/val_trace_$N.txt | head -40
rm -rf $O
