# per-shape kernel table of ONE warm MAgNet training step (arg: magnet_gnn | magnet_cnn_2d | mpnn_2d): gpurun_out/train_trace_<name>.txt
R=$GRAFT_REPO_ROOT; N=${1:-magnet_gnn}; O=$R/gpurun_out/train_trace_$N; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 5 600 rocprofv3 --kernel-trace --output-format csv -d $O -- python3 $R/scripts/dev/one_train_step.py $N > $O/log.txt 2>&1
tail -3 $O/log.txt
python3 $R/scripts/dev/agg_trace.py $O $R/gpurun_out/train_trace_$N.txt | head -60
rm -rf $O
