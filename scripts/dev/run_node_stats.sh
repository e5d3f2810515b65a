# kernel stats of scripts/bench_node_abl.py (node-level kernels at 4 x 256^2 rows)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/node_stats; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 $R/scripts/bench_node_abl.py > $O/log.txt 2>&1
cp $O/*/*_kernel_stats.csv $R/gpurun_out/node_stats.csv; rm -rf $O/*/
