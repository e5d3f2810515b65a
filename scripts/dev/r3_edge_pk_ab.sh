# same-box A/B of the edge kernel with packed-fp32 Swish (product) against the previous one (variants/libmagnet_hip_oldedge.so):
# microbench + headline leg, alternating
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3s; mkdir -p $O; cd $R
HEAD="--steps 20 --warmup 2 --no-cpu-baseline --no-extra-cases --train-steps 0 --no-precision-modes"
for i in 1 2 3; do for v in product oldedge; do
  if [ $v = product ]; then unset MAGNET_LIB_PATH; else export MAGNET_LIB_PATH=$R/magnet_amd/csrc/variants/libmagnet_hip_$v.so; fi
  python scripts/bench_edge16.py 2>/dev/null | grep "^k_mpnn_edge16 l2e=True" | sed "s/^/$v  /"
  python bench.py $HEAD 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v  rollout ms', round(d['ms_per_step'],3), 'edge launch ms', d['roofline']['launch_ms'], 'frac', d['roofline']['frac'])"
done; done
