# samples rocm-smi power / sclk while bench.py runs 300 rollouts with the node-level fusion on and off (is the edge kernel clock-limited?)
R=$GRAFT_REPO_ROOT
for v in 1 0; do
  MAGNET_FUSED_HALVES=$v python3 $R/bench.py --steps 300 --warmup 5 --no-cpu-baseline --train-steps 0 --no-extra-cases > /tmp/pp_$v.json 2>/dev/null &
  pid=$!
  sleep 6
  for i in 1 2 3 4; do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Average Graphics Package Power|Current Socket Graphics Package Power|sclk clock level|Power \(W\)" | head -3 | tr '\n' ' '; echo; sleep 1.5; done
  wait $pid
  python3 -c "import json;d=json.load(open('/tmp/pp_$v.json'));print('fused=$v', d['ms_per_step'], d['roofline']['launch_ms'])"
done
