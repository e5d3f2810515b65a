# kernel trace of the C1 rollout (scripts/bench_small.py): per-kernel stats + the timeline of ONE graph-replayed forward
R=$GRAFT_REPO_ROOT; T=${1:-x}; O=$R/gpurun_out/small_trace_$T; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 $R/scripts/bench_small.py > $O/log.txt 2>&1
cp $O/*/*_kernel_stats.csv $O/kernel_stats.csv
python3 - $O <<'PY'
import csv, glob, sys
O = sys.argv[1]
rows = list(csv.DictReader(open(glob.glob(O + "/*/*_kernel_trace.csv")[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
tail = rows[-420:]                      # the last replayed rollout (about 45 launches x 9 forwards)
with open(O + "/timeline_tail.csv", "w") as f:
    t0 = int(tail[0]["Start_Timestamp"]); prev = t0
    for r in tail:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        f.write(f"{(s - t0) / 1e3:9.1f},{(e - s) / 1e3:7.1f},{(s - prev) / 1e3:6.1f},{r['Kernel_Name'][:70]},{r['Grid_Size']}\n")
        prev = e
PY
rm -rf $O/*/; grep "C1 GPU" $O/log.txt
