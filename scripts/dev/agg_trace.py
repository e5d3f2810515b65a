"""Dev: aggregate a rocprofv3 kernel trace by (kernel, grid size): calls, average, total - the per-shape view the --stats table
averages away.  usage: python scripts/dev/agg_trace.py <dir with *_kernel_trace.csv> <out.txt>"""
import collections
import csv
import glob
import re
import sys

f = glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True)[0]
agg = collections.defaultdict(lambda: [0, 0])
tot = 0
for r in csv.DictReader(open(f)):
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    name = re.sub(r"\(.*", "", r["Kernel_Name"])[:70]
    k = (name, int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])), int(r.get("Grid_Size_Y", 1)))
    agg[k][0] += 1
    agg[k][1] += d
    tot += d
with open(sys.argv[2], "w") as out:
    out.write(f"total kernel time {tot / 1e6:.2f} ms\n")
    for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:70]:
        out.write(f"{k[0]:70s} wg {k[1]:7d} x{k[2]:<3d} calls {n:6d} avg {t / n / 1e3:9.1f} us  total {t / 1e6:8.2f} ms {100 * t / tot:5.1f}%\n")
print(open(sys.argv[2]).read())
