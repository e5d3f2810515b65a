"""top kernels of a rocprofv3 --stats csv: python scripts/dev/top_kernels.py <kernel_stats.csv> [n]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 14
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:n]:
    print(r["Name"][:84].ljust(84), r["Calls"].rjust(6), f'{float(r["AverageNs"]) / 1e3:9.1f} us', f'{float(r["Percentage"]):6.2f} %')
