# HBM traffic of one kernel (name substring $1) inside a python program ($2 ..): two SEPARATE --pmc passes (FETCH_SIZE, WRITE_SIZE; the guide's
# recipe: never together with other trace domains), launches with the kernel's largest grid only -> gpurun_out/traffic_<tag>.txt
# usage: bash scripts/dev/pmc_traffic_kernel.sh k_mpnn_edge_bwd_co scripts/dev/one_train_step.py mpnn_2d
R=$GRAFT_REPO_ROOT; K=$1; shift; TAG=$(echo $K | tr -c 'A-Za-z0-9_' '_'); O=$R/gpurun_out/traffic_$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 5 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/$c -- python3 $R/$@ > $O/$c.log 2>&1
done
python3 - $O "$K" <<'PY' > $R/gpurun_out/traffic_$TAG.txt
import csv, glob, sys, collections
O, K = sys.argv[1], sys.argv[2]
tot = collections.defaultdict(list)
for f in sorted(glob.glob(O + "/*/*/*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if K in r["Kernel_Name"]:
            tot[(r["Counter_Name"], int(r["Grid_Size"]))].append(float(r["Counter_Value"]))
big = max(g for (_, g) in tot)
print(f"kernel {K}, launches with the largest grid ({big} work-items); units and the gfx950 correction as scripts/collect_traffic.py /")
print("MI355X_MICROARCH.md (HBM): the counters are KiB, FETCH_SIZE reports half of the bytes of wide coalesced reads -> doubled")
rd = wr = None
for (c, g), x in sorted(tot.items()):
    if g == big:
        m = sum(x) / len(x) * 1024.0 * (2.0 if c == "FETCH_SIZE" else 1.0)
        print(f"{c:12s} launches {len(x):4d}  mean per launch {m / 1e9:8.3f} GB")
        rd, wr = (m, wr) if c == "FETCH_SIZE" else (rd, m)
if rd is not None and wr is not None:
    print(f"traffic (read + write) per launch {(rd + wr) / 1e9:8.3f} GB")
PY
cat $R/gpurun_out/traffic_$TAG.txt; rm -rf $O
