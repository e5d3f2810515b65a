# Dev: HBM traffic of the kernel behind magnet_linear_bwd128_f32 at 2.1 M rows: two separate --pmc passes (FETCH_SIZE, WRITE_SIZE in KiB;
# gfx950: FETCH_SIZE counts wide coalesced reads at half their bytes -> doubled, MI355X_MICROARCH.md), kernel trace only
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/traffic_linear_bwd; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 5 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/$C -- python3 $R/scripts/dev/bench_linear_bwd.py 2105344 > $O/$C.log 2>&1
done
python3 - $O <<'PY'
import csv, glob, sys, collections
O = sys.argv[1]
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(f"{O}/*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "k_linear_bwd" in n:
            key = ("128-wide" if ("128" in n or "ILi4" in n or "<4>" in n) else "64-wide", r["Counter_Name"])
            acc[key][0] += float(r["Counter_Value"]); acc[key][1] += 1
alg = {"128-wide": 3 * 2105344 * 128 * 4, "64-wide": 3 * 2105344 * 64 * 4}
for (w, c), (v, n) in sorted(acc.items()):
    kib = v / max(n, 1)
    b = kib * 1024 * (2 if c == "FETCH_SIZE" else 1)
    print(f"{w} {c:10s} mean per launch {kib:12.0f} KiB -> {b / 1e9:6.3f} GB ({'doubled' if c == 'FETCH_SIZE' else 'as is'}; {n} launches); algorithmic "
          f"{'reads (dz, x)' if c == 'FETCH_SIZE' else 'writes (dx)'} {alg[w] * (2 if c == 'FETCH_SIZE' else 1) / 3 / 1e9:.3f} GB")
PY
rm -rf $O/FETCH_SIZE $O/WRITE_SIZE
