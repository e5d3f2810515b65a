"""Post-process two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate passes as the TCC slots require) of
bench.py into per-launch HBM traffic of the two roofline kernels.

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --train-steps 0
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --train-steps 0
    python scripts/collect_traffic.py $OUT/fetch $OUT/write profiles/r1/traffic_v7.json

FETCH_SIZE / WRITE_SIZE are reported in KiB.  gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE counts wide
coalesced reads at half their bytes -> doubled here; WRITE_SIZE is taken as is (uncalibrated)."""
import csv
import glob
import json
import sys

KERNELS = {"k_mpnn_edge16": "k_mpnn_edge16", "k_mpnn_edge7": "k_mpnn_edge7", "k_mpnn_edge6": "k_mpnn_edge6", "k_segment_mean<128>": "k_segment_mean<128>"}


def mean_per_launch(d, counter):
    f = glob.glob(d + "/*/*_counter_collection.csv")[0]
    acc = {k: [0.0, 0] for k in KERNELS}
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        for k, pat in KERNELS.items():
            if pat in r["Kernel_Name"]:
                acc[k][0] += float(r["Counter_Value"])
                acc[k][1] += 1
    return {k: (v[0] / v[1] if v[1] else None, v[1]) for k, v in acc.items()}


def main():
    fetch, write, out = sys.argv[1:4]
    fs, ws = mean_per_launch(fetch, "FETCH_SIZE"), mean_per_launch(write, "WRITE_SIZE")
    res = {"unit": "bytes per launch", "how": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), KiB -> bytes, "
                                              "FETCH_SIZE doubled (gfx950 correction), mean over the launches of one bench.py run",
           "kernels": {}}
    for k in KERNELS:
        if fs[k][0] is None or ws[k][0] is None:
            continue
        rd, wr = 2.0 * fs[k][0] * 1024.0, ws[k][0] * 1024.0
        res["kernels"][k] = {"read_bytes": rd, "write_bytes": wr, "traffic": rd + wr, "launches": [fs[k][1], ws[k][1]]}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
