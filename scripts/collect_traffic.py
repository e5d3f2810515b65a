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

# key -> (kernel-name substring, largest_grid_only).  The MAgNet edge chains (the `c5` / `c4b` objects of the default bench line) are
# instances of the generic fused-MLP kernel that also run on the small LR graph and at node level: only the launches with the
# largest grid (= the joint graph's edges) are the roofline kernel.
# (the default line's `c1` leg also launches the MPNN edge kernel, on a 576-edge graph: largest grid only there too)
KERNELS = {"k_mpnn_edge16": ("k_mpnn_edge16", True), "k_mpnn_edge7": ("k_mpnn_edge7", True), "k_mpnn_edge6": ("k_mpnn_edge6", True),
           "k_segment_mean<128>": ("k_segment_mean<128>", False),
           "magnet_gnn:edge_mlp": ("k_mlp_bf16<1, 5, 4, 4, 2, false, true, 2", True),       # fp16x3, 5 Linears 128-wide, LN, CSR mean
           "magnet_cnn_2d:edge_mlp": ("k_mlp_bf16<3, 5, 2, 1, 2, false, true, 2", True),    # one-accumulator fp16x3s, 64-wide hidden, 32-wide output
           "k_mpnn_edge_rows32": ("k_mpnn_edge_rows32", True), "k_mpnn_edge_rowsr": ("k_mpnn_edge_rowsr", True),
           "k_mpnn_edge_rowsg": ("k_mpnn_edge_rowsg", True), "k_mpnn_edge_stream32": ("k_mpnn_edge_stream32", True),
           # the node-level kernels run on every mesh of the line (c1's 100 rows, the reference shapes, 48 x 48 ...): the headline launch
           # is the one with the largest grid - averaged over all of them the figure was 47 MB for a 0.40 GB launch
           "k_node_update2": ("k_node_update2<9, true>", True), "k_node_halves": ("k_node_halves<5>", True)}


def mean_per_launch(d, counter):
    f = glob.glob(d + "/*/*_counter_collection.csv")[0]
    rows = {k: [] for k in KERNELS}
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        for k, (pat, _) in KERNELS.items():
            if pat in r["Kernel_Name"]:
                rows[k].append((int(r["Grid_Size"]), float(r["Counter_Value"])))
    out = {}
    for k, (_, largest) in KERNELS.items():
        v = rows[k]
        if largest and v:
            g = max(x[0] for x in v)
            v = [x for x in v if x[0] == g]
        out[k] = (sum(x[1] for x in v) / len(v) if v else None, len(v), sorted({x[0] for x in v}))
    return out


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
        res["kernels"][k] = {"read_bytes": rd, "write_bytes": wr, "traffic": rd + wr, "launches": [fs[k][1], ws[k][1]],
                             "grid_sizes": fs[k][2], "largest_grid_only": KERNELS[k][1]}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
