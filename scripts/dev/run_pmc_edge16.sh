# SQ counters of the three fp16x3 edge kernels (scripts/bench_edge16.py launches each ~140 times); summary per kernel
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_edge_r3; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
CMD="python3 $R/scripts/bench_edge16.py"
timeout -k 5 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $O/a -- $CMD > $O/a.log 2>&1
timeout -k 5 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT --kernel-trace --output-format csv -d $O/b -- $CMD > $O/b.log 2>&1
timeout -k 5 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS GRBM_GUI_ACTIVE SQ_WAVES --kernel-trace --output-format csv -d $O/c -- $CMD > $O/c.log 2>&1
cd $R && python3 - $O <<'PY'
import csv, glob, sys, collections
O = sys.argv[1]
for p in "abc":
    fs = glob.glob(f"{O}/{p}/*/*_counter_collection.csv")
    if not fs:
        print(p, "no output"); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        n = r["Kernel_Name"]
        if "k_mpnn_edge" in n:
            acc[n[:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for n, d in sorted(acc.items()):
        print(p, n, {k: round(sum(v) / len(v)) for k, v in d.items()}, "launches", len(next(iter(d.values()))))
PY
rm -rf $O/a/*/ $O/b/*/ $O/c/*/
