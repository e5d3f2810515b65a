# Dev: SQ counters of the kernel behind magnet_linear_bwd128_f32 (default generation) at the E-level row count of C5 (three separate --pmc passes, kernel trace only)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_linear_bwd; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_LDS"; do
  i=$((i+1))
  timeout -k 5 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/p$i -- python3 $R/scripts/dev/bench_linear_bwd.py 2105344 > $O/p$i.log 2>&1
done
python3 - $O <<'PY'
import csv, glob, sys, collections
O = sys.argv[1]
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(f"{O}/p*/*/*_counter_collection.csv"):
    rows = list(csv.DictReader(open(f)))
    for r in rows:
        if "k_linear_bwd" in r["Kernel_Name"] and "<2>" not in r["Kernel_Name"] and "ILi2" not in r["Kernel_Name"]:
            a = acc[r["Counter_Name"]]
            a[0] += float(r["Counter_Value"]); a[1] += 1
for k, (v, n) in sorted(acc.items()):
    print(f"{k:28s} mean per launch {v / max(n, 1):16.0f}   ({n} launches)")
PY
rm -rf $O/p*/
