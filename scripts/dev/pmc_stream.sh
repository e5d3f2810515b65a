# SQ counters of k_mpnn_edge_stream32 on the metric graph (30 launches), a few counters per pass -> gpurun_out/pmc_stream.txt
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_stream; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_INSTS_BRANCH" "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA" "SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM"; do
  i=$((i+1))
  timeout -k 5 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/p$i -- python3 $R/scripts/dev/stream_only.py > $O/p$i.log 2>&1
done
python3 - $O <<'PY' > $R/gpurun_out/pmc_stream.txt
import csv, glob, sys, collections
O = sys.argv[1]
tot = collections.defaultdict(list)
for f in sorted(glob.glob(O + "/p*/*/*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if "k_mpnn_edge_stream32" in r["Kernel_Name"]:
            tot[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in tot.items():
    print(f"{k:32s} mean per launch {sum(v) / len(v):16.0f}   ({len(v)} launches)")
PY
cat $R/gpurun_out/pmc_stream.txt
