# SQ counters of one kernel (name substring $1) inside a python program ($2 ..), a few counters per pass -> gpurun_out/pmc_<tag>.txt
# usage: bash scripts/dev/pmc_kernel.sh k_mpnn_edge_bwd_wp scripts/dev/one_train_step.py mpnn_2d
R=$GRAFT_REPO_ROOT; K=$1; shift; TAG=$(echo $K | tr -c 'A-Za-z0-9_' '_'); O=$R/gpurun_out/pmc_$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  timeout -k 5 400 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/p$i -- python3 $R/$@ > $O/p$i.log 2>&1
done
python3 - $O "$K" <<'PY' > $R/gpurun_out/pmc_$TAG.txt
import csv, glob, sys, collections
O, K = sys.argv[1], sys.argv[2]
tot = collections.defaultdict(list)
grid = collections.Counter()
for f in sorted(glob.glob(O + "/p*/*/*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if K in r["Kernel_Name"]:
            tot[(r["Counter_Name"], r["Grid_Size"])].append(float(r["Counter_Value"]))
big = max(int(g) for (_, g) in tot)
print(f"kernel {K}, launches with the largest grid ({big} work-items)")
v = {c: sum(x) / len(x) for (c, g), x in tot.items() if int(g) == big}
for c, x in v.items():
    print(f"{c:28s} mean per launch {x:16.0f}")
if "SQ_WAVE_CYCLES" in v:
    for c in ("SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_LDS"):
        if c in v:
            print(f"{c} / SQ_WAVE_CYCLES = {v[c] / v['SQ_WAVE_CYCLES']:.3f}")
PY
cat $R/gpurun_out/pmc_$TAG.txt; rm -rf $O
