# SQ counters of the fused edge kernel (bench_edge.py launches it ~24 times); results under gpurun_out/pmc_e7*/
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 5 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $R/gpurun_out/pmc_e7a -- python3 $R/scripts/bench_edge.py 256 4 102 > $R/gpurun_out/pmc_e7a.log 2>&1
if [ "$1" = "all" ]; then
timeout -k 5 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT --kernel-trace --output-format csv -d $R/gpurun_out/pmc_e7b -- python3 $R/scripts/bench_edge.py 256 4 102 > $R/gpurun_out/pmc_e7b.log 2>&1
timeout -k 5 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_INSTS_VMEM SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_e7c -- python3 $R/scripts/bench_edge.py 256 4 102 > $R/gpurun_out/pmc_e7c.log 2>&1
fi
