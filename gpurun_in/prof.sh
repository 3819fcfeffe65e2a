cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
CMD="python $R/bench.py --records 2000000 --steps 2 --warmup 1 --no-cpu"
rm -rf $R/gpurun_out/p_*
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $R/gpurun_out/p_pmc1 -- $CMD > /dev/null 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_FLAT --output-format csv -d $R/gpurun_out/p_pmc2 -- $CMD > /dev/null 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE FETCH_SIZE --output-format csv -d $R/gpurun_out/p_pmc3 -- $CMD > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE SQ_IFETCH SQ_INSTS_SMEM SQC_ICACHE_MISSES SQC_ICACHE_HITS --output-format csv -d $R/gpurun_out/p_pmc4 -- $CMD > /dev/null 2>&1
find $R/gpurun_out -name "*counter_collection.csv" | head
