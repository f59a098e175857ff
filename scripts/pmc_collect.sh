#!/bin/bash
# usage: scripts/pmc_collect.sh <tag>   (on the GPU box, from the repo root)
# Separate rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE cannot share a pass; no trace domains besides --kernel-trace), torch-free driver.
tag=$1
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
run() {  # name, counters, driver args...
    local name=$1 ctr=$2; shift 2
    rm -rf /tmp/pmc_$name
    timeout 600 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/pmc_$name -- python $R/scripts/pmc_driver.py "$@" > /tmp/pmc_$name.log 2>&1
    grep PMCINFO /tmp/pmc_$name.log > $R/gpurun_out/${tag}_pmc_${name}.info
    f=$(find /tmp/pmc_$name -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && python $R/scripts/pmc_parse.py "$f" > $R/gpurun_out/${tag}_pmc_${name}.csv
}
run calib_fetch FETCH_SIZE calib
run tracker_fetch FETCH_SIZE tracker 256
run tracker_write WRITE_SIZE tracker 256
run tracker_sq "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" tracker 256
run backend_sq "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT" backend 256
run backend_sq2 "SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES" backend 256
run backend_fetch FETCH_SIZE backend 256
run backend_write WRITE_SIZE backend 256
[ -n "$PMC_SKIP_SPLIT" ] && exit 0   # the split J^T J formulation's kernels (a measured, switched-off alternative: DESIGN.md section 4) on request only
run backend_split_sq "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT" backend_split 256
run backend_split_fetch FETCH_SIZE backend_split 256
run backend_split_write WRITE_SIZE backend_split 256
