#!/bin/bash
# round 6: SQ counters of ba_step in its chain form (GF_BA_CHAIN=1) at 256 and 512 windows (one / two blocks per CU), next to the dense form at 512 -- separate rocprofv3 --pmc passes,
# --kernel-trace only, torch-free driver (scripts/pmc_driver.py)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
CTR="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT"
run() {  # name, chain, B
    rm -rf /tmp/pmc_$1
    GF_BA_COST_ONLY=0 GF_BA_CHAIN=$2 timeout 600 rocprofv3 --pmc $CTR --kernel-trace --output-format csv -d /tmp/pmc_$1 -- python $R/scripts/pmc_driver.py backend $3 > /tmp/pmc_$1.log 2>&1
    f=$(find /tmp/pmc_$1 -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && python $R/scripts/pmc_parse.py "$f" > $R/gpurun_out/r06_pmc_backend_$1_sq.csv
    grep "ba_step" $R/gpurun_out/r06_pmc_backend_$1_sq.csv | cut -c1-160
}
run chain_256 1 256
run chain_512 1 512
run dense_512 0 512
