#!/bin/bash
# (needs library variants built with -DGF_TOPK_THREADS=512 / 256 and a GF_PYR_TAIL_PARTS switch in gf_tracker.hip -- both were temporary; results below)
# select_topk_kernel with 1024 / 512 / 256 threads per sequence and pyr_down_tail_kernel with 4 / 8 / 15 bands: tracker-alone traces
# measured (us per frame of 256 sequences): select_topk 19.1 / 18.0 / 22.3; pyr_down_tail 18.6 / 19.5 / 22.7 -- nothing to gain, the committed forms (1024 threads, 4 bands) stay
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out/r06_exp23
run() {  # name, env...
  local name=$1; shift
  ( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/p23_$name && env "$@" timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p23_$name -- python $R/bench.py --no-cpu-baseline --no-e2e --no-pcie --no-small-batch --no-large-batch --no-other-configs --no-long-run --steps 40 --no-backend > /dev/null 2>&1 )
  f=$(find /tmp/p23_$name -name "*kernel_stats.csv" | head -1)
  echo "== $name"; grep "select_topk\|pyr_down_tail" $f | cut -d, -f1-7 | cut -c1-150
}
run t1024_p4 A=1
run t512_p4 GF_LIB_PATH=$R/ground-fusion_amd/lib/libgf_topk512.so
run t256_p4 GF_LIB_PATH=$R/ground-fusion_amd/lib/libgf_topk256.so
run t1024_p8 GF_PYR_TAIL_PARTS=8
run t1024_p15 GF_PYR_TAIL_PARTS=15
GF_PYR_TAIL_PARTS=8 GF_LIB_PATH=$R/ground-fusion_amd/lib/libgf_topk256.so timeout 600 python -m pytest tests/test_tracker_gpu.py -m gpu -q -x 2>&1 | tail -2
