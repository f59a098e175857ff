#!/bin/bash
# round 6, experiment 6: the drop-in path with more sequences per GPU (honest clock): a sequence's back-end frame has a latency (~5 ms: host phases + solve + marginalisation),
# so window-solves/s = sequences / latency until the GPU or the caller's tracker thread saturates
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r06_exp6; mkdir -p $O
for cfg in "256 2 16" "512 2 16" "512 4 8" "512 4 16" "1024 4 16" "1024 8 8"; do
  set -- $cfg
  GF_GROUP_THREADS=$3 timeout 600 python bench.py --e2e-only --e2e-seqs $1 --e2e-groups $2 > $O/e2e_n$1_g$2_t$3.json 2> $O/e2e_n$1_g$2_t$3.err
  python -c "
import json; r=json.load(open('$O/e2e_n$1_g$2_t$3.json')); print('$cfg', round(r['window_solves_per_s']), [round(x) for x in r['passes_window_solves_per_s']], r['main_thread_ms_per_backend_frame'])"
done
