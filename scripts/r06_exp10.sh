#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r06_exp10; mkdir -p $O
GF_GROUP_TIMING=1 timeout 600 python bench.py --e2e-only > $O/e2e_defaults.json 2> $O/e2e_defaults.err
timeout 600 python bench.py --e2e-only --e2e-groups 1 > $O/e2e_one_group.json 2> $O/e2e_one_group.err
timeout 600 python bench.py --e2e-only --host-threads 8 > $O/e2e_h8.json 2> $O/e2e_h8.err
for f in e2e_defaults e2e_one_group e2e_h8; do python -c "
import json; r=json.load(open('$O/$f.json')); print('$f', round(r['window_solves_per_s']), [round(x) for x in r['passes_window_solves_per_s']], r['group_worker_threads'], r['main_thread_ms_per_backend_frame'])"; done
grep "CPU time summed\|wall-clock anatomy" $O/e2e_defaults.err | tail -4 | cut -c1-520
timeout 900 python -m pytest tests/test_estimator_gpu.py -q -m gpu -x 2>&1 | tail -3
