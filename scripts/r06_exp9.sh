#!/bin/bash
# round 6, experiment 9: the drop-in path with the quota-aware defaults, and with the members' IMU pre-integration as one device launch per frame
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r06_exp9; mkdir -p $O
timeout 600 python bench.py --e2e-only > $O/e2e_defaults.json 2> $O/e2e_defaults.err
GF_GROUP_DEVICE_PREINT=1 timeout 600 python bench.py --e2e-only > $O/e2e_device_preint.json 2> $O/e2e_device_preint.err
GF_GROUP_DEVICE_PREINT=1 GF_GROUP_DEVICE_SWEEPS=1 timeout 600 python bench.py --e2e-only > $O/e2e_device_preint_sweeps.json 2> $O/e2e_device_preint_sweeps.err
timeout 600 python bench.py --e2e-only --e2e-groups 1 > $O/e2e_one_group_defaults.json 2> $O/e2e_one_group_defaults.err
for f in e2e_defaults e2e_device_preint e2e_device_preint_sweeps e2e_one_group_defaults; do python -c "
import json; r=json.load(open('$O/$f.json')); print('$f', round(r['window_solves_per_s']), [round(x) for x in r['passes_window_solves_per_s']], r['group_worker_threads'], r.get('tracker_host_threads'), r['main_thread_ms_per_backend_frame'])"; done
