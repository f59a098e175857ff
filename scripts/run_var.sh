cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pm_$c; timeout 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pm_$c -- python scripts/pmc_lk.py > /tmp/pm_$c.log 2>&1
  grep -h "lk_launches" /tmp/pm_$c.log
  f=$(find /tmp/pm_$c -name "*counter_collection.csv" | head -1)
  lc=$(echo $c | tr A-Z a-z)
  (head -1 $f; grep -E "lk_track_kernel|pyr_|scharr|detect_fused|select_corners" $f) > gpurun_out/r01_k_tracker_pmc_$lc.csv
  python - "$f" "$c" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if r["Counter_Name"] == sys.argv[2]: acc[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    if "gf::" in k: print(sys.argv[2], k, "launches", len(v), "mean", sum(v) / len(v), "max", max(v))
PY
done
