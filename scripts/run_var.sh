cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export GF_NO_TORCH_PRELOAD=1
rm -rf /tmp/pm; timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU --kernel-trace --output-format csv -d /tmp/pm -- python scripts/prof_misc.py > /tmp/pm.log 2>&1
tail -3 /tmp/pm.log
f=$(find /tmp/pm -name "*counter_collection.csv" | head -1); echo $f; head -2 $f
python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for fn in glob.glob('/tmp/pm/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(fn)):
        k = r['Kernel_Name'].split('(')[0].replace('void ', '')
        if 'gfb::' not in k: continue
        acc[k][r['Counter_Name']] += float(r['Counter_Value'])
        if r['Counter_Name'] == 'SQ_WAVE_CYCLES': cnt[k] += 1
for k, v in acc.items():
    n = max(cnt[k], 1)
    print(k, 'launches', n, {c: round(x / n) for c, x in v.items()})
PY
