"""kernel timeline of the last tracker frames of a traced `python bench.py --no-backend --no-e2e --no-cpu-baseline --steps 6`:
`rocprofv3 --kernel-trace --output-format csv -d DIR -- python bench.py ...`, then `python scripts/timeline_tracker.py DIR [n]`"""
import sys, glob, csv
rows = []
for fn in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        n = r["Kernel_Name"]
        if "gf::" in n and "gfb::" not in n:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n.split("(")[0].replace("void gf::", "").replace("gf::", "")))
rows.sort()
n = int(sys.argv[2]) if len(sys.argv) > 2 else 24
rows = rows[-n:]
t0, busy = rows[0][0], rows[0][0]
for s, e, name in rows:
    print("%9.1f us  +%7.1f us  gap %7.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - busy) / 1e3, name))
    busy = max(busy, e)
