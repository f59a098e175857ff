"""kernel timeline of the last batch solve of scripts/prof_misc.py's workload: `rocprofv3 --kernel-trace --output-format csv -d DIR -- python scripts/prof_misc.py`,
then `python scripts/timeline.py DIR` prints start offset, duration, queue and the idle gap before each back-end kernel."""
import sys, glob, csv
rows = []
for fn in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        if "gfb::" in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void gfb::", ""), r.get("Queue_Id", "?")))
rows.sort()
# last solve = from the last ba_setup-less chain: find the last kernel named ba_step with the biggest preceding gap
n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
rows = rows[-n:]
t0 = rows[0][0]
busy_end = rows[0][0]
tot_gap = 0
for s, e, name, q in rows:
    gap = s - busy_end
    if gap > 0:
        tot_gap += gap
    print("%9.1f us  +%7.1f us  q%-3s gap %6.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, q, gap / 1e3, name))
    busy_end = max(busy_end, e)
print("span %.1f us, idle %.1f us" % ((busy_end - t0) / 1e3, tot_gap / 1e3))
