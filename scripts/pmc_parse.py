"""per-kernel averages of a rocprofv3 counter_collection.csv: kernel, counter, launches, mean value per launch, min, max, mean kernel duration [ns]"""
import sys, csv, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(dict)
for r in csv.DictReader(open(sys.argv[1])):
    n = r.get("Kernel_Name", "")
    if not (n.startswith(("gf", "void gf", "calib_"))):
        continue
    key = n.split("(")[0].replace("void ", "")
    acc[key][r["Counter_Name"]].append((int(r.get("Dispatch_Id", 0)), float(r["Counter_Value"])))
    dur[key][int(r.get("Dispatch_Id", 0))] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
print("kernel,counter,launches,mean_per_launch,min,max,mean_duration_ns")
for k in sorted(acc):
    for c in sorted(acc[k]):
        per = collections.defaultdict(float)   # one dispatch may be reported in several rows: sum rows of the same dispatch
        for d, v in acc[k][c]:
            per[d] += v
        # launches that exit at once (finished windows, the finalising ba_step of a solve) are not what a roofline is about: keep the dispatches that ran at
        # least a quarter as long as the longest one of the kernel
        dmax = max(dur[k].values())
        keep = [d for d in per if dur[k][d] >= 0.25 * dmax]
        vals = [per[d] for d in keep]
        print("%s,%s,%d,%.1f,%.1f,%.1f,%.0f" % (k.replace(",", ";"), c, len(vals), sum(vals) / len(vals), min(vals), max(vals), sum(dur[k][d] for d in keep) / len(keep)))
